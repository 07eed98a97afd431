// Drives a scene through the C++ PhysicsWorld facade exactly the way GUIClient does (GUIClient.cpp:3026-3057, 6365-6690):
// new PhysicsObject -> fill fields -> addObject -> activateObject -> think() in a sub-step loop -> read activated_obs.
// Usage: facade_scene <scene.bin (sgp_body_desc[])> <steps> <out.bin (sgp_body_state-like floats)>
#include "PhysicsWorld.h"
#include <utils/Exception.h>
#include "../../include/sgp.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

struct Listener : public PhysicsWorldEventListener
{
	int added = 0, persisted = 0, water = 0;
	void physicsObjectEnteredWater(PhysicsObject&) override { water++; }
	void contactAdded(const JPH::Body&, const JPH::Body&, const JPH::ContactManifold& m) override { added += (int)m.mRelativeContactPointsOn1.size() > 0; }
	void contactPersisted(const JPH::Body&, const JPH::Body&, const JPH::ContactManifold&) override { persisted++; }
};

int main(int argc, char** argv)
{
	if (argc < 4) { fprintf(stderr, "usage\n"); return 2; }
	FILE* f = fopen(argv[1], "rb");
	if (!f) return 2;
	std::vector<sgp_body_desc> descs;
	sgp_body_desc d;
	while (fread(&d, sizeof(d), 1, f) == 1) descs.push_back(d);
	fclose(f);
	const int steps = atoi(argv[2]);
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
		Listener listener;
		if (argc > 4) world->event_listener = &listener;
		std::vector<Reference<PhysicsObject>> obs;
		for (const sgp_body_desc& bd : descs) {
			Reference<PhysicsObject> ob = new PhysicsObject(/*collidable=*/true);
			ob->pos = Vec4f(bd.pos[0], bd.pos[1], bd.pos[2], 1.f);
			ob->rot = Quatf(bd.rot[0], bd.rot[1], bd.rot[2], bd.rot[3]);
			if (bd.shape_type == SGP_SHAPE_BOX && bd.motion_type == SGP_MOTION_STATIC) {       // the ground quad
				ob->shape = PhysicsWorld::createGroundQuadShape(2.f * bd.shape[0]);
				ob->scale = Vec3f(1.f);
			} else if (bd.shape_type == SGP_SHAPE_BOX) { ob->is_cube = true; ob->scale = Vec3f(2 * bd.shape[0], 2 * bd.shape[1], 2 * bd.shape[2]); }
			else if (bd.shape_type == SGP_SHAPE_SPHERE) { ob->is_sphere = true; ob->scale = Vec3f(2 * bd.shape[0]); }
			else { ob->shape = PhysicsWorld::createCapsuleShape(bd.shape[0], bd.shape[1]); ob->scale = Vec3f(1.f); }
			ob->motion_type = bd.motion_type == SGP_MOTION_DYNAMIC ? PhysicsObject::MotionType_dynamic :
				(bd.motion_type == SGP_MOTION_KINEMATIC ? PhysicsObject::MotionType_kinematic : PhysicsObject::MotionType_static);
			ob->mass = bd.mass; ob->friction = bd.friction; ob->restitution = bd.restitution;
			world->addObject(ob);
			world->addObject(ob);                                        // idempotent (PhysicsWorld.cpp:1175)
			if (ob->isDynamic()) world->activateObject(ob);             // GUIClient.cpp:3055-3056
			obs.push_back(ob);
		}
		size_t newly = 0;
		{ Lock lock(world->activated_obs_mutex); newly = world->newly_activated_obs.size(); world->newly_activated_obs.clear(); }
		for (int s = 0; s < steps; ++s) world->think(1.0 / 60.0);
		world->readBackActivatedObjectTransforms();
		size_t n_active;
		{ Lock lock(world->activated_obs_mutex); n_active = world->activated_obs.size(); }
		FILE* o = fopen(argv[3], "wb");
		for (auto& ob : obs) {
			const Vec4f p = world->getPosInJolt(ob);
			const Vec4f v = world->getObjectLinearVelocity(*ob);
			float rec[12] = { ob->pos[0], ob->pos[1], ob->pos[2], ob->rot.v[0], ob->rot.v[1], ob->rot.v[2], ob->rot.v[3], p[0], p[1], p[2], v[0], v[2] };
			fwrite(rec, sizeof(rec), 1, o);
		}
		fclose(o);
		RayTraceResult res;
		world->traceRay(Vec4f(0.3f, 0.2f, 50.f, 1.f), Vec4f(0, 0, -1, 0), 100.f, JPH::BodyID(), res);
		printf("objects %zu newly_activated %zu active %zu contacts_added %d persisted %d ray_hit %d t %.4f\n%s", world->getNumObjects(), newly, n_active,
			listener.added, listener.persisted, res.hit_object != nullptr, res.hit_object ? res.hit_t : -1.f, world->getDiagnostics().c_str());
		// What think() costs on the host beyond the device step (the facade used to allocate and zero-fill event buffers sized to
		// the world's capacity every call): think() with the listener installed and the bare sgp_world_step alternate on the same world.
		{
			auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
			for (auto& ob : obs) if (ob->isDynamic()) world->activateObject(ob);       // keep the pile awake for the measurement
			double t_think = 0, t_step = 0; const int reps = 100;
			for (int s = 0; s < reps; ++s) {
				const double a = now(); world->think(1.0 / 60.0);
				const double b = now(); sgp_world_step(world->world, 1.0f / 60.0f);
				const double c = now();
				t_think += b - a; t_step += c - b;
				for (int kind = SGP_EVENT_ACTIVATED; kind <= SGP_EVENT_CONTACT_PERSISTED; ++kind) { uint32_t n = 0; sgp_world_drain_events(world->world, kind, nullptr, 0, &n); }      // (untimed: the bare step's events are not the next think()'s to deliver)
			}
			printf("think_us %.1f step_us %.1f\n", t_think / reps, t_step / reps);
		}
		// a device-side failure must not pass silently (the reference's think() has no error path; its shape builders throw glare::Exception)
		{
			bool threw = false;
			try { world->think(-1.0); } catch (glare::Exception& e) { threw = true; printf("think(-1) threw: %s\n", e.what().c_str()); }
			if (!threw) printf("think(-1) did not throw\n");
		}
		for (auto& ob : obs) world->removeObject(ob);
		printf("after remove: objects %zu\n", world->getNumObjects());
	} catch (glare::Exception& e) { fprintf(stderr, "glare::Exception: %s\n", e.what().c_str()); return 1; }
	return 0;
}
