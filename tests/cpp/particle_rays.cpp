// A ParticleManager-shaped caller (ParticleManager.cpp:145-274): every frame each particle traces one ray along its velocity for dt
// (traceRay(pos, vel, dt, JPH::BodyID(), results)), bounces off what it hits, else advances.  Run twice over the same particles: through the
// reference's one-ray-per-call facade method and through the batched extension traceRays(); the results must be identical, and the time
// of both is printed (a single traceRay is a kernel launch + a host sync).
#include "PhysicsWorld.h"
#include <utils/Exception.h>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <vector>

struct Particle { Vec4f pos, vel; };

static uint32_t rng_state = 12345u;
static float unitRandom() { rng_state = rng_state * 1664525u + 1013904223u; return (float)(rng_state >> 8) * (1.f / 16777216.f); }

int main()
{
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
		Reference<PhysicsObject> ground = new PhysicsObject(true, PhysicsWorld::createGroundQuadShape(2000.f), nullptr, 0);
		ground->pos = Vec4f(0, 0, -0.5f, 1);
		world->addObject(ground);
		std::vector<Reference<PhysicsObject>> obs;
		for (int i = 0; i < 200; ++i) {      // things for the particles to hit
			Reference<PhysicsObject> ob = new PhysicsObject(true);
			if (i % 2) ob->is_sphere = true; else ob->is_cube = true;
			ob->scale = Vec3f(0.5f + unitRandom()); ob->mass = 10.f; ob->motion_type = PhysicsObject::MotionType_dynamic;
			ob->pos = Vec4f(-15.f + 30.f * unitRandom(), -15.f + 30.f * unitRandom(), 0.6f + 2.f * unitRandom(), 1);
			world->addObject(ob); world->activateObject(ob); obs.push_back(ob);
		}
		for (int s = 0; s < 120; ++s) world->think(1.0 / 60.0);

		const size_t N = 2048;      // the reference's particle cap
		std::vector<Particle> init(N);
		for (size_t i = 0; i < N; ++i) {
			init[i].pos = Vec4f(-15.f + 30.f * unitRandom(), -15.f + 30.f * unitRandom(), 0.3f + 4.f * unitRandom(), 1);
			init[i].vel = Vec4f(-6.f + 12.f * unitRandom(), -6.f + 12.f * unitRandom(), -8.f * unitRandom(), 0);
		}
		const float dt = 1.f / 60.f;
		std::vector<Particle> a = init, b = init;
		size_t hits_a = 0, hits_b = 0;
		double t_serial = 0, t_batch = 0;
		for (int frame = 0; frame < 20; ++frame) {
			// (1) the reference loop: one call per particle
			auto t0 = std::chrono::steady_clock::now();
			for (size_t i = 0; i < N; ++i) {
				RayTraceResult r; r.hit_object = NULL;
				world->traceRay(a[i].pos, a[i].vel, dt, JPH::BodyID(), r);
				if (r.hit_object) {
					++hits_a;
					const Vec4f hitpos = a[i].pos + a[i].vel * r.hit_t;
					const Vec4f n = r.hit_normal_ws;
					a[i].vel = a[i].vel - n * (2.f * dot(a[i].vel, n));
					a[i].pos = hitpos + n * 1.0e-3f + a[i].vel * (dt - r.hit_t);
				} else a[i].pos = a[i].pos + a[i].vel * dt;
			}
			auto t1 = std::chrono::steady_clock::now();
			// (2) the same frame through the batched extension
			std::vector<PhysicsWorld::RayQuery> qs(N); std::vector<RayTraceResult> rs;
			for (size_t i = 0; i < N; ++i) { qs[i].origin = b[i].pos; qs[i].dir = b[i].vel; qs[i].max_t = dt; qs[i].ignore_body_id = JPH::BodyID(); qs[i].collidable_only = false; }
			world->traceRays(qs, rs);
			for (size_t i = 0; i < N; ++i) {
				if (rs[i].hit_object) {
					++hits_b;
					const Vec4f hitpos = b[i].pos + b[i].vel * rs[i].hit_t;
					const Vec4f n = rs[i].hit_normal_ws;
					b[i].vel = b[i].vel - n * (2.f * dot(b[i].vel, n));
					b[i].pos = hitpos + n * 1.0e-3f + b[i].vel * (dt - rs[i].hit_t);
				} else b[i].pos = b[i].pos + b[i].vel * dt;
			}
			auto t2 = std::chrono::steady_clock::now();
			t_serial += std::chrono::duration<double>(t1 - t0).count(); t_batch += std::chrono::duration<double>(t2 - t1).count();
		}
		bool same = hits_a == hits_b;
		for (size_t i = 0; i < N && same; ++i) for (int k = 0; k < 3; ++k) same = same && a[i].pos[k] == b[i].pos[k] && a[i].vel[k] == b[i].vel[k];
		printf("%zu particles x 20 frames: %zu ray hits; one call per ray %.3f ms per frame (%.1f us per ray), batched %.3f ms per frame; identical %d\n",
		       N, hits_a, 1e3 * t_serial / 20, 1e6 * t_serial / 20 / (double)N, 1e3 * t_batch / 20, (int)same);
		return (same && hits_a > 100) ? 0 : 1;
	} catch (glare::Exception& e) { fprintf(stderr, "glare::Exception: %s\n", e.what().c_str()); return 2; }
}
