// A HoverCarPhysics-shaped controller (HoverCarPhysics.cpp:113-348): every sub-step it reads the body's transform and
// velocities through physics_world.physics_system->GetBodyInterface(), applies a hover force (spring to a target height +
// damping) and a yaw torque, exactly the call pattern of the reference controller, then PhysicsWorld::think().
#include "PhysicsWorld.h"
#include "JoltUtils.h"
#include <utils/Exception.h>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/Collision/ObjectLayer.h>
#include <Jolt/Physics/Vehicle/VehicleConstraint.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <cstdio>
#include <cmath>

int main()
{
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
		Reference<PhysicsObject> ground = new PhysicsObject(true, PhysicsWorld::createGroundQuadShape(2000.f), nullptr, 0);
		ground->pos = Vec4f(0, 0, -0.5f, 1);
		world->addObject(ground);
		Reference<PhysicsObject> car = new PhysicsObject(true);
		car->is_cube = true; car->scale = Vec3f(2.f, 4.f, 1.f); car->pos = Vec4f(0, 0, 0.6f, 1); car->mass = 1000.f;
		car->motion_type = PhysicsObject::MotionType_dynamic;
		world->addObject(car);
		world->activateObject(car);
		JPH::BodyInterface& bi = world->physics_system->GetBodyInterface();
		const float target_z = 2.0f, k = 8000.f, c = 3000.f, mass = 1000.f;
		for (int s = 0; s < 600; ++s) {
			bi.ActivateBody(car->jolt_body_id);
			const JPH::Mat44 to_world = bi.GetWorldTransform(car->jolt_body_id);
			const JPH::Vec3 vel = bi.GetLinearVelocity(car->jolt_body_id);
			const JPH::Vec3 up = to_world.GetAxisZ();
			const float z = to_world.GetTranslation().GetZ();
			const float f = mass * 9.81f + k * (target_z - z) - c * vel.GetZ();
			bi.AddForce(car->jolt_body_id, up * f);
			if (s < 120) bi.AddTorque(car->jolt_body_id, JPH::Vec3(0, 0, 400.f));
			world->think(1.0 / 60.0);
		}
		JPH::RVec3 p; JPH::Quat q;
		bi.GetPositionAndRotation(car->jolt_body_id, p, q);
		JPH::Vec3 lv, av;
		bi.GetLinearAndAngularVelocity(car->jolt_body_id, lv, av);
		const float yaw = 2.f * std::atan2(q.GetZ(), q.GetW());
		printf("z %.4f vz %.4f yaw %.4f wz %.4f active %d\n", p.GetZ(), lv.GetZ(), yaw, av.GetZ(), (int)bi.IsActive(car->jolt_body_id));
		// BoatPhysics.cpp:40-43: shape volume through the body lock interface (2 x 4 x 1 box)
		const JPH::Body* locked = world->physics_system->GetBodyLockInterface().TryGetBody(car->jolt_body_id);
		const float volume = locked ? locked->GetShape()->GetVolume() : -1.f;
		printf("volume %.3f\n", volume);
		if (std::fabs(volume - 8.f) > 1e-4f) return 3;
		const bool ok = std::fabs(p.GetZ() - target_z) < 0.05f && std::fabs(lv.GetZ()) < 0.05f && yaw > 0.3f;
		return ok ? 0 : 1;
	} catch (glare::Exception& e) { fprintf(stderr, "glare::Exception: %s\n", e.what().c_str()); return 2; }
}
