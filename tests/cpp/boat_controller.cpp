// A BoatPhysics-shaped controller (BoatPhysics.cpp:35-49,120-267): a box hull floating on the water plane of PhysicsWorld's buoyancy
// sweep.  Every frame, exactly the call pattern of the reference controller: world transform and velocities through
// physics_system->GetBodyInterface(), thrust applied at the propellor point while it is under water (AddForce(id, F, point)), a rudder
// force at the same point proportional to the forward speed, and quadratic water drag scaled by
// PhysicsObject::last_submerged_volume / shape volume (the field think() maintains, PhysicsWorld.cpp:1414-1437); then think().
#include "PhysicsWorld.h"
#include "JoltUtils.h"
#include <utils/Exception.h>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <cstdio>
#include <cmath>

int main()
{
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
		world->setWaterBuoyancyEnabled(true);
		world->setWaterZ(0.f);
		Reference<PhysicsObject> boat = new PhysicsObject(true);
		boat->is_cube = true; boat->scale = Vec3f(2.f, 5.f, 1.f); boat->pos = Vec4f(0, 0, 0.4f, 1);
		boat->mass = 3000.f;                        // 10 m^3 hull: floats with 3000 / (1020 * 10) = 29 % of its volume under water
		boat->motion_type = PhysicsObject::MotionType_dynamic;
		world->addObject(boat);
		world->activateObject(boat);
		JPH::BodyInterface& bi = world->physics_system->GetBodyInterface();
		const JPH::Body* locked = world->physics_system->GetBodyLockInterface().TryGetBody(boat->jolt_body_id);      // BoatPhysics.cpp:40-43
		const float shape_volume = locked ? locked->GetShape()->GetVolume() : 0.f;
		if (std::fabs(shape_volume - 10.f) > 1e-3f) { printf("volume %.3f\n", shape_volume); return 3; }
		const float thrust_force = 9000.f, rudder_factor = 600.f;
		const float front_area = 2.0f * 0.3f, side_area = 5.0f * 0.3f, top_area = 10.f;
		float settle_z = 0, settle_frac = 0, speed_straight = 0, yaw_after_turn = 0, y_before_turn = 0;
		for (int s = 0; s < 1500; ++s) {
			const float forward = s >= 300 ? 1.f : 0.f;                 // settle for 5 s, then full throttle
			const float right = (s >= 900 && s < 1200) ? 1.f : 0.f;    // 5 s of right rudder
			const JPH::Mat44 to_world = bi.GetWorldTransform(boat->jolt_body_id);
			const JPH::Vec3 right_ws = to_world.GetAxisX(), forward_ws = to_world.GetAxisY(), up_ws = to_world.GetAxisZ();
			const JPH::Vec3 lin_vel = bi.GetLinearVelocity(boat->jolt_body_id);
			const float forwards_vel = lin_vel.Dot(forward_ws);
			if (right != 0.f || forward != 0.f) bi.ActivateBody(boat->jolt_body_id);
			const JPH::Vec3 propellor_ws = to_world.GetTranslation() + forward_ws * -2.3f + up_ws * -0.4f;
			if (forward != 0.f && propellor_ws.GetZ() <= world->getWaterZ()) {
				const JPH::Vec3 dir = (forward_ws - up_ws * 0.2f - right_ws * (right * 0.3f)).Normalized();
				bi.AddForce(boat->jolt_body_id, dir * (thrust_force * forward), propellor_ws);
			}
			if (right != 0.f) bi.AddForce(boat->jolt_body_id, right_ws * (-right * forwards_vel * rudder_factor), propellor_ws);
			// drag (BoatPhysics.cpp:232-262)
			const float sub = boat->last_submerged_volume;
			const float v_mag = lin_vel.Length();
			if (sub > 0.f && v_mag > 1.0e-3f) {
				const JPH::Vec3 nv = lin_vel / v_mag;
				const float rho = 1020.f, frac = sub / shape_volume, top_frac = sub < 1.f ? sub * sub * (3.f - 2.f * sub) : 1.f;
				const float Fd = 0.5f * rho * v_mag * v_mag * (0.1f * std::fabs(nv.Dot(forward_ws)) * front_area * frac + 0.5f * std::fabs(nv.Dot(right_ws)) * side_area * frac +
				                                                  0.75f * std::fabs(nv.Dot(up_ws)) * top_area * top_frac);
				bi.AddForce(boat->jolt_body_id, nv * -Fd);
			}
			world->think(1.0 / 60.0);
			if (s >= 120 && s < 300) { settle_z += bi.GetCenterOfMassPosition(boat->jolt_body_id).GetZ() / 180.f; settle_frac += boat->last_submerged_volume / shape_volume / 180.f; }      // mean over the bobbing
			if (s == 899) { speed_straight = bi.GetLinearVelocity(boat->jolt_body_id).Length(); y_before_turn = to_world.GetTranslation().GetY(); }
			if (s == 1199) { const JPH::Quat q = bi.GetRotation(boat->jolt_body_id); yaw_after_turn = 2.f * std::atan2(q.GetZ(), q.GetW()); }
		}
		const JPH::Vec3 p = bi.GetCenterOfMassPosition(boat->jolt_body_id);
		printf("settled: z %.3f submerged fraction %.3f (expected 0.294) | straight: speed %.2f m/s after %.1f m | after the turn: yaw %.2f rad | end %.1f %.1f %.2f underwater %d\n",
		       settle_z, settle_frac, speed_straight, y_before_turn, yaw_after_turn, p.GetX(), p.GetY(), p.GetZ(), (int)boat->underwater);
		const bool floats = std::fabs(settle_frac - 3000.f / (1020.f * 10.f)) < 0.03f && std::fabs(settle_z - (0.5f - 0.294f)) < 0.05f;
		const bool drives = speed_straight > 3.f && speed_straight < 30.f && y_before_turn > 15.f;
		const bool turns = yaw_after_turn < -0.3f;          // right rudder: the stern is pushed left, the bow swings clockwise seen from above
		return (floats && drives && turns && boat->underwater) ? 0 : 1;
	} catch (glare::Exception& e) { fprintf(stderr, "glare::Exception: %s\n", e.what().c_str()); return 2; }
}
