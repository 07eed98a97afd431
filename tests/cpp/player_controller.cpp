// A PlayerPhysics-shaped caller (gui_client/PlayerPhysics.cpp:60-90,253-353): a JPH::CharacterVirtual built the way
// PlayerPhysics::init builds it (capsule r 0.3, cylinder height 1.3, position at the capsule's bottom, supporting volume = lower
// sphere, max strength 1000), driven every frame by the velocity rule of PlayerPhysics::update (on ground: desired + ground
// velocity; in the air: accelerate; gravity always; jump through the ground normal) and ExtendedUpdate with the reference's
// stick-to-floor (0.5 m) and stair (0.4 m) settings, while PhysicsWorld::think steps the world around it.
#include "PhysicsWorld.h"
#include "JoltUtils.h"
#include <utils/Exception.h>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/Collision/ObjectLayer.h>
#include <Jolt/Physics/Character/Character.h>
#include <Jolt/Physics/Character/CharacterVirtual.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <Jolt/Physics/Collision/Shape/CapsuleShape.h>
#include <Jolt/Physics/Collision/Shape/RotatedTranslatedShape.h>
#include <cstdio>
#include <chrono>
#include <string>
#include <cmath>
#include <cstdlib>

static const float SPHERE_RAD = 0.3f, CYLINDER_HEIGHT = 1.3f;

struct Player : public JPH::CharacterContactListener
{
	JPH::CharRef<JPH::CharacterShape> standing_shape;
	std::unique_ptr<JPH::CharacterVirtual> jolt_character;
	bool allow_sliding = true; int contacts_added = 0;
	JPH::TempAllocator temp_allocator;

	void init(PhysicsWorld& physics_world, const JPH::Vec3& bottom_pos)
	{
		standing_shape = JPH::RotatedTranslatedShapeSettings(JPH::Vec3(0, 0, 0.5f * CYLINDER_HEIGHT + SPHERE_RAD), JPH::Quat(0.7071068f, 0, 0, 0.7071068f),
			new JPH::CapsuleShape(0.5f * CYLINDER_HEIGHT, SPHERE_RAD)).Create().Get();
		JPH::CharRef<JPH::CharacterVirtualSettings> settings = new JPH::CharacterVirtualSettings();
		settings->mShape = standing_shape;
		settings->mUp = JPH::Vec3(0, 0, 1);
		settings->mSupportingVolume = JPH::Plane(JPH::Vec3(0, 0, 1), -SPHERE_RAD);
		settings->mMaxStrength = 1000;
		jolt_character.reset(new JPH::CharacterVirtual(settings, bottom_pos, JPH::Quat(), physics_world.physics_system));
		jolt_character->SetListener(this);
	}
	void OnContactAdded(const JPH::CharacterVirtual*, const JPH::BodyID&, const JPH::SubShapeID&, JPH::RVec3Arg, JPH::Vec3Arg, JPH::CharacterContactSettings&) override { ++contacts_added; }
	void OnContactSolve(const JPH::CharacterVirtual* ch, const JPH::BodyID&, const JPH::SubShapeID&, JPH::RVec3Arg, JPH::Vec3Arg n, JPH::Vec3Arg contact_velocity, const JPH::PhysicsMaterial*, JPH::Vec3Arg, JPH::Vec3& new_velocity) override
	{
		// anti-sliding rule of PlayerPhysics::OnContactSolve (:535-545)
		if (!allow_sliding && contact_velocity.LengthSq() < 1.0e-12f && !ch->IsSlopeTooSteep(n)) new_velocity = JPH::Vec3(0, 0, 0);
	}
	// How deep the character stands in the water: 0 dry ... 1 when the water is an eye height above the feet (the rule of PlayerPhysics.cpp:179-195, through the
	// facade's water getters).  From 0.3 on the caller treats the character as swimming.
	static constexpr float EYE_HEIGHT = 1.67f;
	static float submerged_fraction(PhysicsWorld& physics_world, const JPH::Vec3& feet)
	{
		if (!physics_world.getWaterBuoyancyEnabled()) return 0.f;
		return std::fmin(1.f, std::fmax(0.f, (physics_world.getWaterZ() - feet.z) / EYE_HEIGHT));
	}
	// PlayerPhysics::update (:253-353) without flying; the swimming branch (:266-294): a swimmer keeps the vertical part of what it wants, floats up with
	// 1.1 g per submerged fraction and loses up to 20 % of its velocity per frame to the water
	void update(PhysicsWorld& physics_world, const JPH::Vec3& move_desired_vel, bool jump, float dtime)
	{
		allow_sliding = move_desired_vel.LengthSq() != 0.f;
		JPH::Vec3 vel = jolt_character->GetLinearVelocity();
		const float wet = submerged_fraction(physics_world, jolt_character->GetPosition());
		JPH::Vec3 parallel_vel = move_desired_vel; if (wet < 0.3f) parallel_vel.z = 0;
		jolt_character->UpdateGroundVelocity();
		if (jolt_character->IsSupported() && (vel.z - jolt_character->GetGroundVelocity().GetZ()) < 0.1f) vel = parallel_vel + jolt_character->GetGroundVelocity();
		else vel = vel + parallel_vel * dtime;
		vel = vel + JPH::Vec3(0, 0, -9.81f) * dtime;
		vel = vel + JPH::Vec3(0, 0, 9.81f * 1.1f * wet) * dtime;
		vel = vel * (1.f - std::fmin(0.2f, 2.0f * wet * dtime));
		if (vel.z < -100) vel.z = -100;
		if (jump && jolt_character->IsSupported()) {
			const JPH::Vec3 gn = jolt_character->GetGroundNormal();
			const float d = move_desired_vel.x * gn.x + move_desired_vel.y * gn.y + move_desired_vel.z * gn.z;
			vel = (move_desired_vel - gn * d) + jolt_character->GetGroundVelocity() + JPH::Vec3(0, 0, 4.5f);
		}
		jolt_character->SetLinearVelocity(vel);
		JPH::CharacterVirtual::ExtendedUpdateSettings settings;
		settings.mStickToFloorStepDown = JPH::Vec3(0, 0, -0.5f);
		settings.mWalkStairsStepUp = JPH::Vec3(0.0f, 0.0f, 0.4f);
		const JPH::BroadPhaseLayerFilter& bp = physics_world.physics_system->GetDefaultBroadPhaseLayerFilter(1);
		const JPH::ObjectLayerFilter& ol = physics_world.physics_system->GetDefaultLayerFilter(1);
		const JPH::BodyFilter bf; const JPH::ShapeFilter sf;
		if (!by_pieces) { jolt_character->ExtendedUpdate(dtime, physics_world.physics_system->GetGravity(), settings, bp, ol, bf, sf, temp_allocator); return; }
		// What PlayerPhysics::update really compiles (:357-446) is its own spelling-out of ExtendedUpdate from the character's public pieces --
		// GetUp, CancelVelocityTowardsSteepSlopes, Update, StickToFloor, CanWalkStairs, WalkStairs -- so those members have to exist and to
		// add up to the same motion.  The same sequence, in this test's words:
		const JPH::Vec3 up = jolt_character->GetUp(), wanted = jolt_character->GetLinearVelocity();
		jolt_character->SetLinearVelocity(jolt_character->CancelVelocityTowardsSteepSlopes(wanted));
		const JPH::Vec3 before = jolt_character->GetPosition();
		bool left_ground = jolt_character->IsSupported();
		jolt_character->Update(dtime, physics_world.physics_system->GetGravity(), bp, ol, bf, sf, temp_allocator);
		if (jolt_character->IsSupported()) left_ground = false;
		if (left_ground && !settings.mStickToFloorStepDown.IsNearZero() && JPH::Vec3(jolt_character->GetPosition() - before).Dot(up) / dtime <= 1.0e-6f)
			jolt_character->StickToFloor(settings.mStickToFloorStepDown, bp, ol, bf, sf, temp_allocator);
		if (settings.mWalkStairsStepUp.IsNearZero()) return;
		JPH::Vec3 want = wanted * dtime; want -= want.Dot(up) * up;
		const float want_len = want.Length();
		if (!(want_len > 0.0f)) return;
		const JPH::Vec3 dir = want / want_len;
		JPH::Vec3 got = JPH::Vec3(jolt_character->GetPosition() - before); got -= got.Dot(up) * up;
		const float got_len = std::max(0.0f, got.Dot(dir));
		if (got_len + 1.0e-4f < want_len && jolt_character->CanWalkStairs(wanted)) {
			JPH::Vec3 test = -jolt_character->GetGroundNormal(); test -= test.Dot(up) * up; test = test.NormalizedOr(dir);
			if (test.Dot(dir) < settings.mWalkStairsCosAngleForwardContact) test = dir;
			jolt_character->WalkStairs(dtime, settings.mWalkStairsStepUp, dir * std::max(settings.mWalkStairsMinStepForward, want_len - got_len), test * settings.mWalkStairsStepForwardTest,
				settings.mWalkStairsStepDownExtra, bp, ol, bf, sf, temp_allocator);
		}
	}
	bool by_pieces = false;
};

static Reference<PhysicsObject> addBox(PhysicsWorld& w, const Vec3f& size, const Vec4f& pos, PhysicsObject::MotionType mt, float mass = 100.f, const Quatf& rot = Quatf::identity())
{
	Reference<PhysicsObject> ob = new PhysicsObject(true);
	ob->is_cube = true; ob->scale = size; ob->pos = pos; ob->rot = rot; ob->mass = mass; ob->motion_type = mt;
	w.addObject(ob);
	if (mt != PhysicsObject::MotionType_static) w.activateObject(ob);
	return ob;
}

#define CHECK(cond) do { if (!(cond)) { printf("FAILED at step %d: %s  (pos %.3f %.3f %.3f)\n", step, #cond, p.x, p.y, p.z); return 1; } } while (0)

int main(int argc, char** argv)
{
	const bool by_pieces = argc > 1 && std::string(argv[1]) == "pieces";      // drive the character through the pieces of ExtendedUpdate, like PlayerPhysics.cpp:357-446
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
		Reference<PhysicsObject> ground = new PhysicsObject(true, PhysicsWorld::createGroundQuadShape(2000.f), nullptr, 0);
		ground->pos = Vec4f(0, 0, -0.5f, 1);
		world->addObject(ground);
		addBox(*world, Vec3f(2, 4, 0.3f), Vec4f(6, 0, 0.15f, 1), PhysicsObject::MotionType_static);                 // a 0.3 m step across the path (x in [5,7])
		addBox(*world, Vec3f(1, 6, 4), Vec4f(12.5f, 0, 2, 1), PhysicsObject::MotionType_static);                    // a wall at x = 12
		addBox(*world, Vec3f(4, 4, 0.2f), Vec4f(0, -10, 1.5f, 1), PhysicsObject::MotionType_static,
			100.f, Quatf::fromAxisAndAngle(Vec4f(0, 1, 0, 0), -1.15f));                                               // a 66 degree ramp facing -x at y = -10
		Reference<PhysicsObject> crate = addBox(*world, Vec3f(0.8f, 0.8f, 0.8f), Vec4f(-6, 6, 0.4f, 1), PhysicsObject::MotionType_dynamic, 20.f);
		Reference<PhysicsObject> platform = addBox(*world, Vec3f(3, 3, 0.4f), Vec4f(-10, -3, 0.2f, 1), PhysicsObject::MotionType_kinematic);

		Player player;
		player.by_pieces = by_pieces;
		player.init(*world, JPH::Vec3(0, 0, 2.0f));
		const float dt = 1.f / 60.f;
		int step = 0; JPH::Vec3 p;
		double update_us = 0.0; int updates = 0;      // host time of the character's update (its shape queries are blocking calls into the library)
		auto frame = [&](const JPH::Vec3& desired, bool jump = false) {
			world->moveKinematicObject(*platform, Vec4f(-10, -3 + 1.0f * dt * (float)(step + 1), 0.2f, 1), Quatf::identity(), dt);       // platform drifts +y at 1 m/s
			const auto t_up0 = std::chrono::steady_clock::now();
			player.update(*world, desired, jump, dt);
			update_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_up0).count(); ++updates;
			world->think(dt);
			++step; p = player.jolt_character->GetPosition();
			if (getenv("PLAYER_DBG") && step >= atoi(getenv("PLAYER_DBG")) && step < atoi(getenv("PLAYER_DBG")) + 40) { const JPH::Vec3 v = player.jolt_character->GetLinearVelocity(); printf("  step %d pos %.4f %.4f %.4f vel %.3f %.3f %.3f state %d gn %.3f %.3f %.3f ncontacts %d\n", step, p.x, p.y, p.z, v.x, v.y, v.z, (int)player.jolt_character->GetGroundState(), player.jolt_character->GetGroundNormal().x, player.jolt_character->GetGroundNormal().y, player.jolt_character->GetGroundNormal().z, (int)player.jolt_character->GetActiveContacts().size()); }
		};
		// 1. fall and land
		for (int i = 0; i < 90; ++i) frame(JPH::Vec3(0, 0, 0));
		CHECK(player.jolt_character->IsSupported() && std::fabs(p.z) < 0.05f && std::fabs(p.x) < 1e-3f);
		CHECK(player.jolt_character->GetGroundState() == JPH::CharacterVirtual::EGroundState::OnGround && player.jolt_character->GetGroundNormal().z > 0.99f);
		// 2. walk +x at 3 m/s for a second
		for (int i = 0; i < 60; ++i) frame(JPH::Vec3(3, 0, 0));
		CHECK(std::fabs(p.x - 3.0f) < 0.15f && std::fabs(p.z) < 0.05f);
		// 3. on to the 0.3 m step (stairs), across it, and down the other side (stick to floor)
		float max_z = 0;
		for (int i = 0; i < 170; ++i) { frame(JPH::Vec3(3, 0, 0)); max_z = std::fmax(max_z, p.z); if (p.x > 5.6f && p.x < 6.4f) CHECK(std::fabs(p.z - 0.3f) < 0.06f && player.jolt_character->IsSupported()); }
		CHECK(max_z > 0.25f && max_z < 0.45f && p.x > 7.5f && std::fabs(p.z) < 0.05f && player.jolt_character->IsSupported());
		// 4. into the wall: stops a radius (+ padding) in front of it
		for (int i = 0; i < 120; ++i) frame(JPH::Vec3(3, 0, 0));
		CHECK(std::fabs(p.x - (12.0f - SPHERE_RAD)) < 0.06f && std::fabs(p.z) < 0.05f);
		// 5. jump: leaves the ground, comes back
		frame(JPH::Vec3(0, 0, 0), true);
		bool left_ground = false; float apex = 0;
		for (int i = 0; i < 80; ++i) { frame(JPH::Vec3(0, 0, 0)); if (!player.jolt_character->IsSupported()) left_ground = true; apex = std::fmax(apex, p.z); }
		CHECK(left_ground && apex > 0.6f && apex < 1.3f && player.jolt_character->IsSupported() && std::fabs(p.z) < 0.05f);
		// 6. a slope too steep to stand on: walking against it does not climb it
		player.jolt_character->SetPosition(JPH::Vec3(-3.5f, -10, 0.0f)); player.jolt_character->SetLinearVelocity(JPH::Vec3(0, 0, 0));
		for (int i = 0; i < 120; ++i) frame(JPH::Vec3(3, 0, 0));
		CHECK(p.z < 0.35f && p.x < 0.5f);
		// 7. riding the kinematic platform: carried along +y at its speed
		{
			const float py = -3 + 1.0f * dt * (float)step;
			player.jolt_character->SetPosition(JPH::Vec3(-10, py, 0.45f)); player.jolt_character->SetLinearVelocity(JPH::Vec3(0, 0, 0));
			for (int i = 0; i < 30; ++i) frame(JPH::Vec3(0, 0, 0));
			const float y0 = p.y;
			for (int i = 0; i < 120; ++i) frame(JPH::Vec3(0, 0, 0));
			CHECK(player.jolt_character->IsSupported() && std::fabs((p.y - y0) - 2.0f) < 0.15f && std::fabs(p.z - 0.4f) < 0.06f);
			CHECK(std::fabs(player.jolt_character->GetGroundVelocity().y - 1.0f) < 0.05f);
		}
		// 8. pushing a 20 kg crate
		player.jolt_character->SetPosition(JPH::Vec3(-9, 6, 0.0f)); player.jolt_character->SetLinearVelocity(JPH::Vec3(0, 0, 0));
		for (int i = 0; i < 180; ++i) frame(JPH::Vec3(2, 0, 0));
		world->readBackActivatedObjectTransforms();
		const float crate_x = world->getPosInJolt(crate)[0];
		printf("final pos %.3f %.3f %.3f  crate x %.3f  contacts added %d\n", p.x, p.y, p.z, crate_x, player.contacts_added);
		printf("character update: %.1f us on average over %d updates\n", update_us / (double)updates, updates);
		CHECK(crate_x > -5.5f && p.x > -7.0f && p.x < crate_x - 0.5f);
		CHECK(player.contacts_added >= 5);
		// 9. swimming (PlayerPhysics.cpp:182-205,266-294): the water rises to z = 3 over a pit far from everything else.  Dropped in, the character comes up and floats
		//    where buoyancy (1.1 g x submerged fraction) balances gravity -- the water 1 / 1.1 of an eye height above its feet --, a swimmer's wish to go up or down is
		//    honoured (on dry ground it is dropped), and with the water switched off again the character falls to the floor.
		{
			world->setWaterBuoyancyEnabled(true); world->setWaterZ(3.0f);
			CHECK(world->getWaterBuoyancyEnabled() && world->getWaterZ() == 3.0f);
			player.jolt_character->SetPosition(JPH::Vec3(40, 40, 2.6f)); player.jolt_character->SetLinearVelocity(JPH::Vec3(0, 0, 0));
			for (int i = 0; i < 600; ++i) frame(JPH::Vec3(0, 0, 0));
			const float float_z = 3.0f - Player::EYE_HEIGHT / 1.1f;
			CHECK(!player.jolt_character->IsSupported() && std::fabs(p.z - float_z) < 0.05f && std::fabs(player.jolt_character->GetLinearVelocity().z) < 0.05f);
			for (int i = 0; i < 90; ++i) frame(JPH::Vec3(0, 0, -1.5f));      // dive
			CHECK(p.z < float_z - 0.4f && p.z > 0.05f);
			const float dived_z = p.z;
			for (int i = 0; i < 240; ++i) frame(JPH::Vec3(1, 0, 0));          // swim along: comes back up while it moves
			CHECK(p.z > dived_z + 0.2f && p.x > 41.0f);
			world->setWaterBuoyancyEnabled(false);
			for (int i = 0; i < 120; ++i) frame(JPH::Vec3(0, 0, 1.5f));       // no water: the wish to go up means nothing, the character falls and stands
			CHECK(player.jolt_character->IsSupported() && std::fabs(p.z) < 0.05f);
		}
		printf("swimming: ok\n");
		return 0;
	} catch (glare::Exception& e) { fprintf(stderr, "glare::Exception: %s\n", e.what().c_str()); return 2; }
}
