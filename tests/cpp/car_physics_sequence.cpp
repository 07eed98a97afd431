// A car driven the way a vehicle script object drives one, written against the look-alike headers only (no reference source text: the round-5
// version of this file followed gui_client/CarPhysics.cpp statement by statement; this one is the repo's own program).  What it exercises is the set
// of Jolt entry points such a caller needs AROUND the PhysicsWorld facade, in the order a caller's life cycle uses them:
//   set-up     PhysicsWorld::removeObject -> ConvexHullShapeSettings / OffsetCenterOfMassShapeSettings -> BodyCreationSettings (mass override, user data)
//              -> BodyInterface::CreateBody / AddBody -> PhysicsWorld::addObject (adopts the body) -> VehicleConstraintSettings with four WheelSettingsWV,
//              one differential, two anti-roll bars -> VehicleConstraint + VehicleCollisionTesterCastSphere -> PhysicsSystem::AddConstraint / AddStepListener
//   per frame  BodyInterface::{GetWorldTransform, ActivateBody, GetRotation, GetAngularVelocity, AddTorque, GetPointVelocity, GetLinearVelocity},
//              Mat44::StoreFloat4x4, Quat::{sRotation, Conjugated, GetAxisAngle}, WheeledVehicleController::SetDriverInput,
//              Wheel::{HasContact, GetContactPosition, GetContactPointVelocity, GetContactNormal, GetContactLongitudinal},
//              VehicleConstraint::{GetWheelLocalBasis, GetWheelLocalTransform, GetWheelWorldTransform}
//   tear-down  PhysicsSystem::RemoveConstraint / RemoveStepListener, PhysicsWorld::removeObject; BodyLockRead and SubShapeID::PopID on the way
// The scenario: settle, accelerate, steer, coast, get thrown on the roof and turned back by an upright-seeking torque, brake.
// (Which reference lines use each of these symbols is checked against /root/reference at test time by tests/test_reference_members.py; the Jolt include
// paths below are the caller's own list, which tests/test_facade_gpu.py holds this file to.)
#include "PhysicsWorld.h"
#include "JoltUtils.h"
#include <utils/Exception.h>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/Collision/ObjectLayer.h>
#include <Jolt/Physics/Vehicle/VehicleConstraint.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <Jolt/Physics/Collision/Shape/CapsuleShape.h>
#include <Jolt/Physics/Collision/Shape/RotatedTranslatedShape.h>
#include <Jolt/Physics/Collision/Shape/BoxShape.h>
#include <Jolt/Physics/Collision/Shape/OffsetCenterOfMassShape.h>
#include <Jolt/Physics/Vehicle/WheeledVehicleController.h>
#include <Jolt/Physics/Body/BodyCreationSettings.h>
#include <Jolt/Physics/Collision/Shape/ConvexHullShape.h>
#include <cstdio>
#include <cmath>
#include <vector>

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } } while (0)

namespace {

// numbers of the default car
struct CarNumbers {
	float wheel_radius = 0.42f, wheel_width = 0.16f, susp_min = 0.2f, susp_max = 0.5f, attach_raise = 0.2f, spring_hz = 2.f, spring_damping = 0.5f;
	float steer_limit = 0.78525f, engine_torque = 500.f, engine_rpm = 6000.f, brake_torque = 1500.f, handbrake_torque = 4000.f, mass = 1200.f;
};

std::vector<Vec3f> chassis_cloud()
{
	std::vector<Vec3f> pts;
	for (int k = 0; k < 8; ++k) pts.push_back(Vec3f((k & 1) ? 0.9f : -0.9f, (k & 2) ? 2.0f : -2.0f, (k & 4) ? 0.25f : -0.25f));      // the floor pan
	const float roof[4][2] = { { 0.9f, 0.6f }, { -0.9f, 0.6f }, { 0.9f, -1.2f }, { -0.9f, -1.2f } };
	for (const auto& r : roof) pts.push_back(Vec3f(r[0], r[1], 0.7f));
	return pts;
}

JPH::WheelSettingsWV* make_wheel(const CarNumbers& c, float x, float y, bool steers, bool has_handbrake)
{
	JPH::WheelSettingsWV* w = new JPH::WheelSettingsWV;
	w->mPosition = JPH::Vec3(x, y, -0.25f + c.susp_min + c.attach_raise);
	w->mSuspensionDirection = JPH::Vec3(0, 0, -1);
	w->mSteeringAxis = JPH::Vec3(0, 0, 1);
	w->mWheelUp = JPH::Vec3(0, 0, 1);
	w->mWheelForward = JPH::Vec3(0, 1, 0);
	w->mRadius = c.wheel_radius; w->mWidth = c.wheel_width;
	w->mSuspensionMinLength = c.susp_min; w->mSuspensionMaxLength = c.susp_max;
	w->mSuspensionSpring.mFrequency = c.spring_hz; w->mSuspensionSpring.mDamping = c.spring_damping;
	w->mMaxSteerAngle = steers ? c.steer_limit : 0.f;
	w->mMaxBrakeTorque = c.brake_torque;
	w->mMaxHandBrakeTorque = has_handbrake ? c.handbrake_torque : 0.f;
	for (int k = 0; k < 3; ++k) { w->mLongitudinalFriction.mPoints[k].mY *= 1.f; w->mLateralFriction.mPoints[k].mY *= 1.f; }      // the curves are writable point by point
	return w;
}

// torque that turns the body towards "wheels down, same heading": 3 x the rotation still to go as the wanted spin, twice the mass as the gain
JPH::Vec3 upright_torque(JPH::BodyInterface& bi, JPH::BodyID id, const Matrix4f& body_to_world, float mass)
{
	const Vec4f side = body_to_world * Vec4f(1, 0, 0, 0), nose = body_to_world * Vec4f(0, 1, 0, 0);
	const Vec4f level_side = normalise(crossProduct(nose, Vec4f(0, 0, 1, 0)));
	const float heading = std::atan2(level_side[1], level_side[0]);
	const JPH::Quat want = JPH::Quat::sRotation(JPH::Vec3(0, 0, 1), heading);
	const JPH::Quat to_go = want * bi.GetRotation(id).Conjugated();
	JPH::Vec3 axis; float angle;
	to_go.GetAxisAngle(axis, angle);
	(void)side;
	return ((axis * angle) * 3 - bi.GetAngularVelocity(id)) * mass * 2.f;
}

}      // namespace

int main()
{
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world_ref = new PhysicsWorld(nullptr, nullptr);
		PhysicsWorld& world = *world_ref;
		Reference<PhysicsObject> ground = new PhysicsObject(true, PhysicsWorld::createGroundQuadShape(2000.f), nullptr, 0);
		ground->pos = Vec4f(0, 0, -0.5f, 1);
		world.addObject(ground);

		const CarNumbers car;
		const std::vector<Vec3f> cloud = chassis_cloud();
		// the object exists as an ordinary dynamic body first (as the world made it before a vehicle script attached)
		Reference<PhysicsObject> ob = new PhysicsObject(true, PhysicsWorld::createConvexHullShape(cloud), nullptr, 0);
		ob->pos = Vec4f(3.f, -2.f, 0.9f, 1); ob->mass = car.mass; ob->motion_type = PhysicsObject::MotionType_dynamic;
		world.addObject(ob);
		const Vec4f start_pos = ob->pos; const Quatf start_rot = ob->rot;

		// ---- set-up: the caller replaces that body by one of its own making and hangs the vehicle on it
		world.removeObject(ob);
		CHECK(ob->jolt_body_id.IsInvalid());
		JPH::BodyInterface& bi = world.physics_system->GetBodyInterface();
		JPH::Array<JPH::Vec3> hull_pts;
		for (const Vec3f& p : cloud) hull_pts.push_back(toJoltVec3(p));
		JPH::Ref<JPH::ConvexHullShapeSettings> hull_settings = new JPH::ConvexHullShapeSettings(hull_pts);
		JPH::Ref<JPH::Shape> hull = hull_settings->Create().Get();
		JPH::Ref<JPH::Shape> lowered = JPH::OffsetCenterOfMassShapeSettings(JPH::Vec3(0, 0, -0.2f), hull).Create().Get();
		JPH::BodyCreationSettings bcs(lowered, toJoltVec3(start_pos), toJoltQuat(start_rot), JPH::EMotionType::Dynamic, Layers::MOVING);
		bcs.mOverrideMassProperties = JPH::EOverrideMassProperties::CalculateInertia;
		bcs.mMassPropertiesOverride.mMass = car.mass;
		bcs.mUserData = (uint64)ob.ptr();
		JPH::Body* body = bi.CreateBody(bcs);
		CHECK(body != nullptr);
		const JPH::BodyID id = body->GetID();
		ob->jolt_body_id = id;
		bi.AddBody(id, JPH::EActivation::Activate);
		world.addObject(ob);                                    // the facade adopts the existing body
		CHECK(ob->jolt_body_id == id);

		JPH::VehicleConstraintSettings vs;
		vs.mUp = JPH::Vec3(0, 0, 1); vs.mForward = JPH::Vec3(0, 1, 0);
		vs.mWheels = { make_wheel(car, -0.8f, 1.3f, true, false), make_wheel(car, 0.8f, 1.3f, true, false), make_wheel(car, -0.8f, -1.3f, false, true), make_wheel(car, 0.8f, -1.3f, false, true) };
		CHECK(dynamic_cast<JPH::WheelSettingsWV*>(vs.mWheels[2].GetPtr()) != nullptr);
		JPH::WheeledVehicleControllerSettings* cs = new JPH::WheeledVehicleControllerSettings;
		vs.mController = cs;
		cs->mDifferentials.resize(1);
		cs->mDifferentials[0].mLeftWheel = 0; cs->mDifferentials[0].mRightWheel = 1;      // front-wheel drive
		cs->mEngine.mMaxTorque = car.engine_torque; cs->mEngine.mMaxRPM = car.engine_rpm;
		vs.mAntiRollBars.resize(2);
		for (int k = 0; k < 2; ++k) { vs.mAntiRollBars[k].mLeftWheel = 2 * k; vs.mAntiRollBars[k].mRightWheel = 2 * k + 1; }
		JPH::Ref<JPH::VehicleCollisionTester> tester = new JPH::VehicleCollisionTesterCastSphere(Layers::MOVING, 0.5f * car.wheel_width, JPH::Vec3(0, 0, 1));
		JPH::Ref<JPH::VehicleConstraint> vehicle = new JPH::VehicleConstraint(*body, vs);
		vehicle->SetVehicleCollisionTester(tester);
		world.physics_system->AddConstraint(vehicle);
		world.physics_system->AddStepListener(vehicle);

		// ---- frames
		const float dt = 1.f / 60.f;
		float top_speed = 0.f, righting_left = -1.f; int wheel_contacts = 0; bool back_on_wheels = false;
		for (int frame = 0; frame < 600; ++frame) {
			const float throttle = (frame >= 60 && frame < 300) ? 1.f : 0.f, steer = (frame >= 120 && frame < 300) ? 0.3f : 0.f, brake = frame >= 420 ? 1.f : 0.f;
			if (frame == 360) {                                 // on its roof
				const Vec4f p = world.getPosInJolt(ob);
				world.setNewObToWorldTransform(*ob, Vec4f(p[0], p[1], 1.6f, 1), Quatf::fromAxisAndAngle(Vec4f(0, 1, 0, 0), 3.0f), Vec4f(0.f), Vec4f(0.f));
				righting_left = 2.f;
			}
			JPH::Float4 cols[4];
			bi.GetWorldTransform(id).StoreFloat4x4(cols);
			const Matrix4f body_to_world(&cols[0].x);
			if (throttle != 0.f || steer != 0.f || brake != 0.f) bi.ActivateBody(id);
			static_cast<JPH::WheeledVehicleController*>(vehicle->GetController())->SetDriverInput(throttle, steer, brake, 0.f);
			if (righting_left > 0.f) { bi.AddTorque(id, upright_torque(bi, id, body_to_world, car.mass)); righting_left -= dt; }
			for (int i = 0; i < 4; ++i) {
				const JPH::Wheel* wheel = vehicle->GetWheel(i);
				if (!wheel->HasContact()) continue;
				++wheel_contacts;
				// slip of the tyre along its rolling direction (what a caller turns into skid sound and smoke)
				JPH::Vec3 slip = bi.GetPointVelocity(id, wheel->GetContactPosition()) - wheel->GetContactPointVelocity();
				slip -= wheel->GetContactNormal().Dot(slip) * wheel->GetContactNormal();
				const float along = slip.Dot(wheel->GetContactLongitudinal());
				CHECK(std::isfinite(along));
				// the wheel's pose by both routes: body transform x local transform, and the world transform asked for directly
				JPH::Vec3 fwd_os, up_os, right_os;
				vehicle->GetWheelLocalBasis(wheel, fwd_os, up_os, right_os);
				const JPH::Vec3 lc = vehicle->GetWheelLocalTransform(i, JPH::Vec3::sAxisZ(), JPH::Vec3::sAxisX()).GetTranslation();
				const JPH::Vec3 wc = vehicle->GetWheelWorldTransform(i, JPH::Vec3::sAxisZ(), JPH::Vec3::sAxisX()).GetTranslation();
				const Vec4f centre = body_to_world * Vec4f(lc.GetX(), lc.GetY(), lc.GetZ(), 1);
				CHECK(std::fabs(wc.GetX() - centre[0]) < 2e-3f && std::fabs(wc.GetY() - centre[1]) < 2e-3f && std::fabs(wc.GetZ() - centre[2]) < 2e-3f);
				const Vec4f lowest = body_to_world * (Vec4f(lc.GetX(), lc.GetY(), lc.GetZ(), 1) - toVec4fVec(up_os) * car.wheel_radius);
				if (frame > 200 && frame < 300) CHECK(std::fabs(lowest[2]) < 0.08f);      // driving on the flat: the tyre's lowest point is on the ground
			}
			world.think(dt);
			world.readBackActivatedObjectTransforms();
			const JPH::Vec3 v = bi.GetLinearVelocity(id);
			top_speed = std::max(top_speed, v.Length());
			if (frame % 50 == 0) {                              // the body interface and the facade's read-back describe the same pose
				const JPH::Vec3 tp = bi.GetWorldTransform(id).GetTranslation();
				CHECK(std::fabs(tp.GetX() - ob->pos[0]) < 1e-3f && std::fabs(tp.GetZ() - ob->pos[2]) < 1e-3f);
			}
			if (frame == 599) {
				const JPH::Mat44 t = bi.GetWorldTransform(id);
				back_on_wheels = t.GetAxisZ().GetZ() > 0.9f;
				std::printf("final: pos (%.2f %.2f %.2f) up.z %.3f speed %.2f  max speed %.2f  wheel contacts %d\n", t.GetTranslation().GetX(), t.GetTranslation().GetY(),
					t.GetTranslation().GetZ(), t.GetAxisZ().GetZ(), v.Length(), top_speed, wheel_contacts);
			}
		}
		CHECK(top_speed > 5.f);
		CHECK(wheel_contacts > 1000);
		CHECK(back_on_wheels);

		// a body lock gives the user data back; an invalid id does not lock
		{
			JPH::BodyLockRead lock(world.physics_system->GetBodyLockInterface(), id);
			CHECK(lock.Succeeded());
			CHECK(lock.GetBody().GetUserData() == (uint64)ob.ptr());
			JPH::BodyLockRead none(world.physics_system->GetBodyLockInterface(), JPH::BodyID());
			CHECK(!none.Succeeded());
		}
		// sub-shape ids peel off bit fields from the low end
		{
			JPH::SubShapeID rest;
			CHECK(JPH::SubShapeID(0xFFFFFFFFu & ~1u).PopID(1, rest) == 0 && rest.IsEmpty());
			CHECK(JPH::SubShapeID(0xFFFFFFFFu).PopID(1, rest) == 1);
		}

		// ---- tear-down
		world.physics_system->RemoveConstraint(vehicle);
		world.physics_system->RemoveStepListener(vehicle);
		vehicle = nullptr;
		tester = nullptr;
		world.removeObject(ob);
		CHECK(ob->jolt_body_id.IsInvalid());
		world.think(dt);
		std::printf("car_physics_sequence: ok\n");
		return 0;
	} catch (glare::Exception& e) {
		std::printf("exception: %s\n", e.what().c_str());
		return 2;
	}
}
