// A BikePhysics-shaped caller (gui_client/BikePhysics.cpp:124-227,395-617): two WheelSettingsWV (raked front fork), a
// MotorcycleControllerSettings with the reference's lean-spring constants, rear-wheel drive through the 0/1 differential, six
// gears, VehicleCollisionTesterCastCylinder; per sub-step SetDriverInput + EnableLeanController(true) as while a rider is seated.
#include "PhysicsWorld.h"
#include "JoltUtils.h"
#include <utils/Exception.h>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/Collision/ObjectLayer.h>
#include <Jolt/Physics/Vehicle/VehicleConstraint.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <Jolt/Physics/Body/BodyCreationSettings.h>
#include <Jolt/Physics/Vehicle/WheeledVehicleController.h>
#include <Jolt/Physics/Vehicle/MotorcycleController.h>
#include <Jolt/Physics/Collision/Shape/BoxShape.h>
#include <Jolt/Physics/Collision/Shape/OffsetCenterOfMassShape.h>
#include <Jolt/Physics/Collision/Shape/ConvexHullShape.h>
#include <cstdio>
#include <cmath>

int main()
{
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
		Reference<PhysicsObject> ground = new PhysicsObject(true, PhysicsWorld::createGroundQuadShape(2000.f), nullptr, 0);
		ground->pos = Vec4f(0, 0, -0.5f, 1); ground->friction = 1.f;
		world->addObject(ground);
		const float s = 0.18f, wheel_radius = 3.856f / 2 * s, wheel_width = 0.94f * s;
		Reference<PhysicsObject> bike = new PhysicsObject(true);
		bike->is_cube = true; bike->scale = Vec3f(1.7f * s, 9.f * s, 3.2f * s); bike->pos = Vec4f(0, 0, 0.7f, 1); bike->mass = 200.f; bike->restitution = 0.f;
		bike->motion_type = PhysicsObject::MotionType_dynamic;
		world->addObject(bike);
		world->activateObject(bike);

		JPH::VehicleConstraintSettings vehicle;
		vehicle.mUp = JPH::Vec3(0, 0, 1);
		vehicle.mForward = JPH::Vec3(0, 1, 0);
		const float al = std::sqrt(1.87f * 1.87f + 2.37f * 2.37f);
		const JPH::Vec3 steering_axis(0, -1.87f / al, 2.37f / al);
		JPH::WheelSettingsWV* front_wheel = new JPH::WheelSettingsWV;
		front_wheel->mPosition = JPH::Vec3(0, 0.65f, 0.15f);                 // (+0.15: the reference's centre-of-mass offset of -0.15)
		front_wheel->mSuspensionDirection = steering_axis * -1.f;
		front_wheel->mSteeringAxis = steering_axis; front_wheel->mWheelUp = steering_axis; front_wheel->mWheelForward = JPH::Vec3(0, 1, 0);
		front_wheel->mSuspensionMinLength = 0.1f; front_wheel->mSuspensionMaxLength = 0.35f; front_wheel->mSuspensionSpring.mFrequency = 2.0f;
		front_wheel->mRadius = wheel_radius; front_wheel->mWidth = wheel_width; front_wheel->mMaxSteerAngle = JPH::DegreesToRadians(30);
		front_wheel->mMaxHandBrakeTorque = 40000.f; front_wheel->mMaxBrakeTorque = 500.f; front_wheel->mInertia = 0.63f;
		JPH::WheelSettingsWV* rear_wheel = new JPH::WheelSettingsWV;
		rear_wheel->mPosition = JPH::Vec3(0, -0.88f, 0.15f);
		rear_wheel->mSuspensionDirection = JPH::Vec3(0, 0, -1); rear_wheel->mSteeringAxis = JPH::Vec3(0, 0, 1); rear_wheel->mWheelUp = JPH::Vec3(0, 0, 1); rear_wheel->mWheelForward = JPH::Vec3(0, 1, 0);
		rear_wheel->mSuspensionMinLength = 0.1f; rear_wheel->mSuspensionMaxLength = 0.3f; rear_wheel->mSuspensionSpring.mFrequency = 2.5f;
		rear_wheel->mRadius = wheel_radius; rear_wheel->mWidth = wheel_width; rear_wheel->mMaxSteerAngle = 0.f;
		rear_wheel->mMaxHandBrakeTorque = 0; rear_wheel->mMaxBrakeTorque = 700.f;
		vehicle.mWheels = { front_wheel, rear_wheel };
		for (const JPH::Ref<JPH::WheelSettings>& w : vehicle.mWheels) {
			JPH::WheelSettingsWV* wv = dynamic_cast<JPH::WheelSettingsWV*>(w.GetPtr());
			wv->mLongitudinalFriction.mPoints[0].mY = 15; wv->mLongitudinalFriction.mPoints[1].mY = 8; wv->mLongitudinalFriction.mPoints[2].mY = 3;
			wv->mLateralFriction.mPoints[0].mY *= 5.f; wv->mLateralFriction.mPoints[1].mY *= 3.f; wv->mLateralFriction.mPoints[2].mY *= 2.f;
		}
		JPH::MotorcycleControllerSettings* controller_settings = new JPH::MotorcycleControllerSettings();
		vehicle.mController = controller_settings;
		controller_settings->mLeanSpringConstant = 2000.f;
		controller_settings->mLeanSpringIntegrationCoefficient = 2000.f;
		controller_settings->mLeanSpringDamping = 500.f;
		controller_settings->mLeanSmoothingFactor = 0.9f;
		controller_settings->mMaxLeanAngle = JPH::DegreesToRadians(60.f);
		controller_settings->mDifferentials.resize(1);
		controller_settings->mDifferentials[0].mLeftWheel = 0;
		controller_settings->mDifferentials[0].mRightWheel = 1;
		controller_settings->mDifferentials[0].mLeftRightSplit = 1.f;
		controller_settings->mEngine.mMaxTorque = 390;
		controller_settings->mEngine.mMaxRPM = 10000;
		controller_settings->mEngine.mInertia = 0.2f;
		controller_settings->mTransmission.mShiftDownRPM = 5000.0f;
		controller_settings->mTransmission.mShiftUpRPM = 9000.0f;
		controller_settings->mTransmission.mGearRatios = { 2.27f, 1.63f, 1.3f, 1.09f, 0.96f, 0.88f };
		controller_settings->mTransmission.mSwitchTime = 0.2f;

		const JPH::Body bike_body = world->getJoltBody(*bike);
		JPH::Ref<JPH::VehicleConstraint> vehicle_constraint = new JPH::VehicleConstraint(bike_body, vehicle);
		world->physics_system->AddConstraint(vehicle_constraint);
		world->physics_system->AddStepListener(vehicle_constraint.GetPtr());
		JPH::Ref<JPH::VehicleCollisionTester> collision_tester = new JPH::VehicleCollisionTesterCastCylinder(/*Layers::MOVING*/1, 1.f);
		vehicle_constraint->SetVehicleCollisionTester(collision_tester);     // (before or after AddConstraint in the reference: here it must precede it)

		// the tester was set after AddConstraint, as in BikePhysics.cpp:222-228: re-register so the cast radius takes effect
		world->physics_system->RemoveConstraint(vehicle_constraint);
		world->physics_system->AddConstraint(vehicle_constraint);

		JPH::BodyInterface& body_interface = world->physics_system->GetBodyInterface();
		JPH::WheeledVehicleController* controller = static_cast<JPH::WheeledVehicleController*>(vehicle_constraint->GetController());
		float max_abs_roll_straight = 0.f, roll_in_turn = 0.f;
		for (int st = 0; st < 600; ++st) {
			const float forward = st > 30 ? 0.5f : 0.f, right = (st >= 300 && st < 480) ? 0.5f : 0.f;
			if (right != 0.f || forward != 0.f) body_interface.ActivateBody(bike->jolt_body_id);
			controller->SetDriverInput(forward, right, 0.f, 0.f);
			static_cast<JPH::MotorcycleController*>(vehicle_constraint->GetController())->EnableLeanController(true);
			world->think(1.0 / 60.0);
			const JPH::Mat44 t = body_interface.GetWorldTransform(bike->jolt_body_id);
			const JPH::Vec3 up = t.GetAxisZ(), fw = t.GetAxisY();
			// sin(lean) as BikePhysics computes it for its brake multiplier (:430-434), signed towards the right
			const float sin_lean = (up.x * fw.y - up.y * fw.x);
			if (st < 300) max_abs_roll_straight = std::fmax(max_abs_roll_straight, std::fabs(sin_lean));
			if (st == 450) roll_in_turn = sin_lean;
		}
		const JPH::RVec3 p = body_interface.GetPosition(bike->jolt_body_id);
		printf("pos %.2f %.2f %.2f  max|sin lean| straight %.4f  sin lean in turn %.3f  gear %d  steer %.4f\n", p.GetX(), p.GetY(), p.GetZ(), max_abs_roll_straight,
			roll_in_turn, controller->GetCurrentGear(), vehicle_constraint->GetWheel(0)->GetSteerAngle());
		const bool ok = max_abs_roll_straight < 0.05f && roll_in_turn > 0.25f && p.GetX() > 20.f && p.GetZ() > 0.3f && controller->GetCurrentGear() >= 2;
		world->physics_system->RemoveConstraint(vehicle_constraint);
		return ok ? 0 : 1;
	} catch (glare::Exception& e) { fprintf(stderr, "glare::Exception: %s\n", e.what().c_str()); return 2; }
}
