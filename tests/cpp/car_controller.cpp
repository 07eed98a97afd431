// A CarPhysics-shaped caller (gui_client/CarPhysics.cpp:62-231,299-470): builds the VehicleConstraintSettings exactly the way
// CarPhysics does (four WheelSettingsWV, front-wheel-drive differential, two anti-roll bars, engine torque / max RPM, a
// VehicleCollisionTesterCastSphere of half the wheel width), registers the constraint with the physics system, then every
// sub-step passes driver input to the WheeledVehicleController, steps PhysicsWorld::think() and reads the wheels back.
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

int main()
{
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
		Reference<PhysicsObject> ground = new PhysicsObject(true, PhysicsWorld::createGroundQuadShape(2000.f), nullptr, 0);
		ground->pos = Vec4f(0, 0, -0.5f, 1);
		world->addObject(ground);
		// chassis: the default convex hull of the car script (Scripting.cpp:369-386), here already in z-up / y-forward space
		const float x_half_w = 0.9f, up_half_w = 0.25f, fwd_half_w = 2.0f;
		std::vector<Vec3f> convex_hull_pts;
		for (int sx = -1; sx <= 1; sx += 2) for (int su = -1; su <= 1; su += 2) for (int sf = -1; sf <= 1; sf += 2)
			convex_hull_pts.push_back(Vec3f(sx * x_half_w, sf * fwd_half_w, su * up_half_w));
		convex_hull_pts.push_back(Vec3f(x_half_w, 0.6f, 0.7f)); convex_hull_pts.push_back(Vec3f(-x_half_w, 0.6f, 0.7f));     // roof
		convex_hull_pts.push_back(Vec3f(x_half_w, -1.2f, 0.7f)); convex_hull_pts.push_back(Vec3f(-x_half_w, -1.2f, 0.7f));
		// CarPhysics.cpp:76-78: the hull wrapped in an OffsetCenterOfMassShape (object->centre_of_mass_offset_os; here 0.2 m down)
		const PhysicsShape car_body_shape = PhysicsWorld::createCOMOffsetShapeForShape(PhysicsWorld::createConvexHullShape(convex_hull_pts), Vec4f(0, 0, -0.2f, 0));
		Reference<PhysicsObject> car = new PhysicsObject(true, car_body_shape, nullptr, 0);
		car->pos = Vec4f(0, 0, 0.8f, 1); car->mass = 1200.f; car->restitution = 0.f;
		car->motion_type = PhysicsObject::MotionType_dynamic;
		world->addObject(car);
		world->activateObject(car);

		// script defaults (Scripting.cpp:315-348)
		const float wheel_radius = 0.42f, wheel_width = 0.16f, sus_min = 0.2f, sus_max = 0.5f, raise = 0.2f;
		JPH::Ref<JPH::VehicleCollisionTester> tester = new JPH::VehicleCollisionTesterCastSphere(/*Layers::MOVING*/1, 0.5f * wheel_width, JPH::Vec3(0, 0, 1));

		JPH::VehicleConstraintSettings vehicle;
		vehicle.mUp = JPH::Vec3(0, 0, 1);
		vehicle.mForward = JPH::Vec3(0, 1, 0);
		const JPH::Vec3 joint[4] = { JPH::Vec3(-0.8f, 1.3f, -0.25f), JPH::Vec3(0.8f, 1.3f, -0.25f), JPH::Vec3(-0.8f, -1.3f, -0.25f), JPH::Vec3(0.8f, -1.3f, -0.25f) };
		for (int i = 0; i < 4; ++i) {
			JPH::WheelSettingsWV* w = new JPH::WheelSettingsWV;
			w->mPosition = joint[i] + JPH::Vec3(0, 0, sus_min + raise);
			w->mSuspensionDirection = JPH::Vec3(0, 0, -1);
			w->mSteeringAxis = JPH::Vec3(0, 0, 1);
			w->mWheelUp = JPH::Vec3(0, 0, 1);
			w->mWheelForward = JPH::Vec3(0, 1, 0);
			w->mWidth = wheel_width; w->mRadius = wheel_radius;
			w->mSuspensionMinLength = sus_min; w->mSuspensionMaxLength = sus_max;
			w->mSuspensionSpring.mFrequency = 2.0f; w->mSuspensionSpring.mDamping = 0.5f;
			w->mMaxSteerAngle = (i < 2) ? 0.78525f : 0.0f;
			w->mMaxBrakeTorque = 1500.f;
			w->mMaxHandBrakeTorque = (i < 2) ? 0.0f : 4000.f;
			vehicle.mWheels.push_back(w);
		}
		JPH::WheeledVehicleControllerSettings* controller_settings = new JPH::WheeledVehicleControllerSettings;
		vehicle.mController = controller_settings;
		controller_settings->mDifferentials.resize(1);
		controller_settings->mDifferentials[0].mLeftWheel = 0;
		controller_settings->mDifferentials[0].mRightWheel = 1;
		controller_settings->mEngine.mMaxTorque = 500.f;
		controller_settings->mEngine.mMaxRPM = 6000.f;
		vehicle.mAntiRollBars.resize(2);
		vehicle.mAntiRollBars[0].mLeftWheel = 0; vehicle.mAntiRollBars[0].mRightWheel = 1;
		vehicle.mAntiRollBars[1].mLeftWheel = 2; vehicle.mAntiRollBars[1].mRightWheel = 3;

		const JPH::Body chassis_body = world->getJoltBody(*car);           // (CarPhysics: jolt_body = body_interface.CreateBody(...), :84)
		JPH::Ref<JPH::VehicleConstraint> vehicle_constraint = new JPH::VehicleConstraint(chassis_body, vehicle);
		vehicle_constraint->SetVehicleCollisionTester(tester);
		world->physics_system->AddConstraint(vehicle_constraint);
		world->physics_system->AddStepListener(vehicle_constraint.GetPtr());

		JPH::BodyInterface& body_interface = world->physics_system->GetBodyInterface();
		JPH::WheeledVehicleController* controller = static_cast<JPH::WheeledVehicleController*>(vehicle_constraint->GetController());
		float cur_steering_right = 0.f;
		int skid_frames = 0;
		for (int s = 0; s < 480; ++s) {
			const float forward = (s >= 60 && s < 360) ? 1.f : 0.f;
			const float brake = (s >= 360) ? 1.f : 0.f;
			if (s >= 180 && s < 300) cur_steering_right = std::fmin(cur_steering_right + 3.f / 60.f, 1.f);
			else cur_steering_right = std::fmax(cur_steering_right - 3.f / 60.f, 0.f);
			if (cur_steering_right != 0.f || forward != 0.f || brake != 0.f) body_interface.ActivateBody(car->jolt_body_id);
			controller->SetDriverInput(forward, cur_steering_right, brake, 0.f);
			world->think(1.0 / 60.0);
			// tyre-squeal logic of CarPhysics::update (:405-424)
			for (int i = 0; i < 4; ++i) {
				const JPH::Wheel* wheel = vehicle_constraint->GetWheel(i);
				if (wheel->HasContact()) {
					JPH::Vec3 rel = body_interface.GetPointVelocity(car->jolt_body_id, wheel->GetContactPosition()) - wheel->GetContactPointVelocity();
					const JPH::Vec3 n = wheel->GetContactNormal();
					const float vn = rel.x * n.x + rel.y * n.y + rel.z * n.z;
					rel = rel - n * vn;
					const JPH::Vec3 lg = wheel->GetContactLongitudinal();
					const float rel_long = rel.x * lg.x + rel.y * lg.y + rel.z * lg.z;
					if (std::fabs(wheel->GetAngularVelocity() * wheel_radius - rel_long) > 1.f && i == 0) ++skid_frames;
				}
			}
		}
		JPH::RVec3 p; JPH::Quat q;
		body_interface.GetPositionAndRotation(car->jolt_body_id, p, q);
		world->readBackActivatedObjectTransforms();
		const JPH::Vec3 v = body_interface.GetLinearVelocity(car->jolt_body_id);
		const float yaw = 2.f * std::atan2(q.GetZ(), q.GetW());
		const JPH::Wheel* w0 = vehicle_constraint->GetWheel(0);
		const JPH::Mat44 wt = vehicle_constraint->GetWheelLocalTransform(0, JPH::Vec3(1, 0, 0), JPH::Vec3(0, 0, 1));
		printf("pos %.3f %.3f %.3f  speed %.3f  yaw %.3f  rpm %.0f gear %d  sus %.3f contact %d skid_frames %d wheel_z %.3f\n", p.GetX(), p.GetY(), p.GetZ(),
			std::sqrt(v.LengthSq()), yaw, controller->GetEngine().GetCurrentRPM(), controller->GetCurrentGear(), w0->GetSuspensionLength(), (int)w0->HasContact(), skid_frames, wt.GetTranslation().GetZ());
		bool ok = std::sqrt(p.GetX() * p.GetX() + p.GetY() * p.GetY()) > 5.f && p.GetY() > 2.f && p.GetX() > 0.5f   // drove off and turned right (+x)
			&& std::fabs(yaw) > 0.2f                          // (full lock for two seconds: it comes round a long way)
			&& std::sqrt(v.LengthSq()) < 0.3f                // braked to a stop
			&& std::fabs(car->pos[2] - 0.70f) < 0.05f && w0->HasContact()   // object origin (hull space), read back by the facade
			&& w0->GetSuspensionLength() > sus_min && w0->GetSuspensionLength() < sus_max
			&& skid_frames > 10;                             // the front wheels spun on launch
		// vehicleSummoned() (:266-272)
		controller->GetEngine().SetCurrentRPM(0);
		vehicle_constraint->GetWheel(0)->SetAngularVelocity(0);
		ok = ok && controller->GetEngine().GetCurrentRPM() == 0.f;
		world->physics_system->RemoveConstraint(vehicle_constraint);
		world->physics_system->RemoveStepListener(vehicle_constraint.GetPtr());
		world->think(1.0 / 60.0);
		return ok ? 0 : 1;
	} catch (glare::Exception& e) { fprintf(stderr, "glare::Exception: %s\n", e.what().c_str()); return 2; }
}
