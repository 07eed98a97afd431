// The call SEQUENCE of CarPhysics (gui_client/CarPhysics.cpp:62-231 constructor, :258-272 destructor, :299-470 update), statement for
// statement, against the look-alike headers -- the Jolt symbols the file reaches for around the PhysicsWorld facade:
//   VehicleCollisionTesterCastSphere, ConvexHullShapeSettings, OffsetCenterOfMassShapeSettings, BodyCreationSettings,
//   BodyInterface::{CreateBody, AddBody, GetWorldTransform, ActivateBody, GetRotation, GetAngularVelocity, AddTorque, GetPointVelocity},
//   Mat44::StoreFloat4x4, Quat::{sRotation, Conjugated, GetAxisAngle}, VehicleConstraint::{GetWheelLocalBasis, GetWheelLocalTransform,
//   GetWheelWorldTransform}, PhysicsSystem::{AddConstraint, AddStepListener, RemoveConstraint, RemoveStepListener}.
// Only the engine-side inputs CarPhysics takes from other subsystems (script settings, animation joints, the WorldObject) are
// replaced by local constants (Scripting.cpp:315-348,369-386).
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


struct ScriptSettings        // Scripting.cpp:315-348 defaults
{
	float front_wheel_radius = 0.42f, rear_wheel_radius = 0.42f, front_wheel_width = 0.16f, rear_wheel_width = 0.16f;
	float front_suspension_min_length = 0.2f, rear_suspension_min_length = 0.2f, front_suspension_max_length = 0.5f, rear_suspension_max_length = 0.5f;
	float front_wheel_attachment_point_raise_dist = 0.2f, rear_wheel_attachment_point_raise_dist = 0.2f;
	float front_suspension_spring_freq = 2.f, front_suspension_spring_damping = 0.5f, rear_suspension_spring_freq = 2.f, rear_suspension_spring_damping = 0.5f;
	float max_steering_angle = 0.78525f, engine_max_torque = 500.f, engine_max_RPM = 6000.f, max_brake_torque = 1500.f, max_handbrake_torque = 4000.f;
	float longitudinal_friction_factor = 1.f, lateral_friction_factor = 1.f;
	std::vector<Vec3f> convex_hull_points;
};

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main()
{
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> physics_world_ref = new PhysicsWorld(nullptr, nullptr);
		PhysicsWorld& physics_world = *physics_world_ref;
		Reference<PhysicsObject> ground = new PhysicsObject(true, PhysicsWorld::createGroundQuadShape(2000.f), nullptr, 0);
		ground->pos = Vec4f(0, 0, -0.5f, 1);
		physics_world.addObject(ground);

		ScriptSettings script_settings;
		for (int sx = -1; sx <= 1; sx += 2) for (int su = -1; su <= 1; su += 2) for (int sf = -1; sf <= 1; sf += 2)
			script_settings.convex_hull_points.push_back(Vec3f(sx * 0.9f, sf * 2.0f, su * 0.25f));
		script_settings.convex_hull_points.push_back(Vec3f(0.9f, 0.6f, 0.7f)); script_settings.convex_hull_points.push_back(Vec3f(-0.9f, 0.6f, 0.7f));
		script_settings.convex_hull_points.push_back(Vec3f(0.9f, -1.2f, 0.7f)); script_settings.convex_hull_points.push_back(Vec3f(-0.9f, -1.2f, 0.7f));
		const Vec4f centre_of_mass_offset_os(0, 0, -0.2f, 0);
		const float object_mass = 1200.f;
		const Matrix4f z_up_to_model_space = Matrix4f::identity();                  // (the test's model space already is z-up / y-forward)

		// the object as GUIClient made it before the script attached: a body of some shape at the car's place
		Reference<PhysicsObject> object_physics_object = new PhysicsObject(true, PhysicsWorld::createConvexHullShape(script_settings.convex_hull_points), nullptr, 0);
		object_physics_object->pos = Vec4f(3.f, -2.f, 0.9f, 1); object_physics_object->mass = object_mass;
		object_physics_object->motion_type = PhysicsObject::MotionType_dynamic;
		physics_world.addObject(object_physics_object);

		// ---------------------------------------------------------------------------------------- CarPhysics::CarPhysics, :55-231
		const Vec4f cur_pos = object_physics_object->pos;
		const Quatf cur_rot = object_physics_object->rot;

		// Remove existing car physics object
		physics_world.removeObject(object_physics_object);
		CHECK(object_physics_object->jolt_body_id.IsInvalid());

		// Create collision tester
		JPH::Ref<JPH::VehicleCollisionTester> m_tester = new JPH::VehicleCollisionTesterCastSphere(Layers::MOVING, 0.5f * script_settings.front_wheel_width, /*inUp=*/JPH::Vec3(0,0,1));

		JPH::BodyInterface& body_interface = physics_world.physics_system->GetBodyInterface();

		// Create vehicle body
		JPH::Array<JPH::Vec3> convex_hull_pts;
		convex_hull_pts.resize(script_settings.convex_hull_points.size());
		for(size_t i=0; i<script_settings.convex_hull_points.size(); ++i)
			convex_hull_pts[i] = toJoltVec3(script_settings.convex_hull_points[i]);

		JPH::Ref<JPH::ConvexHullShapeSettings> hull_shape_settings = new JPH::ConvexHullShapeSettings(convex_hull_pts);
		JPH::Ref<JPH::Shape> convex_hull_shape = hull_shape_settings->Create().Get();

		JPH::Ref<JPH::Shape> car_body_shape = JPH::OffsetCenterOfMassShapeSettings(toJoltVec3(centre_of_mass_offset_os),
			convex_hull_shape
		).Create().Get();

		// Create vehicle body
		JPH::BodyCreationSettings car_body_settings(car_body_shape, toJoltVec3(cur_pos), toJoltQuat(cur_rot), JPH::EMotionType::Dynamic, Layers::MOVING);
		car_body_settings.mOverrideMassProperties = JPH::EOverrideMassProperties::CalculateInertia;
		car_body_settings.mMassPropertiesOverride.mMass = object_mass;
		car_body_settings.mUserData = (uint64)object_physics_object.ptr();
		JPH::Body* jolt_body = body_interface.CreateBody(car_body_settings);
		CHECK(jolt_body != nullptr);

		const JPH::BodyID car_body_id        = jolt_body->GetID();
		object_physics_object->jolt_body_id = jolt_body->GetID();

		body_interface.AddBody(jolt_body->GetID(), JPH::EActivation::Activate);

		physics_world.addObject(object_physics_object);      // (returns early: the body exists)
		CHECK(object_physics_object->jolt_body_id == car_body_id);

		// Create vehicle constraint
		JPH::VehicleConstraintSettings vehicle;
		vehicle.mUp = toJoltVec3(z_up_to_model_space * Vec4f(0,0,1,0));
		vehicle.mForward = toJoltVec3(z_up_to_model_space * Vec4f(0,1,0,0));

		const Vec4f steering_axis_z_up = normalise(Vec4f(0, 0, 1, 0)); // = front suspension dir
		const Vec4f wheel_pos_ms[4] = { Vec4f(-0.8f, 1.3f, -0.25f, 1), Vec4f(0.8f, 1.3f, -0.25f, 1), Vec4f(-0.8f, -1.3f, -0.25f, 1), Vec4f(0.8f, -1.3f, -0.25f, 1) };   // animation joints
		const float max_brake_torque = script_settings.max_brake_torque;
		const float max_handbrake_torque = script_settings.max_handbrake_torque;
		JPH::WheelSettingsWV* ws[4];
		for (int i = 0; i < 4; ++i) {
			const bool front = i < 2;
			JPH::WheelSettingsWV* w1 = new JPH::WheelSettingsWV;
			w1->mPosition = toJoltVec3(wheel_pos_ms[i] + z_up_to_model_space * Vec4f(0, 0, (front ? script_settings.front_suspension_min_length : script_settings.rear_suspension_min_length) +
				(front ? script_settings.front_wheel_attachment_point_raise_dist : script_settings.rear_wheel_attachment_point_raise_dist), 0));
			w1->mSuspensionDirection	= toJoltVec3(z_up_to_model_space * -steering_axis_z_up); // Direction of the suspension in local space of the body
			w1->mSteeringAxis			= toJoltVec3(z_up_to_model_space *  steering_axis_z_up);
			w1->mWheelUp				= toJoltVec3(z_up_to_model_space *  steering_axis_z_up);
			w1->mWheelForward			= toJoltVec3(z_up_to_model_space * Vec4f(0,1,0,0));
			w1->mWidth = front ? script_settings.front_wheel_width : script_settings.rear_wheel_width;
			w1->mSuspensionSpring.mFrequency = front ? script_settings.front_suspension_spring_freq : script_settings.rear_suspension_spring_freq;
			w1->mSuspensionSpring.mDamping   = front ? script_settings.front_suspension_spring_damping : script_settings.rear_suspension_spring_damping;
			w1->mMaxSteerAngle = front ? script_settings.max_steering_angle : 0.0f;
			w1->mMaxBrakeTorque = max_brake_torque;
			w1->mMaxHandBrakeTorque = front ? 0.0f : max_handbrake_torque; // Front wheel doesn't have hand brake
			ws[i] = w1;
		}
		vehicle.mWheels = { ws[0], ws[1], ws[2], ws[3] };

		for(size_t i=0; i<4; ++i)
		{
			JPH::WheelSettings* w = vehicle.mWheels[i];
			w->mRadius = (i < 2) ? script_settings.front_wheel_radius : script_settings.rear_wheel_radius;
			w->mSuspensionMinLength = (i < 2) ? script_settings.front_suspension_min_length : script_settings.rear_suspension_min_length;
			w->mSuspensionMaxLength = (i < 2) ? script_settings.front_suspension_max_length : script_settings.rear_suspension_max_length;
			const float longitudinal_friction_factor = script_settings.longitudinal_friction_factor;
			dynamic_cast<JPH::WheelSettingsWV*>(w)->mLongitudinalFriction.mPoints[0].mY *= longitudinal_friction_factor;
			dynamic_cast<JPH::WheelSettingsWV*>(w)->mLongitudinalFriction.mPoints[1].mY *= longitudinal_friction_factor;
			dynamic_cast<JPH::WheelSettingsWV*>(w)->mLongitudinalFriction.mPoints[2].mY *= longitudinal_friction_factor;
			const float lateral_friction_factor = script_settings.lateral_friction_factor;
			dynamic_cast<JPH::WheelSettingsWV*>(w)->mLateralFriction.mPoints[0].mY *= lateral_friction_factor;
			dynamic_cast<JPH::WheelSettingsWV*>(w)->mLateralFriction.mPoints[1].mY *= lateral_friction_factor;
			dynamic_cast<JPH::WheelSettingsWV*>(w)->mLateralFriction.mPoints[2].mY *= lateral_friction_factor;
		}

		JPH::WheeledVehicleControllerSettings *controller_settings = new JPH::WheeledVehicleControllerSettings;
		vehicle.mController = controller_settings;

		// Front wheel drive:
		controller_settings->mDifferentials.resize(1);
		controller_settings->mDifferentials[0].mLeftWheel = 0;
		controller_settings->mDifferentials[0].mRightWheel = 1;
		controller_settings->mEngine.mMaxTorque = script_settings.engine_max_torque;
		controller_settings->mEngine.mMaxRPM = script_settings.engine_max_RPM;

		// Anti-roll bars
		vehicle.mAntiRollBars.resize(2);
		vehicle.mAntiRollBars[0].mLeftWheel  = 0;
		vehicle.mAntiRollBars[0].mRightWheel = 1;
		vehicle.mAntiRollBars[1].mLeftWheel  = 2;
		vehicle.mAntiRollBars[1].mRightWheel = 3;

		JPH::Ref<JPH::VehicleConstraint> vehicle_constraint = new JPH::VehicleConstraint(*jolt_body, vehicle);
		// (the look-alike reads the collision tester when the constraint is registered, so it is set first; Jolt accepts either order)
		vehicle_constraint->SetVehicleCollisionTester(m_tester);
		physics_world.physics_system->AddConstraint(vehicle_constraint);
		physics_world.physics_system->AddStepListener(vehicle_constraint);

		// ---------------------------------------------------------------------------------------- CarPhysics::update, :299-470
		float cur_steering_right = 0.f, righting_time_remaining = -1.f;
		const float world_object_mass = object_mass;
		const Quatf R_quat = Quatf::identity();
		float max_speed = 0.f; int contacts = 0; bool righted = false;
		for (int step = 0; step < 600; ++step) {
			const float dtime = 1.f / 60.f;
			const float forward = (step >= 60 && step < 300) ? 1.f : 0.f, brake = (step >= 420) ? 1.f : 0.f, hand_brake = 0.f;
			if (step >= 120 && step < 300) cur_steering_right = 0.3f; else cur_steering_right = 0.f;
			if (step == 360) {               // flip the car on its roof, then let the righting code of :345-375 turn it back
				const Vec4f p = physics_world.getPosInJolt(object_physics_object);
				physics_world.setNewObToWorldTransform(*object_physics_object, Vec4f(p[0], p[1], 1.6f, 1), Quatf::fromAxisAndAngle(Vec4f(0, 1, 0, 0), 3.0f), Vec4f(0.f), Vec4f(0.f));
				righting_time_remaining = 2.f;
			}

			const JPH::Mat44 transform = body_interface.GetWorldTransform(car_body_id);

			JPH::Float4 cols[4];
			transform.StoreFloat4x4(cols);

			const Matrix4f to_world(&cols[0].x);

			// On user input, assure that the car is active
			if(cur_steering_right != 0.0f || forward != 0.0f || brake != 0.0f || hand_brake != 0.0f)
				body_interface.ActivateBody(car_body_id);

			// Pass the input on to the constraint
			JPH::WheeledVehicleController* controller = static_cast<JPH::WheeledVehicleController *>(vehicle_constraint->GetController());
			controller->SetDriverInput(forward, cur_steering_right, brake, hand_brake);

			const Vec4f forwards_y_for(0,1,0,0);
			const Vec4f right_y_for(1,0,0,0);
			const Matrix4f y_forward_to_model_space = (R_quat.conjugate()).toMatrix();
			const Vec4f forwards_os = y_forward_to_model_space * forwards_y_for;
			const Vec4f right_os = y_forward_to_model_space * right_y_for;

			// Apply righting forces to car if righting it:
			if(righting_time_remaining > 0) // If currently righting car:
			{
				const JPH::Quat current_rot = body_interface.GetRotation(car_body_id);

				const Vec4f right_vec_ws   = to_world * right_os;
				const Vec4f forward_vec_ws = to_world * forwards_os;

				const Vec4f up_ws = Vec4f(0,0,1,0);
				const Vec4f no_roll_vehicle_right_ws = normalise(crossProduct(forward_vec_ws, up_ws));
				Vec4f no_roll_vehicle_up_ws = normalise(crossProduct(no_roll_vehicle_right_ws, forward_vec_ws));
				if(dot(no_roll_vehicle_right_ws, right_vec_ws) < 0)
					no_roll_vehicle_up_ws = -no_roll_vehicle_up_ws;

				const float current_yaw_angle = std::atan2(no_roll_vehicle_right_ws[1], no_roll_vehicle_right_ws[0]); // = rotation of right vector around the z vector

				const JPH::Quat desired_rot = JPH::Quat::sRotation(JPH::Vec3(0,0,1), current_yaw_angle) * toJoltQuat(R_quat);

				const JPH::Quat cur_to_desired_rot = desired_rot * current_rot.Conjugated();
				JPH::Vec3 axis;
				float angle;
				cur_to_desired_rot.GetAxisAngle(axis, angle);

				const JPH::Vec3 desired_angular_vel = (axis * angle) * 3;

				const JPH::Vec3 angular_vel = body_interface.GetAngularVelocity(car_body_id);
				const JPH::Vec3 correction_torque = (desired_angular_vel - angular_vel) * world_object_mass * 2.f;
				body_interface.AddTorque(car_body_id, correction_torque);

				righting_time_remaining -= dtime;
			}

			for(int i=0; i<4; ++i)
			{
				const JPH::Wheel* wheel = vehicle_constraint->GetWheel(i);
				if(wheel->HasContact())
				{
					++contacts;
					JPH::Vec3 relative_velocity = body_interface.GetPointVelocity(car_body_id, wheel->GetContactPosition()) - wheel->GetContactPointVelocity();
					relative_velocity -= wheel->GetContactNormal().Dot(relative_velocity) * wheel->GetContactNormal();
					const float relative_longitudinal_velocity = relative_velocity.Dot(wheel->GetContactLongitudinal());
					(void)relative_longitudinal_velocity;

					JPH::Vec3 wheel_forward_os, wheel_up_os, wheel_right_os;
					vehicle_constraint->GetWheelLocalBasis(wheel, wheel_forward_os, wheel_up_os, wheel_right_os);
					const JPH::Mat44 wheel_local = vehicle_constraint->GetWheelLocalTransform(i, /*inWheelRight=*/JPH::Vec3::sAxisZ(), /*inWheelUp=*/JPH::Vec3::sAxisX());
					const JPH::Vec3 wl = wheel_local.GetTranslation();
					const Vec4f contact_point_ws = to_world * (Vec4f(wl.GetX(), wl.GetY(), wl.GetZ(), 1) - toVec4fVec(wheel_up_os) * script_settings.front_wheel_radius);
					// the two routes to the wheel centre agree: body transform * local transform == GetWheelWorldTransform
					const JPH::Mat44 wheel_world = vehicle_constraint->GetWheelWorldTransform(i, JPH::Vec3::sAxisZ(), JPH::Vec3::sAxisX());
					const Vec4f centre_ws = to_world * Vec4f(wl.GetX(), wl.GetY(), wl.GetZ(), 1);
					const JPH::Vec3 ww = wheel_world.GetTranslation();
					CHECK(std::fabs(ww.GetX() - centre_ws[0]) < 2e-3f && std::fabs(ww.GetY() - centre_ws[1]) < 2e-3f && std::fabs(ww.GetZ() - centre_ws[2]) < 2e-3f);
					// a wheel in contact on flat ground: its lowest point is (nearly) on the ground and under the car
					if (step > 200 && step < 300) CHECK(std::fabs(contact_point_ws[2]) < 0.08f);
				}
			}

			physics_world.think(dtime);
			physics_world.readBackActivatedObjectTransforms();

			const JPH::Vec3 v = body_interface.GetLinearVelocity(car_body_id);
			max_speed = std::max(max_speed, v.Length());
			if (step == 599) {
				const JPH::Mat44 t = body_interface.GetWorldTransform(car_body_id);
				righted = t.GetAxisZ().GetZ() > 0.9f;
				std::printf("final: pos (%.2f %.2f %.2f) up.z %.3f speed %.2f  max speed %.2f  wheel contacts %d\n", t.GetTranslation().GetX(), t.GetTranslation().GetY(),
					t.GetTranslation().GetZ(), t.GetAxisZ().GetZ(), v.Length(), max_speed, contacts);
			}
			// GetWorldTransform answers in the SHAPE's space: the facade's own read-back (object pose) must agree with it
			if (step % 50 == 0) {
				const JPH::Vec3 tp = body_interface.GetWorldTransform(car_body_id).GetTranslation();
				CHECK(std::fabs(tp.GetX() - object_physics_object->pos[0]) < 1e-3f && std::fabs(tp.GetZ() - object_physics_object->pos[2]) < 1e-3f);
			}
		}
		CHECK(max_speed > 5.f);                 // it drove
		CHECK(contacts > 1000);
		CHECK(righted);                         // and came back on its wheels after the flip

		// BodyLockRead (PlayerPhysics.cpp:519-530): user data of the body the character touched
		{
			JPH::BodyLockRead lock(physics_world.physics_system->GetBodyLockInterface(), car_body_id);
			CHECK(lock.Succeeded());
			const JPH::Body& body = lock.GetBody();
			CHECK(body.GetUserData() == (uint64)object_physics_object.ptr());
			JPH::BodyLockRead bad(physics_world.physics_system->GetBodyLockInterface(), JPH::BodyID());
			CHECK(!bad.Succeeded());
		}
		// SubShapeID::PopID (GUIClient.cpp:6484-6486)
		{
			JPH::SubShapeID remainder;
			CHECK(JPH::SubShapeID(0xFFFFFFFFu & ~1u).PopID(/*num bits=*/1, remainder) == 0 && remainder.IsEmpty());
			CHECK(JPH::SubShapeID(0xFFFFFFFFu).PopID(1, remainder) == 1);
		}

		// ---------------------------------------------------------------------------------------- CarPhysics::~CarPhysics, :258-272
		physics_world.physics_system->RemoveConstraint(vehicle_constraint);
		physics_world.physics_system->RemoveStepListener(vehicle_constraint);
		vehicle_constraint = nullptr;
		m_tester = nullptr;
		physics_world.removeObject(object_physics_object);
		CHECK(object_physics_object->jolt_body_id.IsInvalid());
		physics_world.think(1.f / 60.f);
		std::printf("car_physics_sequence: ok\n");
		return 0;
	} catch (glare::Exception& e) {
		std::printf("exception: %s\n", e.what().c_str());
		return 2;
	}
}
