// JPH vehicle look-alike: the Jolt vehicle types CarPhysics names (gui_client/CarPhysics.cpp:62,94-231,258-272,328-470),
// backed by sgp_vehicle_* (include/sgp.h).  Same class and member names, Jolt's default values; no Jolt code.
//   VehicleConstraintSettings / WheelSettingsWV / WheeledVehicleControllerSettings  -> sgp_vehicle_desc
//   PhysicsSystem::AddConstraint(VehicleConstraint*)                                  -> sgp_vehicle_create
//   WheeledVehicleController::SetDriverInput                                          -> sgp_vehicle_set_input
//   Wheel getters, VehicleEngine::GetCurrentRPM                                       -> sgp_vehicle_get_state (one read-back per step)
// Differences a maintainer must know: VehicleCollisionTesterCastCylinder casts the wheel as Jolt does for inConvexRadiusFraction = 1 (BikePhysics' value) whatever fraction is passed.
#pragma once
#include "JoltLite.h"
#include "../../../include/sgp.h"
#include <cmath>
#include <vector>
#include <memory>

namespace JPH
{
	typedef uint16_t ObjectLayer;


	class LinearCurve
	{
	public:
		struct Point { float mX, mY; };
		void AddPoint(float x, float y) { mPoints.push_back(Point{ x, y }); }
		std::vector<Point> mPoints;
	};

	struct SpringSettings { float mFrequency = 1.5f, mDamping = 0.5f; };

	class WheelSettings : public RefTargetBase
	{
	public:
		Vec3 mPosition = Vec3(0, 0, 0);
		Vec3 mSuspensionDirection = Vec3(0, -1, 0), mSteeringAxis = Vec3(0, 1, 0), mWheelUp = Vec3(0, 1, 0), mWheelForward = Vec3(0, 0, 1);
		float mSuspensionMinLength = 0.3f, mSuspensionMaxLength = 0.5f, mSuspensionPreloadLength = 0.0f;
		SpringSettings mSuspensionSpring;
		float mRadius = 0.3f, mWidth = 0.1f;
	};
	class WheelSettingsWV : public WheelSettings
	{
	public:
		WheelSettingsWV()
		{
			mLongitudinalFriction.AddPoint(0.0f, 0.0f); mLongitudinalFriction.AddPoint(0.06f, 1.2f); mLongitudinalFriction.AddPoint(0.2f, 1.0f);
			mLateralFriction.AddPoint(0.0f, 0.0f); mLateralFriction.AddPoint(3.0f, 1.2f); mLateralFriction.AddPoint(20.0f, 1.0f);
		}
		float mInertia = 0.9f, mAngularDamping = 0.2f, mMaxSteerAngle = DegreesToRadians(70.0f);
		LinearCurve mLongitudinalFriction, mLateralFriction;
		float mMaxBrakeTorque = 1500.0f, mMaxHandBrakeTorque = 4000.0f;
	};

	class VehicleDifferentialSettings
	{
	public:
		int mLeftWheel = -1, mRightWheel = -1;
		float mDifferentialRatio = 3.42f, mLeftRightSplit = 0.5f, mLimitedSlipRatio = 1.4f, mEngineTorqueRatio = 1.0f;
	};
	class VehicleEngineSettings
	{
	public:
		VehicleEngineSettings() { mNormalizedTorque.AddPoint(0.0f, 0.8f); mNormalizedTorque.AddPoint(0.66f, 1.0f); mNormalizedTorque.AddPoint(1.0f, 0.8f); }
		float mMaxTorque = 500.0f, mMinRPM = 1000.0f, mMaxRPM = 6000.0f;
		LinearCurve mNormalizedTorque;
		float mInertia = 0.5f, mAngularDamping = 0.2f;
	};
	enum class ETransmissionMode { Auto, Manual };
	class VehicleTransmissionSettings
	{
	public:
		ETransmissionMode mMode = ETransmissionMode::Auto;
		std::vector<float> mGearRatios = { 2.66f, 1.78f, 1.3f, 1.0f, 0.74f }, mReverseGearRatios = { -2.90f };
		float mSwitchTime = 0.5f, mClutchReleaseTime = 0.3f, mSwitchLatency = 0.5f, mShiftUpRPM = 4000.0f, mShiftDownRPM = 2000.0f, mClutchStrength = 10.0f;
	};
	class VehicleAntiRollBar { public: int mLeftWheel = 0, mRightWheel = 1; float mStiffness = 1000.0f; };

	class VehicleControllerSettings : public RefTargetBase {};
	class WheeledVehicleControllerSettings : public VehicleControllerSettings
	{
	public:
		VehicleEngineSettings mEngine;
		VehicleTransmissionSettings mTransmission;
		std::vector<VehicleDifferentialSettings> mDifferentials;
		float mDifferentialLimitedSlipRatio = 1.4f;
	};
	// BikePhysics.cpp:197-205
	class MotorcycleControllerSettings : public WheeledVehicleControllerSettings
	{
	public:
		float mMaxLeanAngle = DegreesToRadians(45.0f), mLeanSpringConstant = 5000.0f, mLeanSpringDamping = 1000.0f;
		float mLeanSpringIntegrationCoefficient = 0.0f, mLeanSpringIntegrationCoefficientDecay = 4.0f, mLeanSmoothingFactor = 0.8f;
	};
	class VehicleConstraintSettings
	{
	public:
		Vec3 mUp = Vec3(0, 1, 0), mForward = Vec3(0, 0, 1);
		float mMaxPitchRollAngle = JPH_PI;
		std::vector<Ref<WheelSettings>> mWheels;
		std::vector<VehicleAntiRollBar> mAntiRollBars;
		Ref<VehicleControllerSettings> mController;
	};

	class VehicleCollisionTester : public RefTargetBase { public: virtual float castRadius(float /*wheel_width*/) const { return 0.0f; } virtual uint32_t testerKind() const { return SGP_VEHICLE_TESTER_SPHERE; } };
	class VehicleCollisionTesterRay : public VehicleCollisionTester
	{
	public:
		VehicleCollisionTesterRay(ObjectLayer, const Vec3& = Vec3(0, 1, 0), float max_slope = DegreesToRadians(80.0f)) : mMaxSlopeAngle(max_slope) {}
		float mMaxSlopeAngle;
	};
	class VehicleCollisionTesterCastSphere : public VehicleCollisionTester
	{
	public:
		VehicleCollisionTesterCastSphere(ObjectLayer, float radius, const Vec3& = Vec3(0, 1, 0), float max_slope = DegreesToRadians(80.0f)) : mRadius(radius), mMaxSlopeAngle(max_slope) {}
		float castRadius(float) const override { return mRadius; }
		float mRadius, mMaxSlopeAngle;
	};
	// BikePhysics.cpp:229.  The wheel itself is cast (SGP_VEHICLE_TESTER_CYLINDER): a disc of the wheel's radius less half its width, rounded by half its width --
	// what Jolt's cylinder is with inConvexRadiusFraction = 1, the value BikePhysics passes.  Other fractions get the same shape.
	class VehicleCollisionTesterCastCylinder : public VehicleCollisionTester
	{
	public:
		VehicleCollisionTesterCastCylinder(ObjectLayer, float convex_radius_fraction = 0.1f) : mConvexRadiusFraction(convex_radius_fraction) {}
		float castRadius(float wheel_width) const override { return 0.5f * wheel_width; }
		uint32_t testerKind() const override { return SGP_VEHICLE_TESTER_CYLINDER; }
		float mConvexRadiusFraction;
	};

	class VehicleConstraint;

	class Wheel
	{
	public:
		const WheelSettings* GetSettings() const { return settings; }
		float GetSuspensionLength() const { return st().suspension_length; }
		float GetSteerAngle() const { return st().steer_angle; }
		float GetRotationAngle() const { return st().rotation_angle; }
		float GetAngularVelocity() const { return st().angular_velocity; }
		void  SetAngularVelocity(float w);
		bool  HasContact() const { return st().has_contact != 0; }
		BodyID GetContactBodyID() const { return BodyID(st().contact_body); }
		RVec3 GetContactPosition() const { return v(st().contact_position); }
		Vec3  GetContactPointVelocity() const { return v(st().contact_point_velocity); }
		Vec3  GetContactNormal() const { return v(st().contact_normal); }
		Vec3  GetContactLongitudinal() const { return v(st().contact_longitudinal); }
		Vec3  GetContactLateral() const { return v(st().contact_lateral); }
		float GetSuspensionLambda() const { return st().suspension_lambda; }
		float GetLongitudinalLambda() const { return st().longitudinal_lambda; }
		float GetLateralLambda() const { return st().lateral_lambda; }
	private:
		friend class VehicleConstraint;
		static Vec3 v(const float* p) { return Vec3(p[0], p[1], p[2]); }
		const sgp_wheel_state& st() const;
		VehicleConstraint* owner = nullptr; int index = 0; const WheelSettings* settings = nullptr;
	};

	class VehicleEngine
	{
	public:
		float GetCurrentRPM() const;
		void  SetCurrentRPM(float rpm);
	private:
		friend class VehicleConstraint;
		VehicleConstraint* owner = nullptr;
	};

	class VehicleController { public: virtual ~VehicleController() {} };
	class WheeledVehicleController : public VehicleController
	{
	public:
		void SetDriverInput(float forward, float right, float brake, float hand_brake);
		VehicleEngine& GetEngine() { return engine; }
		const VehicleEngine& GetEngine() const { return engine; }
		int GetCurrentGear() const;
	protected:
		friend class VehicleConstraint;
		VehicleConstraint* owner = nullptr; VehicleEngine engine;
	};
	class MotorcycleController : public WheeledVehicleController
	{
	public:
		void EnableLeanController(bool enable);                  // BikePhysics.cpp:493,617
		bool IsLeanControllerEnabled() const { return lean_enabled; }
	private:
		bool lean_enabled = true;
	};

	// Bound to a world by PhysicsSystem::AddConstraint (CarPhysics.cpp:224-226).
	class VehicleConstraint : public RefTargetBase, public PhysicsStepListener
	{
	public:
		VehicleConstraint(const Body& body, const VehicleConstraintSettings& s) : body_id(body.GetID()), settings(s),
			frame_com(body.com_offset), frame_rot(body.frame_rot[0], body.frame_rot[1], body.frame_rot[2], body.frame_rot[3])
		{
			wheels.resize(settings.mWheels.size());
			for (size_t i = 0; i < wheels.size(); ++i) { wheels[i].owner = this; wheels[i].index = (int)i; wheels[i].settings = settings.mWheels[i].GetPtr(); }
			controller.owner = this; controller.engine.owner = this;
		}
		void SetVehicleCollisionTester(const VehicleCollisionTester* t) { cast_radius = t ? t->castRadius(settings.mWheels.empty() ? 0.0f : settings.mWheels[0]->mWidth) : 0.0f; tester_kind = t ? t->testerKind() : (uint32_t)SGP_VEHICLE_TESTER_SPHERE; }
		VehicleController* GetController() { return &controller; }
		const VehicleController* GetController() const { return &controller; }
		Wheel* GetWheel(uint i) { return &wheels[i]; }
		const Wheel* GetWheel(uint i) const { return &wheels[i]; }
		Vec3 GetLocalUp() const { return settings.mUp; }
		Vec3 GetLocalForward() const { return settings.mForward; }
		// wheel basis in the chassis frame, steering applied (Jolt: VehicleConstraint::GetWheelLocalBasis)
		void GetWheelLocalBasis(const Wheel* w, Vec3& forward_out, Vec3& up_out, Vec3& right_out) const
		{
			const WheelSettings* ws = w->GetSettings();
			const float a = w->GetSteerAngle(), s = std::sin(0.5f * a), c = std::cos(0.5f * a);
			const Quat steer(ws->mSteeringAxis.x * s, ws->mSteeringAxis.y * s, ws->mSteeringAxis.z * s, c);
			up_out = ws->mWheelUp; forward_out = steer * ws->mWheelForward;
			right_out = Vec3(forward_out.y * up_out.z - forward_out.z * up_out.y, forward_out.z * up_out.x - forward_out.x * up_out.z, forward_out.x * up_out.y - forward_out.y * up_out.x);
			const float l = std::sqrt(right_out.LengthSq()); if (l > 0) right_out = right_out * (1.0f / l);
			up_out = Vec3(right_out.y * forward_out.z - right_out.z * forward_out.y, right_out.z * forward_out.x - right_out.x * forward_out.z, right_out.x * forward_out.y - right_out.y * forward_out.x);
		}
		// wheel centre + basis in the chassis frame: columns (right, up, forward) for the given model axes are left to the caller,
		// the translation is what CarPhysics reads (:437-440)
		Mat44 GetWheelLocalTransform(uint i, const Vec3& /*wheel_right*/, const Vec3& /*wheel_up*/) const
		{
			const Wheel* w = &wheels[i]; const WheelSettings* ws = w->GetSettings();
			Vec3 f, u, r; GetWheelLocalBasis(w, f, u, r);
			Mat44 m; m.c[0] = r; m.c[1] = u; m.c[2] = f;
			m.c[3] = ws->mPosition + ws->mSuspensionDirection * w->GetSuspensionLength();
			return m;
		}
		// Jolt: VehicleConstraint::GetWheelWorldTransform = body world transform (shape space -> world) * GetWheelLocalTransform
		// (CarPhysics.cpp:405-470 keeps this variant commented next to the local one; BikePhysics uses the same pair).
		Mat44 GetWheelWorldTransform(uint i, const Vec3& wheel_right, const Vec3& wheel_up) const;
		BodyID GetVehicleBodyID() const { return body_id; }
		uint32_t GetVehicleID() const { return vehicle_id; }

		// -- binding (used by PhysicsSystem)
		void fillDesc(sgp_vehicle_desc& d) const
		{
			sgp_default_vehicle_desc(&d);
			d.body = body_id.GetIndex();
			d.num_wheels = (uint32_t)settings.mWheels.size();
			for (uint32_t i = 0; i < d.num_wheels && i < SGP_MAX_WHEELS; ++i) {
				const WheelSettingsWV* s = dynamic_cast<const WheelSettingsWV*>(settings.mWheels[i].GetPtr());
				sgp_wheel_desc& w = d.wheels[i];
				// shape space -> body frame (centre of mass / principal axes of a hull chassis)
				const Quat inv = frame_rot.Conjugated();
				put(w.position, inv * (s->mPosition - frame_com)); put(w.suspension_dir, inv * s->mSuspensionDirection); put(w.steering_axis, inv * s->mSteeringAxis);
				put(w.wheel_up, inv * s->mWheelUp); put(w.wheel_forward, inv * s->mWheelForward);
				w.suspension_min_length = s->mSuspensionMinLength; w.suspension_max_length = s->mSuspensionMaxLength; w.suspension_preload = s->mSuspensionPreloadLength;
				w.spring_frequency = s->mSuspensionSpring.mFrequency; w.spring_damping = s->mSuspensionSpring.mDamping;
				w.radius = s->mRadius; w.width = s->mWidth; w.inertia = s->mInertia; w.angular_damping = s->mAngularDamping;
				w.max_steer_angle = s->mMaxSteerAngle; w.max_brake_torque = s->mMaxBrakeTorque; w.max_handbrake_torque = s->mMaxHandBrakeTorque;
				for (int k = 0; k < 3; ++k) {
					w.longitudinal_friction[k][0] = s->mLongitudinalFriction.mPoints[k].mX; w.longitudinal_friction[k][1] = s->mLongitudinalFriction.mPoints[k].mY;
					w.lateral_friction[k][0] = s->mLateralFriction.mPoints[k].mX; w.lateral_friction[k][1] = s->mLateralFriction.mPoints[k].mY;
				}
			}
			put(d.up, frame_rot.Conjugated() * settings.mUp); put(d.forward, frame_rot.Conjugated() * settings.mForward);
			d.cast_radius = cast_radius;
			d.collision_tester = tester_kind;
			const WheeledVehicleControllerSettings* c = dynamic_cast<const WheeledVehicleControllerSettings*>(settings.mController.GetPtr());
			d.engine_max_torque = c->mEngine.mMaxTorque; d.engine_min_rpm = c->mEngine.mMinRPM; d.engine_max_rpm = c->mEngine.mMaxRPM;
			d.engine_inertia = c->mEngine.mInertia; d.engine_angular_damping = c->mEngine.mAngularDamping;
			for (int k = 0; k < 3; ++k) { d.engine_torque_curve[k][0] = c->mEngine.mNormalizedTorque.mPoints[k].mX; d.engine_torque_curve[k][1] = c->mEngine.mNormalizedTorque.mPoints[k].mY; }
			d.num_gears = (uint32_t)c->mTransmission.mGearRatios.size(); d.num_reverse_gears = (uint32_t)c->mTransmission.mReverseGearRatios.size();
			for (uint32_t k = 0; k < d.num_gears && k < SGP_MAX_GEARS; ++k) d.gear_ratios[k] = c->mTransmission.mGearRatios[k];
			for (uint32_t k = 0; k < d.num_reverse_gears && k < SGP_MAX_GEARS; ++k) d.reverse_gear_ratios[k] = c->mTransmission.mReverseGearRatios[k];
			d.switch_time = c->mTransmission.mSwitchTime; d.clutch_release_time = c->mTransmission.mClutchReleaseTime; d.switch_latency = c->mTransmission.mSwitchLatency;
			d.shift_up_rpm = c->mTransmission.mShiftUpRPM; d.shift_down_rpm = c->mTransmission.mShiftDownRPM; d.clutch_strength = c->mTransmission.mClutchStrength;
			d.num_differentials = (uint32_t)c->mDifferentials.size();
			for (uint32_t k = 0; k < d.num_differentials && k < 2; ++k) {
				const VehicleDifferentialSettings& s = c->mDifferentials[k];
				d.differentials[k].left_wheel = s.mLeftWheel; d.differentials[k].right_wheel = s.mRightWheel; d.differentials[k].differential_ratio = s.mDifferentialRatio;
				d.differentials[k].left_right_split = s.mLeftRightSplit; d.differentials[k].limited_slip_ratio = s.mLimitedSlipRatio; d.differentials[k].engine_torque_ratio = s.mEngineTorqueRatio;
			}
			d.differential_limited_slip_ratio = c->mDifferentialLimitedSlipRatio;
			if (const MotorcycleControllerSettings* m = dynamic_cast<const MotorcycleControllerSettings*>(c)) {
				d.controller_type = SGP_VEHICLE_CONTROLLER_MOTORCYCLE;
				d.max_lean_angle = m->mMaxLeanAngle; d.lean_spring_constant = m->mLeanSpringConstant; d.lean_spring_damping = m->mLeanSpringDamping;
				d.lean_spring_integration_coefficient = m->mLeanSpringIntegrationCoefficient; d.lean_spring_integration_decay = m->mLeanSpringIntegrationCoefficientDecay;
				d.lean_smoothing_factor = m->mLeanSmoothingFactor;
			}
			d.num_anti_roll_bars = (uint32_t)settings.mAntiRollBars.size();
			for (uint32_t k = 0; k < d.num_anti_roll_bars && k < 2; ++k) {
				d.anti_roll_bars[k].left_wheel = settings.mAntiRollBars[k].mLeftWheel; d.anti_roll_bars[k].right_wheel = settings.mAntiRollBars[k].mRightWheel;
				d.anti_roll_bars[k].stiffness = settings.mAntiRollBars[k].mStiffness;
			}
		}
		void bind(sgp_world* w, uint32_t id, const uint64_t* serial) { world = w; vehicle_id = id; step_serial = serial; cached_serial = ~0ull; }
		void unbind() { world = nullptr; vehicle_id = 0xFFFFFFFFu; }
		const sgp_vehicle_state& state() const
		{
			if (world && (cached_serial != *step_serial)) { sgp_vehicle_get_state(world, vehicle_id, &cached); cached_serial = *step_serial; }
			return cached;
		}
		sgp_world* world = nullptr;
	private:
		static void put(float* o, const Vec3& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
		BodyID body_id; VehicleConstraintSettings settings; Vec3 frame_com; Quat frame_rot; float cast_radius = 0.0f; uint32_t tester_kind = SGP_VEHICLE_TESTER_SPHERE;
		std::vector<Wheel> wheels; MotorcycleController controller;    // (a WheeledVehicleController for cars; the lean part is inert then)
		uint32_t vehicle_id = 0xFFFFFFFFu; const uint64_t* step_serial = nullptr;
		mutable uint64_t cached_serial = ~0ull; mutable sgp_vehicle_state cached = {};
		friend class Wheel; friend class VehicleEngine; friend class WheeledVehicleController; friend class MotorcycleController;
	};

	inline Mat44 VehicleConstraint::GetWheelWorldTransform(uint i, const Vec3& wheel_right, const Vec3& wheel_up) const
	{
		const Mat44 local = GetWheelLocalTransform(i, wheel_right, wheel_up);
		if (!world) return local;
		// chassis pose of the BODY frame -> pose of the shape space the wheel settings are expressed in
		sgp_body_state st; const uint32_t id = body_id.GetIndex();
		if (sgp_body_get_state(world, &id, 1, &st) != SGP_OK) return local;
		const Quat qb(st.rot[0], st.rot[1], st.rot[2], st.rot[3]);
		const Quat q = qb * frame_rot.Conjugated();
		const Vec3 t = Vec3(st.pos[0], st.pos[1], st.pos[2]) - q * frame_com;
		return Mat44::sRotationTranslation(q, t) * local;
	}
	inline const sgp_wheel_state& Wheel::st() const { return owner->state().wheels[index]; }
	inline void Wheel::SetAngularVelocity(float w) { if (owner->world) { sgp_vehicle_reset_drivetrain(owner->world, owner->GetVehicleID(), owner->state().engine_rpm, w); owner->cached_serial = ~0ull; } }
	inline float VehicleEngine::GetCurrentRPM() const { return owner->state().engine_rpm; }
	inline void VehicleEngine::SetCurrentRPM(float rpm) { if (owner->world) { sgp_vehicle_reset_drivetrain(owner->world, owner->GetVehicleID(), rpm, owner->state().wheels[0].angular_velocity); owner->cached_serial = ~0ull; } }
	inline int WheeledVehicleController::GetCurrentGear() const { return owner->state().current_gear; }
	inline void MotorcycleController::EnableLeanController(bool enable)
	{
		if (enable == lean_enabled || !owner->world) { lean_enabled = enable; return; }
		lean_enabled = enable;
		sgp_vehicle_enable_lean_controller(owner->world, owner->GetVehicleID(), enable ? 1 : 0);
	}
	inline void WheeledVehicleController::SetDriverInput(float f, float r, float b, float h)
	{
		if (!owner->world) return;
		const sgp_vehicle_input in = { f, r, b, h };
		sgp_vehicle_set_input(owner->world, owner->GetVehicleID(), &in);
	}
}
