// JPH look-alike subset: exactly the Jolt types the physics FACADE's signatures and GUIClient's listeners name
// (PhysicsWorld.h:77-87,178-185; GUIClient.cpp:10588-10632).  Backed by the sgp C ABI, no Jolt code.
#pragma once
#include <cstdint>
#include <vector>
#include <cmath>
namespace JPH
{
	typedef unsigned int uint;
	class Vec3
	{
	public:
		Vec3() : x(0), y(0), z(0) {}
		Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
		float GetX() const { return x; } float GetY() const { return y; } float GetZ() const { return z; }
		Vec3 operator+(const Vec3& o) const { return Vec3(x + o.x, y + o.y, z + o.z); }
		Vec3 operator-(const Vec3& o) const { return Vec3(x - o.x, y - o.y, z - o.z); }
		Vec3 operator*(float f) const { return Vec3(x * f, y * f, z * f); }
		Vec3 operator/(float f) const { return Vec3(x / f, y / f, z / f); }
		Vec3 operator-() const { return Vec3(-x, -y, -z); }
		float Dot(const Vec3& o) const { return x * o.x + y * o.y + z * o.z; }
		Vec3 Cross(const Vec3& o) const { return Vec3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
		float LengthSq() const { return x * x + y * y + z * z; }
		float Length() const { return std::sqrt(LengthSq()); }
		Vec3 Normalized() const { const float l = Length(); return Vec3(x / l, y / l, z / l); }
		float x, y, z;
	};
	typedef Vec3 RVec3;
	class BodyID
	{
	public:
		static const uint32_t cInvalidBodyID = 0xFFFFFFFFu;
		BodyID() : id(cInvalidBodyID) {}
		explicit BodyID(uint32_t i) : id(i) {}
		bool IsInvalid() const { return id == cInvalidBodyID; }
		uint32_t GetIndexAndSequenceNumber() const { return id; }
		uint32_t GetIndex() const { return id; }
		bool operator==(const BodyID& o) const { return id == o.id; }
		bool operator!=(const BodyID& o) const { return id != o.id; }
	private:
		uint32_t id;
	};
	class Shape { public: float GetVolume() const { return volume; } float volume = 0; };
	// What the listeners read from a body during a contact callback.
	class Body
	{
	public:
		const Shape* GetShape() const { return &shape; }
		Shape shape;
		// Where the simulated body frame (centre of mass, principal axes) sits in the object's shape space; identity except for convex
		// hulls.  Jolt hides this inside the body; PhysicsWorld::getJoltBody() fills it in and VehicleConstraint uses it to express the
		// wheel settings (given in shape space, like JPH::WheelSettings::mPosition) in the body frame.
		Vec3 com_offset; float frame_rot[4] = { 0, 0, 0, 1 };
		Vec3 GetLinearVelocity() const { return lin_vel; }
		uint64_t GetUserData() const { return user_data; }
		BodyID GetID() const { return id; }
		Vec3 lin_vel; uint64_t user_data = 0; BodyID id;
	};
	class ContactManifold
	{
	public:
		RVec3 mBaseOffset;
		Vec3 mWorldSpaceNormal;
		float mPenetrationDepth = 0;
		std::vector<Vec3> mRelativeContactPointsOn1;
	};
	class ContactSettings {};

	class Quat
	{
	public:
		Quat() : x(0), y(0), z(0), w(1) {}
		Quat(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
		float GetX() const { return x; } float GetY() const { return y; } float GetZ() const { return z; } float GetW() const { return w; }
		Quat Conjugated() const { return Quat(-x, -y, -z, w); }
		Vec3 operator*(const Vec3& v) const   // rotate
		{
			const float tx = 2 * (y * v.z - z * v.y), ty = 2 * (z * v.x - x * v.z), tz = 2 * (x * v.y - y * v.x);
			return Vec3(v.x + w * tx + (y * tz - z * ty), v.y + w * ty + (z * tx - x * tz), v.z + w * tz + (x * ty - y * tx));
		}
		float x, y, z, w;
	};
	// column-major rotation + translation, the part of JPH::Mat44 the controllers read
	class Mat44
	{
	public:
		Vec3 GetAxisX() const { return c[0]; } Vec3 GetAxisY() const { return c[1]; } Vec3 GetAxisZ() const { return c[2]; }
		Vec3 GetTranslation() const { return c[3]; }
		Vec3 GetColumn3(int i) const { return c[i]; }
		Vec3 operator*(const Vec3& v) const { return c[0] * v.x + c[1] * v.y + c[2] * v.z + c[3]; }
		Vec3 Multiply3x3(const Vec3& v) const { return c[0] * v.x + c[1] * v.y + c[2] * v.z; }
		Vec3 c[4];
	};
	enum class EActivation { Activate, DontActivate };
}

struct sgp_world;

namespace JPH
{
	// JPH::BodyInterface look-alike: the calls HoverCarPhysics.cpp:113-348,425-480, BoatPhysics.cpp:35-49,134-267,367-385 and
	// GUIClient.cpp:6577-6673 make through physics_world.physics_system->GetBodyInterface(), forwarded to the sgp C ABI.
	// Getters share one cached read-back per body between world mutations (invalidate() is called by PhysicsWorld::think and setters).
	class BodyInterface
	{
	public:
		explicit BodyInterface(sgp_world* w) : world(w), cached_id(0xFFFFFFFFu) {}
		void ActivateBody(const BodyID& id);
		void AddForce(const BodyID& id, const Vec3& force);
		void AddForce(const BodyID& id, const Vec3& force, const RVec3& point);
		void AddTorque(const BodyID& id, const Vec3& torque);
		RVec3 GetPosition(const BodyID& id) const;
		RVec3 GetCenterOfMassPosition(const BodyID& id) const;
		Quat GetRotation(const BodyID& id) const;
		void GetPositionAndRotation(const BodyID& id, RVec3& pos_out, Quat& rot_out) const;
		Mat44 GetWorldTransform(const BodyID& id) const;
		Vec3 GetLinearVelocity(const BodyID& id) const;
		Vec3 GetAngularVelocity(const BodyID& id) const;
		void GetLinearAndAngularVelocity(const BodyID& id, Vec3& lin_out, Vec3& ang_out) const;
		Vec3 GetPointVelocity(const BodyID& id, const RVec3& point) const;
		void SetLinearAndAngularVelocity(const BodyID& id, const Vec3& lin, const Vec3& ang);
		bool IsActive(const BodyID& id) const;
		void invalidate() const { cached_id = 0xFFFFFFFFu; }
	private:
		void fetch(const BodyID& id) const;
		sgp_world* world;
		mutable uint32_t cached_id;
		mutable float st_pos[3], st_rot[4], st_lv[3], st_av[3];
		mutable bool st_active;
	};

	// GetBodyLockInterface().TryGetBody(id)->GetShape()->GetVolume()  (BoatPhysics.cpp:40-43)
	class BodyLockInterface
	{
	public:
		explicit BodyLockInterface(sgp_world* w) : world(w) {}
		Body* TryGetBody(const BodyID& id) const;          // valid until the next TryGetBody call
	private:
		sgp_world* world; mutable Body scratch;
	};

	class BroadPhaseLayerFilter {};
	class ObjectLayerFilter { public: virtual ~ObjectLayerFilter() {} virtual bool ShouldCollide(uint16_t) const { return true; } };
	class DefaultBroadPhaseLayerFilter : public BroadPhaseLayerFilter {};
	class DefaultObjectLayerFilter : public ObjectLayerFilter {};

	class VehicleConstraint;      // Jolt/JoltVehicleLite.h
	class PhysicsStepListener;

	class PhysicsSystem
	{
	public:
		explicit PhysicsSystem(sgp_world* w) : world(w), body_interface(w), body_lock_interface(w), step_serial(0) {}
		const BodyLockInterface& GetBodyLockInterface() const { return body_lock_interface; }
		BodyInterface& GetBodyInterface() { return body_interface; }
		const BodyInterface& GetBodyInterface() const { return body_interface; }
		Vec3 GetGravity() const { return Vec3(0, 0, -9.81f); }   // PhysicsWorld.cpp:520
		// CarPhysics.cpp:224-226,258-262: the vehicle constraint is both a constraint and a step listener in Jolt; here the
		// constraint registration creates / destroys the device-side vehicle and the listener calls are no-ops.
		void AddConstraint(VehicleConstraint* c);
		void RemoveConstraint(VehicleConstraint* c);
		template <class T> void AddStepListener(T*) {}
		template <class T> void RemoveStepListener(T*) {}
		void onStep() { ++step_serial; body_interface.invalidate(); }       // called by PhysicsWorld::think
		// filter factories CharacterVirtual callers pass through (PlayerPhysics.cpp:106-114,344-346): the character queries always use the
		// MOVING object layer's collision set, so these are placeholders
		DefaultBroadPhaseLayerFilter GetDefaultBroadPhaseLayerFilter(uint16_t) const { return DefaultBroadPhaseLayerFilter(); }
		DefaultObjectLayerFilter GetDefaultLayerFilter(uint16_t) const { return DefaultObjectLayerFilter(); }
		sgp_world* world;
	private:
		BodyInterface body_interface;
		BodyLockInterface body_lock_interface;
		uint64_t step_serial;
	};
}
