// JPH look-alike subset: exactly the Jolt types the physics FACADE's signatures and GUIClient's listeners name
// (PhysicsWorld.h:77-87,178-185; GUIClient.cpp:10588-10632).  Backed by the sgp C ABI, no Jolt code.
#pragma once
#include <cstdint>
#include <vector>
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
		float LengthSq() const { return x * x + y * y + z * z; }
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
	// What the listeners read from a body during a contact callback.
	class Body
	{
	public:
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
}
