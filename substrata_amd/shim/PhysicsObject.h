// PhysicsObject -- host-side body record with the reference's field names and defaults
// (/root/reference/gui_client/PhysicsObject.h:52-128, PhysicsObject.cpp:25-60).  The only change against the
// reference declaration: `JPH::Ref<JPH::Shape> jolt_shape` inside PhysicsShape becomes a small POD description of
// the primitive, because this backend collides spheres, boxes and capsules natively (SURVEY.md 8f lists mesh /
// hull / height-field shapes as "next").
#pragma once
#include <maths/Vec4f.h>
#include <maths/Quat.h>
#include <maths/vec3.h>
#include <maths/Matrix4f.h>
#include <utils/ThreadSafeRefCounted.h>
#include <utils/Reference.h>
#include <physics/jscol_aabbox.h>
#include <Jolt/JoltLite.h>
#include <cstddef>

class RayTraceResult;

// Role of PhysicsShape (PhysicsObject.h:33-44).  kind: -1 none, 0 sphere (p0 = r), 1 box (p = half extents), 2 capsule (p0 = r, p1 = half height)
class PhysicsShape
{
public:
	PhysicsShape() : kind(-1), size_B(0) { p[0] = p[1] = p[2] = p[3] = 0.f; }
	js::AABBox getAABBOS() const;
	int kind;
	float p[4];
	size_t size_B;
};

class PhysicsObject : public ThreadSafeRefCounted
{
public:
	GLARE_ALIGNED_16_NEW_DELETE
	friend class PhysicsWorld;

	PhysicsObject(bool collidable);
	PhysicsObject(bool collidable, const PhysicsShape& shape, void* userdata, int userdata_type);
	~PhysicsObject();

	const js::AABBox getAABBoxWS() const;
	const Matrix4f getObToWorldMatrix() const;
	const Matrix4f getWorldToObMatrix() const;

	inline bool isDynamic()   const { return motion_type == MotionType_dynamic; }
	inline bool isKinematic() const { return motion_type == MotionType_kinematic; }

public:
	PhysicsShape shape;
	bool collidable;
	bool is_sensor;
	void* userdata;
	int userdata_type;

	Vec4f pos;
	Quatf rot;
	Vec3f scale;

	Vec4f smooth_translation;
	Quatf smooth_rotation;

	JPH::BodyID jolt_body_id;
	bool is_sphere;
	bool is_cube;

	enum MotionType { MotionType_dynamic, MotionType_kinematic, MotionType_semi_static, MotionType_static };
	MotionType motion_type;

	bool use_zero_linear_drag;
	bool underwater;
	float last_submerged_volume;

	float mass;
	float friction;
	float restitution;
};

typedef Reference<PhysicsObject> PhysicsObjectRef;

struct PhysicsObjectHash
{
	size_t operator() (const PhysicsObjectRef& ob) const { return (size_t)ob.getPointer() >> 3; }
};
