// PhysicsObject -- host-side body record with the reference's field names and defaults
// (/root/reference/gui_client/PhysicsObject.h:52-128, PhysicsObject.cpp:25-60).  The only change against the
// reference declaration: `JPH::Ref<JPH::Shape> jolt_shape` inside PhysicsShape becomes a small POD description of
// the primitive, because this backend collides spheres, boxes and capsules natively (SURVEY.md 8f lists mesh /
// hull / height-field shapes as "next").
#pragma once
#include <memory>
#include <vector>
#include <functional>
#include <maths/Vec4f.h>
#include <maths/Quat.h>
#include <maths/vec3.h>
#include <maths/Matrix4f.h>
#include <utils/ThreadSafeRefCounted.h>
#include <utils/Reference.h>
#include <physics/jscol_aabbox.h>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/Body/BodyID.h>
#include <Jolt/Physics/Collision/Shape/Shape.h>
#include <cstddef>

class RayTraceResult;

// Role of PhysicsShape (PhysicsObject.h:33-44).  kind: -1 none, 0 sphere (p0 = r), 1 box (p = half extents), 2 capsule (p0 = r, p1 = half height)
// Points of a convex hull shape (what createJoltShapeFor...Mesh(..., is_dynamic = true) hands to JPH::ConvexHullShapeSettings) plus
// the device-side hulls already built from them, one per (world, object scale): the shape is shared between bodies like a
// JPH::Ref<JPH::Shape>, the scale is baked into the hull the way JPH::ScaledShape applies it.
struct sgp_world;
struct PhysicsHullData
{
	struct Instance { sgp_world* world; float scale[3]; uint32_t hull_id; float com[3]; float rot[4]; float aabb_min[3], aabb_max[3]; uint32_t users; uint32_t num_vertices; };      // num_vertices: corners the device hull kept (<= 256, JPH::ConvexHullShape::cMaxPointsInHull)
	std::vector<float> points;          // xyz, object space, unscaled
	float com_offset[3] = { 0, 0, 0 };  // OffsetCenterOfMassShape: moves the centre of mass away from the hull's own (object space, unscaled)
	std::vector<Instance> instances;
};

// Triangles of a static mesh shape (what createJoltShapeFor...Mesh(..., is_dynamic = false) hands to JPH::MeshShapeSettings, and the
// triangulated samples of a height field) plus the device-side meshes built from them, one per (world, object scale).
struct PhysicsMeshData
{
	struct Instance { sgp_world* world; float scale[3]; uint32_t mesh_id; uint32_t users; };      // users = bodies made from it; destroyed with the last one
	std::vector<float> vertices;        // xyz, object space, unscaled
	std::vector<uint32_t> indices;      // 3 per triangle, counter-clockwise = front
	std::vector<uint32_t> materials;    // per triangle: the material index a ray hit reports (JPH::IndexedTriangle::mMaterialIndex, PhysicsWorld.cpp:1032-1060); empty = all 0
	std::vector<Instance> instances;
};

class PhysicsShape
{
public:
	PhysicsShape() : kind(-1), size_B(0) { p[0] = p[1] = p[2] = p[3] = 0.f; }
	js::AABBox getAABBOS() const;
	int kind;                            // 0 sphere, 1 box, 2 capsule, 3 convex hull (hull != null), 4 static triangle mesh (mesh != null); -1 with jolt_shape of kind 5 = static compound
	float p[4];
	size_t size_B;
	std::shared_ptr<PhysicsHullData> hull;
	std::shared_ptr<PhysicsMeshData> mesh;
	// The same shape as a JPH::Shape look-alike (what the reference keeps in `JPH::Ref<JPH::Shape> jolt_shape`, PhysicsObject.h:41): set by the
	// PhysicsWorld::create...Shape builders so that callers can hand it to JPH shape settings, and assignable by callers that compose a
	// shape themselves -- MeshBuilding::makePortalMeshes assigns a StaticCompoundShape here (MeshBuilding.cpp:396-413).
	JPH::Ref<JPH::Shape> jolt_shape;
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

	// Hull bodies live in the hull's centre-of-mass / principal-axes frame (what OffsetCenterOfMassShape and the inertia rotation
	// hide inside Jolt): body pos = pos + rot * body_com_os, body rot = rot * body_rot_os.  Identity for every other shape.
	Vec4f body_com_os;
	Quatf body_rot_os;

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

	// device-side shape instances (mesh / hull of this object's scale) the object's body holds: released, and destroyed with their last
	// user, when the body is removed -- the role of the JPH::Ref<JPH::Shape> a Jolt body keeps on its shape
	std::vector<std::function<void()>> shape_instance_releases;
};

typedef Reference<PhysicsObject> PhysicsObjectRef;

struct PhysicsObjectHash
{
	size_t operator() (const PhysicsObjectRef& ob) const { return (size_t)ob.getPointer() >> 3; }
};
