// PhysicsWorld -- the reference's physics facade (/root/reference/gui_client/PhysicsWorld.h:98-218), same class and
// method names, same public members (activated_obs_mutex, activated_obs, newly_activated_obs, event_listener), bound to
// the sgp C ABI (include/sgp.h) instead of JoltPhysics.  Members that exposed raw Jolt objects (physics_system,
// temp_allocator, job_system: PhysicsWorld.h:204-210) do not exist here; INTEGRATION.md lists what that means for the
// callers that reach around the facade.
#pragma once
#include "PhysicsObject.h"
#include <maths/Vec4f.h>
#include <maths/Quat.h>
#include <maths/vec2.h>
#include <utils/ThreadSafeRefCounted.h>
#include <utils/Mutex.h>
#include <utils/HashSet.h>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/Body/BodyID.h>
#include <Jolt/Physics/Body/BodyActivationListener.h>
#include <Jolt/Physics/Collision/ContactListener.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <string>
#include <vector>
#include <cstdint>

namespace glare { class TaskManager; class StackAllocator; class Allocator; }
struct sgp_world;
typedef unsigned char uint8;
typedef uint64_t uint64;
typedef uint32_t uint32;

class RayTraceResult
{
public:
	Vec4f hit_normal_ws;
	const PhysicsObject* hit_object;
	float hit_t;
	unsigned int hit_mat_index;
	Vec2f coords;
};

namespace Layers
{
	static constexpr uint8 NON_MOVING = 0;
	static constexpr uint8 MOVING = 1;
	static constexpr uint8 NON_MOVING_NON_COLLIDABLE = 2;
	static constexpr uint8 MOVING_NON_COLLIDABLE = 3;
	static constexpr uint8 NUM_LAYERS = 4;
};

class PhysicsWorldEventListener
{
public:
	virtual ~PhysicsWorldEventListener() {}
	virtual void physicsObjectEnteredWater(PhysicsObject& ob) {}
	// The reference may call these off the main thread; this backend calls them on the caller's thread at the end of think().
	virtual void contactAdded(const JPH::Body& inBody1, const JPH::Body& inBody2, const JPH::ContactManifold& contact_manifold) {}
	virtual void contactPersisted(const JPH::Body& inBody1, const JPH::Body& inBody2, const JPH::ContactManifold& contact_manifold) {}
};

void computeToWorldAndToObMatrices(const Vec4f& translation, const Quatf& rot_quat, const Vec4f& scale, Matrix4f& ob_to_world_out, Matrix4f& world_to_ob_out);

class PhysicsWorld : public ThreadSafeRefCounted
{
public:
	PhysicsWorld(glare::TaskManager* task_manager, glare::StackAllocator* stack_allocator);
	~PhysicsWorld();

	static void init();

	void setWaterBuoyancyEnabled(bool enabled);
	bool getWaterBuoyancyEnabled() const { return water_buoyancy_enabled; }
	void setWaterZ(float water_z);
	float getWaterZ() const { return water_z; }

	void addObject(const Reference<PhysicsObject>& object);
	void removeObject(const Reference<PhysicsObject>& object);
	void activateObject(const Reference<PhysicsObject>& object);
	void setObjectLayer(const Reference<PhysicsObject>& object, uint8 new_object_layer);

	// Creates a box, centered at (0,0,0), with x and y extent = ground_quad_w, and z extent = 1.
	static PhysicsShape createGroundQuadShape(float ground_quad_w);
	// Not in the reference: the capsule the reference builds inline from JPH::CapsuleShape (PlayerPhysics.cpp:74, AvatarGraphics.cpp:150).
	static PhysicsShape createCapsuleShape(float radius, float half_height);
	// The convex hull createJoltShapeForIndigoMesh / createJoltShapeForBatchedMesh build for a dynamic mesh (PhysicsWorld.cpp:735-1166:
	// JPH::ConvexHullShapeSettings over the mesh vertices) and CarPhysics / BikePhysics build for their bodies (CarPhysics.cpp:66-78);
	// the mesh containers themselves (glare-core) are not part of this tree, so the builder takes the vertex positions.
	// Throws glare::Exception("Error building Jolt shape: ...") for fewer than 4 points; a degenerate cloud is reported when the
	// shape is first added to a world.
	static PhysicsShape createConvexHullShape(const std::vector<Vec3f>& points);
	// The triangle mesh createJoltShapeForIndigoMesh / createJoltShapeForBatchedMesh build for a static object (PhysicsWorld.cpp:735-1017:
	// JPH::MeshShapeSettings over the mesh's vertices and triangles); takes the arrays because the mesh containers are glare-core types.
	// triangle_materials (optional, one per triangle): the material index of the triangle's batch, returned as RayTraceResult::hit_mat_index
	// (PhysicsWorld.cpp:1032-1060,1700-1704).  create_tris_for_mat (optional): "should physics triangles be created for this material?" --
	// triangles of a material whose entry is false are left out (PhysicsWorld.h:124-125, .cpp:1028; MeshBuilding.cpp:392-393); materials
	// beyond the vector's size are kept, like the reference.
	static PhysicsShape createMeshShape(const std::vector<Vec3f>& vertices, const std::vector<uint32>& triangle_indices,
		const std::vector<uint32>* triangle_materials = nullptr, const std::vector<bool>* create_tris_for_mat = nullptr);
	// PhysicsWorld.cpp:1086-1119: a heightfield.getWidth() x getWidth() grid of heights (row-major, sample (x, z) at [z * width + x]) in Jolt's
	// y-up shape space: vertex = (quad_w * x, height, quad_w * z - quad_w * (width - 1)); triangulated here (two triangles per cell, facing +y).
	static PhysicsShape createJoltHeightFieldShape(int vert_res, const std::vector<float>& heightfield, int width, float quad_w);
	// PhysicsWorld.cpp:1138-1153 (OffsetCenterOfMassShapeSettings); implemented for convex hull shapes.
	static PhysicsShape createCOMOffsetShapeForShape(const PhysicsShape& original_shape, const Vec4f& COM_offset);
	static PhysicsShape createScaledAndTranslatedShapeForShape(const PhysicsShape& shape, const Vec3f& translation, const Vec3f& scale);      // PhysicsWorld.h:133

	// What body_interface.CreateBody(...) hands CarPhysics / BikePhysics (CarPhysics.cpp:84-88): the body an already added object is
	// simulated as, for constructing a JPH::VehicleConstraint on it.
	JPH::Body getJoltBody(const PhysicsObject& object) const;

	void think(double dt);

	void setNewObToWorldTransform(PhysicsObject& object, const Vec4f& translation, const Quatf& rot, const Vec4f& scale);
	void setNewObToWorldTransform(PhysicsObject& object, const Vec4f& translation, const Quatf& rot, const Vec4f& linear_vel, const Vec4f& angular_vel);
	void setNewPosition(PhysicsObject& object, const Vec4f& pos);
	Vec4f getObjectLinearVelocity(const PhysicsObject& object) const;
	void setLinearAndAngularVelToZero(PhysicsObject& object);
	void moveKinematicObject(PhysicsObject& object, const Vec4f& translation, const Quatf& rot, float dt);
	void clear();

	struct MemUsageStats { size_t mem; size_t num_meshes; std::vector<int> layer_counts; };
	MemUsageStats getMemUsageStats() const;
	std::string getDiagnostics() const;
	std::string getLoadedMeshes() const;
	const Vec4f getPosInJolt(const Reference<PhysicsObject>& object);
	size_t getNumObjects() const;

	// Debug helpers (PhysicsWorld.h:187-189).  The snapshot is this library's own flat dump (a header + one sgp_body_state per body slot),
	// not Jolt's PhysicsScene stream; computeSizeBForShape reports the bytes the shape description holds.
	void writeJoltSnapshotToDisk(const std::string& path);
	static size_t computeSizeBForShape(const PhysicsShape& shape);
	static size_t computeSizeBForShape(JPH::Ref<JPH::Shape> jolt_shape);      // PhysicsWorld.h:189

	void traceRay(const Vec4f& origin, const Vec4f& dir, float max_t, JPH::BodyID ignore_body_id, RayTraceResult& results_out) const;
	void traceRayAgainstCollidableObs(const Vec4f& origin, const Vec4f& dir, float max_t, JPH::BodyID ignore_body_id, RayTraceResult& results_out) const;
	bool doesRayHitAnything(const Vec4f& origin, const Vec4f& dir, float max_t) const;

	// Extension (not in the reference): many rays in ONE device launch.  A single traceRay costs a kernel launch and a host sync
	// (tens of microseconds) however cheap the ray; ParticleManager::think (ParticleManager.cpp:145-274) traces one ray per particle,
	// up to 2048 per frame, so its loop should collect the rays, call this once and then react to the results.
	struct RayQuery { Vec4f origin, dir; float max_t; JPH::BodyID ignore_body_id; bool collidable_only; };
	void traceRays(const std::vector<RayQuery>& rays, std::vector<RayTraceResult>& results_out) const;

	// What GUIClient.cpp:6581-6690 does through physics_system->GetBodyInterface(): copy the poses of the activated
	// objects back into PhysicsObject::pos / rot (one batched device read instead of one Jolt call per object).
	void readBackActivatedObjectTransforms();
	// BodyInterface::AddForce / AddTorque / AddForce(at point), used by HoverCarPhysics.cpp:113-348, BoatPhysics.cpp:221-267
	void addForce(PhysicsObject& object, const Vec4f& force);
	void addForceAtPoint(PhysicsObject& object, const Vec4f& force, const Vec4f& point);
	void addTorque(PhysicsObject& object, const Vec4f& torque);

public:
	mutable Mutex activated_obs_mutex;
	HashSet<PhysicsObject*> activated_obs GUARDED_BY(activated_obs_mutex);
	HashSet<PhysicsObject*> newly_activated_obs GUARDED_BY(activated_obs_mutex);
	PhysicsWorldEventListener* event_listener;

	sgp_world* world;                   // the C-ABI handle (in place of temp_allocator / job_system)
	JPH::PhysicsSystem* physics_system; // look-alike carrying GetBodyInterface() for the controllers that reach around the facade (PhysicsWorld.h:204)

private:
	void addCompoundObject(const Reference<PhysicsObject>& object, struct sgp_body_desc d);
	void drainActivationEvents();
	bool water_buoyancy_enabled;
	float water_z;
	glare::TaskManager* task_manager;
	glare::StackAllocator* stack_allocator;
	std::vector<PhysicsObject*> id_to_ob;
};

inline void checkRemoveObAndSetRefToNull(PhysicsWorld& physics_world, Reference<PhysicsObject>& physics_object)
{
	if (physics_object) { physics_world.removeObject(physics_object); physics_object = nullptr; }
}
