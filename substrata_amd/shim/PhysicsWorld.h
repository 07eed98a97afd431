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
#include <utils/Array2D.h>
#include <utils/Exception.h>
#include <cstring>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/Body/BodyID.h>
#include <Jolt/Physics/Body/BodyActivationListener.h>
#include <Jolt/Physics/Collision/ContactListener.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <string>
#include <vector>
#include <cstdint>
#include <algorithm>
#include <type_traits>
#include <utility>

namespace glare { class TaskManager; class StackAllocator; class Allocator; }
struct sgp_world;
struct sgp_body_event; struct sgp_contact_event; struct sgp_body_state;
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
	// ---- the reference's own builder signatures (PhysicsWorld.h:122-127).  Indigo::Mesh, BatchedMesh and js::Vector are glare-core / indigo
	// types that are not part of this tree, so the three mesh builders are templates over ANY type exposing the members the reference's code
	// reads (PhysicsWorld.cpp:735-866, 868-1084); with the real headers on the include path the callers' statements -- ModelLoading.cpp:1175,
	// 1686, MeshBuilding.cpp:148,374 -- compile as written.  They reduce to the array builders above.
	//   Indigo::Mesh:  vert_positions[i].{x,y,z}; triangles[i].{vertex_indices[3], tri_mat_index}; quads[i].{vertex_indices[4], mat_index}
	//   BatchedMesh:   vertexSize(), numVerts(), numIndices(), findAttribute(VertAttribute_Position) -> {component_type, offset_B}, vertex_data,
	//                  aabb_os.{min_, span()} (uint16 positions are dequantised over it), index_data, index_type, batches[b].{indices_start,
	//                  num_indices, material_index}.  A skinned mesh (Joints + Weights attributes and animation_data.{nodes, sorted_nodes,
	//                  joint_nodes}) is built in the pose its animation nodes hold, like the reference (PhysicsWorld.cpp:885-947, 814-866):
	//                  round 4 -- a mesh type without an animation_data member is taken as it is.
	template <class IndigoMeshT>
	static PhysicsShape createJoltShapeForIndigoMesh(const IndigoMeshT& mesh, bool build_dynamic_physics_ob, glare::Allocator* mem_allocator = nullptr)
	{
		(void)mem_allocator;
		std::vector<Vec3f> verts(mesh.vert_positions.size());
		for (size_t i = 0; i < verts.size(); ++i) verts[i] = Vec3f(mesh.vert_positions[i].x, mesh.vert_positions[i].y, mesh.vert_positions[i].z);
		if (build_dynamic_physics_ob) return createConvexHullShape(verts);      // Jolt has no dynamic triangle meshes: the convex hull of the vertices
		std::vector<uint32> tris; std::vector<uint32> mats;
		tris.reserve(3 * (mesh.triangles.size() + 2 * mesh.quads.size())); mats.reserve(mesh.triangles.size() + 2 * mesh.quads.size());
		for (size_t i = 0; i < mesh.triangles.size(); ++i) {
			for (int k = 0; k < 3; ++k) tris.push_back((uint32)mesh.triangles[i].vertex_indices[k]);
			mats.push_back((uint32)mesh.triangles[i].tri_mat_index);
		}
		for (size_t i = 0; i < mesh.quads.size(); ++i) {       // a quad = the triangles (0, 1, 2) and (0, 2, 3), both with the quad's material
			const auto& q = mesh.quads[i];
			const uint32 a = (uint32)q.vertex_indices[0], b = (uint32)q.vertex_indices[1], c = (uint32)q.vertex_indices[2], d = (uint32)q.vertex_indices[3];
			tris.insert(tris.end(), { a, b, c, a, c, d });
			mats.push_back((uint32)q.mat_index); mats.push_back((uint32)q.mat_index);
		}
		return createMeshShape(verts, tris, &mats);
	}
	// Skinning of a BatchedMesh before its shape is built (PhysicsWorld.cpp:885-947: "if mesh has joints and weights, take the skinning transform into
	// account"): node -> object matrices down the hierarchy (T R S per node, parents first: sorted_nodes), joint matrix = node matrix x inverse bind
	// matrix, vertex = sum over its four influences of weight x (joint matrix x position); weights are uint8 / uint16 (normalised) or float.
	template <class M, class = void> struct HasAnimationData : std::false_type {};
	template <class M> struct HasAnimationData<M, std::void_t<decltype(std::declval<const M&>().animation_data.joint_nodes)>> : std::true_type {};
	template <class BatchedMeshT>
	static void applySkinTransforms(const BatchedMeshT& mesh, std::vector<Vec3f>& verts)
	{
		if constexpr (HasAnimationData<BatchedMeshT>::value) {
			const auto* joints_attr = mesh.findAttribute(BatchedMeshT::VertAttribute_Joints);
			const auto* weights_attr = mesh.findAttribute(BatchedMeshT::VertAttribute_Weights);
			const auto& anim = mesh.animation_data;
			if (!joints_attr || !weights_attr || anim.joint_nodes.empty()) return;
			const int num_nodes = (int)anim.nodes.size();
			std::vector<Matrix4f> node_to_ob((size_t)num_nodes, Matrix4f::identity());
			for (size_t k = 0; k < anim.sorted_nodes.size(); ++k) {
				const int n = (int)anim.sorted_nodes[k];
				if (n < 0 || n >= num_nodes) throw glare::Exception("createJoltShapeForBatchedMesh: animation node index out of range.");
				const auto& node = anim.nodes[(size_t)n];
				Matrix4f local = node.rot.toMatrix();
				for (int c = 0; c < 3; ++c) { const Vec4f col = local.getColumn(c); local.setColumn(c, Vec4f(col[0] * node.scale[c], col[1] * node.scale[c], col[2] * node.scale[c], 0.f)); }
				local.setColumn(3, Vec4f(node.trans[0], node.trans[1], node.trans[2], 1.f));
				const int parent = (int)node.parent_index;
				if (parent < -1 || parent >= num_nodes) throw glare::Exception("createJoltShapeForBatchedMesh: animation node parent out of range.");
				node_to_ob[(size_t)n] = parent < 0 ? local : node_to_ob[(size_t)parent] * local;
			}
			std::vector<Matrix4f> joint_to_ob(anim.joint_nodes.size());
			for (size_t j = 0; j < joint_to_ob.size(); ++j) {
				const int n = (int)anim.joint_nodes[j];
				if (n < 0 || n >= num_nodes) throw glare::Exception("createJoltShapeForBatchedMesh: joint node index out of range.");
				joint_to_ob[j] = node_to_ob[(size_t)n] * anim.nodes[(size_t)n].inverse_bind_matrix;
			}
			const bool joints_u8 = joints_attr->component_type == BatchedMeshT::ComponentType_UInt8, joints_u16 = joints_attr->component_type == BatchedMeshT::ComponentType_UInt16;
			const bool w_u8 = weights_attr->component_type == BatchedMeshT::ComponentType_UInt8, w_u16 = weights_attr->component_type == BatchedMeshT::ComponentType_UInt16;
			const bool w_f = weights_attr->component_type == BatchedMeshT::ComponentType_Float;
			if (!(joints_u8 || joints_u16) || !(w_u8 || w_u16 || w_f)) throw glare::Exception("createJoltShapeForBatchedMesh: unsupported joint / weight component type.");
			const size_t stride = mesh.vertexSize();
			const unsigned char* data = (const unsigned char*)mesh.vertex_data.data();
			if (!verts.empty() && (verts.size() - 1) * stride + std::max(joints_attr->offset_B + (joints_u8 ? 4u : 8u), weights_attr->offset_B + (w_u8 ? 4u : (w_u16 ? 8u : 16u))) > mesh.vertex_data.size())
				throw glare::Exception("createJoltShapeForBatchedMesh: joint / weight attributes run past the vertex data.");
			for (size_t i = 0; i < verts.size(); ++i) {
				uint32 joint[4]; float weight[4];
				const unsigned char* pj = data + i * stride + joints_attr->offset_B;
				const unsigned char* pw = data + i * stride + weights_attr->offset_B;
				for (int z = 0; z < 4; ++z) {
					if (joints_u8) joint[z] = pj[z]; else { uint16_t t; memcpy(&t, pj + 2 * z, 2); joint[z] = t; }
					if (w_u8) weight[z] = (float)pw[z] * (1.0f / 255.f);
					else if (w_u16) { uint16_t t; memcpy(&t, pw + 2 * z, 2); weight[z] = (float)t * (1.0f / 65535.f); }
					else memcpy(&weight[z], pw + 4 * z, 4);
					if (joint[z] >= joint_to_ob.size()) throw glare::Exception("createJoltShapeForBatchedMesh: vertex joint index out of range.");
				}
				const Vec4f p(verts[i][0], verts[i][1], verts[i][2], 1.f);
				Vec4f acc(0.f, 0.f, 0.f, 0.f);
				for (int z = 0; z < 4; ++z) { const Vec4f q = joint_to_ob[joint[z]] * p; for (int c = 0; c < 4; ++c) acc[c] += q[c] * weight[z]; }
				verts[i] = Vec3f(acc[0], acc[1], acc[2]);
			}
		} else { (void)mesh; (void)verts; }
	}
	template <class BatchedMeshT, class BoolVectorT = std::vector<bool>>
	static PhysicsShape createJoltShapeForBatchedMesh(const BatchedMeshT& mesh, bool build_dynamic_physics_ob, glare::Allocator* mem_allocator = nullptr,
		const BoolVectorT* create_tris_for_mat = nullptr)
	{
		(void)mem_allocator;
		const size_t vert_size_B = mesh.vertexSize(), num_verts = mesh.numVerts();
		const auto* pos_attr = mesh.findAttribute(BatchedMeshT::VertAttribute_Position);
		if (!pos_attr) throw glare::Exception("Pos attribute not present.");
		if (!(pos_attr->component_type == BatchedMeshT::ComponentType_Float || pos_attr->component_type == BatchedMeshT::ComponentType_UInt16))
			throw glare::Exception("PhysicsWorld::createJoltShapeForBatchedMesh(): Pos attribute must have float or uint16 type.");
		const bool pos_is_float = pos_attr->component_type == BatchedMeshT::ComponentType_Float;
		const Vec4f span = mesh.aabb_os.span(), lo = mesh.aabb_os.min_;
		std::vector<Vec3f> verts(num_verts);
		const unsigned char* src = (const unsigned char*)mesh.vertex_data.data();
		for (size_t i = 0; i < num_verts; ++i) {
			const unsigned char* p = src + pos_attr->offset_B + i * vert_size_B;
			if (pos_is_float) { float f[3]; memcpy(f, p, 12); verts[i] = Vec3f(f[0], f[1], f[2]); }
			else { uint16_t q[3]; memcpy(q, p, 6); verts[i] = Vec3f(lo[0] + span[0] / 65535.f * (float)q[0], lo[1] + span[1] / 65535.f * (float)q[1], lo[2] + span[2] / 65535.f * (float)q[2]); }
		}
		applySkinTransforms(mesh, verts);
		if (build_dynamic_physics_ob) return createConvexHullShape(verts);
		std::vector<uint32> tris, mats;
		const unsigned char* idx = (const unsigned char*)mesh.index_data.data();
		for (size_t b = 0; b < mesh.batches.size(); ++b) {
			const uint32 mat_index = (uint32)mesh.batches[b].material_index;
			// "should physics triangles be created for this material?" -- materials beyond the vector are kept (PhysicsWorld.cpp:1028)
			if (create_tris_for_mat && mat_index < create_tris_for_mat->size() && !(*create_tris_for_mat)[mat_index]) continue;
			const size_t i_begin = mesh.batches[b].indices_start, i_end = i_begin + mesh.batches[b].num_indices / 3 * 3;
			for (size_t i = i_begin; i < i_end; ++i) {
				uint32 v;
				if (mesh.index_type == BatchedMeshT::ComponentType_UInt8) v = idx[i];
				else if (mesh.index_type == BatchedMeshT::ComponentType_UInt16) { uint16_t t; memcpy(&t, idx + 2 * i, 2); v = t; }
				else if (mesh.index_type == BatchedMeshT::ComponentType_UInt32) memcpy(&v, idx + 4 * i, 4);
				else throw glare::Exception("Invalid index type.");
				tris.push_back(v);
			}
			mats.insert(mats.end(), (i_end - i_begin) / 3, mat_index);
		}
		return createMeshShape(verts, tris, &mats);
	}
	// PhysicsWorld.h:127 / .cpp:1086-1119: heightfield.getWidth() samples per side (vert_res <= width), Jolt's y-up shape space
	static PhysicsShape createJoltHeightFieldShape(int vert_res, const Array2D<float>& heightfield, float quad_w)
	{
		return createJoltHeightFieldShape(vert_res, std::vector<float>(heightfield.getData(), heightfield.getData() + heightfield.getWidth() * heightfield.getHeight()), (int)heightfield.getWidth(), quad_w);
	}
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
	// per-step scratch of think(): kept between calls, sized to what the steps really produce (never to the world's capacity, never zero-filled)
	std::vector<struct sgp_body_event> body_event_buf;
	std::vector<struct sgp_contact_event> contact_event_buf;
	std::vector<uint32_t> water_ids; std::vector<PhysicsObject*> water_obs; std::vector<struct sgp_body_state> water_states;
};

inline void checkRemoveObAndSetRefToNull(PhysicsWorld& physics_world, Reference<PhysicsObject>& physics_object)
{
	if (physics_object) { physics_world.removeObject(physics_object); physics_object = nullptr; }
}
