// A skinned BatchedMesh through createJoltShapeForBatchedMesh (PhysicsWorld.cpp:885-947: "if mesh has joints and weights, take the skinning
// transform into account"): a bar of two unit cubes stacked along z, the upper cube bound to a joint that is bent over by 90 degrees about y.
// The static triangle shape and the dynamic convex hull must both be those of the BENT bar; a mesh type without animation data, or without
// joint nodes, keeps its bind pose.
#include "PhysicsWorld.h"
#include <utils/Exception.h>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>

struct TestSkinnedMesh
{
	enum ComponentType { ComponentType_Float, ComponentType_Half, ComponentType_UInt8, ComponentType_UInt16, ComponentType_UInt32, ComponentType_PackedNormal };
	enum VertAttributeType { VertAttribute_Position, VertAttribute_Normal, VertAttribute_Joints, VertAttribute_Weights };
	struct VertAttribute { VertAttributeType type; ComponentType component_type; size_t offset_B; };
	struct IndicesBatch { uint32 indices_start, num_indices, material_index; };
	struct Bounds { Vec4f min_, max_; Vec4f span() const { return max_ - min_; } };
	struct AnimationNodeData { Vec4f trans; Quatf rot; Vec4f scale; int parent_index; Matrix4f inverse_bind_matrix; };
	struct AnimationData { std::vector<AnimationNodeData> nodes; std::vector<int> sorted_nodes, joint_nodes; };
	std::vector<VertAttribute> vert_attributes; std::vector<IndicesBatch> batches; std::vector<uint8_t> vertex_data, index_data;
	ComponentType index_type = ComponentType_UInt16; Bounds aabb_os; size_t vert_size = 0;
	AnimationData animation_data;
	size_t vertexSize() const { return vert_size; }
	size_t numVerts() const { return vert_size ? vertex_data.size() / vert_size : 0; }
	size_t numIndices() const { return index_data.size() / 2; }
	const VertAttribute* findAttribute(VertAttributeType t) const { for (const VertAttribute& a : vert_attributes) if (a.type == t) return &a; return nullptr; }
};

// vertex layout: float position (12 B) | uint8 joints x 4 (4 B) | weights x 4 (uint8: 4 B, or float: 16 B)
static TestSkinnedMesh makeBar(bool float_weights)
{
	TestSkinnedMesh m;
	const size_t wsize = float_weights ? 16 : 4;
	m.vert_size = 12 + 4 + wsize;
	m.vert_attributes.push_back({ TestSkinnedMesh::VertAttribute_Position, TestSkinnedMesh::ComponentType_Float, 0 });
	m.vert_attributes.push_back({ TestSkinnedMesh::VertAttribute_Joints, TestSkinnedMesh::ComponentType_UInt8, 12 });
	m.vert_attributes.push_back({ TestSkinnedMesh::VertAttribute_Weights, float_weights ? TestSkinnedMesh::ComponentType_Float : TestSkinnedMesh::ComponentType_UInt8, 16 });
	m.aabb_os.min_ = Vec4f(-0.5f, -0.5f, 0.f, 1.f); m.aabb_os.max_ = Vec4f(0.5f, 0.5f, 2.f, 1.f);
	static const int quads[6][4] = { { 0, 2, 3, 1 }, { 4, 5, 7, 6 }, { 0, 1, 5, 4 }, { 2, 6, 7, 3 }, { 0, 4, 6, 2 }, { 1, 3, 7, 5 } };      // outward-facing, corner index = x + 2 y + 4 z
	std::vector<uint16_t> idx;
	for (int cube = 0; cube < 2; ++cube) {
		for (int c = 0; c < 8; ++c) {
			const float p[3] = { (c & 1) ? 0.5f : -0.5f, (c & 2) ? 0.5f : -0.5f, (float)cube + ((c & 4) ? 1.f : 0.f) };
			std::vector<uint8_t> v(m.vert_size, 0);
			memcpy(&v[0], p, 12);
			v[12] = (uint8_t)cube;                     // first influence: joint 0 for the lower cube, joint 1 for the upper; the other three carry no weight
			if (float_weights) { const float w[4] = { 1.f, 0.f, 0.f, 0.f }; memcpy(&v[16], w, 16); } else v[16] = 255;
			m.vertex_data.insert(m.vertex_data.end(), v.begin(), v.end());
		}
		for (int q = 0; q < 6; ++q) { const int* s = quads[q]; const int b = 8 * cube; for (int k : { s[0], s[1], s[2], s[0], s[2], s[3] }) idx.push_back((uint16_t)(b + k)); }
	}
	m.index_data.resize(idx.size() * 2); memcpy(m.index_data.data(), idx.data(), m.index_data.size());
	m.batches.push_back({ 0u, (uint32)idx.size(), 0u });
	// node 0: the root at the origin; node 1: its child at the height of the joint (z = 1), bent over by 90 degrees about y.  The bind pose had node 1
	// at (0, 0, 1) unrotated, hence its inverse bind matrix is the translation by (0, 0, -1).
	TestSkinnedMesh::AnimationNodeData root = { Vec4f(0, 0, 0, 0), Quatf::identity(), Vec4f(1, 1, 1, 0), -1, Matrix4f::identity() };
	TestSkinnedMesh::AnimationNodeData elbow = { Vec4f(0, 0, 1, 0), Quatf::fromAxisAndAngle(Vec4f(0, 1, 0, 0), 1.5707963f), Vec4f(1, 1, 1, 0), 0, Matrix4f::identity() };
	elbow.inverse_bind_matrix.setColumn(3, Vec4f(0, 0, -1, 1));
	m.animation_data.nodes = { root, elbow };
	m.animation_data.sorted_nodes = { 0, 1 };
	m.animation_data.joint_nodes = { 0, 1 };
	return m;
}

static float rayDown(PhysicsWorld& world, float x, float y)
{
	RayTraceResult r;
	world.traceRay(Vec4f(x, y, 10.f, 1.f), Vec4f(0, 0, -1, 0), 100.f, JPH::BodyID(), r);
	return r.hit_object ? 10.f - r.hit_t : -1000.f;
}

int main()
{
	try {
		PhysicsWorld::init();
		for (int float_weights = 0; float_weights < 2; ++float_weights) {
			Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
			const TestSkinnedMesh bent = makeBar(float_weights != 0);
			Reference<PhysicsObject> ob = new PhysicsObject(true, PhysicsWorld::createJoltShapeForBatchedMesh(bent, /*build_dynamic_physics_ob=*/false, nullptr), nullptr, 0);
			ob->pos = Vec4f(0, 0, 0, 1);
			world->addObject(ob);
			// the upper cube now lies along +x: x in [0, 1], z in [0.5, 1.5]; the lower cube is where it was
			const float z_arm = rayDown(*world, 0.8f, 0.f), z_stub = rayDown(*world, -0.3f, 0.f), z_beyond = rayDown(*world, 1.2f, 0.f);
			printf("weights %s: top of the bent arm %.4f (1.5), of the lower cube %.4f (1.0), beyond the arm %.1f (miss)\n", float_weights ? "float" : "uint8", z_arm, z_stub, z_beyond);
			if (std::fabs(z_arm - 1.5f) > 2e-3f || std::fabs(z_stub - 1.0f) > 2e-3f || z_beyond > -999.f) return 2;

			// the same mesh without joint nodes: the bind pose (a bar 2 m tall, nothing at x = 0.8)
			TestSkinnedMesh straight = bent; straight.animation_data.joint_nodes.clear();
			Reference<PhysicsObject> ob2 = new PhysicsObject(true, PhysicsWorld::createJoltShapeForBatchedMesh(straight, false, nullptr), nullptr, 0);
			ob2->pos = Vec4f(20, 0, 0, 1);
			world->addObject(ob2);
			const float z_top = rayDown(*world, 19.7f, 0.f), z_none = rayDown(*world, 20.8f, 0.f);
			printf("   bind pose: top %.4f (2.0), at x = 0.8 %.1f (miss)\n", z_top, z_none);
			if (std::fabs(z_top - 2.0f) > 2e-3f || z_none > -999.f) return 3;

			// dynamic object: the convex hull of the POSED vertices
			Reference<PhysicsObject> dyn = new PhysicsObject(true, PhysicsWorld::createJoltShapeForBatchedMesh(bent, /*build_dynamic_physics_ob=*/true, nullptr), nullptr, 0);
			dyn->pos = Vec4f(40, 0, 5, 1); dyn->mass = 100.f; dyn->motion_type = PhysicsObject::MotionType_dynamic;
			world->addObject(dyn);
			Reference<PhysicsObject> dyn_straight = new PhysicsObject(true, PhysicsWorld::createJoltShapeForBatchedMesh(straight, true, nullptr), nullptr, 0);
			dyn_straight->pos = Vec4f(60, 0, 5, 1); dyn_straight->mass = 100.f; dyn_straight->motion_type = PhysicsObject::MotionType_dynamic;
			world->addObject(dyn_straight);
			// (before any step: the bodies are where they were put.  The hull of the bent bar reaches x = 1 at z = 1.5 above the object's origin; the straight one ends at x = 0.5)
			const float zd_bent = rayDown(*world, 40.8f, 0.f), zd_straight = rayDown(*world, 60.8f, 0.f), zd_straight_top = rayDown(*world, 60.2f, 0.f);
			const JPH::Body* b1 = world->physics_system->GetBodyLockInterface().TryGetBody(dyn->jolt_body_id);
			const JPH::Body* b2 = world->physics_system->GetBodyLockInterface().TryGetBody(dyn_straight->jolt_body_id);
			const float v_bent = b1 ? b1->GetShape()->GetVolume() : 0.f, v_straight = b2 ? b2->GetShape()->GetVolume() : 0.f;
			printf("   dynamic hulls: bent, top at x = 0.8: %.4f (6.5); straight: %.1f (miss) and %.4f (7.0) at x = 0.2; volumes %.4f / %.4f (2.0 both: the hull of this L happens to have the bar's volume)\n",
				zd_bent, zd_straight, zd_straight_top, v_bent, v_straight);
			if (std::fabs(zd_bent - 6.5f) > 5e-3f || zd_straight > -999.f || std::fabs(zd_straight_top - 7.0f) > 5e-3f) return 4;
			if (std::fabs(v_straight - 2.0f) > 2e-3f || std::fabs(v_bent - 2.0f) > 2e-3f) return 6;
		}
		// an out-of-range joint index is an error, not a read past the matrices
		TestSkinnedMesh bad = makeBar(true); bad.vertex_data[12] = 7;
		bool threw = false;
		try { PhysicsWorld::createJoltShapeForBatchedMesh(bad, false, nullptr); } catch (glare::Exception&) { threw = true; }
		if (!threw) return 5;
		printf("ok\n");
		return 0;
	}
	catch (glare::Exception& e) { printf("exception: %s\n", e.what().c_str()); return 1; }
}
