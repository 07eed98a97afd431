// A small "real world" through the facade: a height-field terrain (createJoltHeightFieldShape, the way TerrainSystem.cpp:1300 builds its
// chunks; y-up shape space turned upright by the object's rotation), a static mesh building (createMeshShape, what
// createJoltShapeForBatchedMesh gives a static object), dynamic boxes / spheres / a hull dropped on both, the player walking up the
// terrain and into the building's wall, rays against the meshes.
#include "PhysicsWorld.h"
#include "JoltUtils.h"
#include <utils/Exception.h>
#include <Jolt/Jolt.h>
#include <Jolt/Physics/Collision/ObjectLayer.h>
#include <Jolt/Physics/Character/Character.h>
#include <Jolt/Physics/Character/CharacterVirtual.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <Jolt/Physics/Collision/Shape/CapsuleShape.h>
#include <Jolt/Physics/Collision/Shape/RotatedTranslatedShape.h>
#include <cstdio>
#include <cmath>
#include <string>
#include <vector>
#include <algorithm>
#include <cstring>

// Stand-ins with the members the reference's builders read from glare-core's BatchedMesh and indigo's Indigo::Mesh (absent from this tree);
// the builders are called with the reference's own signatures (PhysicsWorld.h:122-127).
struct TestBatchedMesh
{
	enum ComponentType { ComponentType_Float, ComponentType_Half, ComponentType_UInt8, ComponentType_UInt16, ComponentType_UInt32, ComponentType_PackedNormal };
	enum VertAttributeType { VertAttribute_Position, VertAttribute_Normal, VertAttribute_Joints, VertAttribute_Weights };
	struct VertAttribute { VertAttributeType type; ComponentType component_type; size_t offset_B; };
	struct IndicesBatch { uint32 indices_start, num_indices, material_index; };
	struct Bounds { Vec4f min_, max_; Vec4f span() const { return max_ - min_; } };
	std::vector<VertAttribute> vert_attributes; std::vector<IndicesBatch> batches; std::vector<uint8_t> vertex_data, index_data;
	ComponentType index_type = ComponentType_UInt16; Bounds aabb_os; size_t vert_size = 0;
	size_t vertexSize() const { return vert_size; }
	size_t numVerts() const { return vert_size ? vertex_data.size() / vert_size : 0; }
	size_t numIndices() const { return index_data.size() / (index_type == ComponentType_UInt8 ? 1 : (index_type == ComponentType_UInt16 ? 2 : 4)); }
	const VertAttribute* findAttribute(VertAttributeType t) const { for (const VertAttribute& a : vert_attributes) if (a.type == t) return &a; return nullptr; }
};
// quantised uint16 positions over the bounds (what BatchedMesh files usually hold), uint16 indices, one batch per material
static TestBatchedMesh makeBatchedMesh(const std::vector<Vec3f>& v, const std::vector<uint32>& tris, const std::vector<uint32>& tri_mats, const Vec4f& lo, const Vec4f& hi)
{
	TestBatchedMesh m;
	m.vert_size = 8; m.vert_attributes.push_back({ TestBatchedMesh::VertAttribute_Position, TestBatchedMesh::ComponentType_UInt16, 0 });
	m.aabb_os.min_ = lo; m.aabb_os.max_ = hi;
	m.vertex_data.resize(v.size() * 8);
	for (size_t i = 0; i < v.size(); ++i) for (int k = 0; k < 3; ++k) { const uint16_t q = (uint16_t)std::lround((v[i][k] - lo[k]) / (hi[k] - lo[k]) * 65535.f); memcpy(&m.vertex_data[i * 8 + 2 * k], &q, 2); }
	uint32 max_mat = 0; for (uint32 x : tri_mats) max_mat = std::max(max_mat, x);
	for (uint32 mat = 0; mat <= max_mat; ++mat) {
		TestBatchedMesh::IndicesBatch b = { (uint32)(m.index_data.size() / 2), 0, mat };
		for (size_t t = 0; t < tri_mats.size(); ++t) if (tri_mats[t] == mat) for (int k = 0; k < 3; ++k) { const uint16_t ix = (uint16_t)tris[3 * t + k]; m.index_data.push_back((uint8_t)(ix & 0xFF)); m.index_data.push_back((uint8_t)(ix >> 8)); b.num_indices++; }
		if (b.num_indices) m.batches.push_back(b);
	}
	return m;
}
struct TestIndigoMesh
{
	struct V3 { float x, y, z; }; struct Triangle { uint32 vertex_indices[3]; uint32 uv_indices[3]; uint32 tri_mat_index; }; struct Quad { uint32 vertex_indices[4]; uint32 uv_indices[4]; uint32 mat_index; };
	std::vector<V3> vert_positions; std::vector<Triangle> triangles; std::vector<Quad> quads;
};

static float terrainHeight(float x, float y) { return 0.8f * std::sin(0.25f * x) * std::cos(0.2f * y) + 0.05f * x; }

int main()
{
	try {
		PhysicsWorld::init();
		Reference<PhysicsWorld> world = new PhysicsWorld(nullptr, nullptr);
		// height field: 64 x 64 samples, 1 m quads.  Shape space (X, height, Z - 63); the object rotation maps y -> z (up), z -> -y,
		// so world x = X, world y = 63 - Z... = sample z index counted downwards from the object's origin
		const int W = 64; const float quad_w = 1.0f;
		Array2D<float> heights(W, W);                                                    // (TerrainSystem.cpp:1300: createJoltHeightFieldShape(res, heightfield, quad_w))
		for (int z = 0; z < W; ++z) for (int x = 0; x < W; ++x) heights.elem(x, z) = terrainHeight((float)x * quad_w - 32.f, 31.f - (float)z * quad_w + 0.f);
		Reference<PhysicsObject> terrain = new PhysicsObject(true, PhysicsWorld::createJoltHeightFieldShape(W, heights, quad_w), nullptr, 0);
		terrain->rot = Quatf::fromAxisAndAngle(Vec4f(1, 0, 0, 0), 1.5707963f);          // y-up shape space -> z-up world
		terrain->pos = Vec4f(-32.f, -32.f, 0.f, 1);                                      // world x = X - 32, world y = -(Z - 63) - 32 = 31 - z index
		world->addObject(terrain);
		// a building: an open box (floor + 4 walls, normals outwards) 6 x 6 x 4 at (10, 10)
		std::vector<Vec3f> bv; std::vector<uint32> bt;
		const float h = 3.f, z0 = -2.f, z1 = 6.f;
		const float cx[4] = { -h, h, h, -h }, cy[4] = { -h, -h, h, h };
		for (int i = 0; i < 4; ++i) { bv.push_back(Vec3f(cx[i], cy[i], z0)); bv.push_back(Vec3f(cx[i], cy[i], z1)); }
		for (int i = 0; i < 4; ++i) { const uint32 a = 2 * i, b = 2 * ((i + 1) % 4); bt.push_back(a); bt.push_back(b); bt.push_back(b + 1); bt.push_back(a); bt.push_back(b + 1); bt.push_back(a + 1); }   // outward-facing walls
		bt.push_back(1); bt.push_back(3); bt.push_back(5); bt.push_back(1); bt.push_back(5); bt.push_back(7);                                                                                          // roof, facing up
		// per-triangle material indices, as createJoltShapeForBatchedMesh stores them from the mesh batches (PhysicsWorld.cpp:1032-1060): wall i -> i + 1, roof -> 5
		std::vector<uint32> bmat;
		for (int i = 0; i < 4; ++i) { bmat.push_back(i + 1); bmat.push_back(i + 1); }
		bmat.push_back(5); bmat.push_back(5);
		// ... as a BatchedMesh (uint16 positions quantised over its bounds, one index batch per material), through the reference's signature
		// (ModelLoading.cpp:1686: createJoltShapeForBatchedMesh(*batched_mesh, /*is dynamic=*/false, mem_allocator))
		const TestBatchedMesh building_mesh = makeBatchedMesh(bv, bt, bmat, Vec4f(-h, -h, z0, 1), Vec4f(h, h, z1, 1));
		Reference<PhysicsObject> building = new PhysicsObject(true, PhysicsWorld::createJoltShapeForBatchedMesh(building_mesh, /*build_dynamic_physics_ob=*/false, /*mem_allocator=*/nullptr), nullptr, 0);
		building->pos = Vec4f(10.f, 10.f, 0.f, 1);
		world->addObject(building);
		// the same mesh with create_tris_for_mat[5] = false (MeshBuilding.cpp:392-393): no roof triangles
		std::vector<bool> create_tris_for_mat(6, true); create_tris_for_mat[5] = false;
		Reference<PhysicsObject> roofless = new PhysicsObject(true, PhysicsWorld::createJoltShapeForBatchedMesh(building_mesh, false, nullptr, &create_tris_for_mat), nullptr, 0);
		roofless->pos = Vec4f(-20.f, -20.f, 0.f, 1);
		world->addObject(roofless);

		// things falling on the terrain and on the building's roof
		std::vector<Reference<PhysicsObject>> obs;
		for (int i = 0; i < 12; ++i) {
			Reference<PhysicsObject> ob = new PhysicsObject(true);
			if (i % 3 == 0) ob->is_sphere = true; else ob->is_cube = true;
			ob->scale = Vec3f(0.8f); ob->mass = 20.f; ob->motion_type = PhysicsObject::MotionType_dynamic;
			const float px = (i < 6) ? (-10.f + 3.5f * i) : (8.5f + 0.9f * (i - 6)), py = (i < 6) ? (-6.f + 2.f * i) : 10.f;
			ob->pos = Vec4f(px, py, (i < 6 ? terrainHeight(px, py) : z1) + 3.f + 0.5f * i, 1);
			world->addObject(ob); world->activateObject(ob); obs.push_back(ob);
		}
		for (int s = 0; s < 420; ++s) world->think(1.0 / 60.0);
		world->readBackActivatedObjectTransforms();
		bool ok = true;
		for (int i = 0; i < 12; ++i) {
			const Vec4f p = world->getPosInJolt(obs[i]);
			const float floor_z = (i < 6) ? terrainHeight(p[0], p[1]) : ((std::fabs(p[0] - 10.f) < 3.f && std::fabs(p[1] - 10.f) < 3.f) ? z1 : terrainHeight(p[0], p[1]));
			const bool fine = p[2] > floor_z + 0.25f && p[2] < floor_z + 1.2f;
			if (!fine) { printf("object %d at %.2f %.2f %.2f, surface %.2f\n", i, p[0], p[1], p[2], floor_z); ok = false; }
		}
		// rays: down onto the terrain, sideways into the building's wall (hit from outside), from inside the building (back faces: no hit on the wall)
		RayTraceResult r;
		world->traceRay(Vec4f(-5, 3, 20, 1), Vec4f(0, 0, -1, 0), 100.f, JPH::BodyID(), r);
		ok = ok && r.hit_object == terrain.ptr() && std::fabs((20.f - r.hit_t) - terrainHeight(-5, 3)) < 0.15f && r.hit_normal_ws[2] > 0.8f;
		world->traceRay(Vec4f(0, 10, 3, 1), Vec4f(1, 0, 0, 0), 100.f, JPH::BodyID(), r);
		ok = ok && r.hit_object == building.ptr() && std::fabs(r.hit_t - 7.f) < 1e-3f && r.hit_normal_ws[0] < -0.99f;
		bool mat_ok = r.hit_mat_index == 4 && r.coords.x == 0.f && r.coords.y == 0.f;      // the -x wall is wall 3 -> material 4; coords stay 0 like the reference (:1693)
		if (!mat_ok) printf("wall material: got %u\n", r.hit_mat_index);
		world->traceRay(Vec4f(10, 10, 3, 1), Vec4f(1, 0, 0, 0), 2.9f, JPH::BodyID(), r);
		ok = ok && r.hit_object == NULL;
		world->traceRay(Vec4f(10, 12, 30, 1), Vec4f(0, 0, -1, 0), 100.f, JPH::BodyID(), r);      // (beside the boxes lying on the roof along y = 10)
		const bool roof_ok = r.hit_object == building.ptr() && r.hit_mat_index == 5 && std::fabs(r.hit_t - 24.f) < 1e-3f;      // the roof
		if (!roof_ok) printf("roof: object %d material %u t %.3f\n", (int)(r.hit_object == building.ptr()), r.hit_mat_index, r.hit_t);
		world->traceRay(Vec4f(10, 1, 3, 1), Vec4f(0, 1, 0, 0), 100.f, JPH::BodyID(), r);
		const bool wall0_ok = r.hit_object == building.ptr() && r.hit_mat_index == 1;                                         // the -y wall is wall 0 -> material 1
		if (!wall0_ok) printf("-y wall: object %d material %u t %.3f\n", (int)(r.hit_object == building.ptr()), r.hit_mat_index, r.hit_t);
		world->traceRay(Vec4f(-5, 3, 20, 1), Vec4f(0, 0, -1, 0), 100.f, JPH::BodyID(), r);
		const bool terrain_ok = r.hit_object == terrain.ptr() && r.hit_mat_index == 0;                                        // no material array -> 0
		if (!terrain_ok) printf("terrain material: object %d material %u\n", (int)(r.hit_object == terrain.ptr()), r.hit_mat_index);
		// the roofless copy: a ray from above goes through where the roof would be and lands on the terrain; its walls are still there
		world->traceRay(Vec4f(-20, -20, 30, 1), Vec4f(0, 0, -1, 0), 100.f, JPH::BodyID(), r);
		const bool through_ok = r.hit_object == terrain.ptr();
		if (!through_ok) printf("roofless from above: terrain %d roofless %d t %.3f\n", (int)(r.hit_object == terrain.ptr()), (int)(r.hit_object == roofless.ptr()), r.hit_t);
		world->traceRay(Vec4f(-30, -20, 3, 1), Vec4f(1, 0, 0, 0), 100.f, JPH::BodyID(), r);
		const bool rwall_ok = r.hit_object == roofless.ptr() && r.hit_mat_index == 4 && std::fabs(r.hit_t - 7.f) < 1e-3f;
		if (!rwall_ok) printf("roofless wall: object %d terrain %d material %u t %.3f\n", (int)(r.hit_object == roofless.ptr()), (int)(r.hit_object == terrain.ptr()), r.hit_mat_index, r.hit_t);
		ok = ok && mat_ok && roof_ok && wall0_ok && terrain_ok && through_ok && rwall_ok;
		{
			// moving a static mesh object keeps it collidable (setNewObToWorldTransform, :546-604): the roofless building goes to (-20, 20), a ray finds
			// its wall there and no longer at the old place, and a box dropped onto the rim of a wall ... lands on the wall's top edge or beside it,
			// so instead drop a box INSIDE: it must come to rest on the terrain between the walls and stay inside them when pushed
			world->setNewObToWorldTransform(*roofless, Vec4f(-20.f, 20.f, 0.f, 1), Quatf::identity(), Vec4f(1.f, 1.f, 1.f, 0));
			world->traceRay(Vec4f(-30, 20, 3, 1), Vec4f(1, 0, 0, 0), 100.f, JPH::BodyID(), r);
			const bool moved = r.hit_object == roofless.ptr() && std::fabs(r.hit_t - 7.f) < 1e-3f && r.hit_mat_index == 4;
			world->traceRay(Vec4f(-30, -20, 3, 1), Vec4f(1, 0, 0, 0), 100.f, JPH::BodyID(), r);
			const bool gone = r.hit_object != roofless.ptr();
			// a ball thrown at the moved wall from outside bounces off it (the wall's collision moved with the object)
			Reference<PhysicsObject> ball = new PhysicsObject(true);
			ball->is_sphere = true; ball->scale = Vec3f(0.6f); ball->mass = 5.f; ball->motion_type = PhysicsObject::MotionType_dynamic; ball->restitution = 0.5f;
			ball->pos = Vec4f(-27.f, 20.f, 4.f, 1);
			world->addObject(ball); world->activateObject(ball);
			world->setNewObToWorldTransform(*ball, ball->pos, Quatf::identity(), Vec4f(12.f, 0, 2.f, 0), Vec4f(0.f));
			float max_x = -1e9f;
			for (int s = 0; s < 45; ++s) { world->think(1.0 / 60.0); max_x = std::fmax(max_x, world->getPosInJolt(ball)[0]); }
			const bool bounced = max_x < -23.f + 0.05f && max_x > -23.6f && world->getObjectLinearVelocity(*ball)[0] < 0.f;
			world->removeObject(ball);
			// a scale change swaps the shape instance (JPH::ScaledShape, :562-601): twice as large, the wall is met 3 m earlier
			world->setNewObToWorldTransform(*roofless, Vec4f(-20.f, 20.f, 0.f, 1), Quatf::identity(), Vec4f(2.f, 2.f, 2.f, 0));
			world->traceRay(Vec4f(-30, 20, 3, 1), Vec4f(1, 0, 0, 0), 100.f, JPH::BodyID(), r);
			const bool scaled = r.hit_object == roofless.ptr() && std::fabs(r.hit_t - 4.f) < 1e-3f && r.hit_mat_index == 4;
			if (!scaled) printf("scaled: object %d terrain %d material %u t %.3f\n", (int)(r.hit_object == roofless.ptr()), (int)(r.hit_object == terrain.ptr()), r.hit_mat_index, r.hit_t);
			if (!(moved && gone && bounced && scaled)) printf("moved mesh: moved %d gone %d bounced %d (max x %.3f) scaled %d\n", (int)moved, (int)gone, (int)bounced, max_x, (int)scaled);
			ok = ok && moved && gone && bounced && scaled;
		}
		{
			// a scripted object: mesh shape + MotionType_kinematic (the reference builds a MeshShape for everything that is not dynamic, :1290, and
			// moves it with moveKinematicObject, :706-731): a lift -- the roof-only slab of the building mesh -- rises with a box on it
			std::vector<bool> only_roof(6, false); only_roof[5] = true;
			Reference<PhysicsObject> lift = new PhysicsObject(true, PhysicsWorld::createJoltShapeForBatchedMesh(building_mesh, false, nullptr, &only_roof), nullptr, 0);
			lift->motion_type = PhysicsObject::MotionType_kinematic;
			lift->pos = Vec4f(30.f, -25.f, 0.f, 1);                    // the slab sits at z = pos.z + 6
			world->addObject(lift);
			const bool lift_added = !lift->jolt_body_id.IsInvalid();
			Reference<PhysicsObject> crate = new PhysicsObject(true);
			crate->is_cube = true; crate->scale = Vec3f(0.8f); crate->mass = 20.f; crate->motion_type = PhysicsObject::MotionType_dynamic; crate->friction = 0.8f;
			crate->pos = Vec4f(30.f, -25.f, 6.6f, 1);
			world->addObject(crate); world->activateObject(crate);
			for (int s = 0; s < 60; ++s) world->think(1.0 / 60.0);     // settles on the slab
			const float z_before = world->getPosInJolt(crate)[2];
			for (int s = 0; s < 120; ++s) {
				const float t = (float)(s + 1) / 60.f;
				world->moveKinematicObject(*lift, Vec4f(30.f + 0.5f * t, -25.f, 1.0f * t, 1), Quatf::identity(), 1.f / 60.f);
				world->think(1.0 / 60.0);
			}
			const Vec4f cp = world->getPosInJolt(crate);
			const bool rode = std::fabs(z_before - 6.4f) < 0.05f && std::fabs(cp[2] - 8.4f) < 0.08f && std::fabs(cp[0] - 31.0f) < 0.25f;      // up 2 m and 1 m along x, carried by friction
			if (!(lift_added && rode)) printf("kinematic mesh lift: added %d, crate z %.3f -> %.3f, x %.3f\n", (int)lift_added, z_before, cp[2], cp[0]);
			ok = ok && lift_added && rode;
		}
		// a decorated unit cube, as GUIClient builds for splat bounds (createScaledAndTranslatedShapeForShape(unit_cube_shape, aabb_min, aabb_span),
		// GUIClient.cpp:4807): the [0,1]^3 cube mesh mapped onto the box [(-1,-2,0), (1,2,1.5)] of an object floating at (-12, 12, 8)
		{
			std::vector<Vec3f> cv; std::vector<uint32> ct;
			for (int i = 0; i < 8; ++i) cv.push_back(Vec3f((float)(i & 1), (float)((i >> 1) & 1), (float)((i >> 2) & 1)));
			const uint32 quads[6][4] = { { 0, 2, 3, 1 }, { 4, 5, 7, 6 }, { 0, 1, 5, 4 }, { 2, 6, 7, 3 }, { 0, 4, 6, 2 }, { 1, 3, 7, 5 } };      // outward-facing
			for (int q = 0; q < 6; ++q) { ct.push_back(quads[q][0]); ct.push_back(quads[q][1]); ct.push_back(quads[q][2]); ct.push_back(quads[q][0]); ct.push_back(quads[q][2]); ct.push_back(quads[q][3]); }
			// (MeshBuilding.cpp:148: createJoltShapeForIndigoMesh(*indigo_mesh, /*build_dynamic_physics_ob=*/false) -- the cube as six quads)
			TestIndigoMesh cube_mesh;
			for (const Vec3f& v : cv) cube_mesh.vert_positions.push_back({ v.x, v.y, v.z });
			for (int q = 0; q < 6; ++q) { TestIndigoMesh::Quad qd = {}; for (int k = 0; k < 4; ++k) qd.vertex_indices[k] = quads[q][k]; qd.mat_index = (uint32)q; cube_mesh.quads.push_back(qd); }
			const PhysicsShape unit_cube = PhysicsWorld::createJoltShapeForIndigoMesh(cube_mesh, /*build_dynamic_physics_ob=*/false);
			// ... and the same mesh as a DYNAMIC object: the convex hull of its vertices (PhysicsWorld.cpp:746-768)
			{
				Reference<PhysicsObject> crate = new PhysicsObject(true, PhysicsWorld::createJoltShapeForIndigoMesh(cube_mesh, /*build_dynamic_physics_ob=*/true), nullptr, 0);
				crate->motion_type = PhysicsObject::MotionType_dynamic; crate->mass = 30.f; crate->pos = Vec4f(40.f, 40.f, 20.f, 1);
				world->addObject(crate); world->activateObject(crate);
				for (int k = 0; k < 30; ++k) world->think(1.0 / 60.0);
				const bool fell = world->getPosInJolt(crate)[2] < 19.f;
				if (!fell) printf("dynamic hull of the cube mesh did not fall\n");
				ok = ok && fell;
			}
			// A finely tessellated DYNAMIC mesh: 200 vertices on an ellipsoid, every one a corner of its hull.  ConvexHullShapeSettings takes them all
			// (PhysicsWorld.cpp:745-768, :979; Jolt keeps up to 256 points) -- rounds 1-4 kept 32.  The object rolls on the terrain and comes to rest above it.
			{
				TestIndigoMesh blob;
				for (int i = 0; i < 200; ++i) {      // a Fibonacci spiral: no two points alike, none inside the hull of the others
					const float z = 1.f - 2.f * ((float)i + 0.5f) / 200.f, r = std::sqrt(1.f - z * z), a = 2.399963f * (float)i;
					blob.vert_positions.push_back({ 0.9f * r * std::cos(a), 0.6f * r * std::sin(a), 0.5f * z });
				}
				for (uint32 i = 0; i + 2 < 200; ++i) { TestIndigoMesh::Triangle t = {}; t.vertex_indices[0] = i; t.vertex_indices[1] = i + 1; t.vertex_indices[2] = i + 2; blob.triangles.push_back(t); }
				Reference<PhysicsObject> ob = new PhysicsObject(true, PhysicsWorld::createJoltShapeForIndigoMesh(blob, /*build_dynamic_physics_ob=*/true), nullptr, 0);
				ob->motion_type = PhysicsObject::MotionType_dynamic; ob->mass = 40.f; ob->pos = Vec4f(6.f, -6.f, terrainHeight(6.f, -6.f) + 3.f, 1);
				world->addObject(ob); world->activateObject(ob);
				const bool all_corners = ob->shape.hull && ob->shape.hull->instances.size() == 1 && ob->shape.hull->instances[0].num_vertices == 200;
				if (!all_corners) printf("dynamic hull of a 200-vertex mesh kept %u vertices\n", ob->shape.hull && !ob->shape.hull->instances.empty() ? ob->shape.hull->instances[0].num_vertices : 0u);
				for (int k = 0; k < 240; ++k) world->think(1.0 / 60.0);
				const Vec4f p = world->getPosInJolt(ob);
				const float above = p[2] - terrainHeight(p[0], p[1]);
				const bool rests = above > 0.3f && above < 1.2f;
				if (!rests) printf("200-vertex hull: %.3f above the terrain at (%.2f, %.2f)\n", above, p[0], p[1]);
				ok = ok && all_corners && rests;
			}
			Reference<PhysicsObject> bounds = new PhysicsObject(true, PhysicsWorld::createScaledAndTranslatedShapeForShape(unit_cube, Vec3f(-1.f, -2.f, 0.f), Vec3f(2.f, 4.f, 1.5f)), nullptr, 0);
			bounds->pos = Vec4f(-12.f, 12.f, 8.f, 1);
			world->addObject(bounds);
			world->traceRay(Vec4f(-12.5f, 13.5f, 30, 1), Vec4f(0, 0, -1, 0), 100.f, JPH::BodyID(), r);
			const bool top = r.hit_object == bounds.ptr() && std::fabs(r.hit_t - (30.f - 9.5f)) < 1e-3f && r.hit_normal_ws[2] > 0.99f;
			world->traceRay(Vec4f(-20.f, 13.9f, 8.7f, 1), Vec4f(1, 0, 0, 0), 100.f, JPH::BodyID(), r);
			const bool side = r.hit_object == bounds.ptr() && std::fabs(r.hit_t - 7.f) < 1e-3f && r.hit_normal_ws[0] < -0.99f;
			world->traceRay(Vec4f(-20.f, 14.1f, 8.7f, 1), Vec4f(1, 0, 0, 0), 12.f, JPH::BodyID(), r);      // just past its +y face: misses it
			const bool miss = r.hit_object != bounds.ptr();
			if (!(top && side && miss)) printf("decorated cube: top %d side %d miss %d\n", (int)top, (int)side, (int)miss);
			ok = ok && top && side && miss;
		}
		{      // debug helpers of the facade (PhysicsWorld.h:187-189)
			world->writeJoltSnapshotToDisk("/tmp/sgp_mesh_world.snap");
			FILE* f = fopen("/tmp/sgp_mesh_world.snap", "rb");
			char magic[9] = { 0 }; const bool snap = f && fread(magic, 1, 8, f) == 8 && std::string(magic) == "SGPSNAP1";
			if (f) fclose(f);
			ok = ok && snap && PhysicsWorld::computeSizeBForShape(building->shape) > sizeof(PhysicsShape);
		}
		{      // computeToWorldAndToObMatrices (PhysicsWorld.cpp:660-704): the two matrices are each other's inverse, also with a zero scale component
			Matrix4f a, b;
			const Quatf q = Quatf::fromAxisAndAngle(normalise(Vec4f(0.3f, -0.5f, 0.8f, 0)), 1.1f);
			computeToWorldAndToObMatrices(Vec4f(3.f, -2.f, 7.f, 1), q, Vec4f(2.f, 0.5f, 1.5f, 0), a, b);
			const Matrix4f prod = a * b;
			float err = 0;
			for (int i = 0; i < 16; ++i) err = std::fmax(err, std::fabs(prod.e[i] - ((i % 5 == 0) ? 1.f : 0.f)));
			const Vec4f back = b * (a * Vec4f(0.25f, -1.f, 2.f, 1));
			Matrix4f c0, c1;
			computeToWorldAndToObMatrices(Vec4f(0, 0, 0, 1), Quatf::identity(), Vec4f(1.f, 0.f, 1.f, 0), c0, c1);
			const bool mats = err < 1e-5f && std::fabs(back[0] - 0.25f) < 1e-5f && std::fabs(back[1] + 1.f) < 1e-5f && std::fabs(back[2] - 2.f) < 1e-5f && c0.e[5] == 1.0e-6f && std::isfinite(c1.e[5]);
			if (!mats) printf("computeToWorldAndToObMatrices: err %g\n", err);
			ok = ok && mats;
		}
		{      // the diagnostics window's counters under the reference's labels (PhysicsWorld.cpp:1578-1604)
			const std::string diag = world->getDiagnostics();
			const char* labels[] = { "Jolt bodies: ", "max bodies: ", "num static bodies: ", "num dynamic bodies: ", "num active dynamic bodies: ", "num kinematic bodies: ",
				"num active kinematic bodies: ", "Active bodies: ", "Meshes:  ", "mem usage: ", "NON_MOVING layer obs:                ", "MOVING layer obs:                    ",
				"NON_MOVING_NON_COLLIDABLE layer obs: ", "MOVING_NON_COLLIDABLE layer obs:     " };
			size_t at = 0; bool labels_ok = true;
			for (const char* l : labels) { const size_t f = diag.find(l, at); if (f == std::string::npos) { labels_ok = false; printf("diagnostics lack '%s'\n", l); break; } at = f; }
			const PhysicsWorld::MemUsageStats mu = world->getMemUsageStats();
			const bool counts_ok = mu.num_meshes >= 4 && mu.mem > 0 && diag.find("num static bodies: 0") == std::string::npos && world->getLoadedMeshes().empty();
			if (!counts_ok) printf("diagnostics:\n%s", diag.c_str());
			ok = ok && labels_ok && counts_ok;
		}
		printf("rays ok %d\n", (int)ok);

		// the player: starts on the terrain west of the building, walks east (+x, uphill on average) into its wall
		struct P : public JPH::CharacterContactListener {} listener;
		JPH::CharRef<JPH::CharacterShape> shape = JPH::RotatedTranslatedShapeSettings(JPH::Vec3(0, 0, 0.65f + 0.3f), JPH::Quat(), new JPH::CapsuleShape(0.65f, 0.3f)).Create().Get();
		JPH::CharRef<JPH::CharacterVirtualSettings> cs = new JPH::CharacterVirtualSettings();
		cs->mShape = shape; cs->mUp = JPH::Vec3(0, 0, 1); cs->mSupportingVolume = JPH::Plane(JPH::Vec3(0, 0, 1), -0.3f); cs->mMaxStrength = 1000;
		JPH::CharacterVirtual player(cs, JPH::Vec3(-8.f, 10.f, terrainHeight(-8.f, 10.f) + 1.5f), JPH::Quat(), world->physics_system);
		player.SetListener(&listener);
		JPH::TempAllocator ta; JPH::CharacterVirtual::ExtendedUpdateSettings ext; ext.mStickToFloorStepDown = JPH::Vec3(0, 0, -0.5f); ext.mWalkStairsStepUp = JPH::Vec3(0, 0, 0.4f);
		float max_err = 0;
		for (int s = 0; s < 600; ++s) {
			JPH::Vec3 vel = player.GetLinearVelocity();
			if (player.IsSupported()) vel = JPH::Vec3(3, 0, 0) + player.GetGroundVelocity(); else vel = vel + JPH::Vec3(3, 0, 0) * (1.f / 60.f);
			vel = vel + JPH::Vec3(0, 0, -9.81f / 60.f);
			player.SetLinearVelocity(vel);
			player.ExtendedUpdate(1.f / 60.f, world->physics_system->GetGravity(), ext, world->physics_system->GetDefaultBroadPhaseLayerFilter(1), world->physics_system->GetDefaultLayerFilter(1), JPH::BodyFilter(), JPH::ShapeFilter(), ta);
			world->think(1.0 / 60.0);
			const JPH::Vec3 pp = player.GetPosition();
			if (s > 60 && player.IsSupported() && pp.x < 6.5f) max_err = std::fmax(max_err, std::fabs(pp.z - terrainHeight(pp.x, pp.y)));
		}
		const JPH::Vec3 pp = player.GetPosition();
		printf("player at %.2f %.2f %.2f (terrain %.2f)  max height error while walking %.3f\n", pp.x, pp.y, pp.z, terrainHeight(pp.x, pp.y), max_err);
		ok = ok && std::fabs(pp.x - (7.f - 0.3f)) < 0.08f && std::fabs(pp.y - 10.f) < 0.3f && max_err < 0.12f && player.IsSupported();
		return ok ? 0 : 1;
	} catch (glare::Exception& e) { fprintf(stderr, "glare::Exception: %s\n", e.what().c_str()); return 2; }
}
