// PhysicsWorld over the sgp C ABI.  Each method follows the reference method of the same name
// (/root/reference/gui_client/PhysicsWorld.cpp, line ranges in the comments) with Jolt calls replaced by ABI calls.
#include "PhysicsWorld.h"
#include <cmath>
#include <Jolt/Physics/Vehicle/VehicleConstraint.h>
#include <Jolt/Physics/Vehicle/WheeledVehicleController.h>
#include <Jolt/Physics/Vehicle/MotorcycleController.h>
#include <utils/Exception.h>
#include "../../include/sgp.h"
#include <algorithm>
#include <cassert>
#include <cstring>
#include <cstdio>

static inline float myClamp(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

// A shape or body the device side refused (capacity, degenerate geometry): the reference would go on silently without the object's collision
// (addObject has no error path, PhysicsWorld.cpp:1178-1189); this backend says so once per kind on stderr so that a world outgrowing its
// capacities does not just lose collision quietly.
static void reportShapeFailure(const char* what)
{
	static int reported = 0;
	if (reported++ < 8) fprintf(stderr, "PhysicsWorld: could not create %s: %s\n", what, sgp_last_error());
}
// the body of `object` now uses the device-side instance: remembered on the object so that removeObject can give it back
static void holdMeshInstance(PhysicsObject& object, const std::shared_ptr<PhysicsMeshData>& data, PhysicsMeshData::Instance* in)
{
	in->users++;
	sgp_world* world = in->world; const uint32_t id = in->mesh_id;
	object.shape_instance_releases.push_back([data, world, id]() {
		for (size_t i = 0; i < data->instances.size(); ++i) if (data->instances[i].world == world && data->instances[i].mesh_id == id) {
			if (--data->instances[i].users == 0) { sgp_mesh_destroy(world, id); data->instances.erase(data->instances.begin() + (long)i); }
			return;
		}
	});
}
static void holdHullInstance(PhysicsObject& object, const std::shared_ptr<PhysicsHullData>& data, PhysicsHullData::Instance* in)
{
	in->users++;
	sgp_world* world = in->world; const uint32_t id = in->hull_id;
	object.shape_instance_releases.push_back([data, world, id]() {
		for (size_t i = 0; i < data->instances.size(); ++i) if (data->instances[i].world == world && data->instances[i].hull_id == id) {
			if (--data->instances[i].users == 0) { sgp_hull_destroy(world, id); data->instances.erase(data->instances.begin() + (long)i); }
			return;
		}
	});
}

// the JPH::Shape look-alike of a facade shape (PhysicsShape::jolt_shape)
static void setJoltShape(PhysicsShape& s)
{
	JPH::Shape* j = new JPH::Shape;
	j->kind = s.kind; j->p[0] = s.p[0]; j->p[1] = s.p[1]; j->p[2] = s.p[2];
	j->hull = s.hull; j->mesh = s.mesh;
	if (s.hull) { j->hull_points = s.hull->points; for (int i = 0; i < 3; ++i) j->com_offset[i] = s.hull->com_offset[i]; }
	if (s.kind == 1) j->volume = 8.f * s.p[0] * s.p[1] * s.p[2];
	s.jolt_shape = j;
}

// PhysicsWorld.cpp:250-273
void PhysicsWorld::init()
{
	const int n = sgp_init();
	if (n <= 0) throw glare::Exception(std::string("PhysicsWorld::init: ") + sgp_last_error());
}

// PhysicsWorld.cpp:462-532.  task_manager / stack_allocator are accepted for source compatibility and unused: the
// step runs as HIP launches on one stream, scratch lives in device arenas (no job system, no temp allocator).
PhysicsWorld::PhysicsWorld(glare::TaskManager* task_manager_, glare::StackAllocator* stack_allocator_)
:	activated_obs(NULL), newly_activated_obs(NULL), event_listener(NULL), world(NULL),
	water_buoyancy_enabled(false), water_z(0), task_manager(task_manager_), stack_allocator(stack_allocator_)
{
	sgp_world_desc d;
	sgp_default_world_desc(&d);        // cMaxBodies 65536 (:492), gravity (0,0,-9.81) (:520), Jolt default settings
	d.max_bodies *= 3;                 // ... of OBJECTS: a mesh object takes three body slots here (the body and two manifold aliases), so 65536 of them still fit
	const int r = sgp_world_create(&d, &world);
	if (r != SGP_OK) { world = NULL; throw glare::Exception(std::string("PhysicsWorld: ") + sgp_last_error()); }
	id_to_ob.resize(d.max_bodies, NULL);
	physics_system = new JPH::PhysicsSystem(world);
}

PhysicsWorld::~PhysicsWorld() { delete physics_system; if (world) sgp_world_destroy(world); }

static void checkSGP(int rc, const char* what);
void PhysicsWorld::setWaterBuoyancyEnabled(bool enabled) { water_buoyancy_enabled = enabled; checkSGP(sgp_world_set_water(world, enabled ? 1 : 0, water_z), "setWaterBuoyancyEnabled"); }
void PhysicsWorld::setWaterZ(float z) { water_z = z; checkSGP(sgp_world_set_water(world, water_buoyancy_enabled ? 1 : 0, water_z), "setWaterZ"); }

// PhysicsWorld.cpp:1123-1135
PhysicsShape PhysicsWorld::createGroundQuadShape(float ground_quad_w)
{
	PhysicsShape s; s.kind = 1; s.p[0] = ground_quad_w / 2; s.p[1] = ground_quad_w / 2; s.p[2] = 0.5f; s.size_B = sizeof(PhysicsShape);
	setJoltShape(s);
	return s;
}
PhysicsShape PhysicsWorld::createCapsuleShape(float radius, float half_height)
{
	PhysicsShape s; s.kind = 2; s.p[0] = radius; s.p[1] = half_height; s.size_B = sizeof(PhysicsShape);
	setJoltShape(s);
	return s;
}

PhysicsShape PhysicsWorld::createConvexHullShape(const std::vector<Vec3f>& points)
{
	if (points.size() < 4) throw glare::Exception("Error building Jolt shape: a convex hull needs at least 4 points");
	PhysicsShape s; s.kind = 3;
	s.hull = std::make_shared<PhysicsHullData>();
	s.hull->points.reserve(points.size() * 3);
	for (size_t i = 0; i < points.size(); ++i) { s.hull->points.push_back(points[i].x); s.hull->points.push_back(points[i].y); s.hull->points.push_back(points[i].z); }
	s.size_B = sizeof(PhysicsShape) + s.hull->points.size() * sizeof(float);
	setJoltShape(s);
	return s;
}

PhysicsShape PhysicsWorld::createMeshShape(const std::vector<Vec3f>& vertices, const std::vector<uint32>& triangle_indices,
	const std::vector<uint32>* triangle_materials, const std::vector<bool>* create_tris_for_mat)
{
	if (vertices.size() < 3 || triangle_indices.size() < 3 || triangle_indices.size() % 3 != 0) throw glare::Exception("Error building Jolt shape: a mesh needs vertices and whole triangles");
	for (size_t i = 0; i < triangle_indices.size(); ++i) if (triangle_indices[i] >= vertices.size()) throw glare::Exception("Error building Jolt shape: triangle index out of range");
	const size_t num_tris = triangle_indices.size() / 3;
	if (triangle_materials && triangle_materials->size() != num_tris) throw glare::Exception("Error building Jolt shape: one material index per triangle expected");
	PhysicsShape s; s.kind = 4;
	s.mesh = std::make_shared<PhysicsMeshData>();
	for (size_t i = 0; i < vertices.size(); ++i) { s.mesh->vertices.push_back(vertices[i].x); s.mesh->vertices.push_back(vertices[i].y); s.mesh->vertices.push_back(vertices[i].z); }
	for (size_t t = 0; t < num_tris; ++t) {
		const uint32 mat = triangle_materials ? (*triangle_materials)[t] : 0;
		// PhysicsWorld.cpp:1028: if(!create_tris_for_mat || (material_index >= create_tris_for_mat->size()) || (*create_tris_for_mat)[material_index])
		if (create_tris_for_mat && mat < create_tris_for_mat->size() && !(*create_tris_for_mat)[mat]) continue;
		for (int k = 0; k < 3; ++k) s.mesh->indices.push_back(triangle_indices[3 * t + k]);
		if (triangle_materials) s.mesh->materials.push_back(mat);
	}
	if (s.mesh->indices.empty()) throw glare::Exception("Error building Jolt shape: no triangles left after the material filter");
	s.size_B = sizeof(PhysicsShape) + s.mesh->vertices.size() * sizeof(float) + (s.mesh->indices.size() + s.mesh->materials.size()) * sizeof(uint32_t);
	setJoltShape(s);
	return s;
}

PhysicsShape PhysicsWorld::createJoltHeightFieldShape(int vert_res, const std::vector<float>& heightfield, int width, float quad_w)
{
	if (width < 2 || vert_res > width || (size_t)width * (size_t)width > heightfield.size()) throw glare::Exception("Error building Jolt heightfield shape: bad sample count");
	const float z_offset = -quad_w * (float)(width - 1);
	std::vector<Vec3f> verts; std::vector<uint32> tris;
	verts.reserve((size_t)width * width);
	for (int z = 0; z < width; ++z) for (int x = 0; x < width; ++x) verts.push_back(Vec3f(quad_w * (float)x, heightfield[(size_t)z * width + x], quad_w * (float)z + z_offset));
	for (int z = 0; z + 1 < width; ++z) for (int x = 0; x + 1 < width; ++x) {
		const uint32 a = (uint32)(z * width + x), b = a + 1, c = a + (uint32)width, d = c + 1;      // a (x,z)  b (x+1,z)  c (x,z+1)  d (x+1,z+1)
		tris.push_back(a); tris.push_back(c); tris.push_back(d);          // facing +y
		tris.push_back(a); tris.push_back(d); tris.push_back(b);
	}
	return createMeshShape(verts, tris);
}

static PhysicsMeshData::Instance* meshInstance(sgp_world* world, const PhysicsShape& shape, const Vec3f& scale)
{
	PhysicsMeshData& m = *shape.mesh;
	for (size_t i = 0; i < m.instances.size(); ++i) {
		PhysicsMeshData::Instance& in = m.instances[i];
		if (in.world == world && in.scale[0] == scale.x && in.scale[1] == scale.y && in.scale[2] == scale.z) return &in;
	}
	std::vector<float> v(m.vertices);
	for (size_t i = 0; i + 2 < v.size(); i += 3) { v[i] *= scale.x; v[i + 1] *= scale.y; v[i + 2] *= scale.z; }
	std::vector<uint32_t> idx(m.indices);
	if (scale.x * scale.y * scale.z < 0.f) for (size_t i = 0; i + 2 < idx.size(); i += 3) std::swap(idx[i + 1], idx[i + 2]);      // a mirroring scale turns the triangles inside out
	sgp_mesh_info info;
	if (sgp_mesh_create_with_materials(world, v.data(), (uint32_t)(v.size() / 3), idx.data(), (uint32_t)(idx.size() / 3),
		m.materials.size() == idx.size() / 3 ? m.materials.data() : nullptr, &info) != SGP_OK) { reportShapeFailure("mesh"); return nullptr; }
	PhysicsMeshData::Instance in; in.world = world; in.scale[0] = scale.x; in.scale[1] = scale.y; in.scale[2] = scale.z; in.mesh_id = info.mesh_id; in.users = 0;
	m.instances.push_back(in);
	return &m.instances.back();
}

// PhysicsWorld.cpp:1155-1166: RotatedTranslatedShape(translation, identity, ScaledShape(shape, scale)) -- a point p of the original
// shape ends up at translation + scale * p.  The decorators are baked into a new vertex list (GUIClient uses this on the unit quad /
// unit cube for text objects and splat bounds, GUIClient.cpp:2117,4807).
PhysicsShape PhysicsWorld::createScaledAndTranslatedShapeForShape(const PhysicsShape& original_shape, const Vec3f& translation, const Vec3f& scale)
{
	if (!(std::fabs(scale.x) > 0.f && std::fabs(scale.y) > 0.f && std::fabs(scale.z) > 0.f)) throw glare::Exception("Error building Jolt shape: degenerate scale");
	PhysicsShape s = original_shape;
	const float sc[3] = { scale.x, scale.y, scale.z }, tr[3] = { translation.x, translation.y, translation.z };
	if (original_shape.kind == 3 && original_shape.hull) {
		s.hull = std::make_shared<PhysicsHullData>();
		s.hull->points = original_shape.hull->points;
		for (size_t i = 0; i < s.hull->points.size(); ++i) s.hull->points[i] = tr[i % 3] + sc[i % 3] * s.hull->points[i];
		for (int i = 0; i < 3; ++i) s.hull->com_offset[i] = sc[i] * original_shape.hull->com_offset[i];
	} else if (original_shape.kind == 4 && original_shape.mesh) {
		s.mesh = std::make_shared<PhysicsMeshData>();
		s.mesh->vertices = original_shape.mesh->vertices;
		for (size_t i = 0; i < s.mesh->vertices.size(); ++i) s.mesh->vertices[i] = tr[i % 3] + sc[i % 3] * s.mesh->vertices[i];
		s.mesh->indices = original_shape.mesh->indices;
		s.mesh->materials = original_shape.mesh->materials;
		if (scale.x * scale.y * scale.z < 0.f) for (size_t i = 0; i + 2 < s.mesh->indices.size(); i += 3) std::swap(s.mesh->indices[i + 1], s.mesh->indices[i + 2]);
	} else throw glare::Exception("Error building Jolt shape: scale / translate decorators are implemented for convex hull and mesh shapes");
	setJoltShape(s);
	return s;
}

PhysicsShape PhysicsWorld::createCOMOffsetShapeForShape(const PhysicsShape& original_shape, const Vec4f& COM_offset)
{
	if (original_shape.kind != 3 || !original_shape.hull) throw glare::Exception("Error building Jolt shape: centre-of-mass offsets are implemented for convex hull shapes only");
	PhysicsShape s = original_shape;
	s.hull = std::make_shared<PhysicsHullData>();
	s.hull->points = original_shape.hull->points;
	for (int i = 0; i < 3; ++i) s.hull->com_offset[i] = original_shape.hull->com_offset[i] + COM_offset[i];
	setJoltShape(s);
	return s;
}

// The device-side hull of `shape` for this world and object scale (JPH::ScaledShape baked into the points), built on first use.
static PhysicsHullData::Instance* hullInstance(sgp_world* world, const PhysicsShape& shape, const Vec3f& scale)
{
	PhysicsHullData& h = *shape.hull;
	for (size_t i = 0; i < h.instances.size(); ++i) {
		PhysicsHullData::Instance& in = h.instances[i];
		if (in.world == world && in.scale[0] == scale.x && in.scale[1] == scale.y && in.scale[2] == scale.z) return &in;
	}
	std::vector<float> pts(h.points);
	for (size_t i = 0; i + 2 < pts.size(); i += 3) { pts[i] *= scale.x; pts[i + 1] *= scale.y; pts[i + 2] *= scale.z; }
	sgp_hull_info info;
	const float off[3] = { h.com_offset[0] * scale.x, h.com_offset[1] * scale.y, h.com_offset[2] * scale.z };
	if (sgp_hull_create_com(world, pts.data(), (uint32_t)(pts.size() / 3), off, &info) != SGP_OK) { reportShapeFailure("convex hull"); return nullptr; }
	PhysicsHullData::Instance in;
	in.users = 0;
	in.world = world; in.scale[0] = scale.x; in.scale[1] = scale.y; in.scale[2] = scale.z; in.hull_id = info.hull_id; in.num_vertices = info.num_vertices;
	memcpy(in.com, info.com, sizeof(in.com)); memcpy(in.rot, info.rot, sizeof(in.rot));
	memcpy(in.aabb_min, info.aabb_min, sizeof(in.aabb_min)); memcpy(in.aabb_max, info.aabb_max, sizeof(in.aabb_max));
	h.instances.push_back(in);
	return &h.instances.back();
}

// object pose <-> body pose (identity unless the body is a hull: body frame = centre of mass / principal axes)
static inline void toBodyPose(const PhysicsObject& ob, const Vec4f& pos, const Quatf& rot, float pos_out[3], float rot_out[4])
{
	const Vec4f p = pos + rot.rotateVector(maskWToZero(ob.body_com_os));
	const Quatf q = rot * ob.body_rot_os;
	for (int i = 0; i < 3; ++i) pos_out[i] = p[i];
	for (int i = 0; i < 4; ++i) rot_out[i] = q.v[i];
}
static inline void toObjectPose(const PhysicsObject& ob, const float body_pos[3], const float body_rot[4], Vec4f& pos_out, Quatf& rot_out)
{
	const Quatf qb(body_rot[0], body_rot[1], body_rot[2], body_rot[3]);
	rot_out = qb * ob.body_rot_os.conjugate();
	const Vec4f c = rot_out.rotateVector(maskWToZero(ob.body_com_os));
	pos_out = Vec4f(body_pos[0] - c[0], body_pos[1] - c[1], body_pos[2] - c[2], 1.f);
}

// PhysicsWorld.cpp:1169-1311
void PhysicsWorld::addObject(const Reference<PhysicsObject>& object)
{
	assert(object->pos.isFinite());
	if (!object->jolt_body_id.IsInvalid()) {          // body already built (:1175), e.g. by CarPhysics through BodyInterface::CreateBody
		if (const JPH::BodyInterface::Frame* f = physics_system->GetBodyInterface().getFrame(object->jolt_body_id)) {
			object->body_com_os = Vec4f(f->com.x, f->com.y, f->com.z, 0.f);
			object->body_rot_os = Quatf(f->rot.x, f->rot.y, f->rot.z, f->rot.w);
		}
		const uint32_t bid = object->jolt_body_id.GetIndex();
		if (bid < id_to_ob.size()) id_to_ob[bid] = object.ptr();
		return;
	}
	sgp_body_desc d;
	sgp_default_body_desc(&d);
	for (int i = 0; i < 3; ++i) d.pos[i] = object->pos[i];
	for (int i = 0; i < 4; ++i) d.rot[i] = object->rot.v[i];
	if (std::fabs(object->scale.x) < 1.0e-7f || std::fabs(object->scale.y) < 1.0e-7f || std::fabs(object->scale.z) < 1.0e-7f) return;   // :1184
	const bool moving = object->motion_type == PhysicsObject::MotionType_dynamic || object->motion_type == PhysicsObject::MotionType_kinematic ||
		object->motion_type == PhysicsObject::MotionType_semi_static;
	if (moving) d.layer = object->collidable ? SGP_LAYER_MOVING : SGP_LAYER_MOVING_NON_COLLIDABLE;               // :1193-1199
	else        d.layer = object->collidable ? SGP_LAYER_NON_MOVING : SGP_LAYER_NON_MOVING_NON_COLLIDABLE;       // :1200-1207
	switch (object->motion_type) {                                                                                // :1209-1217
	case PhysicsObject::MotionType_dynamic:   d.motion_type = SGP_MOTION_DYNAMIC; break;
	case PhysicsObject::MotionType_kinematic: d.motion_type = SGP_MOTION_KINEMATIC; break;
	default:                                  d.motion_type = SGP_MOTION_STATIC; break;
	}
	d.is_sensor = object->is_sensor ? 1 : 0;
	d.friction = myClamp(object->friction, 0.f, 1.f);       // :1236
	d.restitution = myClamp(object->restitution, 0.f, 1.f); // :1237
	d.mass = std::max(0.001f, object->mass);                // :1238
	d.use_zero_linear_drag = object->use_zero_linear_drag ? 1 : 0;
	d.userdata = (uint64)object.ptr();                      // :1241
	d.activate = 0;                                          // EActivation::DontActivate (:1243)
	PhysicsMeshData::Instance* mesh_in = nullptr; PhysicsHullData::Instance* hull_in = nullptr;
	if (object->is_sphere) {              // unit sphere r 0.5, uniform scale = scale.x (:1219-1227)
		d.shape_type = SGP_SHAPE_SPHERE; d.shape[0] = 0.5f * std::fabs(object->scale.x); d.shape[1] = d.shape[2] = 0;
	} else if (object->is_cube) {         // unit cube half 0.5, per-axis scale (:1247-1255)
		d.shape_type = SGP_SHAPE_BOX;
		d.shape[0] = 0.5f * std::fabs(object->scale.x); d.shape[1] = 0.5f * std::fabs(object->scale.y); d.shape[2] = 0.5f * std::fabs(object->scale.z);
	} else {                              // object->shape with a ScaledShape decorator (:1275-1287)
		const PhysicsShape& s = object->shape;
		if (s.jolt_shape.GetPtr() && s.jolt_shape->kind == 5) { addCompoundObject(object, d); return; }      // JPH::StaticCompoundShape (MeshBuilding.cpp:396-413)
		if (s.kind < 0) return;           // shape.jolt_shape == NULL (:1278-1279)
		d.shape_type = s.kind;
		if (s.kind == 4) {
			// JPH::MeshShape: static or kinematic bodies -- the reference builds a mesh shape exactly when the object is not dynamic, and "Jolt doesn't
			// support dynamic bodies with mesh shapes, so change to kinematic" (:1290-1292)
			if (d.motion_type == SGP_MOTION_DYNAMIC) d.motion_type = SGP_MOTION_KINEMATIC;
			PhysicsMeshData::Instance* in = s.mesh ? meshInstance(world, s, object->scale) : nullptr;
			if (!in) return;
			mesh_in = in;
			d.shape[0] = (float)in->mesh_id; d.shape[1] = d.shape[2] = 0;
		} else if (s.kind == 3) {
			PhysicsHullData::Instance* in = s.hull ? hullInstance(world, s, object->scale) : nullptr;
			if (!in) return;              // (silent, like every other rejected add)
			hull_in = in;
			d.shape[0] = (float)in->hull_id; d.shape[1] = d.shape[2] = 0;
			object->body_com_os = Vec4f(in->com[0], in->com[1], in->com[2], 0.f);
			object->body_rot_os = Quatf(in->rot[0], in->rot[1], in->rot[2], in->rot[3]);
			toBodyPose(*object, object->pos, object->rot, d.pos, d.rot);
		} else
		if (s.kind == 1) { d.shape[0] = s.p[0] * std::fabs(object->scale.x); d.shape[1] = s.p[1] * std::fabs(object->scale.y); d.shape[2] = s.p[2] * std::fabs(object->scale.z); }
		else if (s.kind == 0) { d.shape[0] = s.p[0] * std::fabs(object->scale.x); }
		else { d.shape[0] = s.p[0] * std::fabs(object->scale.x); d.shape[1] = s.p[1] * std::fabs(object->scale.z); }
	}
	uint32_t id = SGP_INVALID_ID;
	const int add_rc = sgp_body_add(world, &d, &id);
	if (add_rc != SGP_OK) { if (add_rc == SGP_ERR_CAPACITY) reportShapeFailure("body"); return; }      // silent rejection of bad input, as the reference (:1178-1189)
	if (mesh_in) holdMeshInstance(*object, object->shape.mesh, mesh_in);
	if (hull_in) holdHullInstance(*object, object->shape.hull, hull_in);
	object->jolt_body_id = JPH::BodyID(id);
	if (id < id_to_ob.size()) id_to_ob[id] = object.ptr();
	// the look-alike BodyInterface answers in the shape's space: it needs the body frame of hull bodies
	if (object->shape.kind == 3 && !object->is_sphere && !object->is_cube)
		physics_system->GetBodyInterface().setFrame(object->jolt_body_id, JPH::Vec3(object->body_com_os[0], object->body_com_os[1], object->body_com_os[2]),
			JPH::Quat(object->body_rot_os.v[0], object->body_rot_os.v[1], object->body_rot_os.v[2], object->body_rot_os.v[3]));
}

// A static compound object: every child of the JPH::StaticCompoundShape becomes a child of one sgp compound body, with the object's
// scale applied the way JPH::ScaledShape applies it to a compound (child positions and child shapes scaled per axis; exact for children
// whose rotation is the identity, which is what the reference builds).
void PhysicsWorld::addCompoundObject(const Reference<PhysicsObject>& object, sgp_body_desc d)
{
	if (d.motion_type != SGP_MOTION_STATIC) return;                // StaticCompoundShape on a static object
	const JPH::Shape& comp = *object->shape.jolt_shape;
	const Vec3f sc = object->scale;
	std::vector<sgp_compound_child> children;
	std::vector<std::pair<std::shared_ptr<PhysicsMeshData>, uint32_t>> held_meshes; std::vector<std::pair<std::shared_ptr<PhysicsHullData>, uint32_t>> held_hulls;
	for (const JPH::Shape::SubShape& c : comp.children) {
		sgp_compound_child k; memset(&k, 0, sizeof(k));
		const JPH::Shape& cs = *c.shape;
		JPH::Vec3 pos(c.pos.x * sc.x, c.pos.y * sc.y, c.pos.z * sc.z); JPH::Quat rot = c.rot;
		k.shape_type = cs.kind;
		if (cs.kind == 0) k.shape[0] = cs.p[0] * std::fabs(sc.x);
		else if (cs.kind == 1) { k.shape[0] = cs.p[0] * std::fabs(sc.x); k.shape[1] = cs.p[1] * std::fabs(sc.y); k.shape[2] = cs.p[2] * std::fabs(sc.z); }
		else if (cs.kind == 2) { k.shape[0] = cs.p[0] * std::fabs(sc.x); k.shape[1] = cs.p[1] * std::fabs(sc.z); }
		else if (cs.kind == 4 && cs.mesh) {
			PhysicsShape tmp; tmp.kind = 4; tmp.mesh = cs.mesh;
			PhysicsMeshData::Instance* in = meshInstance(world, tmp, sc);
			if (!in) return;
			held_meshes.push_back(std::make_pair(cs.mesh, in->mesh_id));
			k.shape[0] = (float)in->mesh_id;
		} else if (cs.kind == 3) {
			PhysicsShape tmp; tmp.kind = 3; tmp.hull = cs.hull;
			if (!tmp.hull) { tmp.hull = std::make_shared<PhysicsHullData>(); tmp.hull->points = cs.hull_points; for (int i = 0; i < 3; ++i) tmp.hull->com_offset[i] = cs.com_offset[i]; }
			PhysicsHullData::Instance* in = hullInstance(world, tmp, sc);
			if (!in) return;
			held_hulls.push_back(std::make_pair(tmp.hull, in->hull_id));
			k.shape[0] = (float)in->hull_id;
			pos = pos + rot * JPH::Vec3(in->com[0], in->com[1], in->com[2]);          // the hull's body frame inside the child's frame
			rot = rot * JPH::Quat(in->rot[0], in->rot[1], in->rot[2], in->rot[3]);
		} else return;
		k.pos[0] = pos.x; k.pos[1] = pos.y; k.pos[2] = pos.z; k.rot[0] = rot.x; k.rot[1] = rot.y; k.rot[2] = rot.z; k.rot[3] = rot.w;
		children.push_back(k);
	}
	uint32_t id = SGP_INVALID_ID;
	if (children.empty() || sgp_body_add_compound(world, &d, children.data(), (uint32_t)children.size(), &id) != SGP_OK) { reportShapeFailure("compound body"); return; }
	for (auto& hm : held_meshes) for (PhysicsMeshData::Instance& in : hm.first->instances) if (in.world == world && in.mesh_id == hm.second) { holdMeshInstance(*object, hm.first, &in); break; }
	for (auto& hh : held_hulls) for (PhysicsHullData::Instance& in : hh.first->instances) if (in.world == world && in.hull_id == hh.second) { holdHullInstance(*object, hh.first, &in); break; }
	object->jolt_body_id = JPH::BodyID(id);
	if (id < id_to_ob.size()) id_to_ob[id] = object.ptr();
	physics_system->registerCompound(object->jolt_body_id, (uint32_t)children.size());
}

JPH::Body PhysicsWorld::getJoltBody(const PhysicsObject& object) const
{
	JPH::Body b;
	b.id = object.jolt_body_id; b.user_data = (uint64)&object;
	b.com_offset = JPH::Vec3(object.body_com_os[0], object.body_com_os[1], object.body_com_os[2]);
	for (int i = 0; i < 4; ++i) b.frame_rot[i] = object.body_rot_os.v[i];
	return b;
}

// PhysicsWorld.cpp:1315-1339
void PhysicsWorld::removeObject(const Reference<PhysicsObject>& object)
{
	if (!object->jolt_body_id.IsInvalid()) {
		const uint32_t id = object->jolt_body_id.GetIndex();
		if (physics_system->GetBodyInterface().IsAdded(object->jolt_body_id)) { physics_system->GetBodyInterface().RemoveBody(object->jolt_body_id); physics_system->GetBodyInterface().DestroyBody(object->jolt_body_id); }
		else checkSGP(sgp_body_remove(world, id), "removeObject");
		physics_system->GetBodyInterface().clearFrame(object->jolt_body_id);
		physics_system->registerCompound(object->jolt_body_id, 0);
		for (auto& release : object->shape_instance_releases) release();
		object->shape_instance_releases.clear();
		if (id < id_to_ob.size()) id_to_ob[id] = NULL;
		object->jolt_body_id = JPH::BodyID();
	}
	{
		Lock lock(activated_obs_mutex);
		activated_obs.erase(object.ptr());
		newly_activated_obs.erase(object.ptr());
	}
}

// PhysicsWorld.cpp:1342-1353
void PhysicsWorld::activateObject(const Reference<PhysicsObject>& object)
{
	if (object->jolt_body_id.IsInvalid()) return;
	checkSGP(sgp_body_activate(world, object->jolt_body_id.GetIndex()), "activateObject");
	drainActivationEvents();
}
void PhysicsWorld::setObjectLayer(const Reference<PhysicsObject>& object, uint8 new_object_layer)
{
	if (object->jolt_body_id.IsInvalid()) return;
	checkSGP(sgp_body_set_layer(world, object->jolt_body_id.GetIndex(), new_object_layer), "setObjectLayer");
}

// A device-side failure behind a facade call that has no error path in the reference (think(), the setters): surfaced the way the reference's
// shape builders surface theirs (glare::Exception, PhysicsWorld.cpp:762-763) instead of going on with a world that silently stopped moving.
static void checkSGP(int rc, const char* what)
{
	if (rc != SGP_OK) throw glare::Exception(std::string("PhysicsWorld::") + what + ": " + sgp_last_error());
}

// OnBodyActivated / OnBodyDeactivated (:1448-1486): maintain activated_obs / newly_activated_obs.
// The buffers are members sized to what is actually waiting (sgp_world_event_counts), never to the world's capacity, and never cleared.
void PhysicsWorld::drainActivationEvents()
{
	uint32_t counts[5];
	checkSGP(sgp_world_event_counts(world, counts), "drainActivationEvents");
	for (int kind = SGP_EVENT_ACTIVATED; kind <= SGP_EVENT_DEACTIVATED; ++kind) {
		if (!counts[kind]) continue;
		if (body_event_buf.size() < counts[kind]) body_event_buf.resize(counts[kind] + counts[kind] / 2 + 64);
		uint32_t n = 0;
		checkSGP(sgp_world_drain_events(world, kind, body_event_buf.data(), (uint32_t)body_event_buf.size(), &n), "drainActivationEvents");
		Lock lock(activated_obs_mutex);
		for (uint32_t i = 0; i < n && i < body_event_buf.size(); ++i) {
			PhysicsObject* ob = (PhysicsObject*)body_event_buf[i].userdata;
			if (!ob) continue;                                  // inBodyUserData != 0 (:1452,1475)
			if (kind == SGP_EVENT_ACTIVATED) { activated_obs.insert(ob); newly_activated_obs.insert(ob); }
			else activated_obs.erase(ob);
		}
	}
}

// PhysicsWorld.cpp:1356-1443
void PhysicsWorld::think(double dt)
{
	checkSGP(sgp_world_set_contact_events(world, event_listener ? 1 : 0), "think");
	checkSGP(sgp_world_step(world, (float)dt), "think");         // physics_system->Update((float)dt, 1, ...) (:1363) + buoyancy sweep (:1367-1442)
	physics_system->onStep();                                   // cached body / vehicle read-backs are stale now
	drainActivationEvents();

	uint32_t counts[5];
	checkSGP(sgp_world_event_counts(world, counts), "think");    // (host-side: the step already pulled the lists)
	if (event_listener) {
		// OnContactAdded / OnContactPersisted -> event_listener (:1499-1520), delivered on the calling thread
		for (int kind = SGP_EVENT_CONTACT_ADDED; kind <= SGP_EVENT_CONTACT_PERSISTED; ++kind) {
			if (!counts[kind]) continue;
			if (contact_event_buf.size() < counts[kind]) contact_event_buf.resize(counts[kind] + counts[kind] / 2 + 64);
			uint32_t n = 0;
			checkSGP(sgp_world_drain_events(world, kind, contact_event_buf.data(), (uint32_t)contact_event_buf.size(), &n), "think");
			for (uint32_t i = 0; i < n && i < contact_event_buf.size(); ++i) {
				const sgp_contact_event& e = contact_event_buf[i];
				JPH::Body b1, b2; JPH::ContactManifold m;
				b1.lin_vel = JPH::Vec3(e.lin_vel1[0], e.lin_vel1[1], e.lin_vel1[2]); b1.user_data = e.userdata1; b1.id = JPH::BodyID(e.id1);
				b2.lin_vel = JPH::Vec3(e.lin_vel2[0], e.lin_vel2[1], e.lin_vel2[2]); b2.user_data = e.userdata2; b2.id = JPH::BodyID(e.id2);
				m.mBaseOffset = JPH::Vec3(e.base_offset[0], e.base_offset[1], e.base_offset[2]);
				m.mWorldSpaceNormal = JPH::Vec3(e.normal[0], e.normal[1], e.normal[2]);
				m.mPenetrationDepth = e.penetration;
				for (uint32_t k = 0; k < e.num_points; ++k) m.mRelativeContactPointsOn1.push_back(JPH::Vec3(e.rel_points_on1[k][0], e.rel_points_on1[k][1], e.rel_points_on1[k][2]));
				if (kind == SGP_EVENT_CONTACT_ADDED) event_listener->contactAdded(b1, b2, m); else event_listener->contactPersisted(b1, b2, m);
			}
		}
	}

	if (water_buoyancy_enabled) {
		// underwater / last_submerged_volume / physicsObjectEnteredWater bookkeeping (:1414-1437)
		if (counts[SGP_EVENT_ENTERED_WATER]) {
			const uint32_t want = counts[SGP_EVENT_ENTERED_WATER];
			if (body_event_buf.size() < want) body_event_buf.resize(want + want / 2 + 64);
			uint32_t n = 0;
			checkSGP(sgp_world_drain_events(world, SGP_EVENT_ENTERED_WATER, body_event_buf.data(), (uint32_t)body_event_buf.size(), &n), "think");
			for (uint32_t i = 0; i < n && i < body_event_buf.size(); ++i) {
				PhysicsObject* ob = (PhysicsObject*)body_event_buf[i].userdata;
				if (ob && event_listener) event_listener->physicsObjectEnteredWater(*ob);
			}
		}
		Lock lock(activated_obs_mutex);
		water_ids.clear(); water_obs.clear();
		for (auto it = activated_obs.begin(); it != activated_obs.end(); ++it) if (!(*it)->jolt_body_id.IsInvalid()) { water_ids.push_back((*it)->jolt_body_id.GetIndex()); water_obs.push_back(*it); }
		if (water_states.size() < water_ids.size()) water_states.resize(water_ids.size() + water_ids.size() / 2 + 64);
		if (!water_ids.empty()) {
			checkSGP(sgp_body_get_state(world, water_ids.data(), (uint32_t)water_ids.size(), water_states.data()), "think");
			for (size_t i = 0; i < water_ids.size(); ++i) { water_obs[i]->underwater = water_states[i].underwater != 0; water_obs[i]->last_submerged_volume = water_states[i].submerged_volume; }
		}
	}
}

// GUIClient.cpp:6581-6690 in one batched read
void PhysicsWorld::readBackActivatedObjectTransforms()
{
	Lock lock(activated_obs_mutex);
	std::vector<uint32_t> ids; std::vector<PhysicsObject*> obs;
	for (auto it = activated_obs.begin(); it != activated_obs.end(); ++it) if (!(*it)->jolt_body_id.IsInvalid()) { ids.push_back((*it)->jolt_body_id.GetIndex()); obs.push_back(*it); }
	if (ids.empty()) return;
	std::vector<sgp_body_state> st(ids.size());
	if (sgp_body_get_state(world, ids.data(), (uint32_t)ids.size(), st.data()) != SGP_OK) return;
	for (size_t i = 0; i < ids.size(); ++i) {
		toObjectPose(*obs[i], st[i].pos, st[i].rot, obs[i]->pos, obs[i]->rot);
	}
}

// PhysicsWorld.cpp:546-604
void PhysicsWorld::setNewObToWorldTransform(PhysicsObject& object, const Vec4f& translation, const Quatf& rot_quat, const Vec4f& scale)
{
	assert(translation.isFinite());
	const Vec3f old_scale = object.scale;
	object.pos = translation; object.rot = rot_quat; object.scale = Vec3f(scale);
	if (object.jolt_body_id.IsInvalid()) return;
	const bool is_compound = object.shape.jolt_shape.GetPtr() && object.shape.jolt_shape->kind == 5;
	if (!object.is_sphere && !object.is_cube && (object.shape.kind == 3 || object.shape.kind == 4 || is_compound) &&
		(old_scale.x != object.scale.x || old_scale.y != object.scale.y || old_scale.z != object.scale.z)) {
		// JPH::ScaledShape swap (:562-601): hulls and meshes carry their scale baked into the device-side shape, so a new scale means the
		// shape instance of that scale (built on first use) and a body made from it; like the reference the body ends up activated with
		// zero velocity.  The PhysicsObject keeps its identity (userdata), only its body id changes.
		Reference<PhysicsObject> ref(&object);
		removeObject(ref);
		addObject(ref);
		if (!object.jolt_body_id.IsInvalid()) { sgp_body_activate(world, object.jolt_body_id.GetIndex()); drainActivationEvents(); }
		return;
	}
	float shape[4] = { 0, 0, 0, 0 };
	if (object.is_sphere) shape[0] = 0.5f * std::fabs(scale[0]);                       // sphere forced to uniform scale (:571-572)
	else if (object.is_cube) { shape[0] = 0.5f * std::fabs(scale[0]); shape[1] = 0.5f * std::fabs(scale[1]); shape[2] = 0.5f * std::fabs(scale[2]); }
	else if (object.shape.kind == 1) { shape[0] = object.shape.p[0] * std::fabs(scale[0]); shape[1] = object.shape.p[1] * std::fabs(scale[1]); shape[2] = object.shape.p[2] * std::fabs(scale[2]); }
	else if (object.shape.kind == 0) shape[0] = object.shape.p[0] * std::fabs(scale[0]);
	else { shape[0] = object.shape.p[0] * std::fabs(scale[0]); shape[1] = object.shape.p[1] * std::fabs(scale[2]); }
	float bp[3], br[4];
	toBodyPose(object, translation, rot_quat, bp, br);       // (a hull keeps the scale it was added with: its points are pre-scaled)
	checkSGP(sgp_body_set_pose_shape(world, object.jolt_body_id.GetIndex(), bp, br, shape), "setNewObToWorldTransform");   // zero velocity, new scale, ActivateBody (:553-601)
	physics_system->GetBodyInterface().invalidate();
	drainActivationEvents();
}

// PhysicsWorld.cpp:607-620
void PhysicsWorld::setNewObToWorldTransform(PhysicsObject& object, const Vec4f& pos, const Quatf& rot, const Vec4f& linear_vel, const Vec4f& angular_vel)
{
	assert(pos.isFinite());
	object.pos = pos; object.rot = rot;
	if (object.jolt_body_id.IsInvalid()) return;
	float bp[3], br[4];
	toBodyPose(object, pos, rot, bp, br);
	checkSGP(sgp_body_set_pose_vel(world, object.jolt_body_id.GetIndex(), bp, br, linear_vel.x, angular_vel.x), "setNewObToWorldTransform");
	physics_system->GetBodyInterface().invalidate();
}

// PhysicsWorld.cpp:623-633
void PhysicsWorld::setNewPosition(PhysicsObject& object, const Vec4f& pos)
{
	assert(pos.isFinite());
	object.pos = pos;
	if (object.jolt_body_id.IsInvalid()) return;
	float bp[3], br[4];
	toBodyPose(object, pos, object.rot, bp, br);
	checkSGP(sgp_body_set_pos(world, object.jolt_body_id.GetIndex(), bp), "setNewPosition");
	physics_system->GetBodyInterface().invalidate();
}

// PhysicsWorld.cpp:636-646
Vec4f PhysicsWorld::getObjectLinearVelocity(const PhysicsObject& object) const
{
	if (object.jolt_body_id.IsInvalid()) return Vec4f(0.f);
	const uint32_t id = object.jolt_body_id.GetIndex();
	sgp_body_state st;
	if (sgp_body_get_state(world, &id, 1, &st) != SGP_OK) return Vec4f(0.f);
	return Vec4f(st.lin_vel[0], st.lin_vel[1], st.lin_vel[2], 0.f);
}

// PhysicsWorld.cpp:649-657
void PhysicsWorld::setLinearAndAngularVelToZero(PhysicsObject& object)
{
	const float z[3] = { 0, 0, 0 };
	if (!object.jolt_body_id.IsInvalid()) { sgp_body_set_vel(world, object.jolt_body_id.GetIndex(), z, z); physics_system->GetBodyInterface().invalidate(); }
}

// Object-to-world and world-to-object matrices of a pose with a (possibly non-uniform) scale; same contract as the free function of
// PhysicsWorld.cpp:660-704 (a zero scale component is replaced by 1e-6 so that the inverse exists).  Written element-wise:
//   M = [ R diag(s) | t ],   M^-1 = [ diag(1/s) R^T | -diag(1/s) R^T t ]
void computeToWorldAndToObMatrices(const Vec4f& translation, const Quatf& rot_quat, const Vec4f& scale, Matrix4f& ob_to_world_out, Matrix4f& world_to_ob_out)
{
	float sc[3], inv_sc[3];
	for (int k = 0; k < 3; ++k) { sc[k] = scale[k] != 0.f ? scale[k] : 1.0e-6f; inv_sc[k] = 1.f / sc[k]; }
	const Matrix4f R = rot_quat.toMatrix();
	Matrix4f fwd = Matrix4f::identity(), inv = Matrix4f::identity();
	for (int col = 0; col < 3; ++col)
		for (int row = 0; row < 3; ++row) {
			const float r = R.e[col * 4 + row];
			fwd.e[col * 4 + row] = r * sc[col];              // column `col` of R, stretched by that axis' scale
			inv.e[row * 4 + col] = r * inv_sc[col];          // (R^T) with row `col` divided by the same scale
		}
	for (int row = 0; row < 3; ++row) {
		fwd.e[12 + row] = translation[row];
		inv.e[12 + row] = -(inv.e[row] * translation[0] + inv.e[4 + row] * translation[1] + inv.e[8 + row] * translation[2]);
	}
	ob_to_world_out = fwd;
	world_to_ob_out = inv;
}

// PhysicsWorld.cpp:707-722
void PhysicsWorld::moveKinematicObject(PhysicsObject& object, const Vec4f& translation, const Quatf& rot, float dt)
{
	if (object.jolt_body_id.IsInvalid()) return;
	if (object.motion_type != PhysicsObject::MotionType_kinematic) return;     // "Tried to move a non-kinematic object" guard (:713-720)
	float bp[3], br[4];
	toBodyPose(object, translation, rot, bp, br);
	checkSGP(sgp_body_move_kinematic(world, object.jolt_body_id.GetIndex(), bp, br, dt), "moveKinematicObject");
	physics_system->GetBodyInterface().invalidate();
}

void PhysicsWorld::addForce(PhysicsObject& object, const Vec4f& force) { if (!object.jolt_body_id.IsInvalid()) sgp_body_add_force(world, object.jolt_body_id.GetIndex(), force.x); }
void PhysicsWorld::addForceAtPoint(PhysicsObject& object, const Vec4f& force, const Vec4f& point) { if (!object.jolt_body_id.IsInvalid()) sgp_body_add_force_at(world, object.jolt_body_id.GetIndex(), force.x, point.x); }
void PhysicsWorld::addTorque(PhysicsObject& object, const Vec4f& torque) { if (!object.jolt_body_id.IsInvalid()) sgp_body_add_torque(world, object.jolt_body_id.GetIndex(), torque.x); }

void PhysicsWorld::clear() {}    // stub in the reference too (:1523-1526)

// PhysicsWorld.cpp:1529-1604
PhysicsWorld::MemUsageStats PhysicsWorld::getMemUsageStats() const
{
	sgp_step_stats s; memset(&s, 0, sizeof(s));
	sgp_world_stats(world, &s);
	sgp_body_counts c; memset(&c, 0, sizeof(c));
	sgp_world_body_counts(world, &c);
	MemUsageStats m; m.mem = (size_t)c.shape_bytes; m.num_meshes = c.num_meshes;      // (the reference sums the bytes of the shapes in use, :1553)
	m.layer_counts.assign(s.layer_counts, s.layer_counts + Layers::NUM_LAYERS);
	return m;
}
static std::string niceByteSize(size_t x)
{
	char buf[64];
	if (x < 1024) snprintf(buf, sizeof(buf), "%zu B", x);
	else if (x < (1u << 20)) snprintf(buf, sizeof(buf), "%.3f KB", (double)x / 1024.0);
	else if (x < (1u << 30)) snprintf(buf, sizeof(buf), "%.3f MB", (double)x / 1048576.0);
	else snprintf(buf, sizeof(buf), "%.3f GB", (double)x / 1073741824.0);
	return buf;
}
// the counters of the reference's diagnostics window, under its labels (PhysicsWorld.cpp:1578-1604), followed by this backend's own
std::string PhysicsWorld::getDiagnostics() const
{
	const MemUsageStats stats = getMemUsageStats();
	sgp_step_stats st; memset(&st, 0, sizeof(st));
	sgp_world_stats(world, &st);
	sgp_body_counts c; memset(&c, 0, sizeof(c));
	sgp_world_body_counts(world, &c);
	std::string s;
	s += "Jolt bodies: " + std::to_string(c.num_bodies) + "\n";
	s += "max bodies: " + std::to_string(c.max_bodies) + "\n";
	s += "num static bodies: " + std::to_string(c.num_static) + "\n";
	s += "num dynamic bodies: " + std::to_string(c.num_dynamic) + "\n";
	s += "num active dynamic bodies: " + std::to_string(c.num_active_dynamic) + "\n";
	s += "num kinematic bodies: " + std::to_string(c.num_kinematic) + "\n";
	s += "num active kinematic bodies: " + std::to_string(c.num_active_kinematic) + "\n";
	{
		Lock lock(activated_obs_mutex);
		s += "Active bodies: " + std::to_string(activated_obs.size()) + "\n";
	}
	s += "Meshes:  " + std::to_string(stats.num_meshes) + "\n";
	s += "mem usage: " + niceByteSize(stats.mem) + "\n";
	s += "NON_MOVING layer obs:                " + std::to_string(stats.layer_counts[Layers::NON_MOVING]) + "\n";
	s += "MOVING layer obs:                    " + std::to_string(stats.layer_counts[Layers::MOVING]) + "\n";
	s += "NON_MOVING_NON_COLLIDABLE layer obs: " + std::to_string(stats.layer_counts[Layers::NON_MOVING_NON_COLLIDABLE]) + "\n";
	s += "MOVING_NON_COLLIDABLE layer obs:     " + std::to_string(stats.layer_counts[Layers::MOVING_NON_COLLIDABLE]) + "\n";
	// (not in the reference: what the device step saw last)
	s += "convex hull shapes: " + std::to_string(c.num_hulls) + "\n";
	s += "body pairs: " + std::to_string(st.num_pairs) + ", contact constraints: " + std::to_string(st.num_manifolds) + ", colours: " + std::to_string(st.num_colours) + "\n";
	s += "device mem usage: " + niceByteSize((size_t)st.device_bytes) + "\n";
	return s;
}
std::string PhysicsWorld::getLoadedMeshes() const { return std::string(); }      // (the reference's body is commented out too: it returns "", :1607-1622)

// PhysicsWorld.cpp:1625-1638
const Vec4f PhysicsWorld::getPosInJolt(const Reference<PhysicsObject>& object)
{
	if (object->jolt_body_id.IsInvalid()) return object->pos;
	const uint32_t id = object->jolt_body_id.GetIndex();
	sgp_body_state st;
	if (sgp_body_get_state(world, &id, 1, &st) != SGP_OK) return object->pos;
	Vec4f p; Quatf q;
	toObjectPose(*object, st.pos, st.rot, p, q);
	return p;
}
size_t PhysicsWorld::getNumObjects() const { uint32_t n = 0; sgp_world_num_bodies(world, &n); return n; }

// PhysicsWorld.cpp:1728-1739 (debug): every body slot's state, raw
void PhysicsWorld::writeJoltSnapshotToDisk(const std::string& path)
{
	sgp_step_stats st; memset(&st, 0, sizeof(st));
	sgp_world_stats(world, &st);
	const uint32_t n = (uint32_t)id_to_ob.size();
	std::vector<sgp_body_state> states(n);
	if (n && sgp_world_read_states(world, 0, n, states.data()) != SGP_OK) return;
	FILE* f = fopen(path.c_str(), "wb");
	if (!f) return;
	const char magic[8] = { 'S', 'G', 'P', 'S', 'N', 'A', 'P', '1' };
	const uint32_t header[2] = { n, (uint32_t)sizeof(sgp_body_state) };
	fwrite(magic, 1, 8, f); fwrite(header, sizeof(uint32_t), 2, f);
	if (n) fwrite(states.data(), sizeof(sgp_body_state), n, f);
	fclose(f);
}

// PhysicsWorld.cpp:725-732
size_t PhysicsWorld::computeSizeBForShape(JPH::Ref<JPH::Shape> jolt_shape)
{
	if (!jolt_shape.GetPtr()) return 0;
	size_t b = sizeof(JPH::Shape) + jolt_shape->hull_points.size() * sizeof(float);
	if (jolt_shape->mesh) b += jolt_shape->mesh->vertices.size() * sizeof(float) + (jolt_shape->mesh->indices.size() + jolt_shape->mesh->materials.size()) * sizeof(uint32_t);
	for (const JPH::Shape::SubShape& c : jolt_shape->children) b += computeSizeBForShape(JPH::Ref<JPH::Shape>(const_cast<JPH::Shape*>(c.shape.GetPtr())));
	return b;
}
size_t PhysicsWorld::computeSizeBForShape(const PhysicsShape& shape)
{
	size_t b = sizeof(PhysicsShape);
	if (shape.hull) b += shape.hull->points.size() * sizeof(float);
	if (shape.mesh) b += shape.mesh->vertices.size() * sizeof(float) + shape.mesh->indices.size() * sizeof(uint32_t);
	return b;
}

// PhysicsWorld.cpp:1668-1725
static void doTraceRay(sgp_world* world, const Vec4f& origin, const Vec4f& dir, float max_t, JPH::BodyID ignore_body_id, bool collidable_only, RayTraceResult& results_out)
{
	results_out.hit_object = NULL;
	sgp_ray r; memset(&r, 0, sizeof(r));
	for (int i = 0; i < 3; ++i) { r.origin[i] = origin[i]; r.dir[i] = dir[i]; }
	r.max_t = max_t; r.ignore_id = ignore_body_id.GetIndex(); r.collidable_only = collidable_only ? 1u : 0u;
	sgp_hit h;
	if (sgp_raycast(world, &r, 1, &h) != SGP_OK || h.id == SGP_INVALID_ID) return;
	if (h.userdata != 0) {                                                       // :1687-1690
		results_out.hit_object = (PhysicsObject*)h.userdata;
		results_out.coords = Vec2f(0.f);
		results_out.hit_t = h.t;
		results_out.hit_normal_ws = Vec4f(h.normal[0], h.normal[1], h.normal[2], 0.f);
		results_out.hit_mat_index = h.material;                                 // mesh_shape->GetTriangleUserData(...) for mesh shapes, else 0 (:1698-1704)
	}
}
void PhysicsWorld::traceRay(const Vec4f& origin, const Vec4f& dir, float max_t, JPH::BodyID ignore_body_id, RayTraceResult& results_out) const
{ doTraceRay(world, origin, dir, max_t, ignore_body_id, false, results_out); }
void PhysicsWorld::traceRayAgainstCollidableObs(const Vec4f& origin, const Vec4f& dir, float max_t, JPH::BodyID ignore_body_id, RayTraceResult& results_out) const
{ doTraceRay(world, origin, dir, max_t, ignore_body_id, true, results_out); }
void PhysicsWorld::traceRays(const std::vector<RayQuery>& rays, std::vector<RayTraceResult>& results_out) const
{
	results_out.resize(rays.size());
	if (rays.empty()) return;
	std::vector<sgp_ray> rs(rays.size()); std::vector<sgp_hit> hs(rays.size());
	memset(rs.data(), 0, sizeof(sgp_ray) * rs.size());
	for (size_t k = 0; k < rays.size(); ++k) {
		for (int i = 0; i < 3; ++i) { rs[k].origin[i] = rays[k].origin[i]; rs[k].dir[i] = rays[k].dir[i]; }
		rs[k].max_t = rays[k].max_t; rs[k].ignore_id = rays[k].ignore_body_id.GetIndex(); rs[k].collidable_only = rays[k].collidable_only ? 1u : 0u;
	}
	const bool ok = sgp_raycast(world, rs.data(), (uint32_t)rs.size(), hs.data()) == SGP_OK;
	for (size_t k = 0; k < rays.size(); ++k) {
		RayTraceResult& r = results_out[k];
		r.hit_object = NULL;
		if (!ok || hs[k].id == SGP_INVALID_ID || hs[k].userdata == 0) continue;
		r.hit_object = (PhysicsObject*)hs[k].userdata;
		r.coords = Vec2f(0.f);
		r.hit_t = hs[k].t;
		r.hit_normal_ws = Vec4f(hs[k].normal[0], hs[k].normal[1], hs[k].normal[2], 0.f);
		r.hit_mat_index = hs[k].material;
	}
}
bool PhysicsWorld::doesRayHitAnything(const Vec4f& origin, const Vec4f& dir, float max_t) const
{
	sgp_ray r; memset(&r, 0, sizeof(r));
	for (int i = 0; i < 3; ++i) { r.origin[i] = origin[i]; r.dir[i] = dir[i]; }
	r.max_t = max_t; r.ignore_id = SGP_INVALID_ID;
	sgp_hit h;
	return sgp_raycast(world, &r, 1, &h) == SGP_OK && h.id != SGP_INVALID_ID;
}


// ---------------------------------------------------------------------------------------------------------------
// JPH::BodyInterface look-alike over the C ABI (declared in Jolt/JoltLite.h)

void JPH::BodyInterface::fetch(const BodyID& id) const
{
	const uint32_t i = id.GetIndex();
	if (i == cached_id) return;
	sgp_body_state st;
	memset(&st, 0, sizeof(st)); st.rot[3] = 1.f;
	if (!id.IsInvalid()) sgp_body_get_state(world, &i, 1, &st);
	for (int k = 0; k < 3; ++k) { st_pos[k] = st.pos[k]; st_lv[k] = st.lin_vel[k]; st_av[k] = st.ang_vel[k]; }
	for (int k = 0; k < 4; ++k) st_rot[k] = st.rot[k];
	st_active = st.active != 0;
	cached_id = i;
}
void JPH::BodyInterface::ActivateBody(const BodyID& id) { if (!id.IsInvalid()) { sgp_body_activate(world, id.GetIndex()); invalidate(); } }
void JPH::BodyInterface::AddForce(const BodyID& id, const Vec3& f) { if (!id.IsInvalid()) sgp_body_add_force(world, id.GetIndex(), &f.x); }
void JPH::BodyInterface::AddForce(const BodyID& id, const Vec3& f, const RVec3& p) { if (!id.IsInvalid()) sgp_body_add_force_at(world, id.GetIndex(), &f.x, &p.x); }
void JPH::BodyInterface::AddTorque(const BodyID& id, const Vec3& t) { if (!id.IsInvalid()) sgp_body_add_torque(world, id.GetIndex(), &t.x); }
// Position / rotation of the SHAPE's origin and axes, like Jolt: a hull body is simulated in its centre-of-mass / principal-axes frame
// (frames), so  rot_shape = rot_body * frame.rot^-1,  pos_shape = pos_body - rot_shape * frame.com.
JPH::Quat JPH::BodyInterface::GetRotation(const BodyID& id) const
{
	fetch(id);
	const Quat qb(st_rot[0], st_rot[1], st_rot[2], st_rot[3]);
	const Frame* f = getFrame(id);
	return f ? qb * f->rot.Conjugated() : qb;
}
JPH::RVec3 JPH::BodyInterface::GetPosition(const BodyID& id) const
{
	fetch(id);
	const RVec3 pb(st_pos[0], st_pos[1], st_pos[2]);
	const Frame* f = getFrame(id);
	return f ? pb - GetRotation(id) * f->com : pb;
}
JPH::RVec3 JPH::BodyInterface::GetCenterOfMassPosition(const BodyID& id) const { fetch(id); return RVec3(st_pos[0], st_pos[1], st_pos[2]); }   // the body frame's origin IS the centre of mass
void JPH::BodyInterface::GetPositionAndRotation(const BodyID& id, RVec3& p, Quat& r) const { p = GetPosition(id); r = GetRotation(id); }
JPH::Mat44 JPH::BodyInterface::GetWorldTransform(const BodyID& id) const
{
	const Quat q = GetRotation(id);
	Mat44 m;
	m.c[0] = q * Vec3(1, 0, 0); m.c[1] = q * Vec3(0, 1, 0); m.c[2] = q * Vec3(0, 0, 1); m.c[3] = GetPosition(id);
	return m;
}
JPH::Vec3 JPH::BodyInterface::GetLinearVelocity(const BodyID& id) const { fetch(id); return Vec3(st_lv[0], st_lv[1], st_lv[2]); }
JPH::Vec3 JPH::BodyInterface::GetAngularVelocity(const BodyID& id) const { fetch(id); return Vec3(st_av[0], st_av[1], st_av[2]); }
void JPH::BodyInterface::GetLinearAndAngularVelocity(const BodyID& id, Vec3& l, Vec3& a) const { l = GetLinearVelocity(id); a = GetAngularVelocity(id); }
JPH::Vec3 JPH::BodyInterface::GetPointVelocity(const BodyID& id, const RVec3& p) const
{
	fetch(id);
	const Vec3 r(p.x - st_pos[0], p.y - st_pos[1], p.z - st_pos[2]);
	return Vec3(st_lv[0] + (st_av[1] * r.z - st_av[2] * r.y), st_lv[1] + (st_av[2] * r.x - st_av[0] * r.z), st_lv[2] + (st_av[0] * r.y - st_av[1] * r.x));
}
void JPH::BodyInterface::SetLinearAndAngularVelocity(const BodyID& id, const Vec3& l, const Vec3& a) { if (!id.IsInvalid()) { sgp_body_set_vel(world, id.GetIndex(), &l.x, &a.x); invalidate(); } }
bool JPH::BodyInterface::IsActive(const BodyID& id) const { fetch(id); return st_active; }

void JPH::BodyInterface::setFrame(const BodyID& id, const Vec3& com, const Quat& rot) { Frame f; f.com = com; f.rot = rot; frames[id.GetIndex()] = f; }
void JPH::BodyInterface::clearFrame(const BodyID& id) { frames.erase(id.GetIndex()); }
const JPH::BodyInterface::Frame* JPH::BodyInterface::getFrame(const BodyID& id) const { auto it = frames.find(id.GetIndex()); return it == frames.end() ? nullptr : &it->second; }
JPH::BodyInterface::~BodyInterface() { for (auto& kv : bodies) delete kv.second; }

// CarPhysics.cpp:80-90 / BikePhysics.cpp:107-121
JPH::Body* JPH::BodyInterface::CreateBody(const BodyCreationSettings& s)
{
	const Shape* shape = s.GetShape();
	if (!shape || shape->kind < 0) return nullptr;
	sgp_body_desc d;
	sgp_default_body_desc(&d);
	d.shape_type = shape->kind;
	Vec3 com(0, 0, 0); Quat frot(0, 0, 0, 1);
	if (shape->kind == 3) {
		sgp_hull_info info;
		if (sgp_hull_create_com(world, shape->hull_points.data(), (uint32_t)(shape->hull_points.size() / 3), shape->com_offset, &info) != SGP_OK) return nullptr;
		d.shape[0] = (float)info.hull_id;
		com = Vec3(info.com[0], info.com[1], info.com[2]); frot = Quat(info.rot[0], info.rot[1], info.rot[2], info.rot[3]);
	} else { d.shape[0] = shape->p[0]; d.shape[1] = shape->p[1]; d.shape[2] = shape->p[2]; }
	const Vec3 pb = s.mPosition + s.mRotation * com;
	const Quat qb = s.mRotation * frot;
	d.pos[0] = pb.x; d.pos[1] = pb.y; d.pos[2] = pb.z;
	d.rot[0] = qb.x; d.rot[1] = qb.y; d.rot[2] = qb.z; d.rot[3] = qb.w;
	d.lin_vel[0] = s.mLinearVelocity.x; d.lin_vel[1] = s.mLinearVelocity.y; d.lin_vel[2] = s.mLinearVelocity.z;
	d.ang_vel[0] = s.mAngularVelocity.x; d.ang_vel[1] = s.mAngularVelocity.y; d.ang_vel[2] = s.mAngularVelocity.z;
	d.motion_type = s.mMotionType == EMotionType::Dynamic ? SGP_MOTION_DYNAMIC : (s.mMotionType == EMotionType::Kinematic ? SGP_MOTION_KINEMATIC : SGP_MOTION_STATIC);
	d.layer = (int32_t)s.mObjectLayer;
	d.is_sensor = s.mIsSensor ? 1 : 0; d.allow_sleeping = s.mAllowSleeping ? 1 : 0;
	d.friction = s.mFriction; d.restitution = s.mRestitution; d.linear_damping = s.mLinearDamping; d.angular_damping = s.mAngularDamping; d.gravity_factor = s.mGravityFactor;
	// EOverrideMassProperties::CalculateInertia: the given mass, inertia from the shape (the only mode the callers use; without an override
	// Jolt derives the mass from the shape's volume at density 1000)
	if (s.mOverrideMassProperties != EOverrideMassProperties::CalculateMassAndInertia && s.mMassPropertiesOverride.mMass > 0.0f) d.mass = s.mMassPropertiesOverride.mMass;
	else if (shape->kind != 3) d.mass = 1000.0f * shape->volume;
	d.userdata = s.mUserData;
	d.activate = 0;
	uint32_t id = SGP_INVALID_ID;
	if (sgp_body_add(world, &d, &id) != SGP_OK || id == SGP_INVALID_ID) return nullptr;
	Body* b = new Body;
	b->id = BodyID(id); b->user_data = s.mUserData; b->is_sensor = s.mIsSensor;
	b->shape.kind = shape->kind; b->shape.p[0] = shape->p[0]; b->shape.p[1] = shape->p[1]; b->shape.p[2] = shape->p[2];
	float vol = shape->volume; sgp_body_get_volume(world, id, &vol); b->shape.volume = vol;
	b->com_offset = com; b->frame_rot[0] = frot.x; b->frame_rot[1] = frot.y; b->frame_rot[2] = frot.z; b->frame_rot[3] = frot.w;
	bodies[id] = b;
	if (shape->kind == 3) setFrame(b->id, com, frot);
	invalidate();
	return b;
}
void JPH::BodyInterface::AddBody(const BodyID& id, EActivation activation)
{
	if (id.IsInvalid()) return;
	if (activation == EActivation::Activate) sgp_body_activate(world, id.GetIndex());
	invalidate();
}
void JPH::BodyInterface::RemoveBody(const BodyID& id) { if (!id.IsInvalid()) { sgp_body_remove(world, id.GetIndex()); invalidate(); } }
void JPH::BodyInterface::DestroyBody(const BodyID& id)
{
	auto it = bodies.find(id.GetIndex());
	if (it != bodies.end()) { delete it->second; bodies.erase(it); }
	clearFrame(id);
}


// ---- JPH::PhysicsSystem constraint registration (vehicles) -----------------------------------------------------------
void JPH::PhysicsSystem::AddConstraint(VehicleConstraint* c)
{
	if (!c) return;
	sgp_vehicle_desc d;
	c->fillDesc(d);
	uint32_t id = 0xFFFFFFFFu;
	if (sgp_vehicle_create(world, &d, &id) != SGP_OK) throw glare::Exception(std::string("Error creating vehicle: ") + sgp_last_error());
	c->bind(world, id, &step_serial);
}
void JPH::PhysicsSystem::RemoveConstraint(VehicleConstraint* c)
{
	if (!c || !c->world) return;
	sgp_vehicle_destroy(world, c->GetVehicleID());
	c->unbind();
}

bool JPH::BodyLockInterface::fill(const BodyID& id, Body& out) const
{
	float vol = 0.f;
	if (id.IsInvalid() || sgp_body_get_volume(world, id.GetIndex(), &vol) != SGP_OK) return false;
	sgp_body_state st;
	const uint32_t i = id.GetIndex();
	if (sgp_body_get_state(world, &i, 1, &st) != SGP_OK) return false;
	uint64_t ud = 0; sgp_body_get_userdata(world, i, &ud);
	out.id = id; out.shape.volume = vol; out.lin_vel = Vec3(st.lin_vel[0], st.lin_vel[1], st.lin_vel[2]); out.user_data = ud;
	return true;
}
JPH::Body* JPH::BodyLockInterface::TryGetBody(const BodyID& id) const { return fill(id, scratch) ? &scratch : nullptr; }
JPH::BodyLockRead::BodyLockRead(const BodyLockInterface& iface, const BodyID& id) : ok(iface.fill(id, body)) {}
