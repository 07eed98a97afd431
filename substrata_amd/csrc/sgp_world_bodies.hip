// sgp_world_bodies.hip -- body lifecycle, setters, forces and read-back of the C ABI (PhysicsWorld::addObject :1169-1311, the setters :546-722,
// activation / contact bookkeeping :1448-1520 of /root/reference/gui_client/PhysicsWorld.cpp).  Host side only: edits are queued as commands for k_apply_cmds.
#include "sgp_world_internal.h"

// ---------------------------------------------------------------------------------------------------------------
// bodies

static void mass_properties(int type, const float* p, float mass, float* inv_mass, float* inv_inertia)
{
	// Shape::GetMassProperties scaled to the overridden mass (EOverrideMassProperties::CalculateInertia, PhysicsWorld.cpp:1239)
	float ix, iy, iz;
	if (type == SGP_SHAPE_SPHERE) { const float i = 0.4f * mass * p[0] * p[0]; ix = iy = iz = i; }
	else if (type == SGP_SHAPE_BOX) {
		const float sx = 2.0f * p[0], sy = 2.0f * p[1], sz = 2.0f * p[2];
		const float k = mass / 12.0f;
		ix = k * (sy * sy + sz * sz); iy = k * (sx * sx + sz * sz); iz = k * (sx * sx + sy * sy);
	} else {
		const float r = p[0], H = 2.0f * p[1];
		const float vc = 3.14159265358979323846f * r * r * H;
		const float vs = (4.0f / 3.0f) * 3.14159265358979323846f * r * r * r;
		const float mc = mass * vc / (vc + vs), ms = mass * vs / (vc + vs);
		iz = 0.5f * mc * r * r + 0.4f * ms * r * r;
		ix = mc * (3.0f * r * r + H * H) / 12.0f + ms * (0.4f * r * r + 0.25f * H * H + 0.375f * H * r);
		iy = ix;
	}
	*inv_mass = 1.0f / mass;
	inv_inertia[0] = 1.0f / ix; inv_inertia[1] = 1.0f / iy; inv_inertia[2] = 1.0f / iz;
}

static float bounding_radius(int type, const float* p)
{
	if (type == SGP_SHAPE_SPHERE) return p[0];
	if (type == SGP_SHAPE_BOX) return sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
	return p[0] + p[1];
}

static inline float clamp01(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }

static float host_shape_volume(int type, const float* p)
{
	const float pi = 3.14159265358979323846f;
	if (type == SGP_SHAPE_SPHERE) return (4.0f / 3.0f) * pi * p[0] * p[0] * p[0];
	if (type == SGP_SHAPE_BOX) return 8.0f * p[0] * p[1] * p[2];
	return pi * p[0] * p[0] * (2.0f * p[1]) + (4.0f / 3.0f) * pi * p[0] * p[0] * p[0];
}

#define SGP_LG_MAX_PENDING 64u
static void note_radius(sgp_world* w, uint32_t id, float r)
{
	HostBody& b = w->hb[id];
	const bool was_large = b.flags & BF_LARGE;
	const bool is_large = r > w->desc.large_body_radius;
	b.bound_radius = r;
	if (is_large) b.flags |= BF_LARGE; else b.flags &= ~BF_LARGE;
	if (is_large != was_large || (is_large && (b.flags & BF_ALIVE))) {
		if (is_large && !b.in_large_ids) { w->large_ids.push_back(id); b.in_large_ids = 1; }
		if (!is_large) b.in_large_ids = 0;                      // (its entry in large_ids goes at the next rebuild)
		const bool is_static = (b.flags & BF_MOTION_MASK) == SGP_MOTION_STATIC;
		if (is_large && !was_large && is_static && w->lg_static >= 32u && !w->large_dirty && !b.lg_tomb && b.lg_state == 0 && w->lg_pending < SGP_LG_MAX_PENDING) {
			// a new static large body while a grid stands: onto the linear list until the next rebuild
			w->large_linear.push_back(id); b.lg_state = 2; w->lg_pending++; w->large_list_dirty = true;
		} else w->large_dirty = true;
	}
	if (!is_large) w->max_small_radius = std::max(w->max_small_radius, r);
}

int add_one(sgp_world* w, const sgp_body_desc* d, uint32_t* id_out, bool ghost)
{
	if (!finite3(d->pos) || fabsf(d->pos[0]) > 1.0e9f || fabsf(d->pos[1]) > 1.0e9f || fabsf(d->pos[2]) > 1.0e9f) return SGP_ERR_REJECTED;   // :1178
	if (d->shape_type < 0 || d->shape_type > SGP_SHAPE_MESH) return fail(SGP_ERR_INVALID, "sgp_body_add: bad shape_type");
	const sgd_hull* hull = nullptr;
	const bool is_mesh = d->shape_type == SGP_SHAPE_MESH;
	if (is_mesh) {
		const uint32_t mid = (uint32_t)d->shape[0];
		if (!(d->shape[0] >= 1.0f) || (float)mid != d->shape[0] || mid >= w->meshes.size()) return fail(SGP_ERR_INVALID, "sgp_body_add: bad mesh id");
		if (d->motion_type == SGP_MOTION_DYNAMIC) return fail(SGP_ERR_INVALID, "sgp_body_add: mesh shapes are for static and kinematic bodies (JPH::MeshShape has no mass properties)");
		if (w->meshes[mid].nt == 0) return fail(SGP_ERR_INVALID, "sgp_body_add: the mesh has been destroyed");
	}
	if (d->shape_type == SGP_SHAPE_HULL) {
		const uint32_t hid = (uint32_t)d->shape[0];
		if (!(d->shape[0] >= 1.0f) || (float)hid != d->shape[0] || hid >= w->hulls.size()) return fail(SGP_ERR_INVALID, "sgp_body_add: bad hull id");
		if (w->hulls[hid].nv == 0) return fail(SGP_ERR_INVALID, "sgp_body_add: the hull has been destroyed");
		hull = &w->hulls[hid];
	}
	const int nparam = d->shape_type == SGP_SHAPE_BOX ? 3 : (d->shape_type == SGP_SHAPE_SPHERE ? 1 : ((d->shape_type == SGP_SHAPE_HULL || is_mesh) ? 0 : 2));
	for (int i = 0; i < nparam; ++i) {
		const float lim = (d->shape_type == SGP_SHAPE_CAPSULE && i == 1) ? 0.0f : 0.5e-7f;   // |scale| < 1e-7 on a 0.5 unit shape, :1184
		if (!std::isfinite(d->shape[i]) || d->shape[i] < lim) return SGP_ERR_REJECTED;
	}
	uint32_t id;
	if (is_mesh) {
		// three consecutive slots: the body and its two aliases (second / third contact manifold of a pair) -- the triple a removed mesh body
		// left behind, else fresh ones
		if (!w->free_triples.empty()) { id = w->free_triples.back(); w->free_triples.pop_back(); }
		else { if (w->high + 3 > w->dv.cap_bodies) return fail(SGP_ERR_CAPACITY, "sgp_body_add: max_bodies exceeded"); id = w->high; w->high += 3; }
	}
	else if (!w->free_list.empty()) { id = w->free_list.back(); w->free_list.pop_back(); }
	else { if (w->high >= w->dv.cap_bodies) return fail(SGP_ERR_CAPACITY, "sgp_body_add: max_bodies exceeded"); id = w->high++; }
	BodyCmd c; memset(&c, 0, sizeof(c));
	c.id = id; c.ops = CMD_CREATE;
	memcpy(c.pos, d->pos, sizeof(c.pos)); memcpy(c.rot, d->rot, sizeof(c.rot));
	memcpy(c.linv, d->lin_vel, sizeof(c.linv)); memcpy(c.angv, d->ang_vel, sizeof(c.angv));
	memcpy(c.shape, d->shape, sizeof(c.shape));
	c.friction = clamp01(d->friction);                 // :1236
	c.restitution = clamp01(d->restitution);           // :1237
	c.mass = std::max(0.001f, d->mass);                // :1238
	c.gravity_factor = d->gravity_factor; c.lin_damp = d->linear_damping; c.ang_damp = d->angular_damping;
	c.userdata = d->userdata;
	if (d->motion_type == SGP_MOTION_DYNAMIC) {
		if (hull) {
			// MassProperties of the hull scaled to the overridden mass; the body frame already is the principal frame
			const float density = c.mass / hull->volume;
			c.inv_mass = 1.0f / c.mass;
			c.inv_inertia[0] = 1.0f / (hull->unit_inertia.x * density); c.inv_inertia[1] = 1.0f / (hull->unit_inertia.y * density); c.inv_inertia[2] = 1.0f / (hull->unit_inertia.z * density);
		} else mass_properties(d->shape_type, d->shape, c.mass, &c.inv_mass, c.inv_inertia);
	}
	uint32_t f = BF_ALIVE | ((uint32_t)d->motion_type & BF_MOTION_MASK) | (((uint32_t)d->layer & 0x3u) << BF_LAYER_SHIFT) |
	             (((uint32_t)d->shape_type & 0x7u) << BF_SHAPE_SHIFT);
	if (d->is_sensor) f |= BF_SENSOR;
	if (d->allow_sleeping) f |= BF_ALLOW_SLEEP;
	if (d->use_zero_linear_drag) f |= BF_ZERO_LIN_DRAG;
	if (ghost) f |= BF_GHOST;
	HostBody& hb = w->hb[id];
	hb.flags = f; hb.userdata = d->userdata; hb.ghost = ghost; hb.comp_root = SGP_INVALID_ID; hb.comp_child = 0;
	hb.shape_ref = (is_mesh || d->shape_type == SGP_SHAPE_HULL) ? (uint32_t)d->shape[0] : 0u;
	if (is_mesh) w->mesh_refs[hb.shape_ref]++; else if (hb.shape_ref) w->hull_refs[hb.shape_ref]++;
	if (hb.lg_tomb) w->large_dirty = true;      // the slot of a static large body that left a dead entry in the device grid: the grid is rebuilt before anything can find the newcomer through it
	for (uint32_t k = 1; is_mesh && k <= 2; ++k) if (id + k < w->hb.size() && w->hb[id + k].lg_tomb) w->large_dirty = true;
	note_radius(w, id, is_mesh ? 3.0e38f : (hull ? hull->bound_radius : bounding_radius(d->shape_type, d->shape)));   // (meshes always go through the large-body list)
	hb.volume = is_mesh ? 0.0f : (hull ? hull->volume : host_shape_volume(d->shape_type, d->shape));
	c.flags = hb.flags;
	w->cmds.push_back(c);
	if (is_mesh) for (uint32_t k = 1; k <= 2; ++k) {
		// aliases: same pose and material, flagged large (so never binned) but absent from the large-body list (so never paired or queried)
		BodyCmd a = c; a.id = id + k; a.flags = hb.flags | BF_ALIAS | BF_LARGE;
		HostBody& ha = w->hb[id + k]; ha.flags = a.flags; ha.userdata = d->userdata; ha.ghost = false; ha.bound_radius = 0.0f; ha.volume = 0.0f; ha.comp_root = SGP_INVALID_ID; ha.comp_child = 0;
		w->cmds.push_back(a);
	}
	if (d->activate && d->motion_type != SGP_MOTION_STATIC) { BodyCmd a; memset(&a, 0, sizeof(a)); a.id = id; a.ops = CMD_ACTIVATE; w->cmds.push_back(a); }
	w->n_alive++;
	if (id_out) *id_out = id;
	return SGP_OK;
}

// The host's share of add_one for a body the DEVICE creates from a received record (k_create_from_records; primitive shapes only): the slot, the mirror of the
// flags, radius and volume -- in the order add_one does it, so that slots are handed out exactly as before.  *flags_io: in = the flags add_one would compose from
// the description, out = with BF_LARGE as note_radius decided (what the device must store).
int book_record_body(sgp_world* w, uint32_t* flags_io, uint64_t userdata, float radius, float volume, bool ghost, uint32_t* id_out)
{
	uint32_t id;
	if (!w->free_list.empty()) { id = w->free_list.back(); w->free_list.pop_back(); }
	else { if (w->high >= w->dv.cap_bodies) return fail(SGP_ERR_CAPACITY, "sgp_body_add: max_bodies exceeded"); id = w->high++; }
	HostBody& hb = w->hb[id];
	hb.flags = *flags_io; hb.userdata = userdata; hb.ghost = ghost; hb.comp_root = SGP_INVALID_ID; hb.comp_child = 0; hb.shape_ref = 0u;
	if (hb.lg_tomb) w->large_dirty = true;
	note_radius(w, id, radius);
	hb.volume = volume;
	*flags_io = hb.flags;
	w->n_alive++;
	*id_out = id;
	return SGP_OK;
}

SGP_API int sgp_body_add(sgp_world* w, const sgp_body_desc* d, uint32_t* id_out)
{
	if (!w || !d) return fail(SGP_ERR_INVALID, "sgp_body_add: NULL");
	if (id_out) *id_out = SGP_INVALID_ID;
	return add_one(w, d, id_out, false);
}

SGP_API int sgp_body_add_batch(sgp_world* w, const sgp_body_desc* d, uint32_t n, uint32_t* ids_out)
{
	if (!w || (!d && n)) return fail(SGP_ERR_INVALID, "sgp_body_add_batch: NULL");
	w->cmds.reserve(w->cmds.size() + 2 * (size_t)n);
	for (uint32_t i = 0; i < n; ++i) {
		uint32_t id = SGP_INVALID_ID;
		const int r = add_one(w, &d[i], &id, false);
		if (r != SGP_OK && r != SGP_ERR_REJECTED) return r;
		if (ids_out) ids_out[i] = (r == SGP_OK) ? id : SGP_INVALID_ID;
	}
	return SGP_OK;
}

// world pose of a compound's child: pos = P + R * p_k, rot = R * q_k (plain float arithmetic, written out so that the CPU checker can state the same operations)
static void compound_child_pose(const float P[3], const float R[4], const sgp_compound_child& c, float pos_out[3], float rot_out[4])
{
	const float x = R[0], y = R[1], z = R[2], w_ = R[3];
	const float vx = c.pos[0], vy = c.pos[1], vz = c.pos[2];
	const float tx = 2.0f * (y * vz - z * vy), ty = 2.0f * (z * vx - x * vz), tz = 2.0f * (x * vy - y * vx);
	pos_out[0] = P[0] + (vx + w_ * tx + (y * tz - z * ty));
	pos_out[1] = P[1] + (vy + w_ * ty + (z * tx - x * tz));
	pos_out[2] = P[2] + (vz + w_ * tz + (x * ty - y * tx));
	const float ox = c.rot[0], oy = c.rot[1], oz = c.rot[2], ow = c.rot[3];
	rot_out[0] = w_ * ox + x * ow + y * oz - z * oy;
	rot_out[1] = w_ * oy - x * oz + y * ow + z * ox;
	rot_out[2] = w_ * oz + x * oy - y * ox + z * ow;
	rot_out[3] = w_ * ow - x * ox - y * oy - z * oz;
}
static inline bool is_compound_child(const sgp_world* w, uint32_t id) { return w->hb[id].comp_root != SGP_INVALID_ID && w->hb[id].comp_root != id; }
static inline CompoundRec* compound_of(sgp_world* w, uint32_t id) { auto it = w->compounds.find(id); return it == w->compounds.end() ? nullptr : &it->second; }

SGP_API int sgp_body_remove(sgp_world* w, uint32_t id);
SGP_API int sgp_body_add_compound(sgp_world* w, const sgp_body_desc* base, const sgp_compound_child* children, uint32_t n, uint32_t* id_out)
{
	if (!w || !base || !children || !id_out) return fail(SGP_ERR_INVALID, "sgp_body_add_compound: NULL");
	*id_out = SGP_INVALID_ID;
	if (n < 1 || n > SGP_MAX_COMPOUND_CHILDREN) return fail(SGP_ERR_INVALID, "sgp_body_add_compound: 1..64 children");
	if (base->motion_type != SGP_MOTION_STATIC) return fail(SGP_ERR_INVALID, "sgp_body_add_compound: compound bodies are static (JPH::StaticCompoundShape on a static object)");
	if (!finite3(base->pos) || !finite4(base->rot)) return SGP_ERR_REJECTED;
	for (uint32_t k = 0; k < n; ++k) if (!finite3(children[k].pos) || !finite4(children[k].rot)) return fail(SGP_ERR_INVALID, "sgp_body_add_compound: non-finite child pose");
	CompoundRec rec;
	memcpy(rec.pos, base->pos, 12); memcpy(rec.rot, base->rot, 16);
	for (uint32_t k = 0; k < n; ++k) {
		sgp_body_desc d = *base;
		d.shape_type = children[k].shape_type; memcpy(d.shape, children[k].shape, 16);
		compound_child_pose(base->pos, base->rot, children[k], d.pos, d.rot);
		d.activate = 0;
		uint32_t cid = SGP_INVALID_ID;
		const int r = add_one(w, &d, &cid, false);
		if (r != SGP_OK) {                  // all or nothing
			for (uint32_t j = 0; j < rec.ids.size(); ++j) { w->hb[rec.ids[j]].comp_root = SGP_INVALID_ID; sgp_body_remove(w, rec.ids[j]); }
			return r;
		}
		rec.ids.push_back(cid); rec.children.push_back(children[k]);
	}
	const uint32_t root = rec.ids[0];
	for (uint32_t k = 0; k < n; ++k) { w->hb[rec.ids[k]].comp_root = root; w->hb[rec.ids[k]].comp_child = k; }
	w->n_alive -= (n - 1);                 // one object, however many slots
	w->compounds[root] = std::move(rec);
	*id_out = root;
	return SGP_OK;
}
SGP_API int sgp_body_compound_size(sgp_world* w, uint32_t id, uint32_t* n_out)
{
	if (!live(w, id) || !n_out) return fail(SGP_ERR_BAD_ID, "sgp_body_compound_size: id not live");
	const CompoundRec* c = compound_of(w, id);
	*n_out = c ? (uint32_t)c->ids.size() : 0u;
	return SGP_OK;
}
static inline bool is_mesh_body(const sgp_world* w, uint32_t id) { return ((w->hb[id].flags & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT) == SGP_SHAPE_MESH && !(w->hb[id].flags & BF_ALIAS); }
// queue a pose edit; a static mesh body owns the two alias slots behind it (second / third contact manifold of a pair), which share its pose
static void push_pose_cmd_one(sgp_world* w, const BodyCmd& c)
{
	w->cmds.push_back(c);
	// (a kinematic mesh body -- a scripted door, a lift -- also shares its velocities with them: a contact on the second group of a pair must see the platform move)
	if (is_mesh_body(w, c.id)) for (uint32_t k = 1; k <= 2; ++k) { BodyCmd a = c; a.id = c.id + k; a.ops &= (CMD_SET_POS | CMD_SET_ROT | CMD_SET_VEL | CMD_MOVE_KINEMATIC); if (a.ops) w->cmds.push_back(a); }
}
// ... and a compound moves all its children: each gets the compound's new pose composed with its own
static void push_pose_cmd(sgp_world* w, const BodyCmd& c)
{
	CompoundRec* rec = compound_of(w, c.id);
	if (!rec) { push_pose_cmd_one(w, c); return; }
	if (c.ops & CMD_SET_POS) memcpy(rec->pos, c.pos, 12);
	if (c.ops & CMD_SET_ROT) memcpy(rec->rot, c.rot, 16);
	for (size_t k = 0; k < rec->ids.size(); ++k) {
		BodyCmd a = c; a.id = rec->ids[k];
		a.ops &= ~(CMD_SET_SHAPE | CMD_SET_VEL);
		if (c.ops & (CMD_SET_POS | CMD_SET_ROT)) { a.ops |= CMD_SET_POS | CMD_SET_ROT; compound_child_pose(rec->pos, rec->rot, rec->children[k], a.pos, a.rot); }
		push_pose_cmd_one(w, a);
	}
}
#define REJECT_COMPOUND_CHILD(what) do { if (is_compound_child(w, id)) return fail(SGP_ERR_BAD_ID, what ": the id is a child slot of a compound body; use the compound's id"); } while (0)

SGP_API int sgp_body_remove(sgp_world* w, uint32_t id)
{
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_remove: id not live");
	if (is_compound_child(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_remove: a compound's child is removed with the compound");
	if (CompoundRec* c = compound_of(w, id)) {
		const std::vector<uint32_t> ids = c->ids;
		w->compounds.erase(id);
		for (uint32_t k : ids) w->hb[k].comp_root = SGP_INVALID_ID;
		for (size_t k = 1; k < ids.size(); ++k) { const int r = sgp_body_remove(w, ids[k]); if (r != SGP_OK) return r; w->n_alive++; }
		// (falls through: the first child's slot is removed like any body and accounts for the one object)
	}
	for (uint32_t v = 0; v < w->n_vehicles; ++v) if (w->veh_alive[v] && w->veh_body[v] == id) sgp_vehicle_destroy(w, v);   // a vehicle does not outlive its chassis
	if (w->hb[id].flags & BF_LARGE) {
		HostBody& b = w->hb[id];
		b.in_large_ids = 0;                                    // (large_ids is compacted at the next rebuild)
		if (b.lg_state == 1 && !w->large_dirty && (w->lg_tombs + 1u) * 4u <= w->lg_static) { b.lg_state = 0; b.lg_tomb = 1; w->lg_tombs++; }      // a dead entry stays in the device grid: nothing to do now
		else if (b.lg_state == 2 && !w->large_dirty) { w->large_linear.erase(std::remove(w->large_linear.begin(), w->large_linear.end(), id), w->large_linear.end()); b.lg_state = 0; w->lg_pending--; w->large_list_dirty = true; }
		else w->large_dirty = true;
	}
	if (w->hb[id].flags & BF_ALIAS) return fail(SGP_ERR_BAD_ID, "sgp_body_remove: id not live");
	const bool was_mesh = ((w->hb[id].flags & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT) == SGP_SHAPE_MESH;
	if (w->hb[id].shape_ref) { if (was_mesh) w->mesh_refs[w->hb[id].shape_ref]--; else w->hull_refs[w->hb[id].shape_ref]--; w->hb[id].shape_ref = 0; }
	const uint32_t nslots = was_mesh ? 3u : 1u;
	for (uint32_t k = 0; k < nslots; ++k) { w->hb[id + k].flags = 0; w->cmds.push_back(blank_cmd(id + k, CMD_REMOVE)); }
	if (was_mesh) w->free_triples.push_back(id); else w->free_list.push_back(id);       // a triple stays a triple: the next mesh body reuses it
	w->n_alive--;
	return SGP_OK;
}
SGP_API int sgp_body_activate(sgp_world* w, uint32_t id)
{
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_activate: id not live");
	w->cmds.push_back(blank_cmd(id, CMD_ACTIVATE));
	return SGP_OK;
}
// Body::GetShape()->GetVolume() (BoatPhysics.cpp:40-43)
SGP_API int sgp_body_get_volume(sgp_world* w, uint32_t id, float* volume_out)
{
	if (!live(w, id) || !volume_out) return fail(SGP_ERR_BAD_ID, "sgp_body_get_volume: id not live");
	*volume_out = w->hb[id].volume;
	return SGP_OK;
}
SGP_API int sgp_body_get_userdata(sgp_world* w, uint32_t id, uint64_t* userdata_out)
{
	if (!live(w, id) || !userdata_out) return fail(SGP_ERR_BAD_ID, "sgp_body_get_userdata: id not live");
	uint32_t b = id;
	while (b > 0 && (w->hb[b].flags & BF_ALIAS)) --b;          // an alias slot reports as the body it belongs to
	*userdata_out = w->hb[b].userdata;
	return SGP_OK;
}
SGP_API int sgp_body_set_layer(sgp_world* w, uint32_t id, int32_t layer)
{
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_set_layer: id not live");
	REJECT_COMPOUND_CHILD("sgp_body_set_layer");
	const CompoundRec* rec = compound_of(w, id);
	const size_t n = rec ? rec->ids.size() : 1;
	for (size_t k = 0; k < n; ++k) {
		const uint32_t b = rec ? rec->ids[k] : id;
		BodyCmd c = blank_cmd(b, CMD_SET_LAYER); c.flags = (uint32_t)layer & 0x3u;
		w->hb[b].flags = (w->hb[b].flags & ~BF_LAYER_MASK) | (((uint32_t)layer & 0x3u) << BF_LAYER_SHIFT);
		w->cmds.push_back(c);
	}
	return SGP_OK;
}
SGP_API int sgp_body_set_pose_vel(sgp_world* w, uint32_t id, const float pos[3], const float rot[4], const float lv[3], const float av[3])
{
	REQUIRE_FINITE(pos && rot && lv && av && finite3(pos) && finite4(rot) && finite3(lv) && finite3(av), "sgp_body_set_pose_vel");
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_set_pose_vel: id not live");
	REJECT_COMPOUND_CHILD("sgp_body_set_pose_vel");
	BodyCmd c = blank_cmd(id, CMD_SET_POS | CMD_SET_ROT | CMD_SET_VEL);
	memcpy(c.pos, pos, 12); memcpy(c.rot, rot, 16); memcpy(c.linv, lv, 12); memcpy(c.angv, av, 12);
	push_pose_cmd(w, c);
	return SGP_OK;
}
SGP_API int sgp_body_set_pose_vel_batch(sgp_world* w, const uint32_t* ids, const sgp_pose_vel* recs, uint32_t n)
{
	if (!w || ((!ids || !recs) && n)) return fail(SGP_ERR_INVALID, "sgp_body_set_pose_vel_batch: NULL");
	for (uint32_t i = 0; i < n; ++i) if (!live(w, ids[i])) return fail(SGP_ERR_BAD_ID, "sgp_body_set_pose_vel_batch: id not live");
	for (uint32_t i = 0; i < n; ++i) if (is_compound_child(w, ids[i])) return fail(SGP_ERR_BAD_ID, "sgp_body_set_pose_vel_batch: an id is a child slot of a compound body; use the compound's id");
	for (uint32_t i = 0; i < n; ++i) REQUIRE_FINITE(finite3(recs[i].pos) && finite4(recs[i].rot) && finite3(recs[i].lin_vel) && finite3(recs[i].ang_vel), "sgp_body_set_pose_vel_batch");
	w->cmds.reserve(w->cmds.size() + n);
	for (uint32_t i = 0; i < n; ++i) {
		BodyCmd c = blank_cmd(ids[i], CMD_SET_POS | CMD_SET_ROT | CMD_SET_VEL);
		memcpy(c.pos, recs[i].pos, 12); memcpy(c.rot, recs[i].rot, 16); memcpy(c.linv, recs[i].lin_vel, 12); memcpy(c.angv, recs[i].ang_vel, 12);
		push_pose_cmd(w, c);
	}
	return SGP_OK;
}

// ObjectPhysicsTransformUpdate payload, GUIClient.cpp:7637-7650 (host-side byte packing; x86-64 / little endian)

SGP_API int sgp_body_set_pose_shape(sgp_world* w, uint32_t id, const float pos[3], const float rot[4], const float shape[4])
{
	REQUIRE_FINITE(pos && rot && shape && finite3(pos) && finite4(rot) && finite4(shape), "sgp_body_set_pose_shape");
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_set_pose_shape: id not live");
	REJECT_COMPOUND_CHILD("sgp_body_set_pose_shape");
	BodyCmd c = blank_cmd(id, CMD_SET_POS | CMD_SET_ROT | CMD_SET_VEL | CMD_SET_SHAPE | CMD_ACTIVATE);
	memcpy(c.pos, pos, 12); memcpy(c.rot, rot, 16); memcpy(c.shape, shape, 16);
	const int type = (int)((w->hb[id].flags & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT);
	if (type == SGP_SHAPE_HULL || type == SGP_SHAPE_MESH || compound_of(w, id)) c.ops &= ~CMD_SET_SHAPE;          // hulls, meshes and compounds are pre-scaled (shape.x = table id): only the pose changes
	else {
		note_radius(w, id, bounding_radius(type, shape));
		w->hb[id].volume = host_shape_volume(type, shape);
		c.flags = w->hb[id].flags & BF_LARGE;      // the device copy of the flag follows the host's
	}
	push_pose_cmd(w, c);
	return SGP_OK;
}
SGP_API int sgp_body_set_pos(sgp_world* w, uint32_t id, const float pos[3])
{
	REQUIRE_FINITE(pos && finite3(pos), "sgp_body_set_pos");
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_set_pos: id not live");
	REJECT_COMPOUND_CHILD("sgp_body_set_pos");
	BodyCmd c = blank_cmd(id, CMD_SET_POS); memcpy(c.pos, pos, 12);
	push_pose_cmd(w, c);
	return SGP_OK;
}
SGP_API int sgp_body_set_vel(sgp_world* w, uint32_t id, const float lv[3], const float av[3])
{
	REQUIRE_FINITE(lv && av && finite3(lv) && finite3(av), "sgp_body_set_vel");
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_set_vel: id not live");
	BodyCmd c = blank_cmd(id, CMD_SET_VEL); memcpy(c.linv, lv, 12); memcpy(c.angv, av, 12);
	push_pose_cmd_one(w, c);
	return SGP_OK;
}
SGP_API int sgp_body_move_kinematic(sgp_world* w, uint32_t id, const float tp[3], const float tr[4], float dt)
{
	REQUIRE_FINITE(tp && tr && finite3(tp) && finite4(tr) && std::isfinite(dt), "sgp_body_move_kinematic");
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_move_kinematic: id not live");
	BodyCmd c = blank_cmd(id, CMD_MOVE_KINEMATIC); memcpy(c.pos, tp, 12); memcpy(c.rot, tr, 16); c.dt = dt;
	push_pose_cmd_one(w, c);
	return SGP_OK;
}
SGP_API int sgp_body_add_force(sgp_world* w, uint32_t id, const float f[3])
{
	REQUIRE_FINITE(f && finite3(f), "sgp_body_add_force");
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_add_force: id not live");
	BodyCmd c = blank_cmd(id, CMD_ADD_FORCE); memcpy(c.linv, f, 12);
	w->cmds.push_back(c);
	return SGP_OK;
}
SGP_API int sgp_body_add_force_at(sgp_world* w, uint32_t id, const float f[3], const float p[3])
{
	REQUIRE_FINITE(f && p && finite3(f) && finite3(p), "sgp_body_add_force_at");
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_add_force_at: id not live");
	BodyCmd c = blank_cmd(id, CMD_ADD_FORCE_AT); memcpy(c.linv, f, 12); memcpy(c.pos, p, 12);
	w->cmds.push_back(c);
	return SGP_OK;
}
SGP_API int sgp_body_add_torque(sgp_world* w, uint32_t id, const float t[3])
{
	REQUIRE_FINITE(t && finite3(t), "sgp_body_add_torque");
	if (!live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_body_add_torque: id not live");
	BodyCmd c = blank_cmd(id, CMD_ADD_TORQUE); memcpy(c.angv, t, 12);
	w->cmds.push_back(c);
	return SGP_OK;
}

// Upload the pending edits: grouped by body (submission order kept inside a group), one thread per body.

// ---------------------------------------------------------------------------------------------------------------
// read-back

SGP_API int sgp_body_get_state(sgp_world* w, const uint32_t* ids, uint32_t n, sgp_body_state* out)
{
	if (!w || (!ids && n) || (!out && n)) return fail(SGP_ERR_INVALID, "sgp_body_get_state: NULL");
	hipSetDevice(w->device);
	for (uint32_t i = 0; i < n; ++i) if (!live(w, ids[i])) return fail(SGP_ERR_BAD_ID, "sgp_body_get_state: id not live");
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	if (!n) return SGP_OK;
	const size_t ids_bytes = (sizeof(uint32_t) * n + 15) & ~size_t(15);
	{ int r = ensure_stage(w, ids_bytes + sizeof(sgp_body_state) * n); if (r != SGP_OK) return r; }
	memcpy(w->stage_host, ids, sizeof(uint32_t) * n);
	HIP_TRY(hipMemcpyAsync(w->stage_dev, w->stage_host, sizeof(uint32_t) * n, hipMemcpyHostToDevice, w->stream));
	sgp_body_state* dout = (sgp_body_state*)((char*)w->stage_dev + ids_bytes);
	launch_gather_states(w->dv, (const uint32_t*)w->stage_dev, 0, n, dout, w->stream);
	HIP_TRY(hipMemcpyAsync((char*)w->stage_host + ids_bytes, dout, sizeof(sgp_body_state) * n, hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	memcpy(out, (char*)w->stage_host + ids_bytes, sizeof(sgp_body_state) * n);
	return SGP_OK;
}

SGP_API int sgp_world_read_states(sgp_world* w, uint32_t first, uint32_t n, sgp_body_state* out)
{
	if (!w || (!out && n)) return fail(SGP_ERR_INVALID, "sgp_world_read_states: NULL");
	if ((uint64_t)first + n > w->dv.cap_bodies) return fail(SGP_ERR_INVALID, "sgp_world_read_states: range exceeds max_bodies");
	hipSetDevice(w->device);
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	if (!n) return SGP_OK;
	{ int r = ensure_stage(w, sizeof(sgp_body_state) * n); if (r != SGP_OK) return r; }
	launch_gather_states(w->dv, nullptr, first, n, (sgp_body_state*)w->stage_dev, w->stream);
	HIP_TRY(hipMemcpyAsync(w->stage_host, w->stage_dev, sizeof(sgp_body_state) * n, hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	memcpy(out, w->stage_host, sizeof(sgp_body_state) * n);
	return SGP_OK;
}

// The compacted states of the active bodies land in the pinned staging buffer with ONE host sync: the gather, the counters and a copy sized
// from the previous step's active count (+ slack) are queued together; only a count above that estimate costs a second copy.
// (to_view: the records land in the pinned buffer that only the *_view entry points use, so that a ray cast, a state query or any other call that
//  stages data through stage_host cannot overwrite -- or reallocate -- what a caller is still iterating over)
static int read_active_to_stage(sgp_world* w, uint32_t cap, uint32_t* n_out, uint32_t* m_out, bool poses_only = false, bool to_view = false)
{
	const size_t rec = poses_only ? sizeof(sgp_body_pose) : sizeof(sgp_body_state);
	hipSetDevice(w->device);
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	const uint32_t lim = std::min(cap, w->dv.cap_bodies);
	{ int r = ensure_stage(w, rec * std::max(lim, 1u)); if (r != SGP_OK) return r; }
	void* host_dst = w->stage_host;
	if (to_view) {
		const size_t need = rec * std::max(lim, 1u);
		if (need > w->view_host_bytes) {
			if (w->view_host) { hipStreamSynchronize(w->stream); hipHostFree(w->view_host); w->view_host = nullptr; w->view_host_bytes = 0; }
			HIP_TRY(hipHostMalloc(&w->view_host, need, hipHostMallocDefault));
			w->view_host_bytes = need;
		}
		host_dst = w->view_host;
	}
	HIP_TRY(hipMemsetAsync(&w->dv.ctr->n_read_active, 0, sizeof(uint32_t), w->stream));
	if (poses_only) launch_gather_active_poses(w->dv, w->high, w->stage_dev, lim, w->stream);
	else launch_gather_active(w->dv, w->high, (sgp_body_state*)w->stage_dev, lim, w->stream);
	const uint32_t guess = std::min(lim, w->last_active + w->last_active / 16u + 256u);
	if (guess) HIP_TRY(hipMemcpyAsync(host_dst, w->stage_dev, rec * guess, hipMemcpyDeviceToHost, w->stream));
	{ int r = read_counters(w); if (r != SGP_OK) return r; }      // (the one sync)
	const uint32_t n = w->h_ctr->n_read_active;
	const uint32_t m = std::min(n, lim);
	if (m > guess) {
		HIP_TRY(hipMemcpyAsync((char*)host_dst + rec * guess, (char*)w->stage_dev + rec * guess,
		                       rec * (m - guess), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	*n_out = n; *m_out = m;
	return SGP_OK;
}

SGP_API int sgp_world_read_active(sgp_world* w, sgp_body_state* out, uint32_t cap, uint32_t* n_out)
{
	if (!w || !n_out) return fail(SGP_ERR_INVALID, "sgp_world_read_active: NULL");
	uint32_t n = 0, m = 0;
	{ int r = read_active_to_stage(w, out ? cap : 0u, &n, &m); if (r != SGP_OK) return r; }
	if (m && out) memcpy(out, w->stage_host, sizeof(sgp_body_state) * m);
	*n_out = n;
	return SGP_OK;
}

SGP_API int sgp_world_read_active_view(sgp_world* w, const sgp_body_state** view_out, uint32_t* n_out)
{
	if (!w || !view_out || !n_out) return fail(SGP_ERR_INVALID, "sgp_world_read_active_view: NULL");
	uint32_t n = 0, m = 0;
	{ int r = read_active_to_stage(w, w->dv.cap_bodies, &n, &m, false, true); if (r != SGP_OK) return r; }
	*view_out = (const sgp_body_state*)w->view_host;
	*n_out = m;
	return SGP_OK;
}

SGP_API int sgp_world_read_active_poses_view(sgp_world* w, const sgp_body_pose** view_out, uint32_t* n_out)
{
	if (!w || !view_out || !n_out) return fail(SGP_ERR_INVALID, "sgp_world_read_active_poses_view: NULL");
	uint32_t n = 0, m = 0;
	{ int r = read_active_to_stage(w, w->dv.cap_bodies, &n, &m, true, true); if (r != SGP_OK) return r; }
	*view_out = (const sgp_body_pose*)w->view_host;
	*n_out = m;
	return SGP_OK;
}

SGP_API int sgp_world_body_counts(sgp_world* w, sgp_body_counts* out)
{
	if (!w || !out) return fail(SGP_ERR_INVALID, "sgp_world_body_counts: NULL");
	memset(out, 0, sizeof(*out));
	out->max_bodies = w->dv.cap_bodies;
	for (uint32_t i = 0; i < w->high; ++i) {
		const uint32_t f = w->hb[i].flags;
		if ((f & (BF_ALIVE | BF_ALIAS)) != BF_ALIVE) continue;
		out->num_bodies++;
		const uint32_t m = f & BF_MOTION_MASK;
		if (m == SGP_MOTION_STATIC) out->num_static++; else if (m == SGP_MOTION_DYNAMIC) out->num_dynamic++; else out->num_kinematic++;
	}
	// who is awake lives on the device: one read-back of the active ids
	uint32_t n = 0, m = 0;
	{ int r = read_active_to_stage(w, w->dv.cap_bodies, &n, &m, true); if (r != SGP_OK) return r; }
	const sgp_body_pose* poses = (const sgp_body_pose*)w->stage_host;
	for (uint32_t k = 0; k < m; ++k) {
		const uint32_t id = poses[k].id;
		if (id >= w->high) continue;
		const uint32_t mt = w->hb[id].flags & BF_MOTION_MASK;
		if (mt == SGP_MOTION_DYNAMIC) out->num_active_dynamic++; else if (mt == SGP_MOTION_KINEMATIC) out->num_active_kinematic++;
	}
	for (size_t k = 1; k < w->meshes.size(); ++k) if (w->meshes[k].nt != 0) { out->num_meshes++; out->shape_bytes += sizeof(MeshHeader) + 16ull * w->meshes[k].nv + 16ull * w->meshes[k].nt + sizeof(MeshNode) * (uint64_t)w->meshes[k].n_nodes; }
	for (size_t k = 1; k < w->hulls.size(); ++k) if (w->hulls[k].nv != 0) { out->num_hulls++; out->shape_bytes += sizeof(sgd_hull); }
	return SGP_OK;
}

template <typename T, typename Cmp> static void drain(std::vector<T>& v, void* out, uint32_t cap, uint32_t* n_out, Cmp cmp)
{
	std::sort(v.begin(), v.end(), cmp);
	const uint32_t m = std::min<uint32_t>((uint32_t)v.size(), cap);
	if (out && m) memcpy(out, v.data(), sizeof(T) * m);
	*n_out = (uint32_t)v.size();
	v.clear();
}

SGP_API int sgp_world_drain_events(sgp_world* w, int kind, void* out, uint32_t cap, uint32_t* n_out)
{
	if (!w || !n_out) return fail(SGP_ERR_INVALID, "sgp_world_drain_events: NULL");
	hipSetDevice(w->device);
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	{ int r = collect_events(w); if (r != SGP_OK) return r; }
	auto bcmp = [](const sgp_body_event& a, const sgp_body_event& b) { return a.id < b.id; };
	// total order (a pair can have several events: the manifolds of a body against a mesh, or several steps drained together)
	auto ccmp = [](const sgp_contact_event& a, const sgp_contact_event& b) {
		if (a.id1 != b.id1) return a.id1 < b.id1;
		if (a.id2 != b.id2) return a.id2 < b.id2;
		for (int k = 0; k < 3; ++k) if (a.base_offset[k] != b.base_offset[k]) return a.base_offset[k] < b.base_offset[k];
		for (int k = 0; k < 3; ++k) if (a.normal[k] != b.normal[k]) return a.normal[k] < b.normal[k];
		return a.penetration < b.penetration;
	};
	switch (kind) {
	case SGP_EVENT_ACTIVATED: drain(w->ev_act, out, cap, n_out, bcmp); break;
	case SGP_EVENT_DEACTIVATED: drain(w->ev_deact, out, cap, n_out, bcmp); break;
	case SGP_EVENT_ENTERED_WATER: drain(w->ev_water, out, cap, n_out, bcmp); break;
	case SGP_EVENT_CONTACT_ADDED: drain(w->ev_added, out, cap, n_out, ccmp); break;
	case SGP_EVENT_CONTACT_PERSISTED: drain(w->ev_pers, out, cap, n_out, ccmp); break;
	default: return fail(SGP_ERR_INVALID, "sgp_world_drain_events: bad kind");
	}
	return SGP_OK;
}

SGP_API int sgp_world_event_counts(sgp_world* w, uint32_t counts_out[5])
{
	if (!w || !counts_out) return fail(SGP_ERR_INVALID, "sgp_world_event_counts: NULL");
	hipSetDevice(w->device);
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	{ int r = collect_events(w); if (r != SGP_OK) return r; }
	counts_out[SGP_EVENT_ACTIVATED] = (uint32_t)w->ev_act.size(); counts_out[SGP_EVENT_DEACTIVATED] = (uint32_t)w->ev_deact.size();
	counts_out[SGP_EVENT_ENTERED_WATER] = (uint32_t)w->ev_water.size();
	counts_out[SGP_EVENT_CONTACT_ADDED] = (uint32_t)w->ev_added.size(); counts_out[SGP_EVENT_CONTACT_PERSISTED] = (uint32_t)w->ev_pers.size();
	return SGP_OK;
}

// Test / debug view of the constraints of the last step (sorted by pair key on the host).
struct DumpRec { uint32_t a, b; int32_t colour; int32_t np; float n[3]; float lam_n[4]; float lam_t1[4]; float lam_t2[4]; float bias[4]; };
SGP_API int sgp_world_dump_constraints(sgp_world* w, void* out, uint32_t cap, uint32_t* n_out)
{
	if (!w || !n_out) return fail(SGP_ERR_INVALID, "sgp_world_dump_constraints: NULL");
	hipSetDevice(w->device);
	const uint32_t n = w->n_con;
	*n_out = n;
	const uint32_t m = std::min(n, cap);
	if (!m || !out) return SGP_OK;
	{ int r = ensure_stage(w, sizeof(DumpRec) * n); if (r != SGP_OK) return r; }
	launch_dump_constraints(w->dv, (w->h_sp->parity & 1u) ^ 1u, n, w->stage_dev, n, w->stream);
	HIP_TRY(hipMemcpyAsync(w->stage_host, w->stage_dev, sizeof(DumpRec) * n, hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	DumpRec* r = (DumpRec*)w->stage_host;
	std::sort(r, r + n, [](const DumpRec& x, const DumpRec& y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
	memcpy(out, r, sizeof(DumpRec) * m);
	return SGP_OK;
}

