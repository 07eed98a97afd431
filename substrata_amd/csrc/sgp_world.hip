// sgp_world.hip -- host side of libsgp.so: the C ABI of include/sgp.h over the gfx950 kernels.
//
// Replaces what gui_client/PhysicsWorld.cpp does with Jolt: world construction (:462-532), addObject (:1169-1311),
// setters (:546-722), think (:1356-1443), activation / contact bookkeeping (:1448-1520), ray queries (:1668-1725).
// The job system / temp allocator adapters (:288-457) have no counterpart: work is HIP launches on one stream,
// scratch lives in device arenas allocated once per world.
//
// There is no CPU compute path in this file: without a HIP device every entry point fails with SGP_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <unordered_map>
#include <map>
#include <chrono>
#include "sgp_kernels.h"
#include "sgp_device_vehicle.h"
#include "sgp_hull_build.h"

#define SGP_API extern "C" __attribute__((visibility("default")))

static thread_local std::string g_last_error;
static int g_device_count = -1;

static int fail(int code, const char* what, hipError_t e = hipSuccess)
{
	char buf[512];
	if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
	else snprintf(buf, sizeof(buf), "%s", what);
	g_last_error = buf;
	return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(SGP_ERR_HIP, #expr, e_); } while (0)

static const char* k_class_names[KC_COUNT] = {
	"apply_forces", "bp_cell", "bp_scan", "bp_scatter", "bp_pairs", "bp_large", "narrowphase", "wake",
	"colour_claim", "colour_commit", "colour_count", "setup", "warm_start", "solve_velocity",
	"integrate_pose", "solve_position", "finalize", "island_hook", "island_flag", "sleep_apply", "buoyancy",
	"cache_build", "misc", "edit", "gather", "prep_bodies", "vehicle" };

// ---------------------------------------------------------------------------------------------------------------

struct HostBody {
	uint32_t flags = 0;          // mirror of the static part of the device flags (alive, motion, layer, shape, large)
	uint64_t userdata = 0;
	float bound_radius = 0.0f;
	float volume = 0.0f;         // Shape::GetVolume of the current shape
	bool ghost = false;
	uint32_t shape_ref = 0;                // the mesh / hull id the body references (0 = none): keeps sgp_mesh_destroy / sgp_hull_destroy honest
	uint32_t comp_root = SGP_INVALID_ID;   // child of a static compound body: slot of the compound (= its first child), else invalid
	uint32_t comp_child = 0;               // index among the compound's children
	uint8_t in_large_ids = 0;              // listed in sgp_world::large_ids (no search needed to know)
	uint8_t lg_state = 0;                  // the static large bodies' grid: 0 not in it, 1 in the device grid, 2 waiting on the linear list for the next rebuild
	uint8_t lg_tomb = 0;                   // this id still has a (dead) entry in the device grid: giving the slot to a new body forces the rebuild
};

// A static compound body (sgp_body_add_compound): the slots of its children and their poses in the compound's frame
struct CompoundRec { std::vector<uint32_t> ids; std::vector<sgp_compound_child> children; float pos[3]; float rot[4]; };

struct ProfEvent { int kc; hipEvent_t a, b; };

struct sgp_world {
	sgp_world_desc desc;
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t capture_stream = nullptr;      // launch plans are captured here while the step they belong to already runs, issued eagerly, on `stream`
	DV dv;
	std::vector<void*> allocs;
	uint64_t device_bytes = 0;
	// host mirrors
	std::vector<HostBody> hb;
	std::vector<uint32_t> free_list;
	uint32_t high = 0, n_alive = 0;
	std::vector<uint32_t> large_ids; bool large_dirty = false;      // every large body (host order; may hold ids that have gone: rebuild_large_grid compacts it)
	// round 4: the grid of the static large bodies is rebuilt in full only now and then -- a newcomer waits on the linear list every body walks (lg_pending
	// of them at most), a removed one stays behind as a dead entry (lg_tombs; a query skips what is not alive): streaming one parcel object in or out is
	// a 64-entry list upload or nothing at all, not a read-back and re-sort of 65k bounds (advisor r03)
	uint32_t lg_pending = 0, lg_tombs = 0; bool large_list_dirty = false;
	// what the device sees of them: the static ones in a grid of their own (LargeGrid, rebuilt when the set or a pose in it changes), the rest --
	// moving large bodies, static ones that would fill too many cells -- on the linear list the kernels walk
	std::vector<uint32_t> large_linear;
	LargeGrid* d_lgrid = nullptr; uint32_t* d_lg_start = nullptr; uint32_t* d_lg_items = nullptr; uint32_t cap_lg_items = 0; uint32_t lg_static = 0;
	uint32_t* d_large = nullptr; uint32_t cap_large = 0;
	float max_small_radius = 0.0f;
	uint32_t last_export = 0;                              // records the previous sgp_world_export_boundary produced
	std::unordered_map<uint64_t, uint64_t> ghost_map;      // global id of a ghost -> generation << 32 | local body id (stable across steps)
	uint32_t ghost_gen = 0; bool ghost_map_stale = false; uint64_t ghost_seq_version = 1;      // version: bumped whenever ghost_seq changes (a device copy of the ids knows whether it is current)
	//      // ghost_map is rebuilt from ghost_seq when the general import path needs it
	std::vector<GhostRefresh> ghost_refresh;               // pose refreshes of existing ghosts queued by the last import (uploaded by flush_cmds)
	std::vector<std::pair<uint64_t, uint32_t>> ghost_seq;   // (global id, local id) of the previous import, in its order (fast path of the next one)
	std::unordered_map<uint32_t, CompoundRec> compounds;    // compound id (= first child's slot) -> record
	// pending edits
	std::vector<BodyCmd> cmds;
	// staging
	void* stage_dev = nullptr; size_t stage_dev_bytes = 0;
	void* stage_host = nullptr; size_t stage_host_bytes = 0;
	void* view_host = nullptr; size_t view_host_bytes = 0;       // pinned buffer of sgp_world_read_active[_poses]_view only
	StepCounters* h_ctr = nullptr; StepCounters* h_ctr_dev = nullptr; EventCounters* h_evc = nullptr; EventCounters* h_evc_dev = nullptr;
	bool cache_wiped = false;                                  // the contact cache was emptied by an idle step (step_impl)
	bool dirty_since_step = true;                              // an edit was flushed since the last step (or no step yet)
	bool events_on_device = true;                              // the device event lists may hold something the host vectors do not (a step without read-back, applied edits)
	StepParams sp_uploaded; bool sp_uploaded_valid = false;    // what d_sp holds (upload_sp skips the launch when nothing changed)
	uint32_t last_active = 0xFFFFFFFFu;
	StepParams* h_sp = nullptr; StepParams* d_sp = nullptr;      // pinned host copy / device copy of the per-step scalars
	std::map<std::string, hipGraphExec_t> graphs;              // replayable launch sequences keyed by launch plan
	std::string last_plan_key[2]; uint32_t plan_repeats[2] = { 0, 0 };   // per buffer parity: StepParams (by value in the first launch) flips parity every step
	bool use_graphs = true; bool use_small_world = true; bool use_wake_round = true; uint32_t tail_threshold = 256;
	uint32_t rows_mode_large = 2;          // SGP_ROWS_MODE: the layout worlds of at least compact_rows_min constraints use -- 2 no rows (the lanes rebuild them from the lever arms), 1 compact rows (r x axis only)
	uint32_t compact_rows_min = 1000000;   // SGP_COMPACT_ROWS_MIN: from this many contact constraints on, the velocity rows are stored compact (96 B per point)
	int use_tile_solver = 0;            // SGP_TILE_SOLVER: 0 off, 1 on where the plan finds it applicable (k_ts_solve)
	uint32_t ts_min_constraints = 16384;
	// high colours by component: share of the constraints they may hold (per mille; SGP_HC_BUDGET, 0 = off), and the plan's correction of it
	// hc_k: the first colour that goes to the components (-1: not chosen yet -> the budget rule).  One colour fewer after a step that left a
	// component to the catch-all; one more after a probe (component sizes computed for hc_k - 1 without using them) found that it fits.
	bool use_components = true; uint32_t hc_budget = 160; uint32_t n_cus = 256; uint32_t hc_min_colours = 4; int hc_k = -1; uint32_t hc_bump = 1, hc_since_bump = 0xFFFFu; uint32_t hc_probe_in = 8, hc_probe_gap = 16;
	bool bp_dense_last = false;     // the previous step's broad phase met a halo too large for the small instance of k_bp_pairs
	bool plan_seen = false;         // a step has run: plan_colour_count etc. describe the previous step
	uint32_t graph_launches = 0, eager_steps = 0, idle_steps = 0;
	bool grid_valid = false;                                   // the broad-phase grid matches the current poses (ray queries reuse it)
	// static triangle meshes: host-side headers + pools mirrored on the device (grown on demand)
	std::vector<MeshHeader> meshes; std::vector<float4> mesh_verts; std::vector<uint4> mesh_tris; std::vector<uint32_t> mesh_tri_mat; std::vector<MeshNode> mesh_nodes;
	MeshHeader* d_meshes = nullptr; float4* d_mesh_verts = nullptr; uint4* d_mesh_tris = nullptr; uint32_t* d_mesh_tri_mat = nullptr; MeshNode* d_mesh_nodes = nullptr;
	size_t cap_mesh_verts = 0, cap_mesh_tris = 0, cap_mesh_tri_mat = 0, cap_mesh_nodes = 0;
	// shape lifecycle: bodies referencing each mesh / hull, ids and pool ranges of destroyed shapes waiting for reuse, table capacities (grown on demand)
	std::vector<uint32_t> mesh_refs, hull_refs, free_mesh_ids, free_hull_ids;
	std::vector<std::pair<uint32_t, uint32_t>> free_vert_ranges, free_tri_ranges, free_node_ranges;      // (offset, length)
	size_t cap_mesh_table = 0, cap_hull_table = 0;
	std::vector<uint32_t> free_triples;                   // first slot of freed (mesh body + 2 alias) slot triples
	// convex hull shapes: host copies of the device table (mass properties, radii) -- hull 0 is the +-1 cube template
	std::vector<sgd_hull> hulls; sgd_hull* d_hulls = nullptr;
	// wheeled vehicles: device records (AoS) + host mirror of what the ABI needs without a read-back
	sgd_vehicle* d_vehicles = nullptr; sgp_vehicle_input* d_veh_inputs = nullptr; uint32_t cap_vehicles = 0, n_vehicles = 0;
	float4* d_veh_rows = nullptr; float4* d_veh_head = nullptr;      // the step's rows in the solver's lane-major layout (DV::veh_rows)
	bool fuse_vehicle_solve = true;                                  // SGP_VEHICLE_FUSED=0: the vehicles' rows in launches of their own
	std::vector<uint8_t> veh_alive; std::vector<uint32_t> veh_body; std::vector<sgp_vehicle_input> veh_inputs; bool veh_inputs_dirty = false;
	// events collected on the host until drained
	std::vector<sgp_body_event> ev_act, ev_deact, ev_water;
	std::vector<sgp_contact_event> ev_added, ev_pers;
	// last step
	sgp_step_stats stats;
	uint32_t last_pairs = 0, last_manifolds = 0, n_con = 0;
	uint32_t plan_rounds = 12;                               // launch plan for the next step (from the last step's counters)
	uint32_t plan_round_n[32] = { 0, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u };      // uncoloured manifolds at the start of each round of the last step (no history yet: eight wide rounds)
	uint32_t plan_colour_count[SGP_MAX_COLOURS] = { 0 };
	uint32_t table_alloc = 0, ht_alloc = 0;
	// profiling
	bool profiling = false;
	std::vector<ProfEvent> prof;
	std::vector<hipEvent_t> event_pool; size_t event_next = 0;
	hipEvent_t stage_ev[SGP_NUM_STAGES + 1];
	bool stage_ev_ok = false;
};

template <typename T> static int dev_alloc(sgp_world* w, T*& p, size_t n)
{
	void* q = nullptr;
	const size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
	hipError_t e = hipMalloc(&q, bytes);
	if (e != hipSuccess) return fail(SGP_ERR_HIP, "hipMalloc", e);
	e = hipMemsetAsync(q, 0, bytes, w->stream);
	if (e != hipSuccess) return fail(SGP_ERR_HIP, "hipMemsetAsync", e);
	w->allocs.push_back(q);
	w->device_bytes += bytes;
	p = (T*)q;
	return SGP_OK;
}
#define DEV_ALLOC(ptr, n) do { int r_ = dev_alloc(w, ptr, n); if (r_ != SGP_OK) return r_; } while (0)

static int ensure_stage(sgp_world* w, size_t bytes)
{
	if (bytes > w->stage_dev_bytes) {
		if (w->stage_dev) { hipStreamSynchronize(w->stream); hipFree(w->stage_dev); w->device_bytes -= w->stage_dev_bytes; }
		size_t nb = std::max<size_t>(bytes, 1 << 16); nb = nb + nb / 2;
		HIP_TRY(hipMalloc(&w->stage_dev, nb));
		w->stage_dev_bytes = nb; w->device_bytes += nb;
	}
	if (bytes > w->stage_host_bytes) {
		if (w->stage_host) { hipStreamSynchronize(w->stream); hipHostFree(w->stage_host); }
		size_t nb = std::max<size_t>(bytes, 1 << 16); nb = nb + nb / 2;
		HIP_TRY(hipHostMalloc(&w->stage_host, nb, hipHostMallocDefault));
		w->stage_host_bytes = nb;
	}
	return SGP_OK;
}

static uint32_t next_pow2(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

// ---------------------------------------------------------------------------------------------------------------
// defaults (Jolt v5.3.0 PhysicsSettings; Substrata never overrides them)

SGP_API void sgp_default_settings(sgp_settings* s)
{
	s->num_velocity_steps = 10;
	s->num_position_steps = 2;
	s->baumgarte = 0.2f;
	s->penetration_slop = 0.02f;
	s->speculative_contact_distance = 0.02f;
	s->min_velocity_for_restitution = 1.0f;
	s->max_penetration_distance = 0.2f;
	s->time_before_sleep = 0.5f;
	s->point_velocity_sleep_threshold = 0.03f;
	s->contact_point_preserve_lambda_max_dist_sq = 0.01f * 0.01f;
	s->max_linear_velocity = 500.0f;
	s->max_angular_velocity = 0.25f * 3.14159265358979323846f * 60.0f;
	s->allow_sleeping = 1;
	s->warm_start = 1;
	s->use_body_pair_contact_cache = 1;
	s->body_pair_cache_max_delta_position_sq = 0.001f * 0.001f;
	s->body_pair_cache_cos_max_delta_rotation_div2 = 0.99984769515639123915701155881391f;
}

SGP_API void sgp_default_world_desc(sgp_world_desc* d)
{
	memset(d, 0, sizeof(*d));
	d->max_bodies = 65536;                                       // PhysicsWorld.cpp:492
	d->gravity[0] = 0.0f; d->gravity[1] = 0.0f; d->gravity[2] = -9.81f;   // :520
	d->large_body_radius = 4.0f;
	sgp_default_settings(&d->settings);
}

SGP_API void sgp_default_body_desc(sgp_body_desc* d)
{
	memset(d, 0, sizeof(*d));
	d->rot[3] = 1.0f;
	d->shape_type = SGP_SHAPE_BOX;
	d->shape[0] = d->shape[1] = d->shape[2] = 0.5f;               // unit cube, PhysicsWorld.cpp:1249
	d->motion_type = SGP_MOTION_STATIC;                            // PhysicsObject.cpp:28
	d->layer = SGP_LAYER_NON_MOVING;
	d->mass = 100.0f; d->friction = 0.5f; d->restitution = 0.3f;   // PhysicsObject.cpp:36-38
	d->gravity_factor = 1.0f;
	d->linear_damping = 0.05f; d->angular_damping = 0.05f;
	d->allow_sleeping = 1;
}

SGP_API int sgp_abi_version(void) { return SGP_ABI_VERSION; }
SGP_API const char* sgp_last_error(void) { return g_last_error.c_str(); }
SGP_API const char* sgp_kernel_class_name(int k) { return (k >= 0 && k < KC_COUNT) ? k_class_names[k] : nullptr; }
SGP_API int sgp_abi_sizeof(int which)
{
	switch (which) {
	case 0: return (int)sizeof(sgp_settings); case 1: return (int)sizeof(sgp_world_desc); case 2: return (int)sizeof(sgp_body_desc);
	case 3: return (int)sizeof(sgp_body_state); case 4: return (int)sizeof(sgp_body_event); case 5: return (int)sizeof(sgp_contact_event);
	case 6: return (int)sizeof(sgp_ray); case 7: return (int)sizeof(sgp_hit); case 8: return (int)sizeof(sgp_step_stats);
	case 9: return (int)sizeof(sgp_step_profile); case 10: return (int)sizeof(sgp_ghost_record);
	case 11: return (int)sizeof(sgp_vehicle_desc); case 12: return (int)sizeof(sgp_vehicle_input); case 13: return (int)sizeof(sgp_vehicle_state);
	case 14: return (int)sizeof(sgp_hull_info); case 15: return (int)sizeof(sgp_capsule_query); case 16: return (int)sizeof(sgp_query_contact); case 17: return (int)sizeof(sgp_mesh_info);
	default: return -1;
	}
}

// PhysicsWorld::init(), PhysicsWorld.cpp:250-273: once per process.  Returns the number of HIP devices (>= 1) or an error.
SGP_API int sgp_init(void)
{
	int n = 0;
	const hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) { g_device_count = 0; return fail(SGP_ERR_NO_DEVICE, "no HIP device available (there is no CPU fallback)", e); }
	g_device_count = n;
	return n;
}

// ---------------------------------------------------------------------------------------------------------------
// world construction, PhysicsWorld.cpp:462-532

static int alloc_constraints(sgp_world* w, ConstraintArrays& c, uint32_t cap)
{
	DEV_ALLOC(c.ab, cap); DEV_ALLOC(c.n_fric, cap); DEV_ALLOC(c.key, cap); DEV_ALLOC(c.np_col, cap);
	DEV_ALLOC(c.cdp, cap); DEV_ALLOC(c.cdr, cap); DEV_ALLOC(c.cnl, cap);
	for (int k = 0; k < 4; ++k) {
		DEV_ALLOC(c.r1b[k], cap); DEV_ALLOC(c.r2e[k], cap); DEV_ALLOC(c.lam[k], cap); DEV_ALLOC(c.efft[k], cap);
		DEV_ALLOC(c.loc1[k], cap); DEV_ALLOC(c.loc2[k], cap);
	}
	return SGP_OK;
}

SGP_API int sgp_world_destroy(sgp_world* w);
SGP_API int sgp_vehicle_destroy(sgp_world* w, uint32_t id);

SGP_API int sgp_world_create(const sgp_world_desc* desc, sgp_world** out)
{
	if (!desc || !out || desc->max_bodies == 0) return fail(SGP_ERR_INVALID, "sgp_world_create: bad arguments");
	if (g_device_count < 0) sgp_init();
	if (g_device_count <= 0) return fail(SGP_ERR_NO_DEVICE, "sgp_world_create: no HIP device available (there is no CPU fallback)");
	if (desc->device < 0 || desc->device >= g_device_count) return fail(SGP_ERR_INVALID, "sgp_world_create: bad device ordinal");
	sgp_world* w = new sgp_world();
	w->desc = *desc;
	w->device = desc->device;
	if (w->desc.large_body_radius <= 0.0f) w->desc.large_body_radius = 4.0f;
	if (w->desc.max_body_pairs == 0) w->desc.max_body_pairs = 16u * desc->max_bodies + 1024u;
	if (w->desc.max_manifolds == 0) w->desc.max_manifolds = 8u * desc->max_bodies + 1024u;
	memset(&w->dv, 0, sizeof(w->dv));
	memset(&w->stats, 0, sizeof(w->stats));
	hipError_t e = hipSetDevice(w->device);
	if (e != hipSuccess) { delete w; return fail(SGP_ERR_HIP, "hipSetDevice", e); }
	e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
	if (e != hipSuccess) { delete w; return fail(SGP_ERR_HIP, "hipStreamCreate", e); }
	e = hipStreamCreateWithFlags(&w->capture_stream, hipStreamNonBlocking);
	if (e != hipSuccess) { delete w; return fail(SGP_ERR_HIP, "hipStreamCreate", e); }
	*out = w;   // so that DEV_ALLOC failures can be cleaned by the caller via destroy
	DV& d = w->dv;
	const uint32_t N = desc->max_bodies, P = w->desc.max_body_pairs, M = w->desc.max_manifolds;
	d.cap_bodies = N; d.cap_pairs = P; d.cap_manifolds = M;
	DEV_ALLOC(d.pose, 2 * (size_t)N); DEV_ALLOC(d.vel, 2 * (size_t)N); DEV_ALLOC(d.prop, 2 * (size_t)N); DEV_ALLOC(d.dyn, N);
	DEV_ALLOC(d.force, N); DEV_ALLOC(d.torque, N);
	DEV_ALLOC(d.flags, N); DEV_ALLOC(d.aabb_min, N); DEV_ALLOC(d.aabb_max, N);
	for (int k = 0; k < 3; ++k) DEV_ALLOC(d.sleep_s[k], N);
	DEV_ALLOC(d.sleep_timer, N); DEV_ALLOC(d.submerged, N); DEV_ALLOC(d.userdata, N);
	DEV_ALLOC(d.colour_mask, N); DEV_ALLOC(d.body_con, (size_t)N * SGP_MAX_COLOURS); DEV_ALLOC(d.claim[0], N); DEV_ALLOC(d.claim[1], N);
	DEV_ALLOC(d.veh_claim, N); DEV_ALLOC(d.veh_epoch, 1);
	DEV_ALLOC(d.island, N); DEV_ALLOC(d.island_awake, N); DEV_ALLOC(d.awake_mark, N); DEV_ALLOC(d.export_counts, N / 256 + 2);
	// cells of the broad-phase grid: room for 16 per body slot (clearing and scanning follow the cells a step's grid really has, not this capacity).  The grid covers the bounds of all small bodies with cells of R_max + margin and coarsens them
	// (x 1.5) until it fits this table: a pile that has spread out (config 2 after its tower fell: 60 x 60 x 10 m of 1 m cells) then lands in
	// cells with several bodies each and k_bp_pairs scans hundreds of candidates per body (0.23 ms for 10k boxes with 2 cells per slot, 0.02 ms with 8 or more).
	// the cell arrays: a slot of 64 cells per OCCUPIED tile of the paged grid (sgp_kernels.h), and there are never more occupied tiles than bodies
	d.table_size = std::max(4096u, next_pow2(64u * N));
	DEV_ALLOC(d.cell_hash, N);
	DEV_ALLOC(d.cell_count, d.table_size + 4); DEV_ALLOC(d.cell_start, d.table_size + 4); DEV_ALLOC(d.cell_fill, d.table_size + 4);
	// the page table over the tiles of the bounding box: 8M tiles = 512M cells (a 4 km x 4 km x 70 m world at 1.5 m cells; 32 MB) before the cells have to grow
	// (scaled with the world's capacity, advisor r03: a world of a few hundred bodies pays 256 KB - 1 MB for it, not 32; where a small world is spread
	// so wide that its tiles do not fit, its cells grow -- with few bodies to a cell either way)
	{
		uint32_t dflt = 1u << 16; while (dflt < (1u << 23) && (uint64_t)dflt < 1024ull * (uint64_t)N) dflt <<= 1;
		const char* e = getenv("SGP_GRID_TILE_TABLE"); d.tile_table_size = e && atoi(e) > 0 ? (uint32_t)atoi(e) : dflt;
	}
	DEV_ALLOC(d.tile_slot, d.tile_table_size); DEV_ALLOC(d.tile_of_slot, d.table_size / 64u + 4u);
	if (hipMemsetAsync(d.tile_slot, 0xFF, sizeof(uint32_t) * (size_t)d.tile_table_size, w->stream) != hipSuccess) return fail(SGP_ERR_HIP, "tile table init");
	DEV_ALLOC(w->d_large, N); w->cap_large = N; d.large_ids = w->d_large;
	DEV_ALLOC(w->d_lgrid, 1); DEV_ALLOC(w->d_lg_start, SGP_LG_MAX_CELLS + 1); w->cap_lg_items = 4096; DEV_ALLOC(w->d_lg_items, w->cap_lg_items);
	HIP_TRY(hipMemset(w->d_lgrid, 0, sizeof(LargeGrid)));
	d.lgrid = w->d_lgrid; d.lg_start = w->d_lg_start; d.lg_items = w->d_lg_items;
	{ void* q = nullptr; w->cap_mesh_table = 256; HIP_TRY(hipMalloc(&q, sizeof(MeshHeader) * w->cap_mesh_table)); HIP_TRY(hipMemsetAsync(q, 0, sizeof(MeshHeader) * w->cap_mesh_table, w->stream)); w->d_meshes = (MeshHeader*)q; w->device_bytes += sizeof(MeshHeader) * w->cap_mesh_table; }
	d.meshes = w->d_meshes; d.n_meshes = 1; w->meshes.push_back(MeshHeader{}); w->mesh_refs.push_back(0);
	d.cap_mesh_pairs = P / 4 + 1024; DEV_ALLOC(d.mesh_pairs, 4 * (size_t)d.cap_mesh_pairs); DEV_ALLOC(d.mesh_big, d.cap_mesh_pairs);      // (four lists: one per shape of the other body)
	{ void* q = nullptr; w->cap_hull_table = 64; HIP_TRY(hipMalloc(&q, sizeof(sgd_hull) * w->cap_hull_table)); HIP_TRY(hipMemsetAsync(q, 0, sizeof(sgd_hull) * w->cap_hull_table, w->stream)); w->d_hulls = (sgd_hull*)q; w->device_bytes += sizeof(sgd_hull) * w->cap_hull_table; }
	d.hulls = w->d_hulls;
	d.cap_hull_pairs = P / 4 + 1024; DEV_ALLOC(d.hull_pairs, d.cap_hull_pairs);
	{ void* hw = nullptr; const size_t bytes = (size_t)d.cap_hull_pairs * 64; HIP_TRY(hipMalloc(&hw, bytes)); w->allocs.push_back(hw); w->device_bytes += bytes; d.hull_work = (HullWork*)hw; }      // sizeof(HullWork) = 56
	{
		sgd_hull cube; sgd_hull_cube_template(&cube);
		w->hulls.push_back(cube); w->hull_refs.push_back(1);      // (the cube template is never destroyed)
		HIP_TRY(hipMemcpyAsync(&w->d_hulls[0], &w->hulls[0], sizeof(sgd_hull), hipMemcpyHostToDevice, w->stream));
		d.n_hulls = 1;
	}
	DEV_ALLOC(d.sorted_min, N); DEV_ALLOC(d.sorted_max, N); DEV_ALLOC(d.grid, 1); DEV_ALLOC(d.grid_cells_used, 1);
	DEV_ALLOC(d.bounds_acc, 8); { const int init[8] = { 0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000, 0, 0 }; if (hipMemcpyAsync(d.bounds_acc, init, sizeof(init), hipMemcpyHostToDevice, w->stream) != hipSuccess || hipStreamSynchronize(w->stream) != hipSuccess) return fail(SGP_ERR_HIP, "bounds_acc init"); } DEV_ALLOC(d.scan_block_sums, (d.table_size + 1) / 1024 + 2);
	DEV_ALLOC(d.pairs, P);
	d.cap_wake_pairs = P / 4 + 1024; DEV_ALLOC(d.wake_pairs, d.cap_wake_pairs);
	DEV_ALLOC(d.sleep_label, N); DEV_ALLOC(d.label_wake, N);
	HIP_TRY(hipMemsetAsync(d.label_wake, 0, sizeof(uint32_t) * N, w->stream));
	DEV_ALLOC(d.man_ab, M); DEV_ALLOC(d.man_n, M); DEV_ALLOC(d.man_colour, M); DEV_ALLOC(d.man_prio, M); DEV_ALLOC(d.man_prev, M); DEV_ALLOC(d.man_slot, M);
	DEV_ALLOC(d.hc_root, N); DEV_ALLOC(d.hc_count, N); DEV_ALLOC(d.hc_base, N); DEV_ALLOC(d.hc_rank, M);
	// tile solver: one workgroup per compute unit must be resident, so the tile grid follows the device (256 CUs: 16 x 16)
	{
		int dev = 0, cus = 0;
		if (hipGetDevice(&dev) == hipSuccess) hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
		d.ts_nt = 0; d.ts_gx = d.ts_gy = 0;
#ifdef SGP_EXPERIMENTS      // (the resident tile solver is an experiment: csrc/experiments/sgp_tile_solver.inc; a plain build neither allocates for it nor plans it)
		if (cus >= 256) { d.ts_nt = 256; d.ts_gx = 16; d.ts_gy = 16; } else if (cus >= 128) { d.ts_nt = 128; d.ts_gx = 16; d.ts_gy = 8; } else if (cus >= 64) { d.ts_nt = 64; d.ts_gx = 8; d.ts_gy = 8; }
#endif
		if (d.ts_nt) {
			const size_t E = (size_t)SGP_MAX_COLOURS * d.ts_nt;
			DEV_ALLOC(d.body_tile, N); DEV_ALLOC(d.body_tiles4, N); DEV_ALLOC(d.man_tile, M);
			DEV_ALLOC(d.ts_count, E + 1); DEV_ALLOC(d.ts_start, E + 1); DEV_ALLOC(d.ts_fill, E + 1);
			DEV_ALLOC(d.ts_adj, (size_t)d.ts_nt * 8); DEV_ALLOC(d.ts_wait, d.ts_nt); DEV_ALLOC(d.ts_epoch, (size_t)d.ts_nt * 32); DEV_ALLOC(d.ts_flags, 4);
			DEV_ALLOC(d.ts_at, 2 * (size_t)M); DEV_ALLOC(d.ts_side, 2 * (size_t)M);
		}
	}
	d.cap_hc_list = 2u * M + 4096u; DEV_ALLOC(d.hc_list, d.cap_hc_list); DEV_ALLOC(d.hc_entry, d.cap_hc_list); DEV_ALLOC(d.hc_big_list, 1024);
	DEV_ALLOC(d.ulist[0], M); DEV_ALLOC(d.ulist[1], M);
	for (int k = 0; k < 4; ++k) { DEV_ALLOC(d.man_p1[k], M); DEV_ALLOC(d.man_p2[k], M); }
	DEV_ALLOC(d.rows, (size_t)48 * M);
	{ int r = alloc_constraints(w, d.ca[0], M); if (r != SGP_OK) return r; }
	{ int r = alloc_constraints(w, d.ca[1], M); if (r != SGP_OK) return r; }
	w->ht_alloc = next_pow2(2u * M);
	DEV_ALLOC(d.ht_keys, w->ht_alloc); DEV_ALLOC(d.ht_vals, w->ht_alloc);
	HIP_TRY(hipMemsetAsync(d.ht_keys, 0xFF, sizeof(uint64_t) * w->ht_alloc, w->stream));      // empty contact cache
	d.ht_size = w->ht_alloc;
	DEV_ALLOC(d.ht_cur, 1);
	{ static const uint32_t first = 1024u; HIP_TRY(hipMemcpyAsync(d.ht_cur, &first, sizeof(first), hipMemcpyHostToDevice, w->stream)); }      // (an empty table: any size will do)
	DEV_ALLOC(d.cstarts, SGP_MAX_COLOURS + 2);
	DEV_ALLOC(d.ctr, 1); DEV_ALLOC(d.evc, 1);
	DEV_ALLOC(d.ev_activated, N); DEV_ALLOC(d.ev_deactivated, N); DEV_ALLOC(d.ev_water, N);
	HIP_TRY(hipHostMalloc((void**)&w->h_ctr, sizeof(StepCounters), hipHostMallocMapped));
	HIP_TRY(hipHostGetDevicePointer((void**)&w->h_ctr_dev, w->h_ctr, 0));
	HIP_TRY(hipHostMalloc((void**)&w->h_evc, sizeof(EventCounters), hipHostMallocMapped));
	HIP_TRY(hipHostGetDevicePointer((void**)&w->h_evc_dev, w->h_evc, 0));
	HIP_TRY(hipHostMalloc((void**)&w->h_sp, sizeof(StepParams), hipHostMallocDefault));
	memset(w->h_sp, 0, sizeof(StepParams));
	DEV_ALLOC(w->d_sp, 1);
	d.sp = w->d_sp;
	// (the device's copy holds the parity of the LAST step -- k_step_begin flips it -- i.e. the opposite of h_sp's, which is the next step's)
	{ StepParams init = *w->h_sp; init.parity = w->h_sp->parity ^ 1u; HIP_TRY(hipMemcpyAsync(w->d_sp, &init, sizeof(init), hipMemcpyHostToDevice, w->stream)); HIP_TRY(hipStreamSynchronize(w->stream)); }
	{ const char* e = getenv("SGP_NO_GRAPH"); if (e && e[0] == '1') w->use_graphs = false; }
	{ const char* e = getenv("SGP_NO_SMALL_WORLD"); if (e && e[0] == '1') w->use_small_world = false; }
	{ const char* e = getenv("SGP_NO_WAKE_ROUND"); if (e && e[0] == '1') w->use_wake_round = false; }      // (measurements only: the CPU statement has its own switch)
	{ const char* e = getenv("SGP_DEBUG_FLAGS"); w->dv.dbg_flags = e ? (uint32_t)atoi(e) : 0u; }
#ifdef SGP_EXPERIMENTS
	{ const char* e = getenv("SGP_TILE_SOLVER"); if (e) w->use_tile_solver = atoi(e); }
#endif
	{ const char* e = getenv("SGP_VEHICLE_FUSED"); if (e) w->fuse_vehicle_solve = atoi(e) != 0; }
	{ const char* e = getenv("SGP_COMPACT_ROWS_MIN"); if (e && atoll(e) >= 0) w->compact_rows_min = (uint32_t)atoll(e); }
	{ const char* e = getenv("SGP_ROWS_MODE"); if (e && (atoi(e) == 1 || atoi(e) == 2)) w->rows_mode_large = (uint32_t)atoi(e); }
	{ const char* e = getenv("SGP_TS_MIN_CONSTRAINTS"); if (e && atoi(e) >= 0) w->ts_min_constraints = (uint32_t)atoi(e); }
	{ const char* e = getenv("SGP_HC_BUDGET"); if (e) { const int v = atoi(e); if (v <= 0) w->use_components = false; else w->hc_budget = (uint32_t)v; } }
	{ int dev = 0, cus = 0; if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) w->n_cus = (uint32_t)cus; }
	{ const char* e = getenv("SGP_HC_MIN_COLOURS"); if (e && atoi(e) >= 0) w->hc_min_colours = (uint32_t)atoi(e); }
	{ const char* e = getenv("SGP_TAIL_THRESHOLD"); if (e && atoi(e) > 0) w->tail_threshold = (uint32_t)atoi(e); }
	d.st = desc->settings;
	d.gx = desc->gravity[0]; d.gy = desc->gravity[1]; d.gz = desc->gravity[2];
	w->hb.resize(N);
	HIP_TRY(hipStreamSynchronize(w->stream));
	return SGP_OK;
}

SGP_API int sgp_world_destroy(sgp_world* w)
{
	if (!w) return fail(SGP_ERR_INVALID, "sgp_world_destroy: NULL");
	hipSetDevice(w->device);
	if (w->stream) hipStreamSynchronize(w->stream);
	for (void* p : w->allocs) hipFree(p);
	if (w->stage_dev) hipFree(w->stage_dev);
	if (w->d_meshes) hipFree(w->d_meshes);
	if (w->d_hulls) hipFree(w->d_hulls);
	if (w->d_mesh_verts) hipFree(w->d_mesh_verts);
	if (w->d_mesh_tris) hipFree(w->d_mesh_tris);
	if (w->d_mesh_tri_mat) hipFree(w->d_mesh_tri_mat);
	if (w->d_mesh_nodes) hipFree(w->d_mesh_nodes);
	if (w->d_vehicles) hipFree(w->d_vehicles);
	if (w->d_veh_inputs) hipFree(w->d_veh_inputs);
	hipFree(w->d_veh_rows); hipFree(w->d_veh_head);
	if (w->stage_host) hipHostFree(w->stage_host);
	if (w->view_host) hipHostFree(w->view_host);
	if (w->h_ctr) hipHostFree(w->h_ctr);
	if (w->h_evc) hipHostFree(w->h_evc);
	if (w->h_sp) hipHostFree(w->h_sp);
	for (auto& kv : w->graphs) hipGraphExecDestroy(kv.second);
	for (hipEvent_t ev : w->event_pool) hipEventDestroy(ev);
	if (w->stage_ev_ok) for (int i = 0; i <= SGP_NUM_STAGES; ++i) hipEventDestroy(w->stage_ev[i]);
	if (w->capture_stream) hipStreamDestroy(w->capture_stream);
	if (w->stream) hipStreamDestroy(w->stream);
	delete w;
	return SGP_OK;
}

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
static inline bool finite3(const float* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }
static inline bool finite4(const float* v) { return finite3(v) && std::isfinite(v[3]); }
// The reference only asserts finite inputs in debug builds (PhysicsWorld.cpp:548-556,625,710); a NaN that gets into one body spreads through
// every contact it touches, so the setters refuse it outright.
#define REQUIRE_FINITE(cond, what) do { if (!(cond)) return fail(SGP_ERR_INVALID, what ": non-finite argument"); } while (0)
static inline bool live(const sgp_world* w, uint32_t id) { return w && id < w->high && (w->hb[id].flags & BF_ALIVE); }

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

static int add_one(sgp_world* w, const sgp_body_desc* d, uint32_t* id_out, bool ghost)
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

static BodyCmd blank_cmd(uint32_t id, uint32_t ops) { BodyCmd c; memset(&c, 0, sizeof(c)); c.id = id; c.ops = ops; return c; }
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
// compound ids reported by queries and events: a child's slot -> the compound's id (+ the child index)
static inline uint32_t compound_id_of(const sgp_world* w, uint32_t id, uint32_t* sub_out)
{
	const HostBody& b = w->hb[id];
	if (b.comp_root == SGP_INVALID_ID) { if (sub_out) *sub_out = 0; return id; }
	if (sub_out) *sub_out = b.comp_child;
	return b.comp_root;
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
SGP_API int sgp_physics_update_encode(uint64_t uid, const sgp_body_state* st, double client_time, uint8_t out[SGP_PHYSICS_UPDATE_BYTES])
{
	if (!st || !out) return fail(SGP_ERR_INVALID, "sgp_physics_update_encode: NULL");
	uint8_t* p = out;
	memcpy(p, &uid, 8); p += 8;
	for (int i = 0; i < 3; ++i) { const double v = (double)st->pos[i]; memcpy(p, &v, 8); p += 8; }   // Vec3d world_ob->pos
	memcpy(p, st->rot, 16); p += 16;
	memcpy(p, st->lin_vel, 12); p += 12;
	memcpy(p, st->ang_vel, 12); p += 12;
	memcpy(p, &client_time, 8);
	return SGP_OK;
}
SGP_API int sgp_physics_update_decode(const uint8_t in[SGP_PHYSICS_UPDATE_BYTES], uint64_t* uid_out, sgp_pose_vel* rec, double* client_time_out)
{
	if (!in || !rec) return fail(SGP_ERR_INVALID, "sgp_physics_update_decode: NULL");
	const uint8_t* p = in;
	if (uid_out) memcpy(uid_out, p, 8);
	p += 8;
	for (int i = 0; i < 3; ++i) { double v; memcpy(&v, p, 8); p += 8; rec->pos[i] = (float)v; }
	memcpy(rec->rot, p, 16); p += 16;
	memcpy(rec->lin_vel, p, 12); p += 12;
	memcpy(rec->ang_vel, p, 12); p += 12;
	if (client_time_out) memcpy(client_time_out, p, 8);
	for (int i = 0; i < 3; ++i) if (!std::isfinite(rec->pos[i]) || !std::isfinite(rec->lin_vel[i]) || !std::isfinite(rec->ang_vel[i])) return fail(SGP_ERR_REJECTED, "sgp_physics_update_decode: non-finite field");
	return SGP_OK;
}

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
static int upload_sp(sgp_world* w)
{
	StepParams& sp = *w->h_sp;
	sp.n_slots = w->high;
	sp.n_large = (uint32_t)w->large_linear.size();
	sp.bp_rmax = std::max(0.25f, w->max_small_radius);
	sp.cell_size = sp.bp_rmax + w->dv.st.speculative_contact_distance;
	if (w->sp_uploaded_valid) {
		// nothing changed since the device copy was written (several drains / reads in a row; the first call after a step: k_step_begin wrote it)?
		// dt and the buffer parity are per-step values that only the kernels of a step read, and k_step_begin brings them along by value.
		StepParams cmp = sp; cmp.dt = w->sp_uploaded.dt; cmp.parity = w->sp_uploaded.parity;
		if (memcmp(&w->sp_uploaded, &cmp, sizeof(cmp)) == 0) return SGP_OK;
	}
	{ StepParams up = sp; up.parity = sp.parity ^ 1u; launch_set_params(w->dv, up, w->stream); }      // (device convention: the parity of the last step)
	w->sp_uploaded = sp; w->sp_uploaded_valid = true;
	return SGP_OK;
}

static void invalidate_graphs(sgp_world* w);
// The device's view of the large bodies after the pending edits have been applied (their bounds are read back from the device, which computed
// them): static ones into the grid, the rest on the linear list.  Runs only when the set changed or a static large body moved.
static int rebuild_large_grid(sgp_world* w)
{
	if (!w->large_dirty) {
		if (!w->large_list_dirty) return SGP_OK;
		// only the linear list changed (a static large body waits on it for the next rebuild, or one that waited has gone): the list and its length, no read-back
		w->large_list_dirty = false;
		if (w->large_linear.size() > w->cap_large) return fail(SGP_ERR_CAPACITY, "large-body list full");
		if (!w->large_linear.empty()) HIP_TRY(hipMemcpyAsync(w->d_large, w->large_linear.data(), sizeof(uint32_t) * w->large_linear.size(), hipMemcpyHostToDevice, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));      // (a pageable host vector)
		w->grid_valid = false;
		return upload_sp(w);
	}
	w->large_dirty = false; w->large_list_dirty = false;
	DV& d = w->dv;
	std::vector<uint32_t> stat;
	w->large_linear.clear();
	// (large_ids may list ids that have gone, or one twice after a removal and re-use: keep the live large ones, once)
	{
		std::vector<uint32_t> keep; keep.reserve(w->large_ids.size());
		for (uint32_t id : w->large_ids) { HostBody& b = w->hb[id]; if (b.in_large_ids == 1 && (b.flags & BF_ALIVE) && (b.flags & BF_LARGE)) { keep.push_back(id); b.in_large_ids = 2; } }
		for (uint32_t id : keep) w->hb[id].in_large_ids = 1;
		w->large_ids.swap(keep);
	}
	if (w->lg_tombs) for (HostBody& b : w->hb) b.lg_tomb = 0;
	w->lg_tombs = 0; w->lg_pending = 0;
	for (uint32_t id : w->large_ids) { w->hb[id].lg_state = 0; if ((w->hb[id].flags & BF_MOTION_MASK) == SGP_MOTION_STATIC) stat.push_back(id); else w->large_linear.push_back(id); }
	LargeGrid g; memset(&g, 0, sizeof(g)); g.cell = 1.0f; g.inv_cell = 1.0f; g.nx = g.ny = g.nz = 1;
	std::vector<uint32_t> start, items;
	if (stat.size() < 32) { w->large_linear.insert(w->large_linear.end(), stat.begin(), stat.end()); stat.clear(); }      // (a handful: the list is as good)
	if (!stat.empty()) {
		const uint32_t n = (uint32_t)stat.size();
		const size_t id_bytes = (sizeof(uint32_t) * n + 15) & ~size_t(15);
		{ int r = ensure_stage(w, id_bytes + sizeof(float4) * 2 * (size_t)n); if (r != SGP_OK) return r; }
		memcpy(w->stage_host, stat.data(), sizeof(uint32_t) * n);
		HIP_TRY(hipMemcpyAsync(w->stage_dev, w->stage_host, sizeof(uint32_t) * n, hipMemcpyHostToDevice, w->stream));
		launch_gather_aabbs(d, (const uint32_t*)w->stage_dev, n, (float4*)((char*)w->stage_dev + id_bytes), w->stream);
		HIP_TRY(hipMemcpyAsync((char*)w->stage_host + id_bytes, (char*)w->stage_dev + id_bytes, sizeof(float4) * 2 * (size_t)n, hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		const float4* bb = (const float4*)((char*)w->stage_host + id_bytes);
		// cell edge: twice the median extent, grown until the grid fits its table and an average body fills few cells
		std::vector<float> ext(n);
		for (uint32_t k = 0; k < n; ++k) ext[k] = std::max(std::max(bb[2 * k + 1].x - bb[2 * k].x, bb[2 * k + 1].y - bb[2 * k].y), bb[2 * k + 1].z - bb[2 * k].z);
		std::vector<float> sorted_ext(ext);
		std::nth_element(sorted_ext.begin(), sorted_ext.begin() + n / 2, sorted_ext.end());
		float cell = std::max(2.0f * sorted_ext[n / 2], 0.5f);
		if (!(cell < 1.0e30f)) cell = 1.0e30f;
		auto cell_of = [](float x, float o, float inv, int nn) { return std::min(std::max((int)floorf((x - o) * inv), 0), nn - 1); };      // = lg_cell on the device
		bool placed = false;
		for (int attempt = 0; attempt < 64 && !placed; ++attempt) {
			const float inv = 1.0f / cell;
			// bodies that would fill too many cells stay on the list (the ground, a terrain); the box of the others is the grid
			float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
			std::vector<uint8_t> huge(n, 0);
			for (uint32_t k = 0; k < n; ++k) {
				const float4 mn = bb[2 * k], mx = bb[2 * k + 1];
				const double span = (double)(floorf((mx.x - mn.x) * inv) + 2.0f) * (double)(floorf((mx.y - mn.y) * inv) + 2.0f) * (double)(floorf((mx.z - mn.z) * inv) + 2.0f);
				if (!(span <= (double)SGP_LG_MAX_SPAN) || !std::isfinite(mn.x + mn.y + mn.z + mx.x + mx.y + mx.z)) { huge[k] = 1; continue; }
				lo[0] = std::min(lo[0], mn.x); lo[1] = std::min(lo[1], mn.y); lo[2] = std::min(lo[2], mn.z);
				hi[0] = std::max(hi[0], mx.x); hi[1] = std::max(hi[1], mx.y); hi[2] = std::max(hi[2], mx.z);
			}
			if (!(lo[0] <= hi[0])) break;      // all of them huge
			const double dx = floor((double)(hi[0] - lo[0]) * inv) + 1.0, dy = floor((double)(hi[1] - lo[1]) * inv) + 1.0, dz = floor((double)(hi[2] - lo[2]) * inv) + 1.0;
			if (dx * dy * dz > (double)SGP_LG_MAX_CELLS) { cell *= 1.5f; continue; }
			g.ox = lo[0]; g.oy = lo[1]; g.oz = lo[2]; g.cell = cell; g.inv_cell = inv; g.nx = (int)dx; g.ny = (int)dy; g.nz = (int)dz;
			const size_t ncell = (size_t)g.nx * g.ny * g.nz;
			start.assign(ncell + 1, 0u);
			size_t total = 0;
			auto for_cells = [&](uint32_t k, auto&& fn) {
				const float4 mn = bb[2 * k], mx = bb[2 * k + 1];
				const int x0 = cell_of(mn.x, g.ox, inv, g.nx), x1 = cell_of(mx.x, g.ox, inv, g.nx), y0 = cell_of(mn.y, g.oy, inv, g.ny), y1 = cell_of(mx.y, g.oy, inv, g.ny), z0 = cell_of(mn.z, g.oz, inv, g.nz), z1 = cell_of(mx.z, g.oz, inv, g.nz);
				for (int z = z0; z <= z1; ++z) for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) fn(((size_t)z * g.ny + y) * g.nx + x);
			};
			for (uint32_t k = 0; k < n; ++k) if (!huge[k]) for_cells(k, [&](size_t c) { start[c + 1]++; ++total; });
			for (size_t c = 0; c < ncell; ++c) start[c + 1] += start[c];
			items.resize(total);
			std::vector<uint32_t> fill(start.begin(), start.end() - 1);
			for (uint32_t k = 0; k < n; ++k) { if (huge[k]) w->large_linear.push_back(stat[k]); else { w->hb[stat[k]].lg_state = 1; for_cells(k, [&](size_t c) { items[fill[c]++] = stat[k]; }); } }
			g.n_items = (uint32_t)total;
			placed = true;
		}
		if (!placed) { w->large_linear.insert(w->large_linear.end(), stat.begin(), stat.end()); g.n_items = 0; for (uint32_t id : stat) w->hb[id].lg_state = 0; }
	}
	w->lg_static = g.n_items ? (uint32_t)stat.size() : 0u;
	if (w->large_linear.size() > w->cap_large) return fail(SGP_ERR_CAPACITY, "large-body list full");
	if (items.size() > w->cap_lg_items) {
		// (the captured graphs carry the old pointer)
		HIP_TRY(hipStreamSynchronize(w->stream));
		w->cap_lg_items = (uint32_t)(items.size() + items.size() / 2);
		uint32_t* ni = nullptr;
		HIP_TRY(hipMalloc((void**)&ni, sizeof(uint32_t) * (size_t)w->cap_lg_items));
		// (the previous buffer is freed here, after the synchronisation above: it used to be parked in `allocs` at every growth; advisor r03)
		if (w->d_lg_items) { auto it = std::find(w->allocs.begin(), w->allocs.end(), (void*)w->d_lg_items); if (it != w->allocs.end()) w->allocs.erase(it); hipFree(w->d_lg_items); }
		w->allocs.push_back(ni); w->device_bytes += sizeof(uint32_t) * (size_t)w->cap_lg_items;
		w->d_lg_items = ni; d.lg_items = ni;
		invalidate_graphs(w);
	}
	if (!w->large_linear.empty()) HIP_TRY(hipMemcpyAsync(w->d_large, w->large_linear.data(), sizeof(uint32_t) * w->large_linear.size(), hipMemcpyHostToDevice, w->stream));
	if (g.n_items) {
		HIP_TRY(hipMemcpyAsync(w->d_lg_start, start.data(), sizeof(uint32_t) * start.size(), hipMemcpyHostToDevice, w->stream));
		HIP_TRY(hipMemcpyAsync(w->d_lg_items, items.data(), sizeof(uint32_t) * items.size(), hipMemcpyHostToDevice, w->stream));
	}
	HIP_TRY(hipMemcpyAsync(w->d_lgrid, &g, sizeof(g), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));      // (pageable host vectors)
	w->grid_valid = false;
	return upload_sp(w);
}

static int flush_cmds(sgp_world* w)
{
	hipSetDevice(w->device);
	DV& d = w->dv;
	{ int r = upload_sp(w); if (r != SGP_OK) return r; }
	if (!w->ghost_refresh.empty()) {
		// ghosts never appear in the command queue while their set is unchanged, so the order against the commands below does not matter
		const size_t bytes = w->ghost_refresh.size() * sizeof(GhostRefresh);
		{ int r = ensure_stage(w, bytes); if (r != SGP_OK) return r; }
		memcpy(w->stage_host, w->ghost_refresh.data(), bytes);
		HIP_TRY(hipMemcpyAsync(w->stage_dev, w->stage_host, bytes, hipMemcpyHostToDevice, w->stream));
		launch_ghost_refresh(d, (const GhostRefresh*)w->stage_dev, (uint32_t)w->ghost_refresh.size(), w->stream);
		HIP_TRY(hipStreamSynchronize(w->stream));   // the staging buffer is reused below / by the next call
		w->ghost_refresh.clear();
		w->grid_valid = false;
		w->dirty_since_step = true;
	}
	if (w->cmds.empty()) return rebuild_large_grid(w);
	w->grid_valid = false;
	w->dirty_since_step = true;
	w->events_on_device = true;               // (k_apply_cmds reports activations)
	const size_t n = w->cmds.size();
	// commands of one body must be adjacent and in call order (k_apply_cmds walks runs): a stable sort by id -- skipped when the queue is
	// already ordered, which is what a per-step refresh of thousands of ghosts or snapshots looks like
	std::vector<uint32_t> order(n);
	for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
	bool ordered = true;
	for (size_t i = 1; i < n && ordered; ++i) ordered = w->cmds[i - 1].id <= w->cmds[i].id;
	if (!ordered) {
		// (body id, position in the queue) as one 64-bit key: a plain sort of the keys IS the stable sort by id, without a comparison that walks
		// through the command records (thousands of ghost creations and removals per exchange of a tile: the sort was a third of the flush)
		std::vector<uint64_t> keys(n);
		for (size_t i = 0; i < n; ++i) keys[i] = ((uint64_t)w->cmds[i].id << 32) | (uint64_t)i;
		std::sort(keys.begin(), keys.end());
		for (size_t i = 0; i < n; ++i) order[i] = (uint32_t)(keys[i] & 0xFFFFFFFFull);
	}
	const size_t cmd_bytes = n * sizeof(BodyCmd);
	const size_t run_off = (cmd_bytes + 15) & ~size_t(15);
	const size_t total = run_off + (n + 1) * sizeof(uint32_t);
	{ int r = ensure_stage(w, total); if (r != SGP_OK) return r; }
	BodyCmd* hc = (BodyCmd*)w->stage_host;
	uint32_t* hr = (uint32_t*)((char*)w->stage_host + run_off);
	uint32_t n_runs = 0;
	for (size_t i = 0; i < n; ++i) {
		hc[i] = w->cmds[order[i]];
		if (i == 0 || hc[i].id != hc[i - 1].id) hr[n_runs++] = (uint32_t)i;
		// a static large body that moves or is rescaled sits in other cells of the large bodies' grid afterwards
		if ((hc[i].ops & (CMD_SET_POS | CMD_SET_ROT | CMD_SET_SHAPE)) && hc[i].id < w->hb.size() && (w->hb[hc[i].id].flags & BF_LARGE) && (w->hb[hc[i].id].flags & BF_MOTION_MASK) == SGP_MOTION_STATIC) w->large_dirty = true;
	}
	hr[n_runs] = (uint32_t)n;
	HIP_TRY(hipMemcpyAsync(w->stage_dev, w->stage_host, total, hipMemcpyHostToDevice, w->stream));
	launch_apply_cmds(d, (const BodyCmd*)w->stage_dev, (const uint32_t*)((char*)w->stage_dev + run_off), n_runs, w->stream);
	HIP_TRY(hipStreamSynchronize(w->stream));   // the staging buffer is reused by the next call
	w->cmds.clear();
	return rebuild_large_grid(w);
}

// Pull the device event lists into the host vectors and reset the device counters.
static int collect_events(sgp_world* w, bool counters_fresh = false)
{
	DV& d = w->dv;
	if (!counters_fresh) {
		if (!w->events_on_device) return SGP_OK;      // nothing ran on the device since the lists were last pulled: no copy, no sync
		HIP_TRY(hipMemcpyAsync(w->h_evc, d.evc, sizeof(EventCounters), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	w->events_on_device = false;
	const EventCounters ec = *w->h_evc;
	if (!(ec.n_activated | ec.n_deactivated | ec.n_water | ec.n_contact_added | ec.n_contact_persisted)) return SGP_OK;
	struct L { uint32_t n; uint32_t* dev; std::vector<sgp_body_event>* out; };
	L lists[3] = { { std::min(ec.n_activated, d.cap_bodies), d.ev_activated, &w->ev_act },
	               { std::min(ec.n_deactivated, d.cap_bodies), d.ev_deactivated, &w->ev_deact },
	               { std::min(ec.n_water, d.cap_bodies), d.ev_water, &w->ev_water } };
	for (L& l : lists) {
		if (!l.n) continue;
		{ int r = ensure_stage(w, sizeof(uint32_t) * l.n); if (r != SGP_OK) return r; }
		HIP_TRY(hipMemcpyAsync(w->stage_host, l.dev, sizeof(uint32_t) * l.n, hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		const uint32_t* ids = (const uint32_t*)w->stage_host;
		for (uint32_t k = 0; k < l.n; ++k) { sgp_body_event e; e.id = ids[k]; e._pad = 0; e.userdata = w->hb[ids[k]].userdata; l.out->push_back(e); }
	}
	struct CL { uint32_t n; sgp_contact_event* dev; std::vector<sgp_contact_event>* out; };
	CL cl[2] = { { std::min(ec.n_contact_added, d.cap_contact_events), d.ev_contacts_added, &w->ev_added },
	             { std::min(ec.n_contact_persisted, d.cap_contact_events), d.ev_contacts_persisted, &w->ev_pers } };
	for (CL& l : cl) {
		if (!l.n) continue;
		{ int r = ensure_stage(w, sizeof(sgp_contact_event) * l.n); if (r != SGP_OK) return r; }
		HIP_TRY(hipMemcpyAsync(w->stage_host, l.dev, sizeof(sgp_contact_event) * l.n, hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		const sgp_contact_event* src = (const sgp_contact_event*)w->stage_host;
		for (uint32_t k = 0; k < l.n; ++k) { sgp_contact_event e = src[k];
			while (e.id1 > 0 && (w->hb[e.id1].flags & BF_ALIAS)) --e.id1;       // a mesh body's alias slots report as the mesh body
			while (e.id2 > 0 && (w->hb[e.id2].flags & BF_ALIAS)) --e.id2;
			e.id1 = compound_id_of(w, e.id1, nullptr); e.id2 = compound_id_of(w, e.id2, nullptr);   // a compound's children report as the compound
			e.userdata1 = w->hb[e.id1].userdata; e.userdata2 = w->hb[e.id2].userdata; l.out->push_back(e); }
	}
	HIP_TRY(hipMemsetAsync(d.evc, 0, sizeof(EventCounters), w->stream));
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// profiling helpers

static hipEvent_t pool_event(sgp_world* w)
{
	if (w->event_next == w->event_pool.size()) { hipEvent_t e; hipEventCreate(&e); w->event_pool.push_back(e); }
	return w->event_pool[w->event_next++];
}
struct KScope {
	sgp_world* w; int kc; hipEvent_t a, b; bool on;
	KScope(sgp_world* w_, int kc_) : w(w_), kc(kc_), on(w_->profiling) { if (on) { a = pool_event(w); b = pool_event(w); hipEventRecord(a, w->stream); } }
	~KScope() { if (on) { hipEventRecord(b, w->stream); w->prof.push_back({ kc, a, b }); } }
};
#define STAGE_MARK(i) do { if (w->profiling) hipEventRecord(w->stage_ev[i], w->stream); } while (0)

static int read_counters(sgp_world* w)
{
	HIP_TRY(hipMemcpyAsync(w->h_ctr, w->dv.ctr, sizeof(StepCounters), hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// think(dt), PhysicsWorld.cpp:1356-1443
//
// A step is (1) a LAUNCH PLAN made on the host from the previous step's counters (how many body slots, colouring rounds
// and per-colour launches to issue, with bucketed sizes), (2) the launch sequence of that plan -- issued eagerly, or
// replayed as a hipGraph once the same plan has come up twice -- and (3) ONE host sync that reads the counters back.
// Nothing in the sequence depends on host knowledge of the CURRENT step: kernels size themselves from device counters,
// and catch-all kernels (k_colour_finish, k_solve_tail) keep the result exact when the plan under-estimates.

static uint32_t bucket_up(uint32_t x)
{
	if (x <= 1024) return 1024;
	uint32_t p = 1; while ((p << 1) <= x) p <<= 1;        // largest power of two <= x
	const uint32_t q = p / 4;
	return ((x + q - 1) / q) * q;                          // steps of 25 %
}

struct StepPlan {
	uint32_t nb;                 // body slots covered by the per-body grids
	uint32_t rounds;             // colouring rounds launched before the catch-all
	uint32_t est_pairs, est_man;
	int      tail_first;         // colours [0, tail_first) get their own launch per pass
	uint32_t colour_est[SGP_MAX_COLOURS];
	int      water, contact_events, warm_start, vel_iters, pos_iters;
	uint32_t n_vehicles;
	int      has_meshes;         // some body may be a static triangle mesh: run the mesh-pair narrow phase
	int      has_hulls;          // some body may be a convex hull: run the hull-pair narrow phase
	int      wake_round;         // in-step activation: pair and collide the bodies this step wakes (k_wake_pairs + a second narrow-phase round)
	int      small_colouring;    // the whole colouring in one single-workgroup launch (k_colour_finish builds its own worklist)
	int      small_world;        // warm start + velocity iterations as ONE single-workgroup launch (k_solve_small)
	int      bp_small;           // k_bp_pairs instance with the small LDS footprint (the previous step met no dense halo)
	int      hc_first;           // colours >= hc_first (<= tail_first) are solved by connected component, one launch per pass; -1: off (tail kernel)
	uint32_t hc_est;             // their constraints (previous step)
	int      hc_probe;           // >= 0: this step also computes the component sizes for hc_probe = hc_first - 1 (not used for solving)
	uint32_t hc_probe_est;
	int      tile_solver;        // all velocity iterations in the one resident launch of the tile solver (k_ts_solve)
	int      small_pairs;        // ... with two lanes per constraint (the previous step had <= 384 constraints), else one thread per constraint
	StepParams sp;               // by-value kernel argument of the first launch: part of the key of a captured graph
};

static void make_plan(const sgp_world* w, StepPlan& p)
{
	memset(&p, 0, sizeof(p));
	p.nb = bucket_up(w->high);
	// colouring: a round gets launches of its own (claim + commit over the whole chip) while the previous step still had more than a
	// workgroup's worth of uncoloured manifolds at its start; the remaining rounds -- typically a few hundred manifolds, then a few dozen --
	// run inside the single-workgroup k_colour_finish, which also catches whatever a short plan leaves over
	{
		uint32_t r = 1;
		while (r < 16u && r < w->plan_rounds && w->plan_round_n[r] > SGP_COLOUR_WIDE_MIN) ++r;
		static const uint32_t steps[] = { 1, 2, 3, 4, 6, 8, 12, 16 };
		p.rounds = 16; for (uint32_t k : steps) if (k >= r) { p.rounds = k; break; }
	}
	p.est_pairs = bucket_up(std::max(w->last_pairs + w->last_pairs / 8, 4u * w->high));
	p.est_man = bucket_up(std::max(w->last_manifolds + w->last_manifolds / 8, 2u * w->high));
	int tf = 0;
	while (tf < SGP_OVERFLOW_COLOUR && w->plan_colour_count[tf] > w->tail_threshold) { p.colour_est[tf] = bucket_up(w->plan_colour_count[tf] + w->plan_colour_count[tf] / 8); ++tf; }
	if (tf == 0 && !w->plan_seen && w->high > SGP_SMALL_WORLD_BODIES) {
		// no histogram yet (first step of a large world): rather a dozen launches that may find their colour empty (3 us each) than
		// every constraint of a freshly loaded scene in the single-workgroup tail (100k bodies: 4 ms per pass)
		for (; tf < 12; ++tf) p.colour_est[tf] = bucket_up(w->high / 2u);
	}
	p.tail_first = tf;
	p.water = w->h_sp->water_enabled; p.contact_events = w->h_sp->contact_events;
	p.warm_start = w->dv.st.warm_start; p.vel_iters = w->dv.st.num_velocity_steps; p.pos_iters = w->dv.st.num_position_steps;
	p.n_vehicles = w->n_vehicles;
	p.has_hulls = w->hulls.size() > 1 ? 1 : 0;
	p.has_meshes = w->meshes.size() > 1 ? 1 : 0;
	p.wake_round = w->use_wake_round ? 1 : 0;
	p.small_colouring = (w->last_manifolds <= SGP_SMALL_COLOURING_MANIFOLDS && w->high <= SGP_SMALL_WORLD_BODIES) ? 1 : 0;
	p.small_world = (tf == 0 && w->high <= SGP_SMALL_WORLD_BODIES && w->n_vehicles == 0 && w->use_small_world) ? 1 : 0;
	p.small_pairs = (w->n_con <= 384u || w->n_con > 512u) ? 1 : 0;
	p.bp_small = w->bp_dense_last ? 0 : 1;
	p.hc_first = -1;
	if (!p.small_world && w->use_components && w->n_con != 0) {      // (no histogram yet: the tail kernel takes whatever the first step brings)
		// the high colours: as many of the last colours as hold at most hc_budget (per mille) of the constraints -- few enough that the
		// sub-graph they form is far below its percolation threshold and falls apart into small components (measured: DESIGN.md section 8)
		uint64_t total = 0, sum = 0;
		for (int c = 0; c < SGP_OVERFLOW_COLOUR; ++c) total += w->plan_colour_count[c];
		for (int c = tf; c < SGP_OVERFLOW_COLOUR; ++c) sum += w->plan_colour_count[c];
		int k = tf;
		// first guess: as many of the last colours as hold at most hc_budget (per mille) of the constraints; probes may take it further (a world of
		// scattered objects is all small components: every colour goes to them)
		const uint64_t budget = w->hc_k < 0 ? w->hc_budget : 1000u;
		// ... and never more than one round of workgroups can hold (a workgroup per compute unit: the launch replaces launches that are bound by
		// latency, not throughput -- at a million bodies a colour of 400k constraints is better off in its own, coalesced launch)
		const uint64_t fits = (uint64_t)w->n_cus * 256u * 18u / 25u;      // 256 lane pairs per workgroup, ~0.72 constraints per list entry
		while (k > 0 && (sum + w->plan_colour_count[k - 1]) * 1000u <= total * budget && sum + w->plan_colour_count[k - 1] <= fits) sum += w->plan_colour_count[--k];
		if (w->hc_k >= 0) k = std::min(tf, std::max(k, w->hc_k));
		sum = 0; for (int c = k; c < SGP_OVERFLOW_COLOUR; ++c) sum += w->plan_colour_count[c];
		p.hc_first = k;
		p.hc_est = bucket_up((uint32_t)sum + (uint32_t)(sum / 8));
		p.hc_probe = -1;
		if (w->hc_k >= 0 && w->hc_probe_in == 0 && k > 0 && (sum + w->plan_colour_count[k - 1]) * 1000u <= total * budget && sum + w->plan_colour_count[k - 1] <= fits) {
			p.hc_probe = k - 1;
			const uint64_t ps = sum + w->plan_colour_count[k - 1];
			p.hc_probe_est = bucket_up((uint32_t)ps + (uint32_t)(ps / 8));
		}
		p.tail_first = k;
		// the component launch has a price of its own (the build, 40 us per step; a launch of ~14 us before its first phase): it pays when it
		// replaces several launches -- a world of scattered objects with two or three colours is better off with those launches and the tail
		int used = 0;
		for (int c = k; c < SGP_OVERFLOW_COLOUR; ++c) used += w->plan_colour_count[c] != 0u;
		if (used < (int)w->hc_min_colours) { p.hc_first = -1; p.hc_probe = -1; p.hc_est = 0; p.hc_probe_est = 0; p.tail_first = tf; }
	}
	// the tile solver: a pile large enough that its colour launches are bound by latency, small enough that a tile's bodies fit its LDS table
	// (the table takes what fits and leaves the rest in global memory, so the bound is about speed, not correctness), no vehicle rows between
	// the passes, and enough body slots for k_ts_label's grid to clear the (colour, tile) histogram
	p.tile_solver = (w->use_tile_solver && w->dv.ts_nt && !p.small_world && w->n_vehicles == 0 && p.vel_iters > 0 && w->n_con >= w->ts_min_constraints &&
	                 w->high >= SGP_MAX_COLOURS * w->dv.ts_nt && w->last_active <= w->dv.ts_nt * 1536u && !w->h_sp->compact_rows) ? w->use_tile_solver : 0;      // (2: debugging aid -- the tile order of the slots, solved by the colour launches)
	p.sp = *w->h_sp;
	p.sp.parity = 0u;        // (not part of a plan: the device flips its own)
}

static int enqueue_step(sgp_world* w, const StepPlan& p)
{
	const DV& d = w->dv;
	hipStream_t s = w->stream;
	const uint32_t nb = p.nb;
	STAGE_MARK(0);
	{ KScope k(w, KC_MISC); launch_step_begin(d, p.sp, nb, true, s); }
	STAGE_MARK(1);
	// -- 1/2. broad-phase grid of the current poses (forces do not move bodies), then the step listeners that query it
	//         (VehicleConstraint::OnStep: wheel casts), then forces, then the pair search
	{ KScope k(w, KC_BP_CELL); launch_bp_bounds(d, nb, s); launch_bp_cell(d, nb, s); }
	{ KScope k(w, KC_BP_SCAN); launch_bp_scan(d, s); }
	{ KScope k(w, KC_BP_SCATTER); launch_bp_scatter_large(d, nb, s); }      // (+ the pairs with the large bodies: k_bp_large's work)
	if (p.n_vehicles) { KScope k(w, KC_VEHICLE); launch_vehicle_pre(d, s); }
	{ KScope k(w, KC_BP_PAIRS); launch_bp_pairs(d, p.bp_small, s); }
	STAGE_MARK(2);
	// -- 3. narrow phase, wake-ups, per-body solver records (+ contact events, which see the velocities before the solve)
	{ KScope k(w, KC_NARROWPHASE); launch_narrowphase(d, p.est_pairs, s); if (p.has_hulls) launch_narrowphase_hull(d, s); if (p.has_meshes) launch_narrowphase_mesh(d, p.has_hulls != 0, s); }
	// in-step activation: what the contacts above (or a wheel) woke takes its sleeping island along and collides in this step
	if (p.wake_round) { KScope k(w, KC_NARROWPHASE); launch_wake_round(d, nb, p.has_hulls, p.has_meshes, s); }
	{ KScope k(w, KC_APPLY_FORCES); launch_pre_solve(d, nb, s); }      // sweep 1/3: wake-ups, forces, per-step solver records
	STAGE_MARK(3);
	// -- 4. colouring + constraint setup (k_colour_inherit also resolves every manifold's slot in the previous step's constraints, which the
	//       contact events -- added or persisted? -- and the set-up read)
	{ KScope k(w, KC_COLOUR_CLAIM); launch_colour_inherit(d, p.est_man, s); }
	if (p.contact_events) { KScope k(w, KC_MISC); launch_contact_events(d, p.est_man, s); }
	if (p.small_colouring) {
		// few manifolds: one workgroup runs every colouring round (no per-round launches)
		KScope k(w, KC_COLOUR_COMMIT); launch_colour_finish(d, 0, 1, s);
	} else {
		uint32_t est_unc = p.est_man;
		for (uint32_t round = 0; round < p.rounds; ++round) {
			{ KScope k(w, KC_COLOUR_CLAIM); launch_colour_claim(d, est_unc, round, s); }
			{ KScope k(w, KC_COLOUR_COMMIT); launch_colour_commit(d, est_unc, round, s); }
			if (round >= 1) est_unc = std::max(est_unc - est_unc / 4, 8192u);      // worklists shrink; kernels grid-stride over the rest
		}
		{ KScope k(w, KC_COLOUR_COMMIT); launch_colour_finish(d, p.rounds, 0, s); }
	}
	if (p.tile_solver) {
		{ KScope k(w, KC_COLOUR_COUNT); launch_ts_label(d, nb, s); launch_colour_count_ts(d, p.est_man, s); }
		{ KScope k(w, KC_SETUP); launch_setup_ts(d, p.est_man, s); }
	} else {
		{ KScope k(w, KC_COLOUR_COUNT); launch_colour_count(d, p.est_man, s); }
		{ KScope k(w, KC_SETUP); launch_setup(d, p.est_man, s); }
	}
	if (p.hc_first >= 0 && p.hc_probe >= 0) { KScope k(w, KC_SETUP); launch_hc_probe(d, p.hc_probe, p.hc_probe_est, s); }
	if (p.hc_first >= 0) { KScope k(w, KC_SETUP); launch_hc_build(d, p.hc_first, p.hc_est, s); }
	STAGE_MARK(4);
	// -- 5. warm start + velocity iterations: one launch per planned colour, everything else in the single-workgroup tail
	auto solve_pass = [&](int mode, int kc) {
		// non-contact constraints first: the vehicles' rows ride in the launch of contact colour 0 (no chassis contact is in that colour) when the plan has one
		const bool veh_fused = p.n_vehicles && p.tail_first > 0 && mode != 0 && w->fuse_vehicle_solve;
		if (p.n_vehicles && !veh_fused) { KScope k(w, KC_VEHICLE); launch_vehicle_solve(d, mode, s); }
		for (int c = 0; c < p.tail_first; ++c) { KScope k(w, kc); if (c == 0 && veh_fused) launch_solve_colour_veh(d, c, p.colour_est[c], mode, s, (int)p.sp.compact_rows); else launch_solve_colour(d, c, p.colour_est[c], mode, s, (int)p.sp.compact_rows); }
		if (p.hc_first >= 0) { KScope k(w, kc); launch_solve_hc(d, p.hc_first, p.hc_est, mode, s, (int)p.sp.compact_rows); }      // colours >= tail_first by component + overflow colour
		else { KScope k(w, kc); launch_solve_tail(d, p.tail_first, mode, s, (int)p.sp.compact_rows); }
	};
	if (p.small_world) { KScope k(w, KC_SOLVE_VELOCITY); launch_solve_small(d, p.warm_start, p.vel_iters, p.small_pairs, s); }
	else {
		if (p.warm_start) {
			// vehicle rows first, then every contact constraint of the regular colours (one launch, by body), then the overflow colour
			if (p.n_vehicles) { KScope k(w, KC_VEHICLE); launch_vehicle_solve(d, 0, s); }
			{ KScope k(w, KC_WARM_START); launch_warm_bodies(d, nb, s); }
		}
		if (p.tile_solver == 1) { KScope k(w, KC_SOLVE_VELOCITY); launch_ts_solve(d, p.vel_iters, SGP_MAX_COLOURS, s); }
		else if (p.tile_solver == 3 && p.hc_first >= 0) {
			// the big colours of a pass in the resident tile launch, the sparse high colours by connected component (one launch each per pass)
			for (int it = 0; it < p.vel_iters; ++it) {
				{ KScope k(w, KC_SOLVE_VELOCITY); launch_ts_solve(d, 1, p.tail_first, s); }
				{ KScope k(w, KC_SOLVE_VELOCITY); launch_solve_hc(d, p.hc_first, p.hc_est, 1, s, (int)p.sp.compact_rows); }
			}
		}
		else for (int it = 0; it < p.vel_iters; ++it) solve_pass(1, KC_SOLVE_VELOCITY);
	}
	STAGE_MARK(5);
	// -- 6. the body-array sweep
	{ KScope k(w, KC_INTEGRATE_POSE); launch_integrate_pose(d, nb, s); }
	STAGE_MARK(6);
	// -- 7. position iterations
	for (int it = 0; it < p.pos_iters; ++it) solve_pass(2, KC_SOLVE_POSITION);
	STAGE_MARK(7);
	// -- 8. bounds, sleeping, buoyancy, contact cache
	{ KScope k(w, KC_FINALIZE); launch_finalize(d, nb, s); }
	for (int r = 0; r < SGP_ISLAND_MARK_ROUNDS; ++r) { KScope k(w, KC_ISLAND_HOOK); launch_island_mark(d, p.est_man, r == 0 ? 1 : 0, s); }
	{ KScope k(w, KC_ISLAND_HOOK); launch_island_hook(d, p.est_man, s); }
	{ KScope k(w, KC_ISLAND_FLAG); launch_island_flag(d, p.est_man, s); }
	{ KScope k(w, KC_SLEEP_APPLY); launch_sleep_apply(d, nb, s); }
	if (p.water) { KScope k(w, KC_BUOYANCY); launch_buoyancy(d, nb, s); }
	{
		KScope k(w, KC_CACHE_BUILD);
		launch_cache_build(d, p.est_man, w->h_ctr_dev, w->h_evc_dev, s);      // (+ the counters to host-mapped memory: the step's last launch)
	}
	STAGE_MARK(8);
	return SGP_OK;
}

static int step_impl(sgp_world* w, float dt, bool final_readback)
{
	if (!(dt > 0.0f)) return fail(SGP_ERR_INVALID, "sgp_world_step: dt must be > 0");
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	DV& d = w->dv;
	if (w->high == 0) { memset(&w->stats, 0, sizeof(w->stats)); return SGP_OK; }
	if (w->last_active == 0 && !w->dirty_since_step && !w->profiling) {
		// every body is asleep / static and nothing was edited since the last step: the step is the identity (no forces act on
		// sleeping bodies, no pair has an active member), exactly what PhysicsSystem::Update costs with an empty active list
		sgp_step_stats& st = w->stats;
		const uint32_t nb_ = st.num_bodies;
		uint32_t lc[SGP_NUM_LAYERS]; memcpy(lc, st.layer_counts, sizeof(lc));
		memset(&st, 0, sizeof(st));
		st.num_bodies = nb_; memcpy(st.layer_counts, lc, sizeof(lc)); st.device_bytes = w->device_bytes;
		w->idle_steps++;
		// the first skipped step takes the contact cache with it: a step without an awake body has no contacts, and what wakes up later (in-step
		// activation pairs bodies that were asleep) must not find the constraints of the last step that had some
		if (!w->cache_wiped) { launch_cache_wipe(d, w->stream); w->cache_wiped = true; }
		return SGP_OK;
	}
	w->cache_wiped = false;
	if (w->veh_inputs_dirty && w->n_vehicles) {
		HIP_TRY(hipMemcpyAsync(w->d_veh_inputs, w->veh_inputs.data(), sizeof(sgp_vehicle_input) * w->n_vehicles, hipMemcpyHostToDevice, w->stream));
		w->veh_inputs_dirty = false;
	}
	static const bool timing = getenv("SGP_TIMING") != nullptr;
	static double t_acc[4] = { 0, 0, 0, 0 }; static int t_n = 0;
	auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double tt0 = timing ? now() : 0.0;
	w->h_sp->dt = dt;
	w->h_sp->compact_rows = (w->n_con >= w->compact_rows_min && !(w->high <= SGP_SMALL_WORLD_BODIES)) ? w->rows_mode_large : 0u;      // (decided from the previous step's count; part of the plan's key)
	StepPlan plan;
	make_plan(w, plan);
	const std::string key((const char*)&plan, sizeof(plan));
	bool launched = false;
	if (w->use_graphs && !w->profiling) {
		auto it = w->graphs.find(key);
		if (it == w->graphs.end()) {
			// capture only once the same plan has come up several times in a row (plans churn while a scene is still changing, and a capture costs
			// 0.4 ms -- a fifth of a 100k-body step, where replaying a graph is worth under 1 % over issuing the launches; a small world, whose step IS
			// its launches, captures at the first repeat)
			const uint32_t par = 0u;
			w->plan_repeats[par] = (key == w->last_plan_key[par]) ? w->plan_repeats[par] + 1 : 0;
			if (w->plan_repeats[par] >= (plan.nb <= 2048u ? 1u : 3u)) {
				// THIS step is issued eagerly first; its plan is then captured on a second stream (capturing executes nothing) and instantiated while
				// the device is already at work on the step -- the 0.4-0.6 ms a capture costs the host no longer show in any step
				{ const int r0 = enqueue_step(w, plan); if (r0 != SGP_OK) return r0; w->eager_steps++; launched = true; }
				if (w->graphs.size() >= 16) { for (auto& kv : w->graphs) hipGraphExecDestroy(kv.second); w->graphs.clear(); }
				hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
				HIP_TRY(hipStreamBeginCapture(w->capture_stream, hipStreamCaptureModeThreadLocal));
				hipStream_t run_stream = w->stream; w->stream = w->capture_stream;
				const int r = enqueue_step(w, plan);
				w->stream = run_stream;
				const hipError_t e = hipStreamEndCapture(w->capture_stream, &g);
				if (r != SGP_OK) { if (g) hipGraphDestroy(g); return r; }
				if (e != hipSuccess) return fail(SGP_ERR_HIP, "hipStreamEndCapture", e);
				const hipError_t e2 = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
				hipGraphDestroy(g);
				if (e2 != hipSuccess) return fail(SGP_ERR_HIP, "hipGraphInstantiate", e2);
				w->graphs.emplace(key, ge);
			}
		}
		else { HIP_TRY(hipGraphLaunch(it->second, w->stream)); launched = true; w->graph_launches++; }
	}
	w->last_plan_key[0] = key;
	if (!launched) { const int r = enqueue_step(w, plan); if (r != SGP_OK) return r; w->eager_steps++; }
	w->sp_uploaded = plan.sp; w->sp_uploaded_valid = true;      // (k_step_begin wrote it)
	w->events_on_device = true;
	// -- the ONE host sync of the step: counters, events, and the launch plan for the next step
	const double tt1 = timing ? now() : 0.0;
	HIP_TRY(hipStreamSynchronize(w->stream));
	const double tt2 = timing ? now() : 0.0;
	w->h_sp->parity ^= 1u;                    // the buffer just solved becomes the contact cache of the next step
	w->grid_valid = false;                    // bodies moved after the broad phase of this step
	w->dirty_since_step = false;
	w->last_active = w->h_ctr->n_active;
	const StepCounters c1 = *w->h_ctr;
	const uint32_t n_con = c1.n_constraints;
	if (plan.tile_solver && c1.ts_error) return fail(SGP_ERR_HIP, "sgp_world_step: the tile solver timed out waiting for a neighbouring tile (a workgroup of the resident launch was not running); set SGP_TILE_SOLVER=0");
	w->n_con = n_con;
	w->last_pairs = c1.n_pairs; w->last_manifolds = c1.n_manifolds;
	w->plan_rounds = c1.rounds_used;
	memcpy(w->plan_round_n, c1.round_n, sizeof(w->plan_round_n));
	for (int c = 0; c < SGP_MAX_COLOURS; ++c) w->plan_colour_count[c] = c1.colour_count[c];
	w->plan_seen = true;
	w->bp_dense_last = c1.bp_dense != 0u;
	// the launch plan's first component colour: one colour fewer after a step that left a component to the serial catch-all (which costs
	// more per pass than the launch of one more colour); one more when this step's probe found that the next colour's components fit too
	// (probes get rarer while they fail, up to one in 1024 steps; a success is followed up at once)
	if (plan.hc_first >= 0) {
		int k = plan.hc_first;
		if (c1.hc_n_big > 0u) {
			// (a scene that keeps growing -- a tower coming down -- outruns single steps: the stride doubles while catch-alls follow each other closely)
			w->hc_bump = (w->hc_since_bump < 32u) ? std::min(2u * w->hc_bump, 8u) : 1u;
			w->hc_since_bump = 0;
			// (round 4: a BIG miss -- thousands of constraints in the serial catch-all, 10+ ms in a tile of the collapsing 1M tower -- takes the plan well clear
			// of where it missed at once: the probes bring it back a colour at a time when the scene has calmed down)
			if (c1.hc_n_big > 4096u) w->hc_bump = 8u;
			k = std::min(k + (int)w->hc_bump, (int)SGP_OVERFLOW_COLOUR - 1); w->hc_probe_gap = 64; w->hc_probe_in = 64;
		}
		else if (plan.hc_probe >= 0) {
			if (c1.hc_probe_big == 0u) { k = plan.hc_probe; w->hc_probe_gap = 16; w->hc_probe_in = 16; }
			else { w->hc_probe_gap = std::min(2u * w->hc_probe_gap, 1024u); w->hc_probe_in = w->hc_probe_gap; }
		} else if (w->hc_probe_in) w->hc_probe_in--;
		w->hc_k = k;
		if (w->hc_since_bump < 0xFFFFu) w->hc_since_bump++;
	}
	sgp_step_stats& st = w->stats;
	memset(&st, 0, sizeof(st));
	st.num_bodies = w->n_alive;
	st.num_wake_pairs = std::min(c1.n_wake_pairs, d.cap_wake_pairs);
	st.num_pairs = std::min(c1.n_pairs, d.cap_pairs) + st.num_wake_pairs;
	st.num_manifolds = n_con;
	st.num_contact_points = c1.n_points;
	st.num_colours = c1.n_colours;
	st.num_colour_rounds = c1.rounds_used;
	st.num_overflow_constraints = c1.colour_count[SGP_OVERFLOW_COLOUR];
	st.num_cached_manifolds = c1.n_cached;
	st.num_component_constraints = c1.hc_n; st.num_catch_all_constraints = c1.hc_n_big;
	st.num_deferred_vehicles = c1.veh_deferred;
	st.tile_solver = plan.tile_solver ? (c1.ts_all_adjacent ? 2u : 1u) : 0u;
	st.pairs_dropped = c1.pairs_dropped; st.manifolds_dropped = c1.manifolds_dropped;
	st.device_bytes = w->device_bytes;
	st.num_active = c1.n_active;
	if (final_readback) {
		const size_t a0 = w->ev_act.size(), d0 = w->ev_deact.size();
		{ int r = collect_events(w, /*counters_fresh=*/true); if (r != SGP_OK) return r; }
		st.num_activated = (uint32_t)(w->ev_act.size() - a0);
		st.num_deactivated = (uint32_t)(w->ev_deact.size() - d0);
		for (uint32_t i = 0; i < w->high; ++i) if ((w->hb[i].flags & (BF_ALIVE | BF_ALIAS)) == BF_ALIVE) st.layer_counts[(w->hb[i].flags & BF_LAYER_MASK) >> BF_LAYER_SHIFT]++;
	}
	if (timing) {
		const double tt3 = now();
		t_acc[0] += tt1 - tt0; t_acc[1] += tt2 - tt1; t_acc[2] += tt3 - tt2; t_n++;
		if (t_n == 500 || t_n == 100) {
			fprintf(stderr, "[sgp timing] enqueue %.1f us  sync wait %.1f us  post %.1f us; colouring: %u rounds used, %u planned wide, uncoloured at round start:", t_acc[0] / t_n, t_acc[1] / t_n, t_acc[2] / t_n, c1.rounds_used, plan.rounds);
			for (uint32_t r = 1; r < std::min(c1.rounds_used + 1u, 32u); ++r) fprintf(stderr, " %u", c1.round_n[r]);
			fprintf(stderr, "\n");
			if (t_n == 500) { t_acc[0] = t_acc[1] = t_acc[2] = 0; t_n = 0; }
		}
	}
	return SGP_OK;
}

SGP_API int sgp_world_step(sgp_world* w, float dt)
{
	if (!w) return fail(SGP_ERR_INVALID, "sgp_world_step: NULL");
	hipSetDevice(w->device);
	w->profiling = false;
	return step_impl(w, dt, true);
}

SGP_API int sgp_world_step_n(sgp_world* w, float dt, uint32_t n)
{
	if (!w) return fail(SGP_ERR_INVALID, "sgp_world_step_n: NULL");
	hipSetDevice(w->device);
	w->profiling = false;
	for (uint32_t i = 0; i < n; ++i) { const int r = step_impl(w, dt, i + 1 == n); if (r != SGP_OK) return r; }
	return SGP_OK;
}

SGP_API int sgp_world_step_profiled(sgp_world* w, float dt, sgp_step_profile* out)
{
	if (!w || !out) return fail(SGP_ERR_INVALID, "sgp_world_step_profiled: NULL");
	hipSetDevice(w->device);
	if (!w->stage_ev_ok) { for (int i = 0; i <= SGP_NUM_STAGES; ++i) HIP_TRY(hipEventCreate(&w->stage_ev[i])); w->stage_ev_ok = true; }
	w->profiling = true; w->prof.clear(); w->event_next = 0;
	const int r = step_impl(w, dt, true);
	w->profiling = false;
	if (r != SGP_OK) return r;
	HIP_TRY(hipStreamSynchronize(w->stream));
	memset(out, 0, sizeof(*out));
	if (w->high == 0) return SGP_OK;
	for (int i = 0; i < SGP_NUM_STAGES; ++i) { float ms = 0.0f; hipEventElapsedTime(&ms, w->stage_ev[i], w->stage_ev[i + 1]); out->stage_ms[i] = ms; }
	hipEventElapsedTime(&out->total_ms, w->stage_ev[0], w->stage_ev[SGP_NUM_STAGES]);
	for (const ProfEvent& p : w->prof) { float ms = 0.0f; hipEventElapsedTime(&ms, p.a, p.b); out->kernel_ms[p.kc] += ms; out->kernel_launches[p.kc]++; }
	out->sweep_bodies = w->high;
	out->num_constraints = w->stats.num_manifolds;
	out->num_contact_points = w->stats.num_contact_points;
	out->num_colours = w->stats.num_colours;
	return SGP_OK;
}

// Timing probe (tools/solve_probe.py; not declared in include/sgp.h): average time of `reps` back-to-back launches of the velocity
// iteration of one colour, full (variant 0) or with parts removed (see k_solve_probe).  Leaves the velocities of the world perturbed.
// 1 when the library carries csrc/experiments/* (python -m substrata_amd.build --experiments); not declared in include/sgp.h
SGP_API int sgp_debug_has_experiments(void)
{
#ifdef SGP_EXPERIMENTS
	return 1;
#else
	return 0;
#endif
}
SGP_API int sgp_debug_time_solve(sgp_world* w, int variant, int colour, int reps, float* us_out, uint32_t* count_out)
{
#ifndef SGP_EXPERIMENTS
	return fail(SGP_ERR_INVALID, "sgp_debug_time_solve: this library was built without the experiments (python -m substrata_amd.build --experiments)");
#endif
	if (!w || !us_out || colour < 0 || colour >= SGP_OVERFLOW_COLOUR) return fail(SGP_ERR_INVALID, "sgp_debug_time_solve");
	hipSetDevice(w->device);
	hipEvent_t e0, e1; HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
	const uint32_t est = w->plan_colour_count[colour];
	for (int r = 0; r < 3; ++r) launch_solve_probe(w->dv, variant, colour, est, w->stream);
	HIP_TRY(hipEventRecord(e0, w->stream));
	for (int r = 0; r < reps; ++r) launch_solve_probe(w->dv, variant, colour, est, w->stream);
	HIP_TRY(hipEventRecord(e1, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	float ms = 0.0f; HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
	hipEventDestroy(e0); hipEventDestroy(e1);
	*us_out = 1000.0f * ms / (float)reps;
	if (count_out) *count_out = est;
	return SGP_OK;
}

SGP_API int sgp_world_stats(sgp_world* w, sgp_step_stats* out)
{
	if (!w || !out) return fail(SGP_ERR_INVALID, "sgp_world_stats: NULL");
	*out = w->stats;
	out->num_bodies = w->n_alive;
	out->device_bytes = w->device_bytes;
	return SGP_OK;
}

SGP_API int sgp_world_launch_counts(sgp_world* w, uint32_t* graph_replays_out, uint32_t* eager_steps_out, uint32_t* idle_steps_out)
{
	if (!w) return fail(SGP_ERR_INVALID, "sgp_world_launch_counts: NULL");
	if (graph_replays_out) *graph_replays_out = w->graph_launches;
	if (eager_steps_out) *eager_steps_out = w->eager_steps;
	if (idle_steps_out) *idle_steps_out = w->idle_steps;
	return SGP_OK;
}

SGP_API int sgp_world_num_bodies(sgp_world* w, uint32_t* n_out)
{
	if (!w || !n_out) return fail(SGP_ERR_INVALID, "sgp_world_num_bodies: NULL");
	*n_out = w->n_alive;
	return SGP_OK;
}

SGP_API int sgp_world_set_water(sgp_world* w, int enabled, float z)
{
	if (!w) return fail(SGP_ERR_INVALID, "sgp_world_set_water: NULL");
	w->h_sp->water_enabled = enabled; w->h_sp->water_z = z;
	w->dirty_since_step = true;
	return SGP_OK;
}

SGP_API int sgp_world_set_contact_events(sgp_world* w, int enabled)
{
	if (!w) return fail(SGP_ERR_INVALID, "sgp_world_set_contact_events: NULL");
	hipSetDevice(w->device);
	if (enabled && !w->dv.ev_contacts_added) {
		w->dv.cap_contact_events = w->dv.cap_manifolds;
		DEV_ALLOC(w->dv.ev_contacts_added, w->dv.cap_contact_events);
		DEV_ALLOC(w->dv.ev_contacts_persisted, w->dv.cap_contact_events);
		for (auto& kv : w->graphs) hipGraphExecDestroy(kv.second);
		w->graphs.clear();
	}
	w->h_sp->contact_events = enabled;
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// static triangle meshes (MeshShapeSettings::Create, PhysicsWorld.cpp:735-1166 with is_dynamic = false)

static void invalidate_graphs(sgp_world* w);

template <typename T> static int grow_pool(sgp_world* w, T*& dev, size_t& cap, size_t need, size_t used_before)
{
	if (need <= cap) return SGP_OK;
	size_t nc = std::max<size_t>(need + need / 2, 4096);
	T* nd = nullptr;
	HIP_TRY(hipMalloc((void**)&nd, sizeof(T) * nc));
	if (dev && used_before) HIP_TRY(hipMemcpyAsync(nd, dev, sizeof(T) * used_before, hipMemcpyDeviceToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	if (dev) { hipFree(dev); w->device_bytes -= sizeof(T) * cap; }
	dev = nd; cap = nc; w->device_bytes += sizeof(T) * nc;
	return SGP_OK;
}

// median-split tree over the triangles [first, first + count) of `order`; returns the node index
static uint32_t build_mesh_node(std::vector<MeshNode>& nodes, size_t node_base, std::vector<uint32_t>& order, const std::vector<float>& cen, const std::vector<float>& tmin, const std::vector<float>& tmax, uint32_t first, uint32_t count)
{
	const uint32_t me = (uint32_t)(nodes.size() - node_base);
	nodes.push_back(MeshNode{});
	float mn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, mx[3] = { -3.4e38f, -3.4e38f, -3.4e38f }, cmn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, cmx[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
	for (uint32_t k = first; k < first + count; ++k) for (int a = 0; a < 3; ++a) {
		const uint32_t t = order[k];
		mn[a] = std::min(mn[a], tmin[3 * t + a]); mx[a] = std::max(mx[a], tmax[3 * t + a]);
		cmn[a] = std::min(cmn[a], cen[3 * t + a]); cmx[a] = std::max(cmx[a], cen[3 * t + a]);
	}
	MeshNode nd{};
	nd.mnx = mn[0]; nd.mny = mn[1]; nd.mnz = mn[2]; nd.mxx = mx[0]; nd.mxy = mx[1]; nd.mxz = mx[2];
	int axis = 0; if (cmx[1] - cmn[1] > cmx[axis] - cmn[axis]) axis = 1; if (cmx[2] - cmn[2] > cmx[axis] - cmn[axis]) axis = 2;
	if (count <= 4 || !(cmx[axis] - cmn[axis] > 0.0f)) { nd.left = first; nd.right = 0; nd.count = count; nodes[node_base + me] = nd; return me; }
	const uint32_t mid = first + count / 2;
	std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](uint32_t x, uint32_t y) { return cen[3 * x + axis] < cen[3 * y + axis] || (cen[3 * x + axis] == cen[3 * y + axis] && x < y); });
	nd.count = 0;
	nd.left = build_mesh_node(nodes, node_base, order, cen, tmin, tmax, first, mid - first);
	nd.right = build_mesh_node(nodes, node_base, order, cen, tmin, tmax, mid, first + count - mid);
	nodes[node_base + me] = nd;
	return me;
}

// first-fit from the ranges destroyed shapes gave back, else the end of the pool
static uint32_t take_range(std::vector<std::pair<uint32_t, uint32_t>>& free_ranges, uint32_t len, size_t pool_end)
{
	for (size_t k = 0; k < free_ranges.size(); ++k) if (free_ranges[k].second >= len) {
		const uint32_t off = free_ranges[k].first;
		if (free_ranges[k].second == len) free_ranges.erase(free_ranges.begin() + (long)k); else { free_ranges[k].first += len; free_ranges[k].second -= len; }
		return off;
	}
	return (uint32_t)pool_end;
}
static void give_range(std::vector<std::pair<uint32_t, uint32_t>>& free_ranges, uint32_t off, uint32_t len)
{
	if (!len) return;
	free_ranges.push_back(std::make_pair(off, len));
	std::sort(free_ranges.begin(), free_ranges.end());
	for (size_t k = 0; k + 1 < free_ranges.size();) {       // merge neighbours
		if (free_ranges[k].first + free_ranges[k].second == free_ranges[k + 1].first) { free_ranges[k].second += free_ranges[k + 1].second; free_ranges.erase(free_ranges.begin() + (long)k + 1); } else ++k;
	}
}
template <typename T> static int grow_table(sgp_world* w, T*& dev, size_t& cap, size_t need)
{
	if (need <= cap) return SGP_OK;
	const size_t nc = std::max(need, 2 * cap);
	T* nd = nullptr;
	HIP_TRY(hipMalloc((void**)&nd, sizeof(T) * nc));
	HIP_TRY(hipMemsetAsync(nd, 0, sizeof(T) * nc, w->stream));
	HIP_TRY(hipMemcpyAsync(nd, dev, sizeof(T) * cap, hipMemcpyDeviceToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	hipFree(dev); w->device_bytes += sizeof(T) * (nc - cap);
	dev = nd; cap = nc;
	return SGP_OK;
}

// Which edges of which triangles are ACTIVE (MeshShape::sFindActiveEdges + ActiveEdges::IsEdgeActive with the 5 degree default the reference leaves in
// place, PhysicsWorld.cpp:1028-1060): flags[t] bit k set = edge k (v[k] - v[k + 1]) of triangle t collides with its own normal.  An edge is keyed by its two
// vertex indices: used by one triangle or by more than two -> active; by two -> inactive when concave or when their normals are within the threshold.
// Doubles: the flags must come out the same wherever this runs (tests/test_mesh_parity_gpu.py compares them with the sequential CPU statement's).
#define SGP_ACTIVE_EDGE_COS 0.99619469809f      // cos(5 degrees) as a float (MeshShapeSettings::mActiveEdgeCosThresholdAngle), widened to double for the test below
static void mesh_active_edges(const float* verts, const uint32_t* idx, uint32_t nt, std::vector<uint8_t>& flags)
{
	struct Rec { uint32_t lo, hi, tri, k; };
	std::vector<Rec> e(3 * (size_t)nt);
	flags.assign(nt, 7);
	for (uint32_t t = 0; t < nt; ++t) for (uint32_t k = 0; k < 3; ++k) { const uint32_t a = idx[3 * t + k], b = idx[3 * t + (k + 1) % 3]; e[3 * (size_t)t + k] = Rec{ std::min(a, b), std::max(a, b), t, k }; }
	std::sort(e.begin(), e.end(), [](const Rec& x, const Rec& y) { if (x.lo != y.lo) return x.lo < y.lo; if (x.hi != y.hi) return x.hi < y.hi; if (x.tri != y.tri) return x.tri < y.tri; return x.k < y.k; });
	auto normal = [&](uint32_t t, double n[3]) {
		const float* a = verts + 3 * idx[3 * t]; const float* b = verts + 3 * idx[3 * t + 1]; const float* c = verts + 3 * idx[3 * t + 2];
		const double e1[3] = { (double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2] }, e2[3] = { (double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2] };
		n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
		const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
		if (!(l > 1.0e-30)) return false;
		n[0] /= l; n[1] /= l; n[2] /= l;
		return true;
	};
	const double cos_threshold = (double)SGP_ACTIVE_EDGE_COS;
	for (size_t i = 0; i < e.size(); ) {
		size_t j = i + 1;
		while (j < e.size() && e[j].lo == e[i].lo && e[j].hi == e[i].hi) ++j;
		if (j - i == 2 && e[i].lo != e[i].hi) {
			double n1[3], n2[3];
			if (normal(e[i].tri, n1) && normal(e[i + 1].tri, n2)) {
				const uint32_t va = idx[3 * e[i].tri + e[i].k], vb = idx[3 * e[i].tri + (e[i].k + 1) % 3];       // the edge in the first triangle's winding
				const double d[3] = { (double)verts[3 * vb] - verts[3 * va], (double)verts[3 * vb + 1] - verts[3 * va + 1], (double)verts[3 * vb + 2] - verts[3 * va + 2] };
				const double cosn = n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
				const double cx = n1[1] * n2[2] - n1[2] * n2[1], cy = n1[2] * n2[0] - n1[0] * n2[2], cz = n1[0] * n2[1] - n1[1] * n2[0];
				bool active;
				if (cosn < -0.999848) active = true;                                    // back to back
				else if (cx * d[0] + cy * d[1] + cz * d[2] < 0.0) active = false;       // concave
				else active = cosn < cos_threshold;                                     // convex: active beyond the threshold angle
				if (!active) { flags[e[i].tri] &= (uint8_t)~(1u << e[i].k); flags[e[i + 1].tri] &= (uint8_t)~(1u << e[i + 1].k); }
			}
		}
		i = j;
	}
}

SGP_API int sgp_mesh_create_with_materials(sgp_world* w, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t nt, const uint32_t* tri_mats, sgp_mesh_info* info);
SGP_API int sgp_mesh_create(sgp_world* w, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t nt, sgp_mesh_info* info)
{
	return sgp_mesh_create_with_materials(w, verts, nv, idx, nt, nullptr, info);
}
SGP_API int sgp_mesh_create_with_materials(sgp_world* w, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t nt, const uint32_t* tri_mats, sgp_mesh_info* info)
{
	if (!w || !verts || !idx || !info || nv < 3 || nt < 1) return fail(SGP_ERR_INVALID, "sgp_mesh_create: bad arguments");
	for (uint32_t k = 0; k < 3 * nt; ++k) if (idx[k] >= nv) return fail(SGP_ERR_INVALID, "sgp_mesh_create: vertex index out of range");
	for (uint32_t k = 0; k < 3 * nv; ++k) if (!std::isfinite(verts[k])) return fail(SGP_ERR_INVALID, "sgp_mesh_create: non-finite vertex");
	hipSetDevice(w->device);
	MeshHeader mh{};
	mh.nv = nv; mh.nt = nt;
	mh.vert_off = take_range(w->free_vert_ranges, nv, w->mesh_verts.size());
	mh.tri_off = take_range(w->free_tri_ranges, nt, w->mesh_tris.size());
	if (w->mesh_verts.size() < (size_t)mh.vert_off + nv) w->mesh_verts.resize((size_t)mh.vert_off + nv);
	if (w->mesh_tris.size() < (size_t)mh.tri_off + nt) { w->mesh_tris.resize((size_t)mh.tri_off + nt); w->mesh_tri_mat.resize((size_t)mh.tri_off + nt); }
	float mn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, mx[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
	for (uint32_t k = 0; k < nv; ++k) {
		w->mesh_verts[mh.vert_off + k] = make_float4(verts[3 * k], verts[3 * k + 1], verts[3 * k + 2], 0.0f);
		for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], verts[3 * k + a]); mx[a] = std::max(mx[a], verts[3 * k + a]); }
	}
	mh.mnx = mn[0]; mh.mny = mn[1]; mh.mnz = mn[2]; mh.mxx = mx[0]; mh.mxy = mx[1]; mh.mxz = mx[2];
	std::vector<float> cen(3 * (size_t)nt), tmin(3 * (size_t)nt), tmax(3 * (size_t)nt);
	std::vector<uint32_t> order(nt);
	for (uint32_t t = 0; t < nt; ++t) {
		order[t] = t;
		for (int a = 0; a < 3; ++a) {
			const float p0 = verts[3 * idx[3 * t] + a], p1 = verts[3 * idx[3 * t + 1] + a], p2 = verts[3 * idx[3 * t + 2] + a];
			cen[3 * t + a] = (p0 + p1 + p2) * (1.0f / 3.0f); tmin[3 * t + a] = std::min(p0, std::min(p1, p2)); tmax[3 * t + a] = std::max(p0, std::max(p1, p2));
		}
	}
	{      // the tree is built aside (its size is not known beforehand), then placed in a freed range or at the end of the node pool
		std::vector<MeshNode> nodes;
		build_mesh_node(nodes, 0, order, cen, tmin, tmax, 0, nt);
		mh.n_nodes = (uint32_t)nodes.size();
		mh.node_off = take_range(w->free_node_ranges, mh.n_nodes, w->mesh_nodes.size());
		if (w->mesh_nodes.size() < (size_t)mh.node_off + mh.n_nodes) w->mesh_nodes.resize((size_t)mh.node_off + mh.n_nodes);
		std::copy(nodes.begin(), nodes.end(), w->mesh_nodes.begin() + mh.node_off);
	}
	if (nt >= (1u << 29)) return fail(SGP_ERR_CAPACITY, "sgp_mesh_create: more than 2^29 triangles");
	std::vector<uint8_t> edge_flags;
	mesh_active_edges(verts, idx, nt, edge_flags);
	// (uint4.w of a triangle: its index in the caller's order, and in the top three bits its active-edge flags: MESH_TRI_INDEX / MESH_TRI_EDGES)
	for (uint32_t k = 0; k < nt; ++k) { const uint32_t t = order[k]; w->mesh_tris[mh.tri_off + k] = make_uint4(idx[3 * t], idx[3 * t + 1], idx[3 * t + 2], t | ((uint32_t)edge_flags[t] << 29)); w->mesh_tri_mat[mh.tri_off + k] = tri_mats ? tri_mats[t] : 0u; }
	// upload (pools may move: captured graphs carry the old pointers)
	{ int r = grow_pool(w, w->d_mesh_verts, w->cap_mesh_verts, w->mesh_verts.size(), w->cap_mesh_verts); if (r != SGP_OK) return r; }
	{ int r = grow_pool(w, w->d_mesh_tris, w->cap_mesh_tris, w->mesh_tris.size(), w->cap_mesh_tris); if (r != SGP_OK) return r; }
	{ int r = grow_pool(w, w->d_mesh_tri_mat, w->cap_mesh_tri_mat, w->mesh_tri_mat.size(), w->cap_mesh_tri_mat); if (r != SGP_OK) return r; }
	{ int r = grow_pool(w, w->d_mesh_nodes, w->cap_mesh_nodes, w->mesh_nodes.size(), w->cap_mesh_nodes); if (r != SGP_OK) return r; }
	HIP_TRY(hipMemcpyAsync(w->d_mesh_verts + mh.vert_off, w->mesh_verts.data() + mh.vert_off, sizeof(float4) * nv, hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipMemcpyAsync(w->d_mesh_tris + mh.tri_off, w->mesh_tris.data() + mh.tri_off, sizeof(uint4) * nt, hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipMemcpyAsync(w->d_mesh_tri_mat + mh.tri_off, w->mesh_tri_mat.data() + mh.tri_off, sizeof(uint32_t) * nt, hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipMemcpyAsync(w->d_mesh_nodes + mh.node_off, w->mesh_nodes.data() + mh.node_off, sizeof(MeshNode) * mh.n_nodes, hipMemcpyHostToDevice, w->stream));
	uint32_t id;
	if (!w->free_mesh_ids.empty()) { id = w->free_mesh_ids.back(); w->free_mesh_ids.pop_back(); w->meshes[id] = mh; w->mesh_refs[id] = 0; }
	else { id = (uint32_t)w->meshes.size(); w->meshes.push_back(mh); w->mesh_refs.push_back(0); }
	{ int r = grow_table(w, w->d_meshes, w->cap_mesh_table, w->meshes.size()); if (r != SGP_OK) return r; }
	w->dv.meshes = w->d_meshes;
	HIP_TRY(hipMemcpyAsync(&w->d_meshes[id], &w->meshes[id], sizeof(MeshHeader), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->dv.mesh_verts = w->d_mesh_verts; w->dv.mesh_tris = w->d_mesh_tris; w->dv.mesh_tri_mat = w->d_mesh_tri_mat; w->dv.mesh_nodes = w->d_mesh_nodes; w->dv.n_meshes = (uint32_t)w->meshes.size();
	invalidate_graphs(w);
	memset(info, 0, sizeof(*info));
	info->mesh_id = id; info->num_vertices = nv; info->num_triangles = nt; info->num_nodes = mh.n_nodes;
	memcpy(info->aabb_min, mn, sizeof(mn)); memcpy(info->aabb_max, mx, sizeof(mx));
	return SGP_OK;
}

// JPH::Ref<JPH::Shape> going out of scope: the mesh's table slot and pool ranges become reusable.  Refused while a body still uses it.
// the active-edge bits of a mesh's triangles in the caller's triangle order (tests: compared with the sequential CPU statement's)
SGP_API int sgp_mesh_edge_flags(sgp_world* w, uint32_t mesh_id, uint8_t* out, uint32_t cap)
{
	if (!w || !out || mesh_id < 1 || mesh_id >= w->meshes.size() || w->meshes[mesh_id].nt == 0) return fail(SGP_ERR_BAD_ID, "sgp_mesh_edge_flags: no such mesh");
	const MeshHeader& mh = w->meshes[mesh_id];
	for (uint32_t k = 0; k < mh.nt; ++k) { const uint32_t wv = w->mesh_tris[mh.tri_off + k].w; const uint32_t t = wv & 0x1FFFFFFFu; if (t < cap) out[t] = (uint8_t)(wv >> 29); }
	return SGP_OK;
}

SGP_API int sgp_mesh_destroy(sgp_world* w, uint32_t id)
{
	if (!w || id < 1 || id >= w->meshes.size() || w->meshes[id].nt == 0) return fail(SGP_ERR_BAD_ID, "sgp_mesh_destroy: no such mesh");
	if (w->mesh_refs[id] != 0) return fail(SGP_ERR_REJECTED, "sgp_mesh_destroy: a body still uses the mesh");
	hipSetDevice(w->device);
	const MeshHeader mh = w->meshes[id];
	give_range(w->free_vert_ranges, mh.vert_off, mh.nv); give_range(w->free_tri_ranges, mh.tri_off, mh.nt); give_range(w->free_node_ranges, mh.node_off, mh.n_nodes);
	w->meshes[id] = MeshHeader{};
	HIP_TRY(hipMemcpyAsync(&w->d_meshes[id], &w->meshes[id], sizeof(MeshHeader), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->free_mesh_ids.push_back(id);
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// convex hull shapes (ConvexHullShapeSettings::Create, CarPhysics.cpp:66-78)

static void invalidate_graphs(sgp_world* w);

SGP_API int sgp_hull_create_com(sgp_world* w, const float* pts, uint32_t n, const float* com_offset, sgp_hull_info* info)
{
	if (!w || !pts || !info || n < 4 || n > 100000) return fail(SGP_ERR_INVALID, "sgp_hull_create: bad arguments");
	hipSetDevice(w->device);
	sgd_hull h;
	float com[3], rot[4];
	if (sgd_hull_build(pts, (int)n, com_offset, &h, com, rot) != 0) return fail(SGP_ERR_REJECTED, "sgp_hull_create: degenerate point cloud or too many faces");
	uint32_t id;
	if (!w->free_hull_ids.empty()) { id = w->free_hull_ids.back(); w->free_hull_ids.pop_back(); w->hulls[id] = h; w->hull_refs[id] = 0; }
	else { id = (uint32_t)w->hulls.size(); w->hulls.push_back(h); w->hull_refs.push_back(0); }
	{ int r = grow_table(w, w->d_hulls, w->cap_hull_table, w->hulls.size()); if (r != SGP_OK) return r; }
	w->dv.hulls = w->d_hulls;
	HIP_TRY(hipMemcpyAsync(&w->d_hulls[id], &w->hulls[id], sizeof(sgd_hull), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->dv.n_hulls = (uint32_t)w->hulls.size();
	invalidate_graphs(w);                        // DV travels by value in the captured launches
	memset(info, 0, sizeof(*info));
	info->hull_id = id; info->num_vertices = (uint32_t)h.nv; info->num_faces = (uint32_t)h.nf; info->num_edges = (uint32_t)h.ne;
	memcpy(info->com, com, sizeof(com)); memcpy(info->rot, rot, sizeof(rot));
	info->volume = h.volume;
	info->unit_inertia[0] = h.unit_inertia.x; info->unit_inertia[1] = h.unit_inertia.y; info->unit_inertia[2] = h.unit_inertia.z;
	info->aabb_min[0] = h.aabb_min.x; info->aabb_min[1] = h.aabb_min.y; info->aabb_min[2] = h.aabb_min.z;
	info->aabb_max[0] = h.aabb_max.x; info->aabb_max[1] = h.aabb_max.y; info->aabb_max[2] = h.aabb_max.z;
	return SGP_OK;
}

SGP_API int sgp_hull_create(sgp_world* w, const float* pts, uint32_t n, sgp_hull_info* info) { return sgp_hull_create_com(w, pts, n, nullptr, info); }
SGP_API int sgp_hull_destroy(sgp_world* w, uint32_t id)
{
	if (!w || id < 1 || id >= w->hulls.size() || w->hulls[id].nv == 0) return fail(SGP_ERR_BAD_ID, "sgp_hull_destroy: no such hull");
	if (w->hull_refs[id] != 0) return fail(SGP_ERR_REJECTED, "sgp_hull_destroy: a body still uses the hull");
	hipSetDevice(w->device);
	memset(&w->hulls[id], 0, sizeof(sgd_hull));
	HIP_TRY(hipMemcpyAsync(&w->d_hulls[id], &w->hulls[id], sizeof(sgd_hull), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->free_hull_ids.push_back(id);
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// wheeled vehicles (VehicleConstraint + WheeledVehicleController, CarPhysics.cpp:94-231)

static void invalidate_graphs(sgp_world* w)
{
	for (auto& kv : w->graphs) hipGraphExecDestroy(kv.second);
	w->graphs.clear();
	for (int k = 0; k < 2; ++k) { w->last_plan_key[k].clear(); w->plan_repeats[k] = 0; }
}

SGP_API void sgp_default_vehicle_desc(sgp_vehicle_desc* d)
{
	memset(d, 0, sizeof(*d));
	d->body = SGP_INVALID_ID;
	d->num_wheels = 4;
	for (int i = 0; i < 4; ++i) {
		sgp_wheel_desc* w = &d->wheels[i];
		const bool front = i < 2, left = (i % 2) == 0;
		w->position[0] = left ? -0.8f : 0.8f; w->position[1] = front ? 1.3f : -1.3f; w->position[2] = 0.15f;
		w->suspension_dir[2] = -1.0f; w->steering_axis[2] = 1.0f; w->wheel_up[2] = 1.0f; w->wheel_forward[1] = 1.0f;
		w->suspension_min_length = 0.2f; w->suspension_max_length = 0.5f; w->suspension_preload = 0.0f;     // Scripting.cpp:326-330
		w->spring_frequency = 2.0f; w->spring_damping = 0.5f;                                              // :335-339
		w->radius = 0.42f; w->width = 0.16f;                                                               // :320-324
		w->inertia = 0.9f; w->angular_damping = 0.2f;                                                      // JPH::WheelSettingsWV defaults
		w->max_steer_angle = front ? 0.78525f : 0.0f;                                                      // :342, CarPhysics.cpp:127,153
		w->max_brake_torque = 1500.0f; w->max_handbrake_torque = front ? 0.0f : 4000.0f;                   // :347-348, CarPhysics.cpp:129,155
		const float lf[3][2] = { { 0.0f, 0.0f }, { 0.06f, 1.2f }, { 0.2f, 1.0f } };
		const float tf[3][2] = { { 0.0f, 0.0f }, { 3.0f, 1.2f }, { 20.0f, 1.0f } };
		memcpy(w->longitudinal_friction, lf, sizeof(lf)); memcpy(w->lateral_friction, tf, sizeof(tf));
	}
	d->up[2] = 1.0f; d->forward[1] = 1.0f;
	d->cast_radius = 0.08f;                                                                              // 0.5 * front_wheel_width, CarPhysics.cpp:62
	d->max_slope_angle = 80.0f * 3.14159265358979323846f / 180.0f;
	d->engine_max_torque = 500.0f; d->engine_min_rpm = 1000.0f; d->engine_max_rpm = 6000.0f; d->engine_inertia = 0.5f; d->engine_angular_damping = 0.2f;
	const float ec[3][2] = { { 0.0f, 0.8f }, { 0.66f, 1.0f }, { 1.0f, 0.8f } };
	memcpy(d->engine_torque_curve, ec, sizeof(ec));
	d->num_gears = 5; d->num_reverse_gears = 1;
	const float gr[5] = { 2.66f, 1.78f, 1.3f, 1.0f, 0.74f };
	memcpy(d->gear_ratios, gr, sizeof(gr)); d->reverse_gear_ratios[0] = -2.9f;
	d->switch_time = 0.5f; d->clutch_release_time = 0.3f; d->switch_latency = 0.5f; d->shift_up_rpm = 4000.0f; d->shift_down_rpm = 2000.0f; d->clutch_strength = 10.0f;
	d->num_differentials = 1;                                                                            // front wheel drive, CarPhysics.cpp:191-194
	d->differentials[0].left_wheel = 0; d->differentials[0].right_wheel = 1;
	d->differentials[0].differential_ratio = 3.42f; d->differentials[0].left_right_split = 0.5f; d->differentials[0].limited_slip_ratio = 1.4f; d->differentials[0].engine_torque_ratio = 1.0f;
	d->differentials[1] = d->differentials[0]; d->differentials[1].left_wheel = 2; d->differentials[1].right_wheel = 3;
	d->differential_limited_slip_ratio = 1.4f;
	d->num_anti_roll_bars = 2;                                                                           // CarPhysics.cpp:217-221
	d->anti_roll_bars[0].left_wheel = 0; d->anti_roll_bars[0].right_wheel = 1; d->anti_roll_bars[0].stiffness = 1000.0f;
	d->anti_roll_bars[1].left_wheel = 2; d->anti_roll_bars[1].right_wheel = 3; d->anti_roll_bars[1].stiffness = 1000.0f;
	d->controller_type = SGP_VEHICLE_CONTROLLER_WHEELED;
	d->max_lean_angle = 45.0f * 3.14159265358979323846f / 180.0f; d->lean_spring_constant = 5000.0f; d->lean_spring_damping = 1000.0f;   // JPH::MotorcycleControllerSettings defaults
	d->lean_spring_integration_coefficient = 0.0f; d->lean_spring_integration_decay = 4.0f; d->lean_smoothing_factor = 0.8f; d->lean_steering_limit = 1;
}

static bool vehicle_desc_valid(const sgp_vehicle_desc* d)
{
	if (d->num_wheels < 1 || d->num_wheels > SGP_MAX_WHEELS) return false;
	if (d->num_gears < 1 || d->num_gears > SGP_MAX_GEARS || d->num_reverse_gears < 1 || d->num_reverse_gears > SGP_MAX_GEARS) return false;
	if (d->num_differentials > 2 || d->num_anti_roll_bars > 2) return false;
	for (uint32_t k = 0; k < d->num_differentials; ++k) {
		if (d->differentials[k].left_wheel >= (int)d->num_wheels || d->differentials[k].right_wheel >= (int)d->num_wheels) return false;
		if (!(d->differentials[k].limited_slip_ratio > 1.0f)) return false;
	}
	for (uint32_t k = 0; k < d->num_anti_roll_bars; ++k) {
		const sgp_anti_roll_bar_desc* r = &d->anti_roll_bars[k];
		if (r->left_wheel < 0 || r->right_wheel < 0 || r->left_wheel >= (int)d->num_wheels || r->right_wheel >= (int)d->num_wheels) return false;
	}
	for (uint32_t i = 0; i < d->num_wheels; ++i) {
		const sgp_wheel_desc* w = &d->wheels[i];
		if (!(w->radius > 0.0f) || !(w->inertia > 0.0f) || !(w->suspension_max_length >= w->suspension_min_length) || !(w->suspension_min_length >= 0.0f)) return false;
	}
	if (!(d->engine_inertia > 0.0f) || !(d->engine_max_rpm > 0.0f) || !(d->clutch_release_time > 0.0f) || !(d->differential_limited_slip_ratio > 1.0f)) return false;
	if (d->controller_type != SGP_VEHICLE_CONTROLLER_WHEELED && d->controller_type != SGP_VEHICLE_CONTROLLER_MOTORCYCLE) return false;
	if (d->controller_type == SGP_VEHICLE_CONTROLLER_MOTORCYCLE && !(d->max_lean_angle > 0.0f && d->max_lean_angle < 1.5f)) return false;
	return true;
}

// cos(max slope) by the same fixed polynomial the kernels use for their trigonometry (|x| <= 1.5)
static float host_cos_poly(float x)
{
	if (fabsf(x) > 1.5f) return cosf(x);
	const float x2 = x * x;
	float pc = 2.08767569878681e-9f;
	pc = pc * x2 - 2.75573192239859e-7f;
	pc = pc * x2 + 2.48015873015873e-5f;
	pc = pc * x2 - 1.38888888888889e-3f;
	pc = pc * x2 + 4.16666666666667e-2f;
	pc = pc * x2 - 0.5f;
	pc = pc * x2 + 1.0f;
	return pc;
}

static float host_sin_poly(float x)
{
	if (fabsf(x) > 1.5f) return sinf(x);
	const float x2 = x * x;
	float ps = -2.50521083854417e-8f;
	ps = ps * x2 + 2.75573192239859e-6f;
	ps = ps * x2 - 1.98412698412698e-4f;
	ps = ps * x2 + 8.33333333333333e-3f;
	ps = ps * x2 - 1.66666666666667e-1f;
	ps = ps * x2 + 1.0f;
	return ps * x;
}

static v3 hv3(const float* p) { v3 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; return r; }

static void vehicle_record_from_desc(sgd_vehicle* v, const sgp_vehicle_desc* d)
{
	memset(v, 0, sizeof(*v));
	v->body = d->body; v->alive = 1; v->num_wheels = (int)d->num_wheels;
	for (int i = 0; i < v->num_wheels; ++i) {
		sgd_wheel* w = &v->wheels[i]; const sgp_wheel_desc* s = &d->wheels[i];
		w->position = hv3(s->position); w->suspension_dir = hv3(s->suspension_dir); w->steering_axis = hv3(s->steering_axis);
		w->wheel_up = hv3(s->wheel_up); w->wheel_forward = hv3(s->wheel_forward);
		w->sus_min = s->suspension_min_length; w->sus_max = s->suspension_max_length; w->sus_preload = s->suspension_preload;
		w->spring_freq = s->spring_frequency; w->spring_damp = s->spring_damping;
		w->radius = s->radius; w->width = s->width; w->inertia = s->inertia; w->ang_damping = s->angular_damping;
		w->max_steer = s->max_steer_angle; w->max_brake_torque = s->max_brake_torque; w->max_handbrake_torque = s->max_handbrake_torque;
		memcpy(w->long_fric, s->longitudinal_friction, sizeof(w->long_fric)); memcpy(w->lat_fric, s->lateral_friction, sizeof(w->lat_fric));
		w->suspension_length = w->sus_max; w->contact_body = SGP_INVALID_ID;
	}
	v->up = hv3(d->up); v->forward = hv3(d->forward);
	v->cast_radius = d->cast_radius;
	v->cos_max_slope = host_cos_poly(d->max_slope_angle);
	v->engine_max_torque = d->engine_max_torque; v->engine_min_rpm = d->engine_min_rpm; v->engine_max_rpm = d->engine_max_rpm;
	v->engine_inertia = d->engine_inertia; v->engine_ang_damping = d->engine_angular_damping;
	memcpy(v->engine_curve, d->engine_torque_curve, sizeof(v->engine_curve));
	v->engine_rpm = d->engine_min_rpm;
	v->num_gears = (int)d->num_gears; v->num_reverse_gears = (int)d->num_reverse_gears;
	memcpy(v->gear_ratios, d->gear_ratios, sizeof(v->gear_ratios)); memcpy(v->reverse_gear_ratios, d->reverse_gear_ratios, sizeof(v->reverse_gear_ratios));
	v->switch_time = d->switch_time; v->clutch_release_time = d->clutch_release_time; v->switch_latency = d->switch_latency;
	v->shift_up_rpm = d->shift_up_rpm; v->shift_down_rpm = d->shift_down_rpm; v->clutch_strength = d->clutch_strength;
	v->current_gear = 0; v->clutch_friction = 1.0f;
	v->num_differentials = (int)d->num_differentials;
	for (int k = 0; k < v->num_differentials; ++k) {
		const sgp_differential_desc* s = &d->differentials[k];
		v->differentials[k].left = s->left_wheel; v->differentials[k].right = s->right_wheel; v->differentials[k].ratio = s->differential_ratio;
		v->differentials[k].left_right_split = s->left_right_split; v->differentials[k].limited_slip_ratio = s->limited_slip_ratio;
		v->differentials[k].engine_torque_ratio = s->engine_torque_ratio;
	}
	v->differential_limited_slip_ratio = d->differential_limited_slip_ratio;
	v->num_anti_roll_bars = (int)d->num_anti_roll_bars;
	for (int k = 0; k < v->num_anti_roll_bars; ++k) {
		v->anti_roll_bars[k].left = d->anti_roll_bars[k].left_wheel; v->anti_roll_bars[k].right = d->anti_roll_bars[k].right_wheel;
		v->anti_roll_bars[k].stiffness = d->anti_roll_bars[k].stiffness;
	}
	v->is_motorcycle = d->controller_type == SGP_VEHICLE_CONTROLLER_MOTORCYCLE;
	v->lean_enabled = v->is_motorcycle; v->lean_steering_limit = d->lean_steering_limit != 0;
	v->max_lean_angle = d->max_lean_angle;
	v->tan_max_lean = host_sin_poly(d->max_lean_angle) / host_cos_poly(d->max_lean_angle);
	v->lean_spring_constant = d->lean_spring_constant; v->lean_spring_damping = d->lean_spring_damping;
	v->lean_integration_coefficient = d->lean_spring_integration_coefficient; v->lean_integration_decay = d->lean_spring_integration_decay;
	v->lean_smoothing = d->lean_smoothing_factor;
	v->target_lean.x = 0.0f; v->target_lean.y = 0.0f; v->target_lean.z = 1.0f;
}

static inline bool vehicle_live(const sgp_world* w, uint32_t id) { return w && id < w->n_vehicles && w->veh_alive[id]; }

SGP_API int sgp_vehicle_create(sgp_world* w, const sgp_vehicle_desc* d, uint32_t* id_out)
{
	if (!w || !d || !id_out) return fail(SGP_ERR_INVALID, "sgp_vehicle_create: NULL");
	if (!live(w, d->body) || (w->hb[d->body].flags & BF_MOTION_MASK) != SGP_MOTION_DYNAMIC) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_create: the chassis must be a live dynamic body");
	if (!vehicle_desc_valid(d)) return fail(SGP_ERR_INVALID, "sgp_vehicle_create: bad vehicle description");
	hipSetDevice(w->device);
	uint32_t id = w->n_vehicles;
	for (uint32_t k = 0; k < w->n_vehicles; ++k) if (!w->veh_alive[k]) { id = k; break; }     // lowest free slot
	if (id == w->n_vehicles) {
		if (w->n_vehicles == w->cap_vehicles) {
			// grow the device arrays (the step's captured graphs carry the old pointers)
			const uint32_t nc = w->cap_vehicles ? w->cap_vehicles * 2 : 64;
			sgd_vehicle* nv = nullptr; sgp_vehicle_input* ni = nullptr;
			HIP_TRY(hipMalloc((void**)&nv, sizeof(sgd_vehicle) * nc));
			HIP_TRY(hipMalloc((void**)&ni, sizeof(sgp_vehicle_input) * nc));
			HIP_TRY(hipMemsetAsync(nv, 0, sizeof(sgd_vehicle) * nc, w->stream));
			HIP_TRY(hipMemsetAsync(ni, 0, sizeof(sgp_vehicle_input) * nc, w->stream));
			if (w->n_vehicles) HIP_TRY(hipMemcpyAsync(nv, w->d_vehicles, sizeof(sgd_vehicle) * w->n_vehicles, hipMemcpyDeviceToDevice, w->stream));
			HIP_TRY(hipStreamSynchronize(w->stream));
			// (the row export of the solver passes is rebuilt by every step's controller kernel: nothing to carry over)
			const size_t row_bytes = sizeof(float4) * 16u * 4u * nc, head_bytes = sizeof(float4) * 5u * nc + sizeof(uint32_t) * (nc / 32u + 4u);      // (+ one bit per slot behind the heads: DV::veh_defer_bits)
			float4* nr = nullptr; float4* nh = nullptr;
			HIP_TRY(hipMalloc((void**)&nr, row_bytes));
			HIP_TRY(hipMalloc((void**)&nh, head_bytes));
			HIP_TRY(hipMemsetAsync(nr, 0, row_bytes, w->stream));
			HIP_TRY(hipMemsetAsync(nh, 0, head_bytes, w->stream));
			HIP_TRY(hipStreamSynchronize(w->stream));
			const size_t per_vehicle = sizeof(sgd_vehicle) + sizeof(sgp_vehicle_input) + sizeof(float4) * (16u * 4u + 5u);
			if (w->d_vehicles) { hipFree(w->d_vehicles); hipFree(w->d_veh_inputs); hipFree(w->d_veh_rows); hipFree(w->d_veh_head); w->device_bytes -= per_vehicle * w->cap_vehicles; }
			w->d_vehicles = nv; w->d_veh_inputs = ni; w->d_veh_rows = nr; w->d_veh_head = nh; w->cap_vehicles = nc;
			w->device_bytes += per_vehicle * nc;
			w->veh_inputs_dirty = true;
		}
		w->n_vehicles++;
		w->veh_alive.push_back(0); w->veh_body.push_back(SGP_INVALID_ID); w->veh_inputs.push_back(sgp_vehicle_input{ 0.0f, 0.0f, 0.0f, 0.0f });
	}
	sgd_vehicle rec;
	vehicle_record_from_desc(&rec, d);
	rec.gravity_len = sqrtf(w->dv.gx * w->dv.gx + w->dv.gy * w->dv.gy + w->dv.gz * w->dv.gz);
	HIP_TRY(hipMemcpyAsync(&w->d_vehicles[id], &rec, sizeof(rec), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));                  // `rec` lives on this stack frame
	w->veh_alive[id] = 1; w->veh_body[id] = d->body; w->veh_inputs[id] = sgp_vehicle_input{ 0.0f, 0.0f, 0.0f, 0.0f }; w->veh_inputs_dirty = true;
	w->dv.vehicles = w->d_vehicles; w->dv.vehicle_inputs = w->d_veh_inputs; w->dv.n_vehicles = w->n_vehicles;
	w->dv.veh_rows = w->d_veh_rows; w->dv.veh_head = w->d_veh_head; w->dv.veh_cap = w->cap_vehicles;
	w->dv.veh_defer_bits = (uint32_t*)(w->d_veh_head + 5u * (size_t)w->cap_vehicles);
	{ BodyCmd c = blank_cmd(d->body, CMD_SET_CHASSIS); c.flags = BF_CHASSIS; w->cmds.push_back(c); w->hb[d->body].flags |= BF_CHASSIS; }
	invalidate_graphs(w);
	w->dirty_since_step = true;
	*id_out = id;
	return SGP_OK;
}

SGP_API int sgp_vehicle_destroy(sgp_world* w, uint32_t id)
{
	if (!vehicle_live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_destroy: id not live");
	hipSetDevice(w->device);
	const int zero = 0;
	HIP_TRY(hipMemcpyAsync((char*)&w->d_vehicles[id] + offsetof(sgd_vehicle, alive), &zero, sizeof(int), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->veh_alive[id] = 0;
	// the chassis gets colour 0 back once no live vehicle sits on it
	const uint32_t body = w->veh_body[id];
	bool other = false;
	for (uint32_t k = 0; k < w->n_vehicles; ++k) if (w->veh_alive[k] && w->veh_body[k] == body) other = true;
	if (!other && live(w, body)) { BodyCmd c = blank_cmd(body, CMD_SET_CHASSIS); c.flags = 0; w->cmds.push_back(c); w->hb[body].flags &= ~BF_CHASSIS; }
	w->dirty_since_step = true;
	return SGP_OK;
}

SGP_API int sgp_vehicle_set_inputs(sgp_world* w, uint32_t first, uint32_t n, const sgp_vehicle_input* in)
{
	if (!w || (!in && n)) return fail(SGP_ERR_INVALID, "sgp_vehicle_set_inputs: NULL");
	for (uint32_t k = 0; k < n; ++k) if (!vehicle_live(w, first + k)) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_set_inputs: id not live");
	auto cl = [](float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); };
	for (uint32_t k = 0; k < n; ++k) {
		sgp_vehicle_input c = { cl(in[k].forward, -1.0f, 1.0f), cl(in[k].right, -1.0f, 1.0f), cl(in[k].brake, 0.0f, 1.0f), cl(in[k].hand_brake, 0.0f, 1.0f) };
		w->veh_inputs[first + k] = c;
		// "On user input, assure that the car is active" (CarPhysics.cpp:362-363)
		if ((c.forward != 0.0f || c.right != 0.0f || c.brake != 0.0f || c.hand_brake != 0.0f) && live(w, w->veh_body[first + k])) w->cmds.push_back(blank_cmd(w->veh_body[first + k], CMD_ACTIVATE));
	}
	w->veh_inputs_dirty = true;
	return SGP_OK;
}
SGP_API int sgp_vehicle_set_input(sgp_world* w, uint32_t id, const sgp_vehicle_input* in) { return sgp_vehicle_set_inputs(w, id, 1, in); }

static void hvec_out(float* o, v3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

SGP_API int sgp_vehicle_get_states(sgp_world* w, uint32_t first, uint32_t n, sgp_vehicle_state* out)
{
	if (!w || (!out && n)) return fail(SGP_ERR_INVALID, "sgp_vehicle_get_states: NULL");
	for (uint32_t k = 0; k < n; ++k) if (!vehicle_live(w, first + k)) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_get_states: id not live");
	if (!n) return SGP_OK;
	hipSetDevice(w->device);
	std::vector<sgd_vehicle> recs(n);
	HIP_TRY(hipMemcpyAsync(recs.data(), &w->d_vehicles[first], sizeof(sgd_vehicle) * n, hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	for (uint32_t k = 0; k < n; ++k) {
		const sgd_vehicle* v = &recs[k];
		sgp_vehicle_state* s = &out[k];
		memset(s, 0, sizeof(*s));
		for (int i = 0; i < v->num_wheels; ++i) {
			const sgd_wheel* wh = &v->wheels[i]; sgp_wheel_state* ws = &s->wheels[i];
			ws->suspension_length = wh->suspension_length; ws->steer_angle = wh->steer_angle; ws->rotation_angle = wh->angle; ws->angular_velocity = wh->angular_velocity;
			ws->has_contact = wh->has_contact; ws->contact_body = wh->has_contact ? wh->contact_body : SGP_INVALID_ID;
			if (wh->has_contact) {
				hvec_out(ws->contact_position, wh->contact_pos); hvec_out(ws->contact_normal, wh->contact_normal);
				hvec_out(ws->contact_longitudinal, wh->contact_long); hvec_out(ws->contact_lateral, wh->contact_lat); hvec_out(ws->contact_point_velocity, wh->contact_point_vel);
			}
			ws->suspension_lambda = wh->suspension.lambda + wh->max_up.lambda; ws->longitudinal_lambda = wh->longitudinal.lambda; ws->lateral_lambda = wh->lateral.lambda;
			ws->longitudinal_slip = wh->long_slip; ws->lateral_slip = wh->lat_slip;
		}
		s->engine_rpm = v->engine_rpm; s->current_gear = v->current_gear; s->clutch_friction = v->clutch_friction; s->active = v->active;
	}
	return SGP_OK;
}
SGP_API int sgp_vehicle_get_state(sgp_world* w, uint32_t id, sgp_vehicle_state* out) { return sgp_vehicle_get_states(w, id, 1, out); }

SGP_API int sgp_vehicle_enable_lean_controller(sgp_world* w, uint32_t id, int enabled)
{
	if (!vehicle_live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_enable_lean_controller: id not live");
	hipSetDevice(w->device);
	sgd_vehicle rec;
	HIP_TRY(hipMemcpyAsync(&rec, &w->d_vehicles[id], sizeof(rec), hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	const int on = (rec.is_motorcycle && enabled) ? 1 : 0;
	if (on != rec.lean_enabled) {
		HIP_TRY(hipMemcpyAsync((char*)&w->d_vehicles[id] + offsetof(sgd_vehicle, lean_enabled), &on, sizeof(int), hipMemcpyHostToDevice, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	return SGP_OK;
}

SGP_API int sgp_vehicle_reset_drivetrain(sgp_world* w, uint32_t id, float rpm, float wheel_w)
{
	if (!vehicle_live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_reset_drivetrain: id not live");
	hipSetDevice(w->device);
	sgd_vehicle rec;
	HIP_TRY(hipMemcpyAsync(&rec, &w->d_vehicles[id], sizeof(rec), hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	rec.engine_rpm = rpm;
	for (int i = 0; i < rec.num_wheels; ++i) rec.wheels[i].angular_velocity = wheel_w;
	HIP_TRY(hipMemcpyAsync(&w->d_vehicles[id], &rec, sizeof(rec), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	return SGP_OK;
}

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

// ---------------------------------------------------------------------------------------------------------------
// ray queries, PhysicsWorld.cpp:1668-1725

SGP_API int sgp_raycast(sgp_world* w, const sgp_ray* rays, uint32_t n, sgp_hit* hits)
{
	if (!w || (!rays && n) || (!hits && n)) return fail(SGP_ERR_INVALID, "sgp_raycast: NULL");
	hipSetDevice(w->device);
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	if (!n) return SGP_OK;
	if (!w->grid_valid && w->high) {
		// poses changed since the grid was built (a step integrates after its broad phase; edits move bodies): re-bin
		const DV& d = w->dv; hipStream_t s = w->stream; const uint32_t nb = w->high;
		launch_step_begin(d, *w->h_sp, nb, false, s); w->sp_uploaded = *w->h_sp; w->sp_uploaded_valid = true;
		launch_bp_bounds(d, nb, s); launch_bp_cell(d, nb, s); launch_bp_scan(d, s); launch_bp_scatter(d, nb, s);
		w->grid_valid = true;
	}
	const size_t rb = (sizeof(sgp_ray) * n + 15) & ~size_t(15);
	{ int r = ensure_stage(w, rb + sizeof(sgp_hit) * n); if (r != SGP_OK) return r; }
	memcpy(w->stage_host, rays, sizeof(sgp_ray) * n);
	if (n <= 64) {
		// a handful of rays (the facade's traceRay is n = 1): the kernel reads them from, and writes the hits to, the pinned host buffer
		// directly -- one launch and one sync instead of two copies around it
		launch_raycast(w->dv, (const sgp_ray*)w->stage_host, n, (sgp_hit*)((char*)w->stage_host + rb), w->stream);
		HIP_TRY(hipStreamSynchronize(w->stream));
	} else {
		HIP_TRY(hipMemcpyAsync(w->stage_dev, w->stage_host, sizeof(sgp_ray) * n, hipMemcpyHostToDevice, w->stream));
		sgp_hit* dh = (sgp_hit*)((char*)w->stage_dev + rb);
		launch_raycast(w->dv, (const sgp_ray*)w->stage_dev, n, dh, w->stream);
		HIP_TRY(hipMemcpyAsync((char*)w->stage_host + rb, dh, sizeof(sgp_hit) * n, hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	memcpy(hits, (char*)w->stage_host + rb, sizeof(sgp_hit) * n);
	for (uint32_t k = 0; k < n; ++k) {
		hits[k].userdata = hits[k].id != SGP_INVALID_ID ? w->hb[hits[k].id].userdata : 0;
		hits[k].sub_shape = 0;
		if (hits[k].id != SGP_INVALID_ID) hits[k].id = compound_id_of(w, hits[k].id, &hits[k].sub_shape);
	}
	return SGP_OK;
}

static int ensure_query_grid(sgp_world* w)
{
	if (!w->grid_valid && w->high) {
		// poses changed since the grid was built (a step integrates after its broad phase; edits move bodies): re-bin
		const DV& d = w->dv; hipStream_t s = w->stream; const uint32_t nb = w->high;
		launch_step_begin(d, *w->h_sp, nb, false, s); w->sp_uploaded = *w->h_sp; w->sp_uploaded_valid = true;
		launch_bp_bounds(d, nb, s); launch_bp_cell(d, nb, s); launch_bp_scan(d, s); launch_bp_scatter(d, nb, s);
		w->grid_valid = true;
	}
	return SGP_OK;
}

// CharacterVirtual's CollideShape (PlayerPhysics.cpp:258-353): contacts of capsules with everything within max_separation
SGP_API int sgp_collide_capsules(sgp_world* w, const sgp_capsule_query* qs, uint32_t n, sgp_query_contact* out, uint32_t cap, uint32_t* n_out)
{
	if (!w || (!qs && n) || (!out && cap) || !n_out) return fail(SGP_ERR_INVALID, "sgp_collide_capsules: NULL");
	hipSetDevice(w->device);
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	*n_out = 0;
	if (!n) return SGP_OK;
	ensure_query_grid(w);
	const size_t qb = (sizeof(sgp_capsule_query) * n + 15) & ~size_t(15);
	const size_t ob = sizeof(sgp_query_contact) * std::max(cap, 1u);
	{ int r = ensure_stage(w, qb + ob + 16); if (r != SGP_OK) return r; }
	memcpy(w->stage_host, qs, sizeof(sgp_capsule_query) * n);
	HIP_TRY(hipMemcpyAsync(w->stage_dev, w->stage_host, sizeof(sgp_capsule_query) * n, hipMemcpyHostToDevice, w->stream));
	sgp_query_contact* dout = (sgp_query_contact*)((char*)w->stage_dev + qb);
	uint32_t* dcount = (uint32_t*)((char*)w->stage_dev + qb + ob);
	HIP_TRY(hipMemsetAsync(dcount, 0, sizeof(uint32_t), w->stream));
	launch_collide_capsules(w->dv, (const sgp_capsule_query*)w->stage_dev, n, dout, cap, dcount, w->stream);
	HIP_TRY(hipMemcpyAsync((char*)w->stage_host + qb, dout, ob + 16, hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	const uint32_t cnt = *(const uint32_t*)((char*)w->stage_host + qb + ob);
	const uint32_t m = std::min(cnt, cap);
	sgp_query_contact* h = (sgp_query_contact*)((char*)w->stage_host + qb);
	std::sort(h, h + m, [](const sgp_query_contact& a, const sgp_query_contact& b) {
		if (a.query != b.query) return a.query < b.query;
		if (a.body != b.body) return a.body < b.body;
		return a.sub_shape < b.sub_shape; });         // (the kernel leaves the contact's point index in this field)
	for (uint32_t i = 0; i < m; ++i) { h[i].userdata = w->hb[h[i].body].userdata; h[i].body = compound_id_of(w, h[i].body, &h[i].sub_shape); }
	memcpy(out, h, sizeof(sgp_query_contact) * m);
	*n_out = cnt;
	return SGP_OK;
}

SGP_API int sgp_spherecast(sgp_world* w, const sgp_ray* rays, const float* radii, uint32_t n, sgp_hit* hits)
{
	if (!w || (n && (!rays || !radii || !hits))) return fail(SGP_ERR_INVALID, "sgp_spherecast: NULL");
	hipSetDevice(w->device);
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	if (!n) return SGP_OK;
	ensure_query_grid(w);
	const size_t rb = (sizeof(sgp_ray) * n + 15) & ~size_t(15), fb = (sizeof(float) * n + 15) & ~size_t(15);
	{ int r = ensure_stage(w, rb + fb + sizeof(sgp_hit) * n); if (r != SGP_OK) return r; }
	memcpy(w->stage_host, rays, sizeof(sgp_ray) * n);
	memcpy((char*)w->stage_host + rb, radii, sizeof(float) * n);
	HIP_TRY(hipMemcpyAsync(w->stage_dev, w->stage_host, rb + sizeof(float) * n, hipMemcpyHostToDevice, w->stream));
	sgp_hit* dh = (sgp_hit*)((char*)w->stage_dev + rb + fb);
	launch_spherecast(w->dv, (const sgp_ray*)w->stage_dev, (const float*)((char*)w->stage_dev + rb), n, dh, w->stream);
	HIP_TRY(hipMemcpyAsync((char*)w->stage_host + rb + fb, dh, sizeof(sgp_hit) * n, hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	memcpy(hits, (char*)w->stage_host + rb + fb, sizeof(sgp_hit) * n);
	for (uint32_t k = 0; k < n; ++k) {
		hits[k].userdata = hits[k].id != SGP_INVALID_ID ? w->hb[hits[k].id].userdata : 0;
		hits[k].sub_shape = 0;
		if (hits[k].id != SGP_INVALID_ID) hits[k].id = compound_id_of(w, hits[k].id, &hits[k].sub_shape);
	}
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// multi-GPU tiles

SGP_API int sgp_world_export_boundary(sgp_world* w, const float lo[3], const float hi[3], float margin, sgp_ghost_record* out, uint32_t cap, uint32_t* n_out)
{
	if (!w || !n_out) return fail(SGP_ERR_INVALID, "sgp_world_export_boundary: NULL");
	hipSetDevice(w->device);
	static const bool timing = getenv("SGP_TIMING") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	const uint32_t lim = std::min(cap, w->dv.cap_bodies);
	{ int r = ensure_stage(w, sizeof(sgp_ghost_record) * std::max(lim, 1u)); if (r != SGP_OK) return r; }
	launch_export_boundary(w->dv, w->high, make_float3(lo[0], lo[1], lo[2]), make_float3(hi[0], hi[1], hi[2]), margin,
	                       (sgp_ghost_record*)w->stage_dev, lim, &w->dv.ctr->n_export, w->stream);
	// one sync in the common case: the counters and as many records as the previous call produced (+ 25 %) come back together
	uint32_t guess = out ? std::min(lim, w->last_export + w->last_export / 4 + 64u) : 0u;
	if (guess) HIP_TRY(hipMemcpyAsync(w->stage_host, w->stage_dev, sizeof(sgp_ghost_record) * guess, hipMemcpyDeviceToHost, w->stream));
	{ int r = read_counters(w); if (r != SGP_OK) return r; }
	const uint32_t n = w->h_ctr->n_export, m = std::min(n, lim);
	w->last_export = n;
	const auto t1 = std::chrono::steady_clock::now();
	if (m && out) {
		if (m > guess) {
			HIP_TRY(hipMemcpyAsync((char*)w->stage_host + sizeof(sgp_ghost_record) * guess, (char*)w->stage_dev + sizeof(sgp_ghost_record) * guess,
			                       sizeof(sgp_ghost_record) * (m - guess), hipMemcpyDeviceToHost, w->stream));
			HIP_TRY(hipStreamSynchronize(w->stream));
		}
		// the kernel wrote the records in ascending body id (k_export_count + k_export_boundary): the order of the exchange is deterministic
		memcpy(out, w->stage_host, sizeof(sgp_ghost_record) * m);
	}
	if (timing) { const auto t2 = std::chrono::steady_clock::now(); fprintf(stderr, "[sgp timing] export_boundary: device part %.1f us, sort + copy of %u records %.1f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count(), m, std::chrono::duration<double, std::micro>(t2 - t1).count()); }
	*n_out = n;
	return SGP_OK;
}

// Where the poses of an import come from when the records are already on the device (sgp_tiles_*): the device copy of the records and a
// device array for the local body id of every record (grown here); surviving ghosts are then refreshed by ONE kernel, not by commands.
struct GhostDeviceSource { const sgp_ghost_record* d_recs; uint32_t** d_ids; uint32_t* cap_ids; uint64_t* ids_version; };

// (skip: per RECORD, 1 = not a ghost here (an immigrant of the same exchange); the id array stays aligned with the records, such entries hold "no body")
static int upload_ghost_ids(sgp_world* w, const GhostDeviceSource* dev, const uint8_t* skip = nullptr, uint32_t n_records = 0)
{
	const uint32_t n = skip ? n_records : (uint32_t)w->ghost_seq.size();
	std::vector<uint32_t> ids(n);
	if (skip) { size_t g = 0; for (uint32_t k = 0; k < n; ++k) ids[k] = skip[k] ? SGP_INVALID_ID : w->ghost_seq[g++].second; }
	else for (uint32_t k = 0; k < n; ++k) ids[k] = w->ghost_seq[k].second;
	if (n > *dev->cap_ids) {
		if (*dev->d_ids) { HIP_TRY(hipStreamSynchronize(w->stream)); hipFree(*dev->d_ids); }
		*dev->cap_ids = n + n / 2 + 1024;
		HIP_TRY(hipMalloc((void**)dev->d_ids, sizeof(uint32_t) * (size_t)*dev->cap_ids));
	}
	if (n) { HIP_TRY(hipMemcpyAsync(*dev->d_ids, ids.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, w->stream)); HIP_TRY(hipStreamSynchronize(w->stream)); }      // (`ids` is pageable memory going out of scope)
	*dev->ids_version = skip ? ~0ull : w->ghost_seq_version;      // (an array with holes serves this import only: the next one that finds the set unchanged uploads the plain list)
	return SGP_OK;
}

static int make_ghost(sgp_world* w, const sgp_ghost_record& r, uint32_t* id_out)
{
	sgp_body_desc d; sgp_default_body_desc(&d);
	memcpy(d.pos, r.pos, 12); memcpy(d.rot, r.rot, 16); memcpy(d.lin_vel, r.lin_vel, 12); memcpy(d.ang_vel, r.ang_vel, 12);
	d.shape_type = r.shape_type; memcpy(d.shape, r.shape, 16);
	d.motion_type = SGP_MOTION_KINEMATIC;      // velocity driven, infinite mass for this tile's solve
	// layer and sensor flag of the original: a sensor or a non-collidable body near the border must not become a solid obstacle next door
	d.layer = (int32_t)(r.flags & SGP_GHOST_FLAG_LAYER_MASK);
	if (d.layer == SGP_LAYER_NON_MOVING) d.layer = SGP_LAYER_MOVING;                               // (a kinematic ghost lives on a moving layer)
	if (d.layer == SGP_LAYER_NON_MOVING_NON_COLLIDABLE) d.layer = SGP_LAYER_MOVING_NON_COLLIDABLE;
	d.is_sensor = (r.flags & SGP_GHOST_FLAG_SENSOR) ? 1 : 0;
	d.mass = r.mass; d.friction = r.friction; d.restitution = r.restitution;
	d.activate = 1; d.userdata = r.userdata;       // a ray or an event that meets the ghost names the object, like its owner would
	*id_out = SGP_INVALID_ID;
	return add_one(w, &d, id_out, true);
}

static int import_ghosts_impl(sgp_world* w, const sgp_ghost_record* in_all, uint32_t n_all, const GhostDeviceSource* dev, const uint8_t* skip = nullptr, const uint64_t* gids = nullptr, uint32_t gid_stride = 0);
// (skip[k] = 1: record k is no ghost -- an immigrant riding in the same exchange -- and is left out; with a device source the records stay where they are and
// the id array has a hole there)
struct GhostView {      // the ghost records of an import: all of them, or those a mask lets through (by index: nothing is copied)
	const sgp_ghost_record* base; const uint32_t* idx; uint32_t n;
	const uint64_t* gids; uint32_t gid_stride;      // the records' global ids packed (16-byte keys of the exchange), or NULL: the diff then walks 16 bytes per record, not 128
	const sgp_ghost_record& operator[](uint32_t k) const { return idx ? base[idx[k]] : base[k]; }
	uint64_t gid(uint32_t k) const { const uint32_t r = idx ? idx[k] : k; return gids ? gids[(size_t)r * gid_stride] : base[r].global_id; }
};
static int import_ghosts_view(sgp_world* w, const GhostView& in, uint32_t n, const GhostDeviceSource* dev, const uint8_t* skip, uint32_t n_all);
static int import_ghosts_impl(sgp_world* w, const sgp_ghost_record* in_all, uint32_t n_all, const GhostDeviceSource* dev, const uint8_t* skip, const uint64_t* gids, uint32_t gid_stride)
{
	if (!skip) { GhostView v = { in_all, nullptr, n_all, gids, gid_stride }; return import_ghosts_view(w, v, n_all, dev, nullptr, n_all); }
	std::vector<uint32_t> idx; idx.reserve(n_all);
	for (uint32_t k = 0; k < n_all; ++k) if (!skip[k]) idx.push_back(k);
	GhostView v = { in_all, idx.data(), (uint32_t)idx.size(), gids, gid_stride };
	return import_ghosts_view(w, v, v.n, dev, skip, n_all);
}
static int import_ghosts_view(sgp_world* w, const GhostView& in, uint32_t n, const GhostDeviceSource* dev, const uint8_t* skip, uint32_t n_all)
{
	// ghosts keep their local id while they stay in the set, so the contact cache (keyed by body ids) keeps warm-starting.
	// ghost_seq: (global id, local id) of the previous import, in its order
	// 1. the usual case: the same ghosts as in the previous import, in the same order -- no bookkeeping, just refresh their poses
	if (n == w->ghost_seq.size() && n > 0) {
		bool same = true;
		for (uint32_t k = 0; k < n && same; ++k) same = in.gid(k) == w->ghost_seq[k].first && live(w, w->ghost_seq[k].second);
		if (same) {
			if (dev) {
				{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
				if (skip || *dev->ids_version != w->ghost_seq_version) { int r = upload_ghost_ids(w, dev, skip, n_all); if (r != SGP_OK) return r; }      // (the set was last changed by an import that did not come through here; or the records hold immigrants between the ghosts)
				launch_ghost_refresh_records(w->dv, dev->d_recs, *dev->d_ids, n_all, w->stream);
				w->grid_valid = false; w->dirty_since_step = true;
				return SGP_OK;
			}
			// a later import before the next flush supersedes an earlier one: the refresh list holds one record per ghost
			w->ghost_refresh.resize(n);
			for (uint32_t k = 0; k < n; ++k) {
				GhostRefresh& c = w->ghost_refresh[k];
				c.id = w->ghost_seq[k].second;
				memcpy(c.pos, in[k].pos, 12); memcpy(c.rot, in[k].rot, 16); memcpy(c.linv, in[k].lin_vel, 12); memcpy(c.angv, in[k].ang_vel, 12);
			}
			return SGP_OK;
		}
	}
	w->ghost_refresh.clear();
	w->cmds.reserve(w->cmds.size() + n);
	std::vector<std::pair<uint64_t, uint32_t>> seq(n, std::pair<uint64_t, uint32_t>(0, SGP_INVALID_ID));
	std::vector<uint32_t> gone;
	auto refresh_cmd = [&](uint32_t id, const sgp_ghost_record& r) {
		if (dev) return;                       // refreshed from the device copy of the records below
		BodyCmd c = blank_cmd(id, CMD_SET_POS | CMD_SET_ROT | CMD_SET_VEL | CMD_ACTIVATE);
		memcpy(c.pos, r.pos, 12); memcpy(c.rot, r.rot, 16); memcpy(c.linv, r.lin_vel, 12); memcpy(c.angv, r.ang_vel, 12);
		w->cmds.push_back(c);
	};
	// 2. both the old and the new sequence ascending in global id (what every exchange produces: by source rank, then by the source's body
	//    id): a two-pointer diff finds who stayed, who is new and who left, without hashing.  New ghosts take their slots in record order,
	//    leavers are removed afterwards in ascending id order -- the same allocation order as the general path below.
	bool ascending = true;
	for (uint32_t k = 1; k < n && ascending; ++k) ascending = in.gid(k - 1) < in.gid(k);
	for (size_t k = 1; k < w->ghost_seq.size() && ascending; ++k) ascending = w->ghost_seq[k - 1].first < w->ghost_seq[k].first;
	if (ascending) {
		const std::vector<std::pair<uint64_t, uint32_t>>& old = w->ghost_seq;
		size_t i = 0, j = 0;
		while (i < n || j < old.size()) {
			if (j == old.size() || (i < n && in.gid(i) < old[j].first)) {
				uint32_t id; const int r = make_ghost(w, in[i], &id);
				if (r != SGP_OK && r != SGP_ERR_REJECTED) return r;
				seq[i] = std::make_pair(in.gid(i), r == SGP_OK ? id : SGP_INVALID_ID); ++i;
			} else if (i == n || old[j].first < in.gid(i)) {
				if (old[j].second != SGP_INVALID_ID && live(w, old[j].second)) gone.push_back(old[j].second);
				++j;
			} else {
				uint32_t id = old[j].second;
				if (id != SGP_INVALID_ID && live(w, id)) refresh_cmd(id, in[i]);
				else { const int r = make_ghost(w, in[i], &id); if (r != SGP_OK && r != SGP_ERR_REJECTED) return r; if (r != SGP_OK) id = SGP_INVALID_ID; }
				seq[i] = std::make_pair(in.gid(i), id); ++i; ++j;
			}
		}
		w->ghost_map_stale = true;
	} else {
		// 3. general: hash map global id -> (generation of the last import that contained it, local id)
		if (w->ghost_map_stale) {
			w->ghost_map.clear();
			for (const auto& e : w->ghost_seq) if (e.second != SGP_INVALID_ID) w->ghost_map[e.first] = ((uint64_t)w->ghost_gen << 32) | e.second;
			w->ghost_map_stale = false;
		}
		const uint32_t gen = ++w->ghost_gen;
		for (uint32_t k = 0; k < n; ++k) {
			seq[k].first = in.gid(k);
			auto it = w->ghost_map.find(in.gid(k));
			if (it != w->ghost_map.end() && live(w, (uint32_t)it->second)) {
				const uint32_t id = (uint32_t)it->second;
				refresh_cmd(id, in[k]);
				it->second = ((uint64_t)gen << 32) | id;
				seq[k].second = id;
				continue;
			}
			uint32_t id; const int r = make_ghost(w, in[k], &id);
			if (r == SGP_OK) { w->ghost_map[in.gid(k)] = ((uint64_t)gen << 32) | id; seq[k].second = id; }
			else if (r != SGP_ERR_REJECTED) return r;
		}
		// whatever was not refreshed by this import left the ghost set
		for (auto it = w->ghost_map.begin(); it != w->ghost_map.end();) {
			if ((uint32_t)(it->second >> 32) != gen) { if (live(w, (uint32_t)it->second)) gone.push_back((uint32_t)it->second); it = w->ghost_map.erase(it); }
			else ++it;
		}
	}
	// leavers: removed in ascending id order (deterministic free-list order)
	std::sort(gone.begin(), gone.end());
	for (uint32_t id : gone) sgp_body_remove(w, id);
	w->ghost_seq.swap(seq);
	w->ghost_seq_version++;
	if (dev && n) {
		// new ghosts and removals reach the device first, then ONE kernel gives every ghost of the set its pose from the received records (a rejected
		// record -- non-finite pose ... -- has "no body" in the id array, like an immigrant's: the kernel passes over it)
		{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
		{ int r = upload_ghost_ids(w, dev, skip, n_all); if (r != SGP_OK) return r; }
		launch_ghost_refresh_records(w->dv, dev->d_recs, *dev->d_ids, n_all, w->stream);
		w->grid_valid = false; w->dirty_since_step = true;
	}
	return SGP_OK;
}

SGP_API int sgp_world_import_ghosts(sgp_world* w, const sgp_ghost_record* in, uint32_t n)
{
	if (!w || (!in && n)) return fail(SGP_ERR_INVALID, "sgp_world_import_ghosts: NULL");
	hipSetDevice(w->device);
	return import_ghosts_impl(w, in, n, nullptr);
}

// ---- host-side routing of exported records (tiles.py) -------------------------------------------------------------------------

static inline bool in_box(const float* p, const float* lo, const float* hi, float pad)
{
	return p[0] >= lo[0] - pad && p[0] < hi[0] + pad && p[1] >= lo[1] - pad && p[1] < hi[1] + pad && p[2] >= lo[2] - pad && p[2] < hi[2] + pad;
}

SGP_API int sgp_tiles_route(const sgp_ghost_record* recs, uint32_t n, uint32_t my_rank, const float* boxes, uint32_t n_tiles, float pad,
                            sgp_ghost_record* send_out, uint32_t cap, uint32_t* send_counts,
                            uint32_t* emigrant_ids, uint32_t emigrant_cap, uint32_t* n_emigrants)
{
	if ((!recs && n) || !boxes || !send_counts || !n_emigrants || my_rank >= n_tiles) return fail(SGP_ERR_INVALID, "sgp_tiles_route: bad arguments");
	const float* mylo = boxes + 6 * (size_t)my_rank; const float* myhi = mylo + 3;
	// flags per record: emigrant?
	std::vector<uint8_t> emig(n, 0);
	uint32_t ne = 0;
	for (uint32_t k = 0; k < n; ++k) {
		bool taker = false;      // (same rule as route_mask on the device: without a tile that contains the centre the body stays where it is)
		for (uint32_t r = 0; r < n_tiles && !taker; ++r) if (r != my_rank) taker = in_box(recs[k].pos, boxes + 6 * (size_t)r, boxes + 6 * (size_t)r + 3, 0.0f);
		if (n_tiles > 1 && taker && (recs[k].motion_type & 0xFFu) == SGP_MOTION_DYNAMIC && !(recs[k].flags & SGP_GHOST_FLAG_CHASSIS) && !in_box(recs[k].pos, mylo, myhi, 0.0f)) {
			emig[k] = 1;
			if (ne < emigrant_cap && emigrant_ids) emigrant_ids[ne] = (uint32_t)(recs[k].global_id & 0xFFFFFFFFull);
			++ne;
		}
	}
	*n_emigrants = ne;
	if (ne > emigrant_cap) return fail(SGP_ERR_CAPACITY, "sgp_tiles_route: emigrant list too small");
	uint32_t w = 0;
	for (uint32_t r = 0; r < n_tiles; ++r) {
		send_counts[r] = 0;
		if (r == my_rank) continue;
		const float* lo = boxes + 6 * (size_t)r; const float* hi = lo + 3;
		for (uint32_t k = 0; k < n; ++k) {
			if (!in_box(recs[k].pos, lo, hi, pad)) continue;
			if (w >= cap || !send_out) return fail(SGP_ERR_CAPACITY, "sgp_tiles_route: send buffer too small");
			sgp_ghost_record o = recs[k];
			o.global_id |= (uint64_t)my_rank << 40;
			if (emig[k]) o.motion_type = SGP_MOTION_DYNAMIC | SGP_GHOST_TAKE_OWNERSHIP;
			send_out[w++] = o;
			++send_counts[r];
		}
	}
	return SGP_OK;
}

SGP_API int sgp_tiles_split(const sgp_ghost_record* in, uint32_t n, const float lo[3], const float hi[3],
                            sgp_ghost_record* ghosts_out, uint32_t* n_ghosts, sgp_ghost_record* immigrants_out, uint32_t* n_immigrants)
{
	if ((!in && n) || !lo || !hi || !n_ghosts || !n_immigrants || (n && (!ghosts_out || !immigrants_out))) return fail(SGP_ERR_INVALID, "sgp_tiles_split: bad arguments");
	uint32_t g = 0, m = 0;
	for (uint32_t k = 0; k < n; ++k) {
		if (in[k].motion_type & SGP_GHOST_TAKE_OWNERSHIP) { if (in_box(in[k].pos, lo, hi, 0.0f)) immigrants_out[m++] = in[k]; }
		else ghosts_out[g++] = in[k];
	}
	*n_ghosts = g; *n_immigrants = m;
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------

SGP_API int sgp_world_device_array(sgp_world* w, int which, void** dev_ptr_out, uint32_t* count_out)
{
	if (!w || !dev_ptr_out) return fail(SGP_ERR_INVALID, "sgp_world_device_array: NULL");
	void* p = nullptr;
	switch (which) { case 0: p = w->dv.pose; break; case 1: p = w->dv.vel; break;
	default: return fail(SGP_ERR_INVALID, "sgp_world_device_array: bad index"); }
	*dev_ptr_out = p;
	if (count_out) *count_out = w->high;
	return SGP_OK;
}

SGP_API int sgp_world_stream(sgp_world* w, void** stream_out)
{
	if (!w || !stream_out) return fail(SGP_ERR_INVALID, "sgp_world_stream: NULL");
	*stream_out = (void*)w->stream;
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// sgp_tiles_*: the per-step ghost exchange of the spatial tiles below the C ABI.
//
//   export + ROUTING on the device (k_route_count / scan / write: one record per (body, destination), segmented by destination)
//   -> per-destination counts all-gathered over RCCL (ncclAllGather, device buffers) and read back with ONE small copy (header +
//      counts matrix + emigrant ids)
//   -> the records travel device to device: grouped ncclSend / ncclRecv over xGMI straight out of the send buffer's segments
//      (or, for several tiles driven by one process, plain device-to-device copies)
//   -> the receiving tile refreshes its ghosts.  While the set of ghosts is what it was the step before (the steady state), a kernel
//      applies the poses straight from the received records; only when the set changed (or bodies immigrate) do the records come to
//      the host, which owns the body slots.
// RCCL is bound at run time (dlopen): libsgp.so carries no link-time dependency on it, a single-GPU user never loads it.
#include <dlfcn.h>

namespace {
typedef struct { char internal[128]; } sgp_nccl_unique_id;
typedef void* sgp_nccl_comm;
enum { SGP_NCCL_UINT8 = 1, SGP_NCCL_UINT32 = 3 };          // ncclDataType_t (rccl.h): ncclUint8 = 1, ncclUint32 = 3
struct RcclApi {
	void* lib = nullptr; bool tried = false;
	int (*GetUniqueId)(sgp_nccl_unique_id*) = nullptr;
	int (*CommInitRank)(sgp_nccl_comm*, int, sgp_nccl_unique_id, int) = nullptr;
	int (*CommDestroy)(sgp_nccl_comm) = nullptr;
	int (*AllGather)(const void*, void*, size_t, int, sgp_nccl_comm, hipStream_t) = nullptr;
	int (*Send)(const void*, size_t, int, int, sgp_nccl_comm, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, sgp_nccl_comm, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	int (*CommCount)(sgp_nccl_comm, int*) = nullptr;
};
RcclApi g_rccl;

// The prototypes above are hand-declared so that libsgp.so builds and loads without RCCL.  Where <rccl/rccl.h> is installed at build time
// they are checked against it: same number of parameters, every parameter and the result of the same size and kind (pointer / integer or
// enum / class passed by value), and the enumerators this file passes as integers.  A mismatch is a compile error, not a first-run surprise.
#if __has_include(<rccl/rccl.h>)
}
#include <rccl/rccl.h>
#include <type_traits>
namespace {
template <class A, class B> constexpr bool sgp_abi_same_arg()
{
	return sizeof(A) == sizeof(B) && std::is_pointer<A>::value == std::is_pointer<B>::value && std::is_class<A>::value == std::is_class<B>::value &&
	       (std::is_integral<A>::value || std::is_enum<A>::value) == (std::is_integral<B>::value || std::is_enum<B>::value);
}
template <class F, class G> struct sgp_abi_same : std::false_type {};
template <class R, class... A, class S, class... B> struct sgp_abi_same<R (*)(A...), S (*)(B...)>
{
	template <bool same_arity, class Dummy = void> struct args { static constexpr bool value = false; };
	template <class Dummy> struct args<true, Dummy> { static constexpr bool value = (sgp_abi_same_arg<A, B>() && ... && true); };
	static constexpr bool value = sgp_abi_same_arg<R, S>() && args<sizeof...(A) == sizeof...(B)>::value;
};
#define SGP_CHECK_RCCL(member, fn) static_assert(sgp_abi_same<decltype(RcclApi::member), decltype(&fn)>::value, "hand-declared prototype of " #fn " does not match <rccl/rccl.h>")
SGP_CHECK_RCCL(GetUniqueId, ncclGetUniqueId);
SGP_CHECK_RCCL(CommInitRank, ncclCommInitRank);
SGP_CHECK_RCCL(CommDestroy, ncclCommDestroy);
SGP_CHECK_RCCL(AllGather, ncclAllGather);
SGP_CHECK_RCCL(Send, ncclSend);
SGP_CHECK_RCCL(Recv, ncclRecv);
SGP_CHECK_RCCL(GroupStart, ncclGroupStart);
SGP_CHECK_RCCL(GroupEnd, ncclGroupEnd);
SGP_CHECK_RCCL(GetErrorString, ncclGetErrorString);
SGP_CHECK_RCCL(CommCount, ncclCommCount);
static_assert(sizeof(ncclUniqueId) == sizeof(sgp_nccl_unique_id) && NCCL_UNIQUE_ID_BYTES == SGP_TILES_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
static_assert((int)ncclUint8 == SGP_NCCL_UINT8 && (int)ncclUint32 == SGP_NCCL_UINT32 && (int)ncclSuccess == 0, "ncclDataType_t / ncclResult_t values");
#endif

bool rccl_load()
{
	if (g_rccl.tried) return g_rccl.lib != nullptr;
	g_rccl.tried = true;
	// the copy this process already has (PyTorch brings its own), else the system's
	const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
	for (const char* n : names) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (g_rccl.lib) break; }
	if (!g_rccl.lib) for (const char* n : names) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.lib) break; }
	if (!g_rccl.lib) return false;
	bool ok = true;
	auto sym = [&](const char* name) { void* p = dlsym(g_rccl.lib, name); if (!p) ok = false; return p; };
	g_rccl.GetUniqueId = (int (*)(sgp_nccl_unique_id*))sym("ncclGetUniqueId");
	g_rccl.CommInitRank = (int (*)(sgp_nccl_comm*, int, sgp_nccl_unique_id, int))sym("ncclCommInitRank");
	g_rccl.CommDestroy = (int (*)(sgp_nccl_comm))sym("ncclCommDestroy");
	g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, sgp_nccl_comm, hipStream_t))sym("ncclAllGather");
	g_rccl.Send = (int (*)(const void*, size_t, int, int, sgp_nccl_comm, hipStream_t))sym("ncclSend");
	g_rccl.Recv = (int (*)(void*, size_t, int, int, sgp_nccl_comm, hipStream_t))sym("ncclRecv");
	g_rccl.GroupStart = (int (*)())sym("ncclGroupStart");
	g_rccl.GroupEnd = (int (*)())sym("ncclGroupEnd");
	g_rccl.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
	if (!ok) { g_rccl.lib = nullptr; return false; }
	g_rccl.CommCount = (int (*)(sgp_nccl_comm, int*))dlsym(g_rccl.lib, "ncclCommCount");      // optional: only reported in sgp_tiles_stats
	return true;
}
int rccl_fail(const char* what, int rc)
{
	g_last_error = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
	return SGP_ERR_HIP;
}
#define RCCL_TRY(call, what) do { const int rc_ = (call); if (rc_ != 0) return rccl_fail(what, rc_); } while (0)
}

#define SGP_TILES_EMIG_INLINE 512          // emigrant ids that come back with the header copy

struct sgp_tiles {
	sgp_world* w = nullptr;
	uint32_t rank = 0, n_tiles = 1;
	TileRoute route;
	sgp_nccl_comm comm = nullptr;
	// device
	uint32_t* d_block_counts = nullptr; uint32_t* d_block_offsets = nullptr; uint32_t cap_blocks = 0;
	char* d_ctl = nullptr;                 // [RouteHeader][counts matrix n_tiles x SGP_MAX_TILES... see ctl_bytes][emigrant ids]
	sgp_ghost_record* d_send = nullptr; uint32_t cap_send = 0;
	sgp_ghost_record* d_recv = nullptr; uint32_t cap_recv = 0;
	uint32_t* d_emig = nullptr; uint32_t cap_emig = 0;
	uint32_t* d_seq_ids = nullptr; uint32_t cap_seq = 0; uint64_t ids_version = 0;      // local body id of ghost k of the current ghost set (device copy, for the refresh kernel)
	// host (pinned)
	char* h_ctl = nullptr; sgp_ghost_record* h_recv = nullptr; uint32_t cap_h_recv = 0;
	uint4* d_keys = nullptr; uint32_t cap_keys = 0; void* h_keys = nullptr; uint32_t cap_h_keys = 0;      // (global id, ownership flag) of the received records
	// last exchange
	std::vector<uint32_t> recv_counts, recv_offsets;
	std::vector<uint64_t> seq_gids;        // global ids of the ghosts of the previous import, in order
	bool seq_valid = false;
	sgp_tiles_stats stats;
	std::vector<sgp_migration> migrations;
	// re-tiling (sgp_tiles_rebalance): this tile's histogram, everybody's (RCCL all-gather), the pinned host copy
	uint32_t* d_hist = nullptr; uint32_t* d_hist_all = nullptr; uint32_t* h_hist = nullptr;
};
static size_t tiles_matrix_off() { return sizeof(RouteHeader); }
static size_t tiles_emig_off(uint32_t n_tiles) { return sizeof(RouteHeader) + sizeof(uint32_t) * (size_t)n_tiles * n_tiles; }
static size_t tiles_ctl_bytes(uint32_t n_tiles) { return tiles_emig_off(n_tiles) + sizeof(uint32_t) * SGP_TILES_EMIG_INLINE; }

SGP_API int sgp_tiles_unique_id(uint8_t out[SGP_TILES_UNIQUE_ID_BYTES])
{
	if (!out) return fail(SGP_ERR_INVALID, "sgp_tiles_unique_id: NULL");
	if (!rccl_load()) return fail(SGP_ERR_HIP, "sgp_tiles_unique_id: RCCL (librccl.so) not found");
	sgp_nccl_unique_id id;
	RCCL_TRY(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
	static_assert(sizeof(id) == SGP_TILES_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
	memcpy(out, &id, sizeof(id));
	return SGP_OK;
}

template <typename T> static int tiles_grow(sgp_world* w, T*& p, uint32_t& cap, uint32_t need, bool keep = false)
{
	if (need <= cap) return SGP_OK;
	const uint32_t nc = std::max(need + need / 2, 4096u);
	T* q = nullptr;
	HIP_TRY(hipMalloc((void**)&q, sizeof(T) * (size_t)nc));
	if (p) { HIP_TRY(hipStreamSynchronize(w->stream)); if (keep && cap) HIP_TRY(hipMemcpy(q, p, sizeof(T) * (size_t)cap, hipMemcpyDeviceToDevice)); hipFree(p); }
	p = q; cap = nc;
	return SGP_OK;
}

SGP_API int sgp_tiles_destroy(sgp_tiles* t)
{
	if (!t) return SGP_OK;
	if (t->w) { hipSetDevice(t->w->device); hipStreamSynchronize(t->w->stream); }
	if (t->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(t->comm);
	hipFree(t->d_block_counts); hipFree(t->d_block_offsets); hipFree(t->d_ctl); hipFree(t->d_send); hipFree(t->d_recv); hipFree(t->d_emig); hipFree(t->d_seq_ids);
	if (t->h_ctl) hipHostFree(t->h_ctl);
	if (t->h_recv) hipHostFree(t->h_recv);
	if (t->h_keys) hipHostFree(t->h_keys);
	hipFree(t->d_keys);
	hipFree(t->d_hist); hipFree(t->d_hist_all); if (t->h_hist) hipHostFree(t->h_hist);
	delete t;
	return SGP_OK;
}

SGP_API int sgp_tiles_create(sgp_world* w, uint32_t rank, uint32_t n_tiles, const float* boxes, float margin, float radius_pad, const uint8_t* unique_id, sgp_tiles** out)
{
	if (!w || !boxes || !out || n_tiles < 1 || n_tiles > SGP_MAX_TILES || rank >= n_tiles) return fail(SGP_ERR_INVALID, "sgp_tiles_create: bad arguments (1..64 tiles)");
	*out = nullptr;
	hipSetDevice(w->device);
	sgp_tiles* t = new sgp_tiles();
	t->w = w; t->rank = rank; t->n_tiles = n_tiles;
	memset(&t->route, 0, sizeof(t->route));
	memcpy(t->route.boxes, boxes, sizeof(float) * 6 * n_tiles);
	t->route.n_tiles = n_tiles; t->route.my_rank = rank; t->route.margin = margin; t->route.pad = margin + radius_pad;
	memset(&t->stats, 0, sizeof(t->stats));
	const size_t cb = tiles_ctl_bytes(n_tiles);
	if (hipMalloc((void**)&t->d_ctl, cb) != hipSuccess || hipHostMalloc((void**)&t->h_ctl, cb, hipHostMallocDefault) != hipSuccess) { sgp_tiles_destroy(t); return fail(SGP_ERR_HIP, "sgp_tiles_create: allocation"); }
	hipMemset(t->d_ctl, 0, cb); memset(t->h_ctl, 0, cb);
	t->recv_counts.assign(n_tiles, 0); t->recv_offsets.assign(n_tiles, 0);
	if (unique_id) {      // (a one-tile communicator is legal: it lets a single GPU run the whole collective path, bench.py --force-comm)
		if (!rccl_load()) { sgp_tiles_destroy(t); return fail(SGP_ERR_HIP, "sgp_tiles_create: RCCL (librccl.so) not found"); }
		sgp_nccl_unique_id id; memcpy(&id, unique_id, sizeof(id));
		const auto t0 = std::chrono::steady_clock::now();
		const int rc = g_rccl.CommInitRank(&t->comm, (int)n_tiles, id, (int)rank);
		if (rc != 0) { sgp_tiles_destroy(t); return rccl_fail("ncclCommInitRank", rc); }
		t->stats.comm_init_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
		int seen = 0;
		if (g_rccl.CommCount && g_rccl.CommCount(t->comm, &seen) == 0) t->stats.comm_ranks = (uint32_t)seen;
		if (t->stats.comm_ranks && t->stats.comm_ranks != n_tiles) { sgp_tiles_destroy(t); return fail(SGP_ERR_HIP, "sgp_tiles_create: the RCCL communicator does not have one rank per tile"); }
	}
	*out = t;
	return SGP_OK;
}

// phase 1: export + routing kernels (stream order), header and emigrant ids still on the device
static int tiles_launch_route(sgp_tiles* t)
{
	sgp_world* w = t->w;
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	const uint32_t blocks = w->high ? (w->high + 255u) / 256u : 1u;
	const uint32_t cols = t->n_tiles + 1;
	if (blocks * cols > t->cap_blocks) {
		uint32_t c1 = t->cap_blocks, c2 = t->cap_blocks;
		{ int r = tiles_grow(w, t->d_block_counts, c1, blocks * cols); if (r != SGP_OK) return r; }
		{ int r = tiles_grow(w, t->d_block_offsets, c2, blocks * cols); if (r != SGP_OK) return r; }
		t->cap_blocks = std::min(c1, c2);
	}
	if (!t->cap_send) { int r = tiles_grow(w, t->d_send, t->cap_send, 16384u); if (r != SGP_OK) return r; }
	if (!t->cap_emig) { int r = tiles_grow(w, t->d_emig, t->cap_emig, 4096u); if (r != SGP_OK) return r; }
	launch_route_export(w->dv, w->high, t->route, t->d_block_counts, t->d_block_offsets, (RouteHeader*)t->d_ctl, t->d_send, t->cap_send, t->d_emig, t->cap_emig, w->stream);
	// the first emigrant ids ride along with the header copy
	HIP_TRY(hipMemcpyAsync(t->d_ctl + tiles_emig_off(t->n_tiles), t->d_emig, sizeof(uint32_t) * std::min<uint32_t>(SGP_TILES_EMIG_INLINE, t->cap_emig), hipMemcpyDeviceToDevice, w->stream));
	return SGP_OK;
}

// phase 2 (after the control block is on the host): capacity check, emigrants leave this world
static int tiles_after_header(sgp_tiles* t, bool* redo)
{
	sgp_world* w = t->w;
	*redo = false;
	const RouteHeader* h = (const RouteHeader*)t->h_ctl;
	if (h->total > t->cap_send || h->n_emigrants > t->cap_emig) {       // more boundary bodies than the buffers hold: grow, route again
		if (h->total > t->cap_send) { int r = tiles_grow(w, t->d_send, t->cap_send, h->total); if (r != SGP_OK) return r; }
		if (h->n_emigrants > t->cap_emig) { int r = tiles_grow(w, t->d_emig, t->cap_emig, h->n_emigrants); if (r != SGP_OK) return r; }
		*redo = true;
		return SGP_OK;
	}
	t->stats.exported = h->total; t->stats.emigrated = h->n_emigrants;
	if (h->n_emigrants) {
		std::vector<uint32_t> ids(h->n_emigrants);
		const uint32_t inl = std::min<uint32_t>(h->n_emigrants, SGP_TILES_EMIG_INLINE);
		memcpy(ids.data(), t->h_ctl + tiles_emig_off(t->n_tiles), sizeof(uint32_t) * inl);
		if (h->n_emigrants > inl) { HIP_TRY(hipMemcpy(ids.data() + inl, t->d_emig + inl, sizeof(uint32_t) * (h->n_emigrants - inl), hipMemcpyDeviceToHost)); }
		// owned dynamic bodies whose centre has left the tile: removed here, re-created by the tile that contains them (their record is already
		// in the send buffer, flagged SGP_GHOST_TAKE_OWNERSHIP); the caller learns about it through sgp_tiles_drain_migrations
		for (uint32_t id : ids) {
			if (!live(w, id)) continue;
			sgp_migration m; memset(&m, 0, sizeof(m)); m.userdata = w->hb[id].userdata; m.old_id = id; m.new_id = SGP_INVALID_ID; m.direction = SGP_MIGRATION_OUT;
			t->migrations.push_back(m);
			const int r = sgp_body_remove(w, id); if (r != SGP_OK) return r;
		}
	}
	return SGP_OK;
}

// phase 4: what arrived (n records in d_recv, by source rank) becomes this world's ghost set (+ immigrants)
static int tiles_import(sgp_tiles* t, uint32_t n)
{
	sgp_world* w = t->w;
	t->stats.received = n;
	// steady state: the same ghosts as last step in the same order, nobody immigrating -> poses go from the received records to the bodies
	// on the device; the host only sees 16 bytes per record (global id + ownership flag, packed by a kernel), not the 128-byte records
	struct GhostKey { uint64_t global_id; uint32_t motion_type, pad; };
	if (n) {
		{ int r = tiles_grow(w, t->d_keys, t->cap_keys, n); if (r != SGP_OK) return r; }
		if (n > t->cap_h_keys) {
			if (t->h_keys) hipHostFree(t->h_keys);
			t->cap_h_keys = n + n / 2 + 1024;
			HIP_TRY(hipHostMalloc((void**)&t->h_keys, 16 * (size_t)t->cap_h_keys, hipHostMallocDefault));
		}
		launch_pack_ghost_keys(t->d_recv, n, t->d_keys, w->stream);
		HIP_TRY(hipMemcpyAsync(t->h_keys, t->d_keys, 16 * (size_t)n, hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		const GhostKey* keys = (const GhostKey*)t->h_keys;
		bool same = n == w->ghost_seq.size();
		for (uint32_t k = 0; k < n && same; ++k) same = !(keys[k].motion_type & SGP_GHOST_TAKE_OWNERSHIP) && keys[k].global_id == w->ghost_seq[k].first && live(w, w->ghost_seq[k].second);
		if (same) {
			{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
			GhostDeviceSource dev = { t->d_recv, &t->d_seq_ids, &t->cap_seq, &t->ids_version };
			if (t->ids_version != w->ghost_seq_version) { int r = upload_ghost_ids(w, &dev); if (r != SGP_OK) return r; }
			launch_ghost_refresh_records(w->dv, t->d_recv, t->d_seq_ids, n, w->stream);
			w->grid_valid = false; w->dirty_since_step = true;
			t->stats.ghosts = n; t->stats.immigrated = 0; t->stats.fast_imports++;
			return SGP_OK;
		}
	}
	// the set changed (or bodies immigrate): the records themselves come to the host, which owns the body slots
	if (n > t->cap_h_recv) {
		if (t->h_recv) hipHostFree(t->h_recv);
		t->cap_h_recv = n + n / 2 + 1024;
		HIP_TRY(hipHostMalloc((void**)&t->h_recv, sizeof(sgp_ghost_record) * (size_t)t->cap_h_recv, hipHostMallocDefault));
	}
	if (n) {
		HIP_TRY(hipMemcpyAsync(t->h_recv, t->d_recv, sizeof(sgp_ghost_record) * (size_t)n, hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	// who is a ghost, who immigrates (flagged records addressed to another tile are dropped)
	const float* lo = t->route.boxes + 6 * t->rank; const float* hi = lo + 3;
	const GhostKey* keys = (const GhostKey*)t->h_keys;      // (n > 0: packed above; the scans below read 16 bytes per record instead of 128)
	bool plain = true;
	for (uint32_t k = 0; k < n && plain; ++k) plain = !(keys[k].motion_type & SGP_GHOST_TAKE_OWNERSHIP);
	const size_t seq_before = w->ghost_seq.size();
	if (plain) {
		// ghosts only: the poses stay on the device -- the host compares global ids (and creates / removes the few bodies that entered or left
		// the set), one kernel refreshes every ghost from the received records
		bool unchanged = n == seq_before;            // (no ghosts before, none now: nothing for the host to do either)
		for (uint32_t k = 0; k < n && unchanged; ++k) unchanged = keys[k].global_id == w->ghost_seq[k].first;
		GhostDeviceSource dev = { t->d_recv, &t->d_seq_ids, &t->cap_seq, &t->ids_version };
		{ int rc = import_ghosts_impl(w, t->h_recv, n, &dev, nullptr, (const uint64_t*)t->h_keys, 2); if (rc != SGP_OK) return rc; }
		t->stats.ghosts = n; t->stats.immigrated = 0;
		if (unchanged) t->stats.fast_imports++; else t->stats.slow_imports++;
		return SGP_OK;
	}
	// bodies immigrate with this exchange: their records sit between the ghosts'.  The ghosts still take the device path (by index: no record is copied, no
	// refresh command is made -- a tile of the collapsing tower holds 25 000 ghosts and receives immigrants in EVERY step: 3.5 MB of records copied and
	// 25 000 commands built, uploaded and applied per step was most of the exchange's 1.2 ms, profiles/r04_tiles_import.md)
	std::vector<uint8_t> skip(n, 0);
	std::vector<const sgp_ghost_record*> immigrants;
	uint32_t n_ghosts = 0;
	for (uint32_t k = 0; k < n; ++k) {
		if (!(keys[k].motion_type & SGP_GHOST_TAKE_OWNERSHIP)) { ++n_ghosts; continue; }
		const sgp_ghost_record& r = t->h_recv[k];
		skip[k] = 1;
		if (in_box(r.pos, lo, hi, 0.0f)) immigrants.push_back(&r);
	}
	{
		GhostDeviceSource dev = { t->d_recv, &t->d_seq_ids, &t->cap_seq, &t->ids_version };
		int rc = import_ghosts_impl(w, t->h_recv, n, &dev, skip.data(), (const uint64_t*)t->h_keys, 2); if (rc != SGP_OK) return rc;
	}
	uint32_t n_imm = 0;
	for (const sgp_ghost_record* pr : immigrants) {
		const sgp_ghost_record& r = *pr;
		sgp_body_desc d; sgp_default_body_desc(&d);
		memcpy(d.pos, r.pos, 12); memcpy(d.rot, r.rot, 16); memcpy(d.lin_vel, r.lin_vel, 12); memcpy(d.ang_vel, r.ang_vel, 12);
		d.shape_type = r.shape_type; memcpy(d.shape, r.shape, 16);
		d.motion_type = SGP_MOTION_DYNAMIC;
		d.layer = (int32_t)(r.flags & SGP_GHOST_FLAG_LAYER_MASK);
		d.is_sensor = (r.flags & SGP_GHOST_FLAG_SENSOR) ? 1 : 0; d.allow_sleeping = (r.flags & SGP_GHOST_FLAG_ALLOW_SLEEP) ? 1 : 0; d.use_zero_linear_drag = (r.flags & SGP_GHOST_FLAG_ZERO_DRAG) ? 1 : 0;
		d.mass = r.mass; d.friction = r.friction; d.restitution = r.restitution;
		d.gravity_factor = r.gravity_factor; d.linear_damping = r.linear_damping; d.angular_damping = r.angular_damping;
		d.userdata = r.userdata; d.activate = 1;
		uint32_t id = SGP_INVALID_ID;
		const int rc = add_one(w, &d, &id, false);
		// the previous owner has already let go of the body: failing to take it over must not pass silently
		if (rc != SGP_OK) return fail(rc == SGP_ERR_REJECTED ? SGP_ERR_INVALID : rc, "sgp_tiles_exchange: could not take over a migrating body (raise max_bodies; hull / mesh ids must mean the same shape on every tile)");
		sgp_migration m; memset(&m, 0, sizeof(m)); m.userdata = r.userdata; m.old_id = (uint32_t)(r.global_id & 0xFFFFFFFFull); m.new_id = id; m.direction = SGP_MIGRATION_IN; m.peer = (uint32_t)(r.global_id >> 40);
		t->migrations.push_back(m);
		++n_imm;
	}
	t->stats.immigrated = n_imm; t->stats.ghosts = n_ghosts; t->stats.slow_imports++;
	return SGP_OK;
}

SGP_API int sgp_tiles_exchange(sgp_tiles* t)
{
	if (!t || !t->w) return fail(SGP_ERR_INVALID, "sgp_tiles_exchange: NULL");
	sgp_world* w = t->w;
	hipSetDevice(w->device);
	const uint32_t T = t->n_tiles;
	const auto t_begin = std::chrono::steady_clock::now();
	struct Stamp { sgp_tiles* t; std::chrono::steady_clock::time_point t0; ~Stamp() { t->stats.last_exchange_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); t->stats.exchanges++; t->stats.total_exchange_ms += t->stats.last_exchange_ms; } } stamp = { t, t_begin };
	if (T > 1 && !t->comm) return fail(SGP_ERR_INVALID, "sgp_tiles_exchange: created without a communicator (use sgp_tiles_exchange_group for tiles of one process)");
	uint32_t* d_matrix = (uint32_t*)(t->d_ctl + tiles_matrix_off());
	// Routing is local and its COUNTS do not depend on the buffer sizes, so the all-gather runs exactly once per exchange; a rank whose
	// send / emigrant buffers were too small grows them and re-runs only its own routing kernels -- the other ranks never notice, and no rank
	// can return between the all-gather and the matching send / recv (which would leave its peers waiting in ncclRecv for ever).
	for (int attempt = 0; attempt < 3; ++attempt) {
		{ int r = tiles_launch_route(t); if (r != SGP_OK) return r; }
		// every rank's per-destination counts: one small all-gather on device buffers
		if (t->comm && attempt == 0) RCCL_TRY(g_rccl.AllGather(t->d_ctl /* RouteHeader::seg_count comes first */, d_matrix, T, SGP_NCCL_UINT32, t->comm, w->stream), "ncclAllGather");
		HIP_TRY(hipMemcpyAsync(t->h_ctl, t->d_ctl, tiles_ctl_bytes(T), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		bool redo = false;
		{ int r = tiles_after_header(t, &redo); if (r != SGP_OK) return r; }
		if (!redo) break;
		t->stats.route_retries++;
		if (attempt == 2) return fail(SGP_ERR_CAPACITY, "sgp_tiles_exchange: send buffer");      // (unreachable: the second attempt has the sizes the first one reported)
	}
	const RouteHeader* h = (const RouteHeader*)t->h_ctl;
	const uint32_t* matrix = (const uint32_t*)(t->h_ctl + tiles_matrix_off());          // [source][destination]
	uint32_t n_recv = 0;
	for (uint32_t r = 0; r < T; ++r) { t->recv_counts[r] = (T > 1 && r != t->rank) ? matrix[(size_t)r * T + t->rank] : 0u; t->recv_offsets[r] = n_recv; n_recv += t->recv_counts[r]; }
	{ int r = tiles_grow(w, t->d_recv, t->cap_recv, std::max(n_recv, 1u)); if (r != SGP_OK) return r; }
	if (T > 1) {
		RCCL_TRY(g_rccl.GroupStart(), "ncclGroupStart");
		for (uint32_t r = 0; r < T; ++r) {
			if (r == t->rank) continue;
			if (h->seg_count[r]) RCCL_TRY(g_rccl.Send(t->d_send + h->seg_start[r], sizeof(sgp_ghost_record) * (size_t)h->seg_count[r], SGP_NCCL_UINT8, (int)r, t->comm, w->stream), "ncclSend");
			if (t->recv_counts[r]) RCCL_TRY(g_rccl.Recv(t->d_recv + t->recv_offsets[r], sizeof(sgp_ghost_record) * (size_t)t->recv_counts[r], SGP_NCCL_UINT8, (int)r, t->comm, w->stream), "ncclRecv");
		}
		RCCL_TRY(g_rccl.GroupEnd(), "ncclGroupEnd");
	}
	t->stats.sent = h->total;
	return tiles_import(t, n_recv);
}

// Several tiles driven by ONE process (one GPU or several): the same exchange with plain device-to-device copies in place of RCCL.
SGP_API int sgp_tiles_exchange_group(sgp_tiles** ts, uint32_t n)
{
	if (!ts || !n) return fail(SGP_ERR_INVALID, "sgp_tiles_exchange_group: NULL");
	for (uint32_t i = 0; i < n; ++i) if (!ts[i] || ts[i]->n_tiles != n || ts[i]->rank != i) return fail(SGP_ERR_INVALID, "sgp_tiles_exchange_group: pass all tiles, in rank order");
	for (int attempt = 0; attempt < 3; ++attempt) {
		for (uint32_t i = 0; i < n; ++i) { hipSetDevice(ts[i]->w->device); int r = tiles_launch_route(ts[i]); if (r != SGP_OK) return r;
			HIP_TRY(hipMemcpyAsync(ts[i]->h_ctl, ts[i]->d_ctl, tiles_ctl_bytes(n), hipMemcpyDeviceToHost, ts[i]->w->stream)); }
		bool any_redo = false;
		for (uint32_t i = 0; i < n; ++i) { hipSetDevice(ts[i]->w->device); HIP_TRY(hipStreamSynchronize(ts[i]->w->stream)); const RouteHeader* h = (const RouteHeader*)ts[i]->h_ctl; if (h->total > ts[i]->cap_send || h->n_emigrants > ts[i]->cap_emig) any_redo = true; }
		if (any_redo) {        // grow whoever was short, route everyone again (nothing has been removed yet)
			for (uint32_t i = 0; i < n; ++i) { const RouteHeader* h = (const RouteHeader*)ts[i]->h_ctl; hipSetDevice(ts[i]->w->device);
				if (h->total > ts[i]->cap_send) { int r = tiles_grow(ts[i]->w, ts[i]->d_send, ts[i]->cap_send, h->total); if (r != SGP_OK) return r; }
				if (h->n_emigrants > ts[i]->cap_emig) { int r = tiles_grow(ts[i]->w, ts[i]->d_emig, ts[i]->cap_emig, h->n_emigrants); if (r != SGP_OK) return r; } }
			if (attempt == 2) return fail(SGP_ERR_CAPACITY, "sgp_tiles_exchange_group: send buffer");
			continue;
		}
		break;
	}
	for (uint32_t i = 0; i < n; ++i) { bool redo = false; hipSetDevice(ts[i]->w->device); int r = tiles_after_header(ts[i], &redo); if (r != SGP_OK) return r; }
	for (uint32_t dst = 0; dst < n; ++dst) {
		sgp_tiles* t = ts[dst];
		hipSetDevice(t->w->device);
		uint32_t n_recv = 0;
		for (uint32_t src = 0; src < n; ++src) { const RouteHeader* hs = (const RouteHeader*)ts[src]->h_ctl; t->recv_counts[src] = src == dst ? 0u : hs->seg_count[dst]; t->recv_offsets[src] = n_recv; n_recv += t->recv_counts[src]; }
		{ int r = tiles_grow(t->w, t->d_recv, t->cap_recv, std::max(n_recv, 1u)); if (r != SGP_OK) return r; }
		for (uint32_t src = 0; src < n; ++src) {
			if (!t->recv_counts[src]) continue;
			const RouteHeader* hs = (const RouteHeader*)ts[src]->h_ctl;
			HIP_TRY(hipMemcpyAsync(t->d_recv + t->recv_offsets[src], ts[src]->d_send + hs->seg_start[dst], sizeof(sgp_ghost_record) * (size_t)t->recv_counts[src], hipMemcpyDeviceToDevice, t->w->stream));
		}
		t->stats.sent = ((const RouteHeader*)t->h_ctl)->total;
		{ int r = tiles_import(t, n_recv); if (r != SGP_OK) return r; }
	}
	return SGP_OK;
}

// ---- re-tiling by body count ------------------------------------------------------------------------------------------------------
// A static split of a scene that moves -- BASELINE config 4 is a tower that falls out of its upper tiles -- leaves tiles without work.  The grid keeps
// its topology (gx x gy x gz, tile = ix + gx (iy + gy iz)); its planes move to the quantiles of where the OWNED bodies are: the x planes from all
// bodies, the y planes of every x slab from that slab's bodies, the z planes of every (x, y) column from that column's.  Four small rounds (bounds,
// then one histogram of SGP_TILE_HIST_BINS bins per axis and group), each a kernel + an all-gather of a few KB + one read-back; every rank derives the
// same planes from the same gathered counts.  Bodies then change owner through the ordinary migration of the next exchange.
#define TILE_HIST_MAX_GROUPS 16
static int tiles_hist_buffers(sgp_tiles* t)
{
	const size_t one = sizeof(uint32_t) * TILE_HIST_MAX_GROUPS * SGP_TILE_HIST_BINS;
	if (!t->d_hist) { HIP_TRY(hipMalloc((void**)&t->d_hist, one)); HIP_TRY(hipMalloc((void**)&t->d_hist_all, one * t->n_tiles)); HIP_TRY(hipHostMalloc((void**)&t->h_hist, one * t->n_tiles, hipHostMallocDefault)); }
	return SGP_OK;
}
// one round on the tiles of this process (one with a communicator, or all of a group): sum[k] = counts over every tile (level 0: min / max as ordered ints)
static int tiles_hist_round(sgp_tiles** ts, uint32_t n_local, const TilePlanes& tp, int level, uint32_t len, std::vector<uint64_t>& sum, int bounds[6])
{
	const uint32_t T = ts[0]->n_tiles;
	for (uint32_t i = 0; i < n_local; ++i) {
		sgp_tiles* t = ts[i]; sgp_world* w = t->w;
		hipSetDevice(w->device);
		{ int r = tiles_hist_buffers(t); if (r != SGP_OK) return r; }
		{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
		if (level == 0) { int* h = (int*)t->h_hist; for (int k = 0; k < 6; ++k) h[k] = k < 3 ? 0x7FFFFFFF : (int)0x80000000; h[6] = h[7] = 0; HIP_TRY(hipMemcpyAsync(t->d_hist, h, 32, hipMemcpyHostToDevice, w->stream)); }
		else HIP_TRY(hipMemsetAsync(t->d_hist, 0, sizeof(uint32_t) * len, w->stream));
		if (w->high) launch_tiles_hist(w->dv, w->high, tp, level, t->d_hist, w->stream);
		if (t->comm) {
			RCCL_TRY(g_rccl.AllGather(t->d_hist, t->d_hist_all, len, SGP_NCCL_UINT32, t->comm, w->stream), "ncclAllGather (re-tiling)");
			HIP_TRY(hipMemcpyAsync(t->h_hist, t->d_hist_all, sizeof(uint32_t) * (size_t)len * T, hipMemcpyDeviceToHost, w->stream));
		} else HIP_TRY(hipMemcpyAsync(t->h_hist, t->d_hist, sizeof(uint32_t) * len, hipMemcpyDeviceToHost, w->stream));
	}
	for (uint32_t i = 0; i < n_local; ++i) { hipSetDevice(ts[i]->w->device); HIP_TRY(hipStreamSynchronize(ts[i]->w->stream)); }
	sum.assign(len, 0);
	for (int k = 0; k < 6; ++k) bounds[k] = k < 3 ? 0x7FFFFFFF : (int)0x80000000;
	auto fold = [&](const uint32_t* h) {
		if (level == 0) { const int* b = (const int*)h; for (int k = 0; k < 3; ++k) { bounds[k] = std::min(bounds[k], b[k]); bounds[3 + k] = std::max(bounds[3 + k], b[3 + k]); } }
		else for (uint32_t k = 0; k < len; ++k) sum[k] += h[k];
	};
	if (ts[0]->comm) for (uint32_t r = 0; r < T; ++r) fold(ts[0]->h_hist + (size_t)r * len);
	else for (uint32_t i = 0; i < n_local; ++i) fold(ts[i]->h_hist);
	return SGP_OK;
}
static inline float ordered_int_to_float(int i) { const int v = i >= 0 ? i : i ^ 0x7FFFFFFF; float f; memcpy(&f, &v, 4); return f; }
// the g - 1 planes that cut a histogram into g parts of equal count (linear inside a bin); an empty histogram is cut evenly
static void quantile_planes(const uint64_t* h, float lo, float hi, uint32_t g, float* planes)
{
	uint64_t total = 0; for (uint32_t b = 0; b < SGP_TILE_HIST_BINS; ++b) total += h[b];
	const double bw = ((double)hi - (double)lo) / SGP_TILE_HIST_BINS;
	for (uint32_t k = 1; k < g; ++k) {
		if (!total) { planes[k - 1] = (float)(lo + ((double)hi - lo) * k / g); continue; }
		const double target = (double)total * k / g;
		uint64_t cum = 0; uint32_t b = 0;
		while (b + 1 < SGP_TILE_HIST_BINS && (double)(cum + h[b]) < target) { cum += h[b]; ++b; }
		const double frac = h[b] ? (target - (double)cum) / (double)h[b] : 0.5;
		planes[k - 1] = (float)(lo + (b + std::min(1.0, std::max(0.0, frac))) * bw);
	}
	for (uint32_t k = 1; k + 1 < g; ++k) if (planes[k] < planes[k - 1]) planes[k] = planes[k - 1];
}
static int tiles_rebalance_impl(sgp_tiles** ts, uint32_t n_local, uint32_t gx, uint32_t gy, uint32_t gz, int by_contacts)
{
	const uint32_t T = ts[0]->n_tiles;
	if (!gx || !gy || !gz || gx > 4 || gy > 4 || gz > 4 || gx * gy * gz != T) return fail(SGP_ERR_INVALID, "sgp_tiles_rebalance: the grid must have one cell per tile (at most 4 per axis)");
	TilePlanes tp; memset(&tp, 0, sizeof(tp)); tp.gx = gx; tp.gy = gy; tp.gz = gz; tp.by_contacts = by_contacts ? 1u : 0u;
	std::vector<uint64_t> sum; int bounds[6];
	{ int r = tiles_hist_round(ts, n_local, tp, 0, 8, sum, bounds); if (r != SGP_OK) return r; }
	if (bounds[0] > bounds[3]) return SGP_OK;                         // nobody owns a dynamic body: nothing to balance
	for (int a = 0; a < 3; ++a) { tp.glo[a] = ordered_int_to_float(bounds[a]); tp.ghi[a] = ordered_int_to_float(bounds[3 + a]); const float pad = 1.0e-3f * (1.0f + fabsf(tp.ghi[a] - tp.glo[a])); tp.glo[a] -= pad; tp.ghi[a] += pad; }
	{ int r = tiles_hist_round(ts, n_local, tp, 1, SGP_TILE_HIST_BINS, sum, bounds); if (r != SGP_OK) return r; }
	quantile_planes(sum.data(), tp.glo[0], tp.ghi[0], gx, tp.xp);
	{ int r = tiles_hist_round(ts, n_local, tp, 2, gx * SGP_TILE_HIST_BINS, sum, bounds); if (r != SGP_OK) return r; }
	for (uint32_t ix = 0; ix < gx; ++ix) quantile_planes(sum.data() + (size_t)ix * SGP_TILE_HIST_BINS, tp.glo[1], tp.ghi[1], gy, tp.yp + 4 * ix);
	{ int r = tiles_hist_round(ts, n_local, tp, 3, gx * gy * SGP_TILE_HIST_BINS, sum, bounds); if (r != SGP_OK) return r; }
	float zp[16 * 4]; memset(zp, 0, sizeof(zp));
	for (uint32_t c = 0; c < gx * gy; ++c) quantile_planes(sum.data() + (size_t)c * SGP_TILE_HIST_BINS, tp.glo[2], tp.ghi[2], gz, zp + 4 * c);
	const float big = 1.0e9f;
	float boxes[6 * SGP_MAX_TILES];
	for (uint32_t r = 0; r < T; ++r) {
		const uint32_t ix = r % gx, iy = (r / gx) % gy, iz = r / (gx * gy);
		float* lo = boxes + 6 * r; float* hi = lo + 3;
		lo[0] = ix ? tp.xp[ix - 1] : -big; hi[0] = ix + 1 < gx ? tp.xp[ix] : big;
		lo[1] = iy ? tp.yp[4 * ix + iy - 1] : -big; hi[1] = iy + 1 < gy ? tp.yp[4 * ix + iy] : big;
		lo[2] = iz ? zp[4 * (ix + gx * iy) + iz - 1] : -big; hi[2] = iz + 1 < gz ? zp[4 * (ix + gx * iy) + iz] : big;
	}
	for (uint32_t i = 0; i < n_local; ++i) { memcpy(ts[i]->route.boxes, boxes, sizeof(float) * 6 * T); ts[i]->stats.rebalances++; }
	return SGP_OK;
}
SGP_API int sgp_tiles_rebalance(sgp_tiles* t, uint32_t gx, uint32_t gy, uint32_t gz, int by_contacts)
{
	if (!t || !t->w) return fail(SGP_ERR_INVALID, "sgp_tiles_rebalance: NULL");
	if (t->n_tiles > 1 && !t->comm) return fail(SGP_ERR_INVALID, "sgp_tiles_rebalance: created without a communicator (use sgp_tiles_rebalance_group for tiles of one process)");
	return tiles_rebalance_impl(&t, 1, gx, gy, gz, by_contacts);
}
SGP_API int sgp_tiles_rebalance_group(sgp_tiles** ts, uint32_t n, uint32_t gx, uint32_t gy, uint32_t gz, int by_contacts)
{
	if (!ts || !n) return fail(SGP_ERR_INVALID, "sgp_tiles_rebalance_group: NULL");
	for (uint32_t i = 0; i < n; ++i) if (!ts[i] || ts[i]->n_tiles != n || ts[i]->rank != i || ts[i]->comm) return fail(SGP_ERR_INVALID, "sgp_tiles_rebalance_group: pass all tiles of the (communicator-less) group, in rank order");
	return tiles_rebalance_impl(ts, n, gx, gy, gz, by_contacts);
}
SGP_API int sgp_tiles_get_boxes(sgp_tiles* t, float* boxes_out)
{
	if (!t || !boxes_out) return fail(SGP_ERR_INVALID, "sgp_tiles_get_boxes: NULL");
	memcpy(boxes_out, t->route.boxes, sizeof(float) * 6 * t->n_tiles);
	return SGP_OK;
}

SGP_API int sgp_tiles_get_stats(sgp_tiles* t, sgp_tiles_stats* out)
{
	if (!t || !out) return fail(SGP_ERR_INVALID, "sgp_tiles_get_stats: NULL");
	*out = t->stats;
	return SGP_OK;
}

SGP_API int sgp_tiles_drain_migrations(sgp_tiles* t, sgp_migration* out, uint32_t cap, uint32_t* n_out)
{
	if (!t || !n_out || (!out && cap)) return fail(SGP_ERR_INVALID, "sgp_tiles_drain_migrations: NULL");
	const uint32_t n = (uint32_t)t->migrations.size(), m = std::min(n, cap);
	if (m) memcpy(out, t->migrations.data(), sizeof(sgp_migration) * m);
	*n_out = n;
	t->migrations.erase(t->migrations.begin(), t->migrations.begin() + m);
	return SGP_OK;
}

// Self test of the run-time RCCL binding on ONE GPU (tests/test_tiles_parity_gpu.py): a one-rank communicator, an all-gather, and a grouped
// ncclSend / ncclRecv of `n_records` records from this rank to itself, compared byte for byte.  Not declared in include/sgp.h.
SGP_API int sgp_tiles_selftest_rccl(sgp_world* w, uint32_t n_records)
{
	if (!w || !n_records) return fail(SGP_ERR_INVALID, "sgp_tiles_selftest_rccl: bad arguments");
	hipSetDevice(w->device);
	if (!rccl_load()) return fail(SGP_ERR_HIP, "RCCL (librccl.so) not found");
	sgp_nccl_unique_id id;
	RCCL_TRY(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
	sgp_nccl_comm comm = nullptr;
	RCCL_TRY(g_rccl.CommInitRank(&comm, 1, id, 0), "ncclCommInitRank");
	const size_t bytes = sizeof(sgp_ghost_record) * (size_t)n_records;
	unsigned char *a = nullptr, *b = nullptr; uint32_t *c = nullptr;
	HIP_TRY(hipMalloc((void**)&a, bytes)); HIP_TRY(hipMalloc((void**)&b, bytes)); HIP_TRY(hipMalloc((void**)&c, 64));
	std::vector<unsigned char> src(bytes), dst(bytes, 0);
	for (size_t i = 0; i < bytes; ++i) src[i] = (unsigned char)((i * 2654435761u) >> 13);
	const uint32_t row[4] = { 11, 22, 33, 44 }; uint32_t got[4] = { 0, 0, 0, 0 };
	HIP_TRY(hipMemcpy(a, src.data(), bytes, hipMemcpyHostToDevice)); HIP_TRY(hipMemset(b, 0, bytes)); HIP_TRY(hipMemcpy(c, row, 16, hipMemcpyHostToDevice));
	int rc = g_rccl.AllGather(c, c + 8, 4, SGP_NCCL_UINT32, comm, w->stream);
	if (rc == 0) rc = g_rccl.GroupStart();
	if (rc == 0) rc = g_rccl.Send(a, bytes, SGP_NCCL_UINT8, 0, comm, w->stream);
	if (rc == 0) rc = g_rccl.Recv(b, bytes, SGP_NCCL_UINT8, 0, comm, w->stream);
	if (rc == 0) rc = g_rccl.GroupEnd();
	hipStreamSynchronize(w->stream);
	hipMemcpy(dst.data(), b, bytes, hipMemcpyDeviceToHost); hipMemcpy(got, c + 8, 16, hipMemcpyDeviceToHost);
	hipFree(a); hipFree(b); hipFree(c);
	g_rccl.CommDestroy(comm);
	if (rc != 0) return rccl_fail("RCCL self test", rc);
	if (memcmp(src.data(), dst.data(), bytes) != 0 || memcmp(row, got, 16) != 0) return fail(SGP_ERR_HIP, "RCCL self test: payload mismatch");
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Network physics snapshots: the de-jitter ring and the insertion schedule (include/sgp.h has the reference map).  Host-side state only.
#include <map>
struct SnapshotRing {
	struct Entry { sgp_pose_vel rec; double client_time, local_time; };
	Entry slots[SGP_SNAPSHOT_HISTORY];
	uint32_t next_snapshot_i = 0, next_insertable_snapshot_i = 0;
	double transmission_time_offset = 0.0;
	uint32_t idle_expires = 0;      // expire() calls this ring has seen without ever holding a snapshot
};
struct sgp_snapshot_queue { std::map<uint64_t, SnapshotRing> rings; };      // ordered: the playback order is ascending uid, deterministic

SGP_API int sgp_snapshot_queue_create(sgp_snapshot_queue** out)
{
	if (!out) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_create: NULL");
	*out = new sgp_snapshot_queue();
	return SGP_OK;
}
SGP_API int sgp_snapshot_queue_destroy(sgp_snapshot_queue* q) { delete q; return SGP_OK; }

SGP_API int sgp_snapshot_queue_push(sgp_snapshot_queue* q, uint64_t uid, const sgp_pose_vel* rec, double client_time, double local_time)
{
	if (!q || !rec) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_push: NULL");
	REQUIRE_FINITE(finite3(rec->pos) && finite4(rec->rot) && finite3(rec->lin_vel) && finite3(rec->ang_vel), "sgp_snapshot_queue_push");
	SnapshotRing& r = q->rings[uid];
	SnapshotRing::Entry& e = r.slots[r.next_snapshot_i % (uint32_t)SGP_SNAPSHOT_HISTORY];      // the oldest slot is overwritten, pending or not
	e.rec = *rec; e.client_time = client_time; e.local_time = local_time;
	r.next_snapshot_i++;
	return SGP_OK;
}
SGP_API int sgp_snapshot_queue_push_wire(sgp_snapshot_queue* q, const uint8_t msg[SGP_PHYSICS_UPDATE_BYTES], double local_time)
{
	if (!q || !msg) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_push_wire: NULL");
	uint64_t uid = 0; sgp_pose_vel rec; double t = 0.0;
	{ const int r = sgp_physics_update_decode(msg, &uid, &rec, &t); if (r != SGP_OK) return r; }
	return sgp_snapshot_queue_push(q, uid, &rec, t, local_time);
}

SGP_API int sgp_snapshot_queue_ownership(sgp_snapshot_queue* q, uint64_t uid, double global_time_now, double ownership_change_global_time, int renewal)
{
	if (!q) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_ownership: NULL");
	SnapshotRing& r = q->rings[uid];
	const double offset = global_time_now - ownership_change_global_time;      // receiver's clock minus sender's clock at the same event
	if (renewal) { if (r.transmission_time_offset == 0.0) r.transmission_time_offset = offset; }
	else { r.transmission_time_offset = offset; r.next_insertable_snapshot_i = r.next_snapshot_i; }      // a new owner: what the old one queued is void
	return SGP_OK;
}

SGP_API int sgp_snapshot_queue_poll(sgp_snapshot_queue* q, double global_time, double padding_delay, uint64_t* uids_out, sgp_pose_vel* recs_out, uint32_t cap, uint32_t* n_out)
{
	if (!q || !n_out || (cap && (!uids_out || !recs_out))) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_poll: NULL");
	uint32_t n = 0;
	for (auto& kv : q->rings) {
		SnapshotRing& r = kv.second;
		if (!(r.next_insertable_snapshot_i < r.next_snapshot_i)) continue;                       // nothing pending
		const SnapshotRing::Entry& e = r.slots[r.next_insertable_snapshot_i % (uint32_t)SGP_SNAPSHOT_HISTORY];
		const double desired_insertion_time = e.client_time + r.transmission_time_offset + padding_delay;
		if (!(global_time >= desired_insertion_time)) continue;
		if (n < cap) { uids_out[n] = kv.first; recs_out[n] = e.rec; r.next_insertable_snapshot_i++; }      // (beyond cap: stays pending, reported in *n_out)
		++n;
	}
	*n_out = n;
	return SGP_OK;
}

SGP_API int sgp_snapshot_queue_expire(sgp_snapshot_queue* q, double local_time_now, double max_age, uint32_t* n_out)
{
	if (!q) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_expire: NULL");
	for (auto it = q->rings.begin(); it != q->rings.end();) {
		SnapshotRing& r = it->second;
		const bool has_any = r.next_snapshot_i > 0;
		const double last = has_any ? r.slots[(r.next_snapshot_i - 1) % (uint32_t)SGP_SNAPSHOT_HISTORY].local_time : -1.0e300;
		// (a ring that an ownership message created and no transform update ever filled has no time stamp to age by: it goes after 1024 calls -- the caller
		// expires once per frame -- so that the map cannot grow without bound on a long-running client; advisor r03)
		if (has_any ? (local_time_now - last > max_age) : (++it->second.idle_expires > 1024u)) it = q->rings.erase(it); else ++it;
	}
	if (n_out) *n_out = (uint32_t)q->rings.size();
	return SGP_OK;
}

SGP_API int sgp_snapshot_queue_peek(sgp_snapshot_queue* q, uint64_t uid, uint32_t* next_snapshot_i, uint32_t* next_insertable_snapshot_i, double* transmission_time_offset)
{
	if (!q) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_peek: NULL");
	auto it = q->rings.find(uid);
	if (it == q->rings.end()) return fail(SGP_ERR_BAD_ID, "sgp_snapshot_queue_peek: uid not tracked");
	if (next_snapshot_i) *next_snapshot_i = it->second.next_snapshot_i;
	if (next_insertable_snapshot_i) *next_insertable_snapshot_i = it->second.next_insertable_snapshot_i;
	if (transmission_time_offset) *transmission_time_offset = it->second.transmission_time_offset;
	return SGP_OK;
}
