// sgp_world_internal.h -- what the host-side files of libsgp.so share: the world record (sgp_world), error reporting, device allocation helpers and
// the handful of functions one file defines and another calls.  Not part of the ABI (include/sgp.h is); nothing here is exported.
//   sgp_world.hip            defaults, world construction, the launch plan and the step (think), command flush, events
//   sgp_world_bodies.hip     body lifecycle, setters, forces, read-back (addObject / setters / getters of PhysicsWorld)
//   sgp_world_shapes.hip     meshes, convex hulls, vehicles
//   sgp_world_queries.hip    rays, capsule queries, sphere casts
//   sgp_world_tiles.hip      ghost import / export, the tile exchange over RCCL, re-tiling
//   sgp_world_snapshots.hip  the network snapshot codec and the de-jitter queue (host only)
#pragma once
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

extern thread_local std::string g_last_error;      // (defined in sgp_world.hip)
extern int g_device_count;

inline int fail(int code, const char* what, hipError_t e = hipSuccess)
{
	char buf[512];
	if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
	else snprintf(buf, sizeof(buf), "%s", what);
	g_last_error = buf;
	return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(SGP_ERR_HIP, #expr, e_); } while (0)

inline const char* const k_class_names[KC_COUNT] = {
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
	std::vector<uint4> rec_creates;      // bodies the device is to create from received records: (record index, body slot, flags, -), launched by the import that queued them
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
	bool dirty_since_step = true;                              // an edit was flushed since the last step (or no step yet)
	bool events_on_device = true;                              // the device event lists may hold something the host vectors do not (a step without read-back, applied edits)
	StepParams sp_uploaded; bool sp_uploaded_valid = false;    // what d_sp holds (upload_sp skips the launch when nothing changed)
	uint32_t last_active = 0xFFFFFFFFu;
	StepParams* h_sp = nullptr; StepParams* d_sp = nullptr;      // pinned host copy / device copy of the per-step scalars
	std::map<std::string, hipGraphExec_t> graphs;              // replayable launch sequences keyed by launch plan
	std::string last_plan_key[2]; uint32_t plan_repeats[2] = { 0, 0 };   // per buffer parity: StepParams (by value in the first launch) flips parity every step
	bool use_graphs = true; bool use_small_world = true; bool use_wake_round = true; uint32_t tail_threshold = 256;
	uint32_t rows_mode_default = 2;        // SGP_ROWS_MODE_DEFAULT: the layout below compact_rows_min constraints -- 1 compact rows: r x axis stored (96 B per point), I (r x axis) rebuilt by the lane from the
	                                       // step's world-inverse-inertia record (DV::iw, round 5): config 3 496 -> 509, config 5 626 -> 652, config 2 915 -> 939 steps/s against 0 = full rows (192 B per point)
	bool rows_in_small_worlds = false;     // SGP_ROWS_IN_SMALL_WORLDS=1 with SGP_NO_SMALL_WORLD=1 (tools/fuzz_parity.py): worlds of up to 2048 bodies, which otherwise keep full rows, take the layouts below
	uint32_t rows_mode2_min = 65536;       // ... and below this many constraints compact rows (1) where the default says none (2): a world of 20k constraints spends its passes in the component and tail kernels, which keep a constraint for ten iterations and have the rows' use ten times (config 2: 942 against 934 steps/s)
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
	// single-query mailbox (sgp_raycast with n = 1): host-mapped block + whether a server wave is (believed to be) resident on the stream
	RayMailbox* ray_mb = nullptr; bool ray_server_on = false; bool ray_server_enabled = true; uint32_t ray_seq = 0, ray_gen = 0;
	uint32_t ray_server_launches = 0, ray_server_rays = 0;
	bool last_step_idle = false;       // the last step was skipped (every body asleep, nothing edited): no vehicle took part in it, whatever its record says
	bool grid_valid = false;                                   // the broad-phase grid matches the current poses (ray queries reuse it)
	// static triangle meshes: host-side headers + pools mirrored on the device (grown on demand)
	std::vector<MeshHeader> meshes; std::vector<float4> mesh_verts; std::vector<uint4> mesh_tris; std::vector<uint32_t> mesh_tri_mat; std::vector<MeshNode> mesh_nodes;
	MeshHeader* d_meshes = nullptr; float4* d_mesh_verts = nullptr; uint4* d_mesh_tris = nullptr; uint32_t* d_mesh_tri_mat = nullptr; MeshNode* d_mesh_nodes = nullptr;
	size_t cap_mesh_verts = 0, cap_mesh_tris = 0, cap_mesh_tri_mat = 0, cap_mesh_nodes = 0;
	// shape lifecycle: bodies referencing each mesh / hull, ids and pool ranges of destroyed shapes waiting for reuse, table capacities (grown on demand)
	std::vector<uint32_t> mesh_refs, hull_refs, free_mesh_ids, free_hull_ids;
	uint32_t n_big_hulls = 0;      // live hulls of more than SGD_HULL_SMALL_VERTS vertices: their pairs go through k_narrowphase_hull_big (part of the step plan)
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
	bool veh_cylinder_seen = false;                            // some vehicle casts its wheels as cylinders (SGP_VEHICLE_TESTER_CYLINDER): k_vehicle_cast's instance with that search in it
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

template <typename T> inline int dev_alloc(sgp_world* w, T*& p, size_t n)
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

inline int ensure_stage(sgp_world* w, size_t bytes)
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

inline uint32_t next_pow2(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }


// ---- helpers used by several files -----------------------------------------------------------------------------------------------------------
static inline bool finite3(const float* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }
static inline bool finite4(const float* v) { return finite3(v) && std::isfinite(v[3]); }
// The reference only asserts finite inputs in debug builds (PhysicsWorld.cpp:548-556,625,710); a NaN that gets into one body spreads through
// every contact it touches, so the setters refuse it outright.
#define REQUIRE_FINITE(cond, what) do { if (!(cond)) return fail(SGP_ERR_INVALID, what ": non-finite argument"); } while (0)
static inline bool live(const sgp_world* w, uint32_t id) { return w && id < w->high && (w->hb[id].flags & BF_ALIVE); }

static BodyCmd blank_cmd(uint32_t id, uint32_t ops) { BodyCmd c; memset(&c, 0, sizeof(c)); c.id = id; c.ops = ops; return c; }
// compound ids reported by queries and events: a child's slot -> the compound's id (+ the child index)
static inline uint32_t compound_id_of(const sgp_world* w, uint32_t id, uint32_t* sub_out)
{
	const HostBody& b = w->hb[id];
	if (b.comp_root == SGP_INVALID_ID) { if (sub_out) *sub_out = 0; return id; }
	if (sub_out) *sub_out = b.comp_child;
	return b.comp_root;
}

// defined in sgp_world.hip
void ray_server_stop(sgp_world* w);      // tells a resident ray server to leave (no wait: what is launched next runs after it); called by flush_cmds
void invalidate_graphs(sgp_world* w);
int flush_cmds(sgp_world* w);
int collect_events(sgp_world* w, bool counters_fresh = false);
int read_counters(sgp_world* w);
// defined in sgp_world_bodies.hip
int add_one(sgp_world* w, const sgp_body_desc* d, uint32_t* id_out, bool ghost);
int book_record_body(sgp_world* w, uint32_t* flags_io, uint64_t userdata, float radius, float volume, bool ghost, uint32_t* id_out);      // (the host's share of add_one for a body created on the device from a record)
