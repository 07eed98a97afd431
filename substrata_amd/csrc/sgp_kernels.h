// sgp_kernels.h -- device-side data layout + kernel declarations of the gfx950 rigid-body step.
//
// One sgp_world owns one set of SoA arrays in HBM (struct DeviceArrays).  Everything a kernel needs is passed as a
// by-value view (struct DV) so launches carry plain pointers only.
//
// Stage map of the step behind PhysicsWorld::think (/root/reference/gui_client/PhysicsWorld.cpp:1356-1443) and where each stage lives:
//   sgp_k_broadphase.hip   K2/K3  k_step_begin (bounds), k_bp_cell (paged cell grid), k_scan_*, k_bp_scatter_large, k_bp_pairs + layer filter (PhysicsWorld.cpp:160-189)
//   sgp_k_narrowphase.hip  K4     k_narrowphase (sphere / box / capsule manifolds, sgp_device_collide.h), k_wake_pairs (in-step activation), k_narrowphase_hull*
//   sgp_k_mesh.hip         K4     k_narrowphase_mesh<8|64, kinds>: (body, static mesh) pairs by lanes
//   sgp_k_constraints.hip  K5/K6  k_colour_* (deterministic colouring), k_setup (contact-cache match, constraint rows), k_cache_build, k_contact_events
//   sgp_k_solve.hip        K7     k_warm_bodies, k_solve_colour<0|1|2>, k_hc_* + k_solve_hc (high colours by connected component), k_solve_tail*, k_solve_small*
//   sgp_k_sweep.hip        K8/K1/K9/A3  the body-array sweep k_pre_solve + k_integrate_pose + k_finalize, k_island_*, k_sleep_apply, k_buoyancy (PhysicsWorld.cpp:1367-1442)
//   sgp_k_vehicle.hip      (f)1   k_vehicle_cast / controller, the vehicle rows inside the solver passes (k_solve_colour_veh)
//   sgp_k_queries.hip      A7     k_raycast, k_collide_capsules, k_spherecast
//   sgp_k_edits.hip        A5/A6  k_apply_cmds, k_ghost_refresh, read-back
//   sgp_k_tiles.hip        (e)    tile export / routing, re-tiling histograms
//   sgp_dev_*.h            device-inline functions shared between stage files (sgp_dev_all.h includes them in dependency order)
// All body state is SoA float4 / 32-byte records in HBM; every per-body kernel is a coalesced 16 B/lane sweep.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sgp.h"

// ---- body flags (uint32 per body) -----------------------------------------------------------------------------
#define BF_MOTION_MASK   0x3u        // SGP_MOTION_*
#define BF_LAYER_SHIFT   2
#define BF_LAYER_MASK    (0x3u << BF_LAYER_SHIFT)
#define BF_ALIVE         (1u << 4)
#define BF_ACTIVE        (1u << 5)
#define BF_SENSOR        (1u << 6)
#define BF_ALLOW_SLEEP   (1u << 7)
#define BF_ZERO_LIN_DRAG (1u << 8)
#define BF_LARGE         (1u << 9)   // skips the hashed grid (ground quad etc.)
#define BF_UNDERWATER    (1u << 10)
#define BF_GHOST         (1u << 11)  // owned by another tile (multi-GPU), simulated as velocity-driven
#define BF_HAS_FORCE     (1u << 13)  // force / torque accumulators hold something: k_pre_solve reads and clears them
#define BF_CACHE_INVALID (1u << 12)  // created or reshaped since the last step: its pairs do not reuse cached manifolds (cleared by k_pre_solve)
#define BF_SHAPE_SHIFT   18          // SGP_SHAPE_* (3 bits)
#define BF_SHAPE_MASK    (0x7u << BF_SHAPE_SHIFT)
#define BF_CHASSIS       (1u << 22)  // chassis of a live vehicle: colour 0 of the contact colouring is the vehicle's own (its rows are solved next to the first contact colour)
#define BF_ALIAS         (1u << 21)  // second / third slot of a static mesh body: carries contact manifolds only (never binned, never queried)
#define BF_WAKE          (1u << 14)  // scratch: touched by an active body this step
#define BF_CAN_SLEEP     (1u << 15)  // scratch: sleep test result
#define BF_MOVABLE_PREV  (1u << 16)  // movable (dynamic and awake) when the PREVIOUS step coloured its constraints
#define BF_MOVABLE_CUR   (1u << 17)  // same, this step
#define BF_AWAKE_STEP    (1u << 23)  // awake (active, not static) in the step being taken, once its collision detection has woken what it wakes (k_pre_solve); until the next k_pre_solve
#define BF_FRESH         (1u << 24)  // was BF_CACHE_INVALID when the step being taken began (k_pre_solve): no cache entry of this slot's pairs is carried over (k_cache_build)

#define SGP_LABEL(slot, gen) ((uint32_t)(slot) | ((uint32_t)(gen) << 25))      // (max_bodies < 2^25)
#define SGP_LABEL_SLOT(l) ((l) & 0x1FFFFFFu)
#define SGP_SMALL_COLOURING_MANIFOLDS 4096   // at most this many manifolds last step: colouring rounds run inside one workgroup
#define SGP_ISLAND_MARK_ROUNDS 3   // marking rounds before the island union-find (k_island_mark)
#define SGP_MAX_COLOURS      64
#define SGP_OVERFLOW_COLOUR  63
// DV::man_prev of a manifold: slot of the pair's constraint in the previous step's buffer (28 bits; MAN_PREV_NONE: it had none) | that constraint's point count
// << 28 (what the set-up needs of its header: no gather for it) | MAN_PREV_REUSED: the manifold itself was taken from the body-pair contact cache
#define MAN_PREV_SLOT_MASK 0x0FFFFFFFu
#define MAN_PREV_NONE      0x0FFFFFFFu
#define MAN_PREV_PNP_SHIFT 28
#define MAN_PREV_REUSED    0x80000000u
#define SGP_COLOUR_WIDE_MIN  2048u   // a colouring round with fewer uncoloured manifolds than this runs inside k_colour_finish (one workgroup)

// kernel classes for the per-kernel profile
enum {
	KC_APPLY_FORCES = 0, KC_BP_CELL, KC_BP_SCAN, KC_BP_SCATTER, KC_BP_PAIRS, KC_BP_LARGE, KC_NARROWPHASE, KC_WAKE,
	KC_COLOUR_CLAIM, KC_COLOUR_COMMIT, KC_COLOUR_COUNT, KC_SETUP, KC_WARM_START, KC_SOLVE_VELOCITY,
	KC_INTEGRATE_POSE, KC_SOLVE_POSITION, KC_FINALIZE, KC_ISLAND_HOOK, KC_ISLAND_FLAG, KC_SLEEP_APPLY, KC_BUOYANCY,
	KC_CACHE_BUILD, KC_MISC, KC_EDIT, KC_GATHER, KC_PREP_BODIES, KC_VEHICLE, KC_COUNT
};

// Dense broad-phase grid of the current step (written by k_bp_grid_params).
// Round 4: the grid is PAGED.  Cells are grouped in 4 x 4 x 4 tiles; a dense page table over the TILES of the bounding box (DV::tile_slot, 1/64 of what a
// dense cell table over the same box would need) maps a tile to a compact slot -- or to nothing: only tiles that hold a body get one, in the order their
// first body arrives --, and cell (x, y, z) lives at slot * 64 + ((z & 3) * 4 + (y & 3)) * 4 + (x & 3) of the cell arrays.  Counts, scans, the cell-sorted
// records and k_bp_pairs' workgroups therefore scale with the OCCUPIED tiles: forty piles cost the same side by side or two kilometres apart (the dense
// grid had to coarsen its cells to fit its table, and walked the empty tiles in between).
#define BP_TILE_NONE    0xFFFFFFFFu
#define BP_TILE_PENDING 0xFFFFFFFEu
struct BpGrid {
	int   min_x, min_y, min_z, max_x, max_y, max_z;   // ordered-int encoded float bounds of the small bodies' AABB centres
	float ox, oy, oz, inv_cell, cell;
	int   nx, ny, nz;                                 // cells of the bounding box (what coordinates clamp to)
	uint32_t n_cells;                                 // != 0: a grid exists
	int   tnx, tny, tnz;                              // tiles of the bounding box = the page table's dimensions
};

// One static triangle mesh in the pools.
struct MeshHeader { uint32_t vert_off, nv, tri_off, nt, node_off, n_nodes; float mnx, mny, mnz, mxx, mxy, mxz; };
// Node of a mesh's bounding-volume tree (node 0 = root).  Inner node (count == 0): children `left`, `right`.  Leaf: triangles
// [left, left + count) of the mesh's tree-ordered triangle array (uint4.w of a triangle = its index in the caller's order).
struct MeshNode { float mnx, mny, mnz; uint32_t left; float mxx, mxy, mxz; uint32_t right; uint32_t count; uint32_t pad[3]; };

// Static large bodies (every static mesh, every static body beyond the broad phase's large-body radius: the buildings of a parcel grid) in a uniform
// grid of their own, built by the host whenever that set changes (rebuild_large_grid): a body sits in every cell its bounds overlap.  What stays
// on the linear large-body list are the moving large bodies and the static ones that would fill too many cells (the ground quad, a terrain).
#define SGP_LG_MAX_CELLS 262144
#define SGP_LG_MAX_SPAN 256        // cells one body may fill
struct LargeGrid { float ox, oy, oz, cell, inv_cell; int nx, ny, nz; uint32_t n_items; uint32_t pad[3]; };

// Device-side counters of one step (read back once per step).
struct StepCounters {
	uint32_t n_pairs;
	uint32_t n_manifolds;        // narrow-phase hits (incl. sensors)
	uint32_t n_constraints;      // manifolds that became contact constraints
	uint32_t n_points;
	uint32_t n_hull_pairs;       // pairs with a convex hull, deferred to k_narrowphase_hull
	uint32_t n_cached;           // manifolds taken from the body-pair contact cache
	uint32_t ucount[2];          // sizes of the two uncoloured worklists (round parity)
	uint32_t rounds_used;        // colouring rounds that found work
	uint32_t n_colours;          // highest used colour + 1 (overflow colour excluded)
	uint32_t pairs_dropped;
	uint32_t manifolds_dropped;
	uint32_t n_active;
	uint32_t n_read_active;
	uint32_t n_export;
	uint32_t n_mesh_pairs[4];    // pairs with a static mesh, deferred to k_narrowphase_mesh: one list per shape of the other body (sphere / box / capsule / hull), so that the
	                             // eight pairs a wave takes at a time run the same code (a sphere is tested in closed form, a box by a separating-axis search with clipping)
	uint32_t n_mesh_big;         // ... of them with more candidate triangles than eight lanes should take (k_narrowphase_mesh<64>)
	uint32_t hc_class[9];        // high-colour components per size class (k_hc_alloc)
	uint32_t hc_entries;         // entries of the component list (classes padded to whole workgroups)
	uint32_t hc_n;               // constraints of the high colours
	uint32_t hc_n_big;           // ... of them in components too large for a workgroup (solved by the catch-all)
	uint32_t bp_dense;           // some tile's halo held more records than the small instance of k_bp_pairs stages in LDS
	uint32_t hc_probe_big;       // launch-plan probe: constraints of components too large for a workgroup if one more colour went to the components
	uint32_t hc_done;            // workgroups of the running solve launch that have finished (the last one runs the catch-all and clears it)
	uint32_t n_tiles_used;       // occupied tiles of this step's grid (slots handed out by k_bp_cell)
	uint32_t veh_deferred;       // vehicles that share a movable body (a dynamic body under a wheel, a chassis a wheel stands on) with a vehicle of lower index: solved after the others, in index order
	uint32_t veh_done;           // workgroups of the running vehicle-row launch that have finished (the last one solves the deferred vehicles and clears it)
	// in-step activation (k_wake_pairs): the pairs of the bodies this step wakes, and where the second narrow-phase round starts in the hull / mesh lists
	uint32_t n_wake_pairs, n_woken, hull_base, mesh_base[4], mesh_big_base;
	uint32_t any_awake;          // somebody (dynamic or kinematic) is awake in this step (plain store of 1 by k_pre_solve).  A step nobody is awake in is the identity and leaves the
	                             //   contact cache as it found it: no table wipe (k_island_mark), no rebuild, and the buffer parity goes back (k_cache_build) -- a pile that fell
	                             //   asleep as a whole finds its contacts when it wakes (round 6; VERDICT r05 'a pile that wakes later starts cold')
	uint32_t wake_any;           // some sleeping body was marked for wake-up this step (plain store of 1: k_wake_pairs has nothing to do otherwise)
	uint32_t tickets[4];         // last_block(): workgroups of k_colour_count / k_warm_bodies / k_cache_build that have finished
	uint32_t ts_error;           // tile solver: a tile gave up waiting for a neighbour (k_step_end copies ts_flags[0])
	uint32_t ts_all_adjacent;    // tile solver: some body was touched by more than four tiles
	uint32_t round_n[32];        // uncoloured manifolds at the start of colouring round r (the host plans the next step's wide rounds from it)
	uint32_t colour_count[SGP_MAX_COLOURS];
	uint32_t colour_fill[SGP_MAX_COLOURS];
	// (colour, point-count class) buckets: a colour's slots are laid out by point count (4, 3, 2, <= 1 points: the longest first) so that a wave holds manifolds of one
	// length -- a wave runs as long as its longest manifold, and unsorted nearly every wave of a mixed pile held a four-point one
	uint32_t cnp_count[SGP_MAX_COLOURS * 4];
	uint32_t cnp_start[SGP_MAX_COLOURS * 4];
	uint32_t cnp_fill[SGP_MAX_COLOURS * 4];
};

// Event counters: NOT cleared at step start (edits between steps also raise activation events); the host drains them.
struct EventCounters {
	uint32_t n_activated;
	uint32_t n_deactivated;
	uint32_t n_water;
	uint32_t n_contact_added;
	uint32_t n_contact_persisted;
	uint32_t pad[3];
};

// Host -> device body edit command (applied in order, one thread per body run).
#define CMD_SET_POS      (1u << 0)
#define CMD_SET_ROT      (1u << 1)
#define CMD_SET_VEL      (1u << 2)
#define CMD_SET_SHAPE    (1u << 3)
#define CMD_ADD_FORCE    (1u << 4)
#define CMD_ADD_TORQUE   (1u << 5)
#define CMD_ADD_FORCE_AT (1u << 6)
#define CMD_ACTIVATE     (1u << 7)
#define CMD_SET_LAYER    (1u << 8)
#define CMD_REMOVE       (1u << 9)
#define CMD_MOVE_KINEMATIC (1u << 10)
#define CMD_CREATE       (1u << 11)
#define CMD_SET_CHASSIS  (1u << 12)   // flags & BF_CHASSIS = the new value

// Per-step pose refresh of an existing ghost body (tiles): what a SET_POS | SET_ROT | SET_VEL | ACTIVATE command does, in 56 bytes
struct GhostRefresh { uint32_t id; float pos[3]; float rot[4]; float linv[3]; float angv[3]; };

struct BodyCmd {
	uint32_t id;
	uint32_t ops;
	float pos[3];      // SET_POS / MOVE_KINEMATIC target / ADD_FORCE_AT point
	float rot[4];
	float linv[3];     // SET_VEL / ADD_FORCE(_AT) force
	float angv[3];     // SET_VEL / ADD_TORQUE torque
	float shape[4];
	float dt;          // MOVE_KINEMATIC
	// CREATE only
	float inv_mass, mass;
	float inv_inertia[3];
	float friction, restitution, gravity_factor, lin_damp, ang_damp;
	uint32_t flags;
	uint32_t pad_;
	uint64_t userdata;
};

// Per-step scalars, device resident: kernels read them through DV::sp so that the launch arguments of a step never
// change and the whole step can be replayed as a hipGraph.  The host writes the pinned copy, a copy node uploads it.
struct StepParams {
	float    dt;
	uint32_t n_slots;          // high-water body slot count
	uint32_t n_large;
	float    cell_size;        // requested broad-phase cell edge = bp_rmax + speculative margin (the grid may coarsen it)
	float    bp_rmax;          // largest bounding radius of the small bodies
	int      water_enabled; float water_z;
	int      contact_events;
	uint32_t parity;           // which of DV::ca[] is the current constraint buffer (the other one is the contact cache)
	uint32_t compact_rows;     // row layout of the velocity iterations: 0 full (192 B per point), 1 r x axis only (96 B; I (r x axis) rebuilt by the lane that needs it), 2 none (everything rebuilt from the lever arms r1b / r2e and efft): bandwidth-bound worlds
	uint32_t pad[6];
};

// Constraint (contact manifold) SoA, double buffered (current step / previous step = contact cache).
// Round 6 measured what these kernels are bound by (profiles/r06_pmc_calibration.md, profiles/r06d_sq_counters.md): a gather moves a whole 128-byte line
// whatever it asks for, and a kernel with a thread per manifold is paced by the NUMBER of per-lane 16-byte accesses its waves issue (every scattered
// access is a tag look-up of its own), not by the lines behind them.  So: nothing is stored twice, the point count of the previous constraint travels in
// DV::man_prev (no header gather), body ids and np_col are one 16-byte header (one load in the solver, one store in the set-up), and what only the
// body-pair contact cache reads is one 64-byte record per slot (three accesses, one line).
#define PREC_F4 4       // float4 per slot of ConstraintArrays::prec: [0] relative rotation conj(q1) q2, [1] (relative position in body 1's frame xyz, normal in body 2's frame x),
                        // [2] (normal-in-2 y, z, -, -), [3] -: the pose of body 2 relative to body 1 and the normal WHEN THE MANIFOLD WAS COMPUTED (polytope pairs only)
struct ConstraintArrays {
	uint4*    hdr;         // body ids a < b (the pair key is a << 32 | b), np_col = np | colour << 8 | persisted << 16 (| NPCOL_CATCH_ALL), -
	float4*   n_fric;      // normal xyz, combined friction w
	float4*   r1b[4];      // r1 xyz, bias w
	float4*   r2e[4];      // r2 xyz, eff_n w
	float4*   lam[4];      // lam_n, lam_t1, lam_t2, -
	float2*   efft[4];     // eff_t1, eff_t2
	float4*   loc1[4];     // contact point in body-1 frame
	float4*   loc2[4];     // contact point in body-2 frame
	float4*   prec;        // the body-pair contact cache's record (above), PREC_F4 per slot
};

#define POSE_F4 4u            // float4 per body in DV::pose (pose record + property record)
#define VEL_F4 4u             // float4 per body in DV::vel (velocity record + the step's world inverse inertia)
struct DV {
	StepParams* sp;
	uint32_t cap_bodies, cap_pairs, cap_manifolds;
	// bodies
	// Per-body state as 64-byte records (four float4 each, record i at [4 i] .. [4 i + 3]), one array per purpose: a constraint gathers ONE 128-byte line per
	// body and purpose (a gather moves a whole line whatever it asks for: profiles/r06_pmc_calibration.md), a sweep kernel streams the records it needs.
	float4* pose;              // POSE_F4 float4 per body: [position xyz, inverse mass (0 unless dynamic)] [rotation quaternion] -- the position iterations correct these two in
	                           //   place -- then the properties, [+2] = [local inverse inertia diagonal xyz, restitution], [+3] = [shape parameters xyz, friction]: the narrow phase,
	                           //   k_setup and the position iterations read pose and properties together (round 6: one gathered line per body instead of two)
	float4* vel;               // 64-byte records (VEL_F4 float4 per body, record i at [VEL_F4 i]): [linear velocity xyz, EFFECTIVE inverse mass of the step (0 unless
	                           //   dynamic and awake; k_pre_solve)] [angular velocity xyz, -]: THE velocity storage, what the velocity iterations gather and scatter; then the
	                           //   world inverse inertia of the step, [+2] = (xx, xy, xz, yy), [+3] = (yz, zz, -, -), written by k_pre_solve when the step's rows are compact
	                           //   (StepParams::compact_rows != 0) for bodies that can move: a lane rebuilding I (r x axis) finds it in the SAME 128-byte line as the velocities it
	                           //   gathers anyway (round 6: at 1 M bodies a velocity launch moved 784 B per constraint for 340 algorithmic -- four gathered lines, two now)
	float4* dyn;               // one float4 per body: linear damping, angular damping, gravity factor, inverse mass (again; k_pre_solve reads nothing else of the pose)
	float4* force;             // accumulated force xyz, - (read only for bodies flagged BF_HAS_FORCE)
	float4* torque;            // accumulated torque xyz, mass w
	uint32_t* flags;
	float4* aabb_min;
	float4* aabb_max;
	float4* sleep_s[3];        // sleep test spheres: centre xyz, radius w
	float*  sleep_timer;
	uint32_t* sleep_label;     // LABEL(slot, generation): the island a body fell asleep with (the root its union-find ended with; itself from creation): bodies that share it wake together
	uint32_t* slot_gen;        // [slot] how often the slot has been given to a new body (7 bits used): a label made from a slot is current only while that body lives in it --
	                           //   a body created in the slot a removed island root left must not share the wake label of that island's sleepers (round 6)
	uint32_t* label_wake;      // [label] = the step epoch (*veh_epoch) in which a body with that label was woken
	uint2*  wake_pairs; uint32_t cap_wake_pairs;      // pairs of the woken bodies with what was not awake when the step began
	float*  submerged;
	uint64_t* userdata;        // mUserData of the body (the caller's PhysicsObject*): travels with the body when its ownership migrates to another tile
	uint64_t* colour_mask;
	float4* warm;              // [body][colour][2]: what the warm start of the body's constraint of that colour adds to its velocity record (linear, angular), valid where
	                           //   colour_mask[body] has the bit (k_setup writes it, k_warm_bodies adds a body's records in colour order: a body's colours are its lowest ones, so
	                           //   its records are a few consecutive 128-byte lines -- round 5: ~19 scattered lines per body through a (body, colour) -> slot table)
	uint64_t* claim[2];
	uint32_t* island;
	uint32_t* island_awake;
	uint32_t* export_counts;   // per 256-body block: bodies the tile export picks (k_export_count)
	uint32_t* awake_mark;      // per body: 1 = sleepy but known to stay awake this step (k_island_mark)
	// tile solver (k_ts_*): ts_nt tiles = workgroups of the one resident velocity-iteration launch, a ts_gx x ts_gy grid over the world (0: unavailable)
	uint32_t ts_nt, ts_gx, ts_gy;
	uint8_t*  body_tile;       // tile of every body (k_ts_label)
	uint64_t* body_tiles4;     // up to four tile ids (16 bits each, 0xFFFF free): the tiles whose constraints touch the body this step
	uint8_t*  man_tile;        // tile of every manifold's constraint
	uint32_t* ts_count;        // [colour * ts_nt + tile] constraints; ts_start: their first slot (+ 1 entry: the total); ts_fill: scratch of the slot assignment
	uint32_t* ts_start;
	uint32_t* ts_fill;
	uint32_t* ts_adj;          // [tile][8]: bit U set = this tile and tile U touch a common body
	uint64_t* ts_wait;         // [tile]: colours in which the tile touches a shared body (it waits for its neighbours before those phases)
	uint32_t* ts_epoch;        // [tile * 32]: phases the tile has completed (one 128 B line per tile)
	uint32_t* ts_flags;        // [0] a tile gave up waiting (the step fails), [1] some body is touched by more than four tiles (all tiles neighbours)
	uint16_t* ts_at;           // [2 * slot + side]: where that side's body lives during the launch: LDS slot, TS_AT_SHARED or TS_AT_IMMOVABLE
	uint8_t*  ts_side;         // [2 * slot + side]: 0 immovable, 1 private to the constraint's tile, 2 shared
	// broad phase
	uint32_t table_size;       // power of two
	uint32_t* cell_hash;       // per body: linear cell index in the dense grid (0xFFFFFFFF = not binned)
	uint32_t* cell_count;      // per bucket (+1)
	uint32_t* cell_start;      // per bucket (+1), exclusive scan of cell_count
	uint32_t* cell_fill;
	uint32_t* scan_block_sums;
	float4*   sorted_min;      // cell-sorted copy: aabb min xyz, flags (bits) w
	float4*   sorted_max;      // cell-sorted copy: aabb max xyz, body id (bits) w
	struct BpGrid* grid;       // per-step dense grid parameters (device)
	int*      bounds_acc;      // [6] ordered-int min / max of the small bodies' AABB centres, accumulated by k_step_begin, consumed by k_bp_cell, reset by k_bp_scatter
	uint32_t* grid_cells_used; // cells of the most recent grid = 64 x its occupied tiles (what the next step has to clear of the cell tables)
	uint32_t* tile_slot;       // the page table: [tile of the bounding box] -> slot, BP_TILE_NONE, or BP_TILE_PENDING while its first body fetches a slot
	uint32_t* tile_of_slot;    // [slot] -> tile (index into tile_slot): the tiles k_bp_pairs walks, and what the next step resets of the page table
	uint32_t  tile_table_size; // entries of tile_slot
	const uint32_t* large_ids;
	uint2*    pairs;
	// narrow phase output (manifolds, unordered)
	uint2*    man_ab;
	float4*   man_n;           // normal xyz, np (as int bits) w
	float4*   man_p1[4];
	float4*   man_p2[4];
	int32_t*  man_colour;      // -1 uncoloured, -2 not a constraint (sensor)
	uint32_t* ulist[2];        // worklists of still-uncoloured manifolds, double buffered by round parity
	uint64_t* man_prio;
	uint32_t* man_slot;        // constraint slot of the manifold (k_setup_slots)
	uint32_t* man_prev;        // slot of the same pair's constraint in the previous step's buffer (MAN_PREV_NONE if none), bit 31: the manifold was
	                           // taken from the body-pair contact cache (k_narrowphase) -- one hash look-up per manifold, shared by every later kernel
	// high colours by connected component (k_hc_*)
	uint32_t* hc_root;         // [cap_bodies] union-find parent (bodies of the high colours that can move)
	uint32_t* hc_count;        // [cap_bodies] constraints of the component (at its root)
	uint32_t* hc_base;         // [cap_bodies] class << 28 | index within the class (at its root), HC_BIG: catch-all
	uint32_t* hc_rank;         // [cap_manifolds] rank of the constraint within its component
	uint32_t* hc_list;         // constraint slot per lane pair of the solve launch (k_hc_scatter) ...
	uint4*    hc_entry;        // ... ordered by colour within a workgroup's share, with the constraint's np_col and bodies (k_hc_sort): what k_solve_hc reads
	uint32_t  cap_hc_list;
	uint32_t* hc_big_list;     // constraints of components too large for a workgroup (k_hc_scatter; up to HC_BIG_LIST = 1024 of them)
	// constraints
	uint32_t dbg_flags;        // SGP_DEBUG_FLAGS (developer switches): bit 0 = tail kernel without its register-resident path
	float4* rows;              // velocity-iteration rows, [point 0..3][axis n,t1,t2][4][cap_manifolds] (k_setup): r1 x axis (w: bias for n),
	                           //   r2 x axis (w: effective mass of the axis), I1 (r1 x axis), I2 (r2 x axis)
	ConstraintArrays ca[2];    // [sp->parity] = this step's constraints, the other = previous step's (contact cache)
	uint4* ht; uint32_t ht_size;   // contact cache: 16-byte entries (pair key low, high, slot in the previous step's constraints, its np_col) -- one line per probe, and the
	                           //   colour a manifold may inherit comes with the probe (ht_size: allocated entries, a power of two); key ~0 = empty
	uint32_t* cache_total;     // [2] per constraint buffer (parity): entries the contact cache holds in it -- the step's constraints [0, n_constraints) and behind them the contacts of
	                           //   sleeping pairs carried over from step to step (k_cache_build; never solved, counted or reported): set to n_constraints by k_island_mark, grown by the carry
	uint32_t* ht_cur;          // entries the last rebuild used (device scalar, a power of two <= ht_size): what look-ups mask with
	uint32_t* cstarts;         // [SGP_MAX_COLOURS + 1] first constraint slot of every colour (device-side exclusive scan)
	// counters / events
	StepCounters* ctr;
	EventCounters* evc;
	uint32_t* ev_activated; uint32_t* ev_deactivated; uint32_t* ev_water;
	sgp_contact_event* ev_contacts_added; sgp_contact_event* ev_contacts_persisted; uint32_t cap_contact_events;
	// convex hull shapes (sgp_device_hull.h): fixed-capacity table, hull 0 = the +-1 cube template every box is a scaled copy of
	const struct sgd_hull_s* hulls; uint32_t n_hulls;
	uint2* hull_pairs; uint32_t cap_hull_pairs;
	struct HullWork* hull_work;      // [cap_hull_pairs] pair + result of the axis search
	// static triangle meshes: headers + pooled vertices / triangles / tree nodes (mesh frame = body frame)
	const struct MeshHeader* meshes; uint32_t n_meshes;
	const float4* mesh_verts; const uint4* mesh_tris; const uint32_t* mesh_tri_mat; const struct MeshNode* mesh_nodes;     // mesh_tri_mat: user data (material index) per tree-ordered triangle
	const LargeGrid* lgrid; const uint32_t* lg_start; const uint32_t* lg_items;      // the static large bodies' grid (cell c: items [lg_start[c], lg_start[c + 1]))
	uint2* mesh_pairs; uint32_t cap_mesh_pairs; uint32_t* mesh_big;      // mesh_pairs: four lists of cap_mesh_pairs each (by the other body's shape); mesh_big: indices into mesh_pairs
	// wheeled vehicles (sgp_device_vehicle.h): AoS, one record per vehicle slot
	struct sgd_vehicle* vehicles; uint32_t n_vehicles; const sgp_vehicle_input* vehicle_inputs;
	// the rows of the step as the solver passes read them (k_vehicle_controller exports, veh_quad_solve consumes): 16 chunks per wheel, [chunk][4 vehicle + wheel]; 5 float4 per vehicle
	float4* veh_rows; float4* veh_head; uint32_t veh_cap;
	// the movable bodies a vehicle's rows act on this step (its chassis, the dynamic bodies under its wheels): [body] = step epoch << 32 | ~vehicle index, by
	// atomicMax (k_vehicle_cast) -- the lowest vehicle index of the current step wins, entries of earlier steps lose against any of this step (no clearing);
	// the contact colouring keeps such a body out of colour 0, a vehicle that lost a claim is deferred (veh_defer_bits, one bit per vehicle slot)
	uint64_t* veh_claim; uint32_t* veh_epoch; uint32_t* veh_defer_bits;
	// settings (fixed after world creation)
	sgp_settings st;
	float gx, gy, gz;
};

// ---- launch wrappers (each defined at the end of the stage file that holds its kernels) ---------------------------------------------------------------
// `nb` = number of body slots the per-body grids must cover (a bucketed upper bound of StepParams::n_slots)
// first / last launch of a step: per-step scalars in by value + scratch reset; counters out to host-mapped memory
void launch_step_begin(const DV& d, const StepParams& sp, uint32_t nb, bool reset_step_scratch, hipStream_t s);
void launch_set_params(const DV& d, const StepParams& sp, hipStream_t s);
void launch_step_end(const DV& d, StepCounters* host_mapped, EventCounters* host_events, hipStream_t s);
void launch_fill_u64(uint64_t* p, uint64_t v, size_t n, hipStream_t s);
void launch_pre_solve(const DV& d, uint32_t nb, hipStream_t s);      // wake-ups + forces + per-step solver records (after the narrow phase)
void launch_bp_bounds(const DV& d, uint32_t nb, hipStream_t s);
void launch_bp_cell(const DV& d, uint32_t nb, hipStream_t s);
void launch_bp_scan(const DV& d, hipStream_t s);
void launch_bp_scatter(const DV& d, uint32_t nb, hipStream_t s);
void launch_bp_pairs(const DV& d, int small_lds, hipStream_t s);      // small_lds: the instance with room for 4 workgroups per compute unit (sparse scenes)
void launch_bp_large(const DV& d, uint32_t nb, hipStream_t s);
void launch_bp_scatter_large(const DV& d, uint32_t nb, hipStream_t s);      // both in one launch (the step's path)
void launch_narrowphase(const DV& d, uint32_t n_pairs_upper, hipStream_t s);
void launch_wake_round(const DV& d, uint32_t nb, int has_hulls, bool has_meshes, hipStream_t s);      // in-step activation: k_wake_pairs + the narrow phase of its pairs
void launch_narrowphase_hull(const DV& d, bool big_hulls, hipStream_t s);     // only worlds with hull shapes; big_hulls: some hull has more than 32 vertices (k_narrowphase_hull_big)
void launch_narrowphase_mesh(const DV& d, bool has_hulls, hipStream_t s);
void launch_narrowphase_mesh_blocks(const DV& d, bool has_hulls, uint32_t blocks, hipStream_t s);      // (the same four launches with a grid of `blocks` workgroups)     // only worlds with mesh shapes; has_hulls: also the instances for hull bodies
void launch_colour_inherit(const DV& d, uint32_t n_man, hipStream_t s);
void launch_colour_claim(const DV& d, uint32_t n_man, uint32_t round, hipStream_t s);
void launch_colour_commit(const DV& d, uint32_t n_man, uint32_t round, hipStream_t s);
void launch_colour_count(const DV& d, uint32_t n_man, hipStream_t s);
void launch_colour_finish(const DV& d, uint32_t first_round, int build_list, hipStream_t s);
void launch_setup(const DV& d, uint32_t n_man, hipStream_t s);
void launch_ts_label(const DV& d, uint32_t nb, hipStream_t s);
void launch_colour_count_ts(const DV& d, uint32_t n_man, hipStream_t s);
void launch_setup_ts(const DV& d, uint32_t n_man, hipStream_t s);
void launch_ts_solve(const DV& d, int passes, int colour_end, hipStream_t s);      // colours < colour_end, `passes` times
// mode: 0 warm start, 1 velocity iteration, 2 position iteration.  est = expected constraints of that colour (grid sizing only)
void launch_solve_colour(const DV& d, int colour, uint32_t est, int mode, hipStream_t s, int compact_rows);      // compact_rows: StepParams::compact_rows of the step (selects the kernel)
void launch_solve_probe(const DV& d, int variant, int colour, uint32_t est, hipStream_t s);
void launch_warm_bodies(const DV& d, uint32_t nb, hipStream_t s);
void launch_solve_tail(const DV& d, int first_colour, int mode, hipStream_t s, int compact_rows = 0);
// small worlds: warm start + all velocity iterations in one single-workgroup launch (needs n_slots <= SGP_SMALL_WORLD_BODIES)
#define SGP_SMALL_WORLD_BODIES 2048
void launch_hc_build(const DV& d, int first_colour, uint32_t est, hipStream_t s);      // components of the colours >= first_colour (after launch_setup)
void launch_hc_probe(const DV& d, int probe_colour, uint32_t probe_est, hipStream_t s);      // before launch_hc_build: component sizes if the components started at probe_colour
void launch_solve_hc(const DV& d, int first_colour, uint32_t est, int mode, hipStream_t s, int compact_rows);   // one pass over them + the overflow colour
void launch_solve_small(const DV& d, int warm_start, int iterations, int lane_pairs, hipStream_t s);      // lane_pairs: two lanes per constraint (<= 384 constraints stay in registers), else one (<= 512)
void launch_integrate_pose(const DV& d, uint32_t nb, hipStream_t s);
void launch_finalize(const DV& d, uint32_t nb, hipStream_t s);
void launch_island_mark(const DV& d, uint32_t n_con, int clear_cache, hipStream_t s);      // clear_cache: the launch also empties the contact-cache table (first round)
void launch_island_hook(const DV& d, uint32_t n_con, hipStream_t s);
void launch_island_flag(const DV& d, uint32_t n_con, hipStream_t s);
void launch_sleep_apply(const DV& d, uint32_t nb, hipStream_t s);
void launch_buoyancy(const DV& d, uint32_t nb, hipStream_t s);
void launch_cache_build(const DV& d, uint32_t n_con, StepCounters* host_mapped, EventCounters* host_events, hipStream_t s);      // + the step's counters to the host (its last workgroup)
void launch_contact_events(const DV& d, uint32_t n_man, hipStream_t s);
void launch_ghost_refresh(const DV& d, const GhostRefresh* recs, uint32_t n, hipStream_t s);
void launch_apply_cmds(const DV& d, const BodyCmd* cmds, const uint32_t* run_start, uint32_t n_runs, hipStream_t s);
void launch_gather_aabbs(const DV& d, const uint32_t* ids, uint32_t n, float4* out, hipStream_t s);      // out[2 k], out[2 k + 1] = bounds of body ids[k]
void launch_gather_states(const DV& d, const uint32_t* ids, uint32_t first, uint32_t n, sgp_body_state* out, hipStream_t s);
void launch_gather_active(const DV& d, uint32_t nb, sgp_body_state* out, uint32_t cap, hipStream_t s);
void launch_gather_active_poses(const DV& d, uint32_t nb, void* out, uint32_t cap, hipStream_t s);      // sgp_body_pose records (32 B)
void launch_dump_constraints(const DV& d, uint32_t which, uint32_t n_con, void* out, uint32_t cap, hipStream_t s);
void launch_vehicle_pre(const DV& d, bool cylinder_testers, hipStream_t s);
void launch_vehicle_solve(const DV& d, int mode, hipStream_t s);
void launch_solve_colour_veh(const DV& d, int colour, uint32_t est, int mode, hipStream_t s, int compact_rows);      // a contact colour whose first workgroups solve the vehicles' rows (mode 1, 2)      // mode as launch_solve_colour
// Single-query mailbox (round 5): PhysicsWorld::traceRay is called one ray at a time by unchanged callers (ParticleManager.cpp:164: up to 2048 per frame;
// HoverCarPhysics.cpp:348), and a launch + a host sync per ray is ~23 us.  While such a caller is at it, ONE resident wave (k_ray_server) answers rays
// handed over through this block of host-mapped, coherent memory: the host writes the ray and bumps `req_seq`, the wave (polling with system-scope
// loads) traces it with the very function k_raycast runs per thread and publishes the hit + `done_seq`.  The wave leaves when told to (`stop`, set by
// whatever next touches the world's stream) or after `idle_ticks` without a request, so nothing ever waits for it longer than that.
struct RayMailbox {
	// line 0 (64 B), host -> device: the wave reads it with ONE 64-byte load (lane l < 16 takes word l), so a new `req_seq` comes with its ray (the host
	// writes the ray first; a cache line is read as a whole)
	uint32_t req_seq, stop_gen; sgp_ray ray; uint32_t pad0[16 - 2 - sizeof(sgp_ray) / 4];      // stop_gen: every server of this generation or older must leave (never reset: a new server has a newer generation)
	// line 1 (64 B), device -> host: written with ONE 64-byte store, the sequence number at both ends (the host takes the hit when both match)
	uint32_t done_seq, exited_gen; sgp_hit hit; uint32_t pad1[16 - 2 - sizeof(sgp_hit) / 4 - 1]; uint32_t done_seq2;      // exited_gen: generation of the last server that left
};
static_assert(sizeof(RayMailbox) == 128 && sizeof(sgp_ray) == 36 && sizeof(sgp_hit) == 48, "RayMailbox: two 64-byte lines");
void launch_ray_server(const DV& d, RayMailbox* mb, uint32_t first_seq, uint32_t generation, uint64_t idle_ticks, uint64_t max_ticks, hipStream_t s);
void launch_raycast(const DV& d, const sgp_ray* rays, uint32_t n, sgp_hit* hits, hipStream_t s);
void launch_collide_capsules(const DV& d, const sgp_capsule_query* q, uint32_t n, sgp_query_contact* out, uint32_t cap, uint32_t* count, hipStream_t s);
void launch_spherecast(const DV& d, const sgp_ray* rays, const float* radii, uint32_t n, sgp_hit* hits, hipStream_t s);
void launch_export_boundary(const DV& d, uint32_t nb, float3 lo, float3 hi, float margin, sgp_ghost_record* out, uint32_t cap, uint32_t* count, hipStream_t s);

// ---- tile exchange with the routing on the device (sgp_tiles_*) ----------------------------------------------------
#define SGP_MAX_TILES 64
// by-value kernel argument: every tile's region + this tile's rank
struct TileRoute { float boxes[6 * SGP_MAX_TILES]; uint32_t n_tiles, my_rank; float margin, pad; };
// what the routing kernels leave for the host (and for the counts all-gather): records per destination, where each destination's segment
// starts in the send buffer, how many owned bodies emigrate
struct RouteHeader { uint32_t seg_count[SGP_MAX_TILES]; uint32_t seg_start[SGP_MAX_TILES]; uint32_t n_emigrants; uint32_t total; uint32_t pad[2]; };
// Re-tiling by body count (sgp_tiles_rebalance): the split planes found so far + what the histogram kernel needs
#define SGP_TILE_HIST_BINS 1024
struct TilePlanes { uint32_t gx, gy, gz, by_contacts; float glo[3], ghi[3]; float xp[4]; float yp[16]; };      // x planes [gx - 1]; y planes of x slab ix at [4 ix ..]
// level 0: bounds of the owned dynamic bodies' centres (six ordered ints, atomicMin / Max); level 1 / 2 / 3: histogram of x / y (per x slab) / z (per (x, y) column)
void launch_tiles_hist(const DV& d, uint32_t nb, const TilePlanes& tp, int level, uint32_t* out, hipStream_t s);
// block_counts / block_offsets: [block][n_tiles + 1] (last column: emigrants)
// gather_row: what this rank contributes to the exchange's all-gather, [seg_count[0 .. n_tiles), status, cap_recv]: status = host_status if that is
// not ROUTE_OK, else ROUTE_REDO when the records or emigrants do not fit cap / emigrant_cap (decided on the device: no host round trip before the collective)
#define ROUTE_OK 0u
#define ROUTE_REDO 1u       // this rank must grow a buffer and route again
#define ROUTE_FAILED 2u     // this rank cannot take part any more (a growth failed): every rank returns an error from this exchange
void launch_route_export(const DV& d, uint32_t nb, const TileRoute& t, uint32_t* block_counts, uint32_t* block_offsets, RouteHeader* header,
                         sgp_ghost_record* out, uint32_t cap, uint32_t* emigrant_ids, uint32_t emigrant_cap, uint32_t* gather_row, uint32_t cap_recv, uint32_t host_status, hipStream_t s);
// ghost pose refresh straight from received records: record k refreshes body ids[k] (the unchanged-ghost-set fast path, no host copy of the poses)
void launch_pack_ghost_keys(const sgp_ghost_record* recs, uint32_t n, void* out_uint4, void* aux_uint4, const float* region_lo, const float* region_hi, hipStream_t s);
// bodies created on the device from received records: list = uint4 (record index, body slot, flags, -) per newcomer (the host chose the slots)
void launch_create_from_records(const DV& d, const sgp_ghost_record* recs, const void* list_uint4, uint32_t n, float gravity_factor, float lin_damp, float ang_damp, hipStream_t s);
#define GKEY_VALID      (1u << 0)
#define GKEY_SHAPE_SHIFT 1
#define GKEY_FLAGS_SHIFT 4
#define GKEY_IN_REGION  (1u << 10)
void launch_ghost_refresh_records(const DV& d, const sgp_ghost_record* recs, const uint32_t* ids, uint32_t n, hipStream_t s);
