// sgp_world.hip -- host side of libsgp.so: the C ABI of include/sgp.h over the gfx950 kernels.
//
// Replaces what gui_client/PhysicsWorld.cpp does with Jolt: world construction (:462-532), addObject (:1169-1311),
// setters (:546-722), think (:1356-1443), activation / contact bookkeeping (:1448-1520), ray queries (:1668-1725).
// The job system / temp allocator adapters (:288-457) have no counterpart: work is HIP launches on one stream,
// scratch lives in device arenas allocated once per world.
//
// There is no CPU compute path in this file: without a HIP device every entry point fails with SGP_ERR_NO_DEVICE.
#include "sgp_world_internal.h"

thread_local std::string g_last_error;
int g_device_count = -1;

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
	DEV_ALLOC(c.hdr, cap); DEV_ALLOC(c.n_fric, cap);
	DEV_ALLOC(c.prec, (size_t)cap * PREC_F4);      // the body-pair contact cache's records: 64 bytes per slot
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
	if (desc->max_bodies >= (1u << 25)) return fail(SGP_ERR_CAPACITY, "sgp_world_create: max_bodies must be below 2^25 (the cell arrays hold 64 cells per body slot in 32-bit indices)");      // (64 N and its power of two overflowed silently)
	sgp_world* w = new sgp_world();
	w->desc = *desc;
	w->device = desc->device;
	if (w->desc.large_body_radius <= 0.0f) w->desc.large_body_radius = 4.0f;
	if (w->desc.max_body_pairs == 0) w->desc.max_body_pairs = 16u * desc->max_bodies + 1024u;
	if (w->desc.max_manifolds == 0) w->desc.max_manifolds = std::min(8u * desc->max_bodies + 1024u, MAN_PREV_NONE - 1u);
	if (w->desc.max_manifolds >= MAN_PREV_NONE) { delete w; return fail(SGP_ERR_CAPACITY, "sgp_world_create: max_manifolds must be below 2^28 (a manifold's reference to its previous constraint keeps 28 bits for the slot)"); }
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
	DEV_ALLOC(d.pose, POSE_F4 * (size_t)N); DEV_ALLOC(d.vel, VEL_F4 * (size_t)N); DEV_ALLOC(d.dyn, N);
	DEV_ALLOC(d.force, N); DEV_ALLOC(d.torque, N);
	DEV_ALLOC(d.flags, N); DEV_ALLOC(d.aabb_min, N); DEV_ALLOC(d.aabb_max, N);
	for (int k = 0; k < 3; ++k) DEV_ALLOC(d.sleep_s[k], N);
	DEV_ALLOC(d.sleep_timer, N); DEV_ALLOC(d.submerged, N); DEV_ALLOC(d.userdata, N);
	DEV_ALLOC(d.colour_mask, N); DEV_ALLOC(d.warm, (size_t)N * SGP_MAX_COLOURS * 2); DEV_ALLOC(d.claim[0], N); DEV_ALLOC(d.claim[1], N);
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
	DEV_ALLOC(d.sleep_label, N); DEV_ALLOC(d.label_wake, N); DEV_ALLOC(d.slot_gen, N);
	HIP_TRY(hipMemsetAsync(d.label_wake, 0, sizeof(uint32_t) * N, w->stream));
	HIP_TRY(hipMemsetAsync(d.slot_gen, 0, sizeof(uint32_t) * N, w->stream));
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
	DEV_ALLOC(d.ht, w->ht_alloc);
	HIP_TRY(hipMemsetAsync(d.ht, 0xFF, sizeof(uint4) * w->ht_alloc, w->stream));      // empty contact cache (key ~0)
	d.ht_size = w->ht_alloc;
	DEV_ALLOC(d.ht_cur, 1);
	DEV_ALLOC(d.cache_total, 2); HIP_TRY(hipMemsetAsync(d.cache_total, 0, sizeof(uint32_t) * 2, w->stream));
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
	{ const char* e = getenv("SGP_NO_RAY_SERVER"); if (e && e[0] == '1') w->ray_server_enabled = false; }      // (single rays then cost a launch + a sync each)
	{ const char* e = getenv("SGP_NO_WAKE_ROUND"); if (e && e[0] == '1') w->use_wake_round = false; }      // (measurements only: the CPU statement has its own switch)
	{ const char* e = getenv("SGP_DEBUG_FLAGS"); w->dv.dbg_flags = e ? (uint32_t)atoi(e) : 0u; }
#ifdef SGP_EXPERIMENTS
	{ const char* e = getenv("SGP_TILE_SOLVER"); if (e) w->use_tile_solver = atoi(e); }
#endif
	{ const char* e = getenv("SGP_VEHICLE_FUSED"); if (e) w->fuse_vehicle_solve = atoi(e) != 0; }
	{ const char* e = getenv("SGP_COMPACT_ROWS_MIN"); if (e && atoll(e) >= 0) w->compact_rows_min = (uint32_t)atoll(e); }
	{ const char* e = getenv("SGP_ROWS_MODE"); if (e && (atoi(e) == 1 || atoi(e) == 2)) w->rows_mode_large = (uint32_t)atoi(e); }
	// (differential testing: a world of a few hundred bodies with the row layouts of the large ones -- only without the small-world kernel, which reads full rows)
	{ const char* e = getenv("SGP_ROWS_IN_SMALL_WORLDS"); if (e && e[0] == '1' && !w->use_small_world) w->rows_in_small_worlds = true; }
	{ const char* e = getenv("SGP_ROWS_MODE2_MIN"); if (e && atoll(e) >= 0) w->rows_mode2_min = (uint32_t)atoll(e); }
	{ const char* e = getenv("SGP_ROWS_MODE_DEFAULT"); if (e && atoi(e) >= 0 && atoi(e) <= 2) w->rows_mode_default = (uint32_t)atoi(e); }
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
	ray_server_stop(w);
	if (w->stream) hipStreamSynchronize(w->stream);
	if (w->ray_mb) hipHostFree(w->ray_mb);
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

void ray_server_stop(sgp_world* w)
{
	if (!w->ray_server_on) return;
	__atomic_store_n(&w->ray_mb->stop_gen, w->ray_gen, __ATOMIC_RELEASE);      // (never taken back: the next server is a newer generation)
	w->ray_server_on = false;
}

int flush_cmds(sgp_world* w)
{
	ray_server_stop(w);      // (every entry point that launches on the world's stream comes through here first)
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
int collect_events(sgp_world* w, bool counters_fresh)
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

int read_counters(sgp_world* w)
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
	int      veh_cylinder;       // some vehicle casts its wheels as cylinders: the cast kernel's instance that holds that search (twice the registers)
	int      has_meshes;         // some body may be a static triangle mesh: run the mesh-pair narrow phase
	int      has_hulls;          // some body may be a convex hull: run the hull-pair narrow phase (2: a hull of more than 32 vertices exists -- k_narrowphase_hull_big too)
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
	p.veh_cylinder = w->veh_cylinder_seen ? 1 : 0;
	p.has_hulls = w->hulls.size() > 1 ? (w->n_big_hulls ? 2 : 1) : 0;
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
	if (p.n_vehicles) { KScope k(w, KC_VEHICLE); launch_vehicle_pre(d, p.veh_cylinder != 0, s); }
	{ KScope k(w, KC_BP_PAIRS); launch_bp_pairs(d, p.bp_small, s); }
	STAGE_MARK(2);
	// -- 3. narrow phase, wake-ups, per-body solver records (+ contact events, which see the velocities before the solve)
	{ KScope k(w, KC_NARROWPHASE); launch_narrowphase(d, p.est_pairs, s); if (p.has_hulls) launch_narrowphase_hull(d, p.has_hulls == 2, s); if (p.has_meshes) launch_narrowphase_mesh(d, p.has_hulls != 0, s); }
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
		w->idle_steps++; w->last_step_idle = true;
		// (the contact cache stays as the last step with somebody awake left it: what wakes up later finds the contacts it fell asleep with)
		return SGP_OK;
	}
	w->last_step_idle = false;
	if (w->veh_inputs_dirty && w->n_vehicles) {
		HIP_TRY(hipMemcpyAsync(w->d_veh_inputs, w->veh_inputs.data(), sizeof(sgp_vehicle_input) * w->n_vehicles, hipMemcpyHostToDevice, w->stream));
		w->veh_inputs_dirty = false;
	}
	static const bool timing = getenv("SGP_TIMING") != nullptr;
	static double t_acc[4] = { 0, 0, 0, 0 }; static int t_n = 0;
	auto now = []() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
	const double tt0 = timing ? now() : 0.0;
	w->h_sp->dt = dt;
	w->h_sp->compact_rows = ((w->high <= SGP_SMALL_WORLD_BODIES && !w->rows_in_small_worlds) || w->n_con == 0u) ? 0u : (w->n_con >= w->compact_rows_min ? w->rows_mode_large : (w->rows_mode_default == 2u && w->n_con < w->rows_mode2_min ? 1u : w->rows_mode_default));      // (decided from the previous step's count, part of the plan's key; no constraints: full rows cost nothing and k_pre_solve need not write the inertia records)
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
	if (w->h_ctr->any_awake) w->h_sp->parity ^= 1u;  // the buffer just solved becomes the contact cache of the next step (a step nobody was awake in keeps the one it found: k_cache_build)
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
	out->row_layout = w->h_sp->compact_rows;
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

