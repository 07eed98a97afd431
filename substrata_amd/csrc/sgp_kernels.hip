// sgp_kernels.hip -- hand-written gfx950 kernels of the rigid-body step behind PhysicsWorld::think
// (/root/reference/gui_client/PhysicsWorld.cpp:1356-1443).  Stage map:
//   k_pre_solve           K8a  MotionProperties::ApplyForceTorqueAndDragInternal (+ wake-ups, per-step solver records)
//   k_bp_*                K2/K3 spatial-hash broad phase (uniform grid, hashed buckets) + layer filter (PhysicsWorld.cpp:160-189)
//   k_narrowphase         K4   sphere/box/capsule manifolds (sgp_device_collide.h)
//   k_colour_*, k_setup   K5/K6 contact cache match (warm start), deterministic colouring, constraint properties
//   k_solve_colour<0|1>, k_solve_tail  K7  warm start + sequential impulses, one colour per launch
//   k_integrate_pose      K8b  the body-array sweep: x += v dt, q <- rot(w dt) q
//   k_solve_colour<2>     K7b  Baumgarte position iterations
//   k_finalize, k_island_*, k_sleep_apply  K1 + K9  AABB refresh, sleep spheres, island sleeping
//   k_buoyancy            A3   Substrata's own water sweep (PhysicsWorld.cpp:1367-1442)
// All body state is SoA float4 in HBM; every per-body kernel is a coalesced 16 B/lane sweep.
#include "sgp_kernels.h"
#include <algorithm>
#include "sgp_device_collide.h"
#include "sgp_device_vehicle.h"
#include "sgp_device_mesh.h"

#define TPB 256

// ---------------------------------------------------------------------------------------------------------------
// small helpers

SGP_DEV const ConstraintArrays& CUR(const DV& d) { return d.ca[d.sp->parity & 1]; }
SGP_DEV const ConstraintArrays& PRV(const DV& d) { return d.ca[(d.sp->parity & 1) ^ 1]; }

SGP_DEV uint32_t f_motion(uint32_t f) { return f & BF_MOTION_MASK; }
SGP_DEV uint32_t f_layer(uint32_t f) { return (f & BF_LAYER_MASK) >> BF_LAYER_SHIFT; }
SGP_DEV uint32_t f_shape(uint32_t f) { return (f & BF_SHAPE_MASK) >> BF_SHAPE_SHIFT; }
SGP_DEV bool f_movable(uint32_t f) { return (f & (BF_ALIVE | BF_ACTIVE)) == (BF_ALIVE | BF_ACTIVE) && f_motion(f) == SGP_MOTION_DYNAMIC; }
// Workgroups are dealt to the eight XCDs in turn (each with an L2 of its own).  For a kernel that walks a list whose neighbours share data -- pairs in
// broad-phase tile order, manifolds, a colour's slots -- workgroup b takes chunk xcd_block() instead of b: one XCD then works through a contiguous
// eighth of the list.  The grid must be a multiple of eight.  (Pays in the colour launches, +2 % on config 3; the streaming kernels k_narrowphase and
// k_setup got SLOWER with it -- 85 -> 93 us, 157 -> 188 us -- and keep the plain order.)
SGP_DEV uint32_t xcd_block() { return (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3); }
SGP_DEV bool f_active_for_pairs(uint32_t f) { return (f & (BF_ALIVE | BF_ACTIVE)) == (BF_ALIVE | BF_ACTIVE) && f_motion(f) != SGP_MOTION_STATIC; }

// MyObjectLayerPairFilter, PhysicsWorld.cpp:160-189
SGP_DEV bool layers_collide(uint32_t l1, uint32_t l2)
{
	if (l1 == SGP_LAYER_NON_MOVING) return l2 == SGP_LAYER_MOVING;
	if (l1 == SGP_LAYER_MOVING) return l2 != SGP_LAYER_NON_MOVING_NON_COLLIDABLE && l2 != SGP_LAYER_MOVING_NON_COLLIDABLE;
	return false;
}

SGP_DEV const sgd_hull* body_hull(const DV& d, float4 sh) { return &d.hulls[(uint32_t)sh.x]; }

SGP_DEV v3 shape_local_half(const DV& d, uint32_t type, float4 sh)
{
	if (type == SGP_SHAPE_MESH) return V3(1.0f, 1.0f, 1.0f);        // (static: never asked for sleep points)
	if (type == SGP_SHAPE_HULL) {
		const sgd_hull* h = body_hull(d, sh);
		return V3(fmaxf(fabsf(h->aabb_min.x), fabsf(h->aabb_max.x)), fmaxf(fabsf(h->aabb_min.y), fabsf(h->aabb_max.y)), fmaxf(fabsf(h->aabb_min.z), fabsf(h->aabb_max.z)));
	}
	if (type == SGP_SHAPE_SPHERE) return V3(sh.x, sh.x, sh.x);
	if (type == SGP_SHAPE_BOX) return V3(sh.x, sh.y, sh.z);
	return V3(sh.x, sh.x, sh.y + sh.x);
}

SGP_DEV float shape_volume(const DV& d, uint32_t type, float4 sh)
{
	if (type == SGP_SHAPE_MESH) return 0.0f;
	if (type == SGP_SHAPE_HULL) return body_hull(d, sh)->volume;
	if (type == SGP_SHAPE_SPHERE) return (4.0f / 3.0f) * 3.14159265358979323846f * sh.x * sh.x * sh.x;
	if (type == SGP_SHAPE_BOX) return 8.0f * sh.x * sh.y * sh.z;
	return 3.14159265358979323846f * sh.x * sh.x * (2.0f * sh.y) + (4.0f / 3.0f) * 3.14159265358979323846f * sh.x * sh.x * sh.x;
}

SGP_DEV void compute_aabb(const DV& d, uint32_t type, float4 sh, v3 pos, quat q, v3& mn, v3& mx)
{
	v3 e;
	if (type == SGP_SHAPE_MESH) {
		const MeshHeader mh = d.meshes[(uint32_t)sh.x];
		const m33 R = quat_to_m33(q);
		v3 lo = V3(3.4e38f, 3.4e38f, 3.4e38f), hi = V3(-3.4e38f, -3.4e38f, -3.4e38f);
		for (int k = 0; k < 8; ++k) {
			const v3 c = V3((k & 1) ? mh.mxx : mh.mnx, (k & 2) ? mh.mxy : mh.mny, (k & 4) ? mh.mxz : mh.mnz);
			const v3 p = m33_mul(R, c);
			lo = V3(fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z)); hi = V3(fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z));
		}
		mn = v3_add(pos, lo); mx = v3_add(pos, hi);
		return;
	}
	if (type == SGP_SHAPE_HULL) {
		const sgd_hull* h = body_hull(d, sh);
		const m33 R = quat_to_m33(q);
		v3 lo = V3(3.4e38f, 3.4e38f, 3.4e38f), hi = V3(-3.4e38f, -3.4e38f, -3.4e38f);
		for (int i = 0; i < h->nv; ++i) {
			const v3 p = m33_mul(R, h->verts[i]);
			lo = V3(fminf(lo.x, p.x), fminf(lo.y, p.y), fminf(lo.z, p.z)); hi = V3(fmaxf(hi.x, p.x), fmaxf(hi.y, p.y), fmaxf(hi.z, p.z));
		}
		mn = v3_add(pos, lo); mx = v3_add(pos, hi);
		return;
	}
	if (type == SGP_SHAPE_SPHERE) e = V3(sh.x, sh.x, sh.x);
	else {
		const m33 R = quat_to_m33(q);
		if (type == SGP_SHAPE_BOX) {
			e = V3(fabsf(R.c0.x) * sh.x + fabsf(R.c1.x) * sh.y + fabsf(R.c2.x) * sh.z,
			       fabsf(R.c0.y) * sh.x + fabsf(R.c1.y) * sh.y + fabsf(R.c2.y) * sh.z,
			       fabsf(R.c0.z) * sh.x + fabsf(R.c1.z) * sh.y + fabsf(R.c2.z) * sh.z);
		} else {
			e = V3(fabsf(R.c2.x) * sh.y + sh.x, fabsf(R.c2.y) * sh.y + sh.x, fabsf(R.c2.z) * sh.y + sh.x);
		}
	}
	mn = v3_sub(pos, e);
	mx = v3_add(pos, e);
}

// Body::GetSleepTestPoints
SGP_DEV void sleep_points(const DV& d, uint32_t type, float4 sh, v3 pos, quat q, v3 out[3])
{
	const v3 ext = shape_local_half(d, type, sh);
	const m33 R = quat_to_m33(q);
	int lowest = 0;
	if (ext.y < v3_get(ext, lowest)) lowest = 1;
	if (ext.z < v3_get(ext, lowest)) lowest = 2;
	const int i1 = lowest == 0 ? 1 : 0;
	const int i2 = lowest == 2 ? 1 : 2;
	out[0] = pos;
	out[1] = v3_add(pos, v3_scale(m33_col(R, i1), v3_get(ext, i1)));
	out[2] = v3_add(pos, v3_scale(m33_col(R, i2), v3_get(ext, i2)));
}

SGP_DEV void reset_sleep(const DV& d, uint32_t i, uint32_t type, float4 sh, v3 pos, quat q)
{
	v3 p[3];
	sleep_points(d, type, sh, pos, q, p);
	d.sleep_s[0][i] = F4(p[0], 0.0f);
	d.sleep_s[1][i] = F4(p[1], 0.0f);
	d.sleep_s[2][i] = F4(p[2], 0.0f);
	d.sleep_timer[i] = 0.0f;
}

SGP_DEV void push_event(uint32_t* list, uint32_t* counter, uint32_t cap, uint32_t id)
{
	const uint32_t k = atomicAdd(counter, 1u);
	if (k < cap) list[k] = id;
}

// "The last workgroup to finish does what needs everybody's results": every thread of the workgroup calls this at the end of the kernel's parallel part;
// true (for the whole workgroup) in the workgroup that took the last ticket.  The tickets live in StepCounters (zeroed by the step's first launch) and
// are used once per step each.  A dependent kernel boundary costs ~4.5 us at the launch floor; this costs a fence and an atomic.
SGP_DEV bool last_block(uint32_t* ticket)
{
	__shared__ uint32_t s_last_ticket;
	__syncthreads();
	if (threadIdx.x == 0) { __threadfence(); s_last_ticket = atomicAdd(ticket, 1u); }
	__syncthreads();
	const bool last = s_last_ticket == gridDim.x - 1u;
	if (last) __threadfence();          // what the other workgroups wrote (and this compute unit may still hold older copies of)
	return last;
}

SGP_DEV int float_to_ordered(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
SGP_DEV float ordered_to_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

// First launch of a step: the per-step scalars arrive BY VALUE (no upload node), the per-step counters, grid bounds and
// scratch arrays are reset by this one grid-stride kernel (instead of half a dozen runtime memset nodes).
// Round 4: the same launch also finds the bounds of the small bodies' AABB centres (the broad-phase grid's extent), which was a launch of its own: the first
// `bounds_blocks` workgroups sweep the bodies and fold their six extrema into DV::bounds_acc (ordered-int atomics; reset by k_bp_scatter, which runs after
// the grid parameters have been derived from them).
__global__ void __launch_bounds__(TPB) k_step_begin(DV d, StepParams sp, uint32_t nb, int reset_scratch, uint32_t bounds_blocks)
{
	const uint32_t tid = blockIdx.x * TPB + threadIdx.x, stride = gridDim.x * TPB;
	if (tid == 0) {
		// the buffer parity lives on the device and flips with every step (reset_scratch: a step, not a re-binning between steps): were it a by-value
		// argument, every launch plan would need two captured graphs -- one per parity -- and a plan change would cost two captures
		const uint32_t par = reset_scratch ? (d.sp->parity ^ 1u) : d.sp->parity;
		*d.sp = sp;
		d.sp->parity = par;
		*d.veh_epoch = *d.veh_epoch + 1u;      // (device side: the by-value step parameters are part of a captured graph's key and must not change from step to step)
	}
	uint32_t* c = (uint32_t*)d.ctr;
	for (uint32_t i = tid; i < sizeof(StepCounters) / 4; i += stride) c[i] = 0;
	// the cell tables: only what the previous grid used (everything above it is still zero; the table has room for far more cells than a step uses)
	const uint32_t used = min(*d.grid_cells_used, d.table_size) + 4u;
	for (uint32_t i = tid; i < used; i += stride) { d.cell_count[i] = 0; d.cell_fill[i] = 0; }
	// ... and of the page table: the entries of the tiles that held a slot (everything else already says "none")
	for (uint32_t sl = tid; sl < (used - 4u) / 64u; sl += stride) d.tile_slot[d.tile_of_slot[sl]] = BP_TILE_NONE;
	if (reset_scratch) {
		const uint32_t n = min(nb, d.cap_bodies);
		for (uint32_t i = tid; i < n; i += stride) { d.colour_mask[i] = 0ull; d.claim[0][i] = ~0ull; d.claim[1][i] = ~0ull; }
	}
	if (blockIdx.x >= bounds_blocks) return;          // (workgroup-uniform)
	// (a grid-stride loop over few workgroups: every workgroup ends with six atomics on the same six words, and those serialise)
	float mnx = 3.0e38f, mny = 3.0e38f, mnz = 3.0e38f, mxx = -3.0e38f, mxy = -3.0e38f, mxz = -3.0e38f;
	const uint32_t n_slots = sp.n_slots;
	for (uint32_t i = tid; i < n_slots; i += bounds_blocks * TPB) {
		const uint32_t f = d.flags[i];
		if ((f & BF_ALIVE) && !(f & BF_LARGE)) {
			const float4 mn = d.aabb_min[i], mx = d.aabb_max[i];
			const float cx = (mn.x + mx.x) * 0.5f, cy = (mn.y + mx.y) * 0.5f, cz = (mn.z + mx.z) * 0.5f;
			mnx = fminf(mnx, cx); mny = fminf(mny, cy); mnz = fminf(mnz, cz); mxx = fmaxf(mxx, cx); mxy = fmaxf(mxy, cy); mxz = fmaxf(mxz, cz);
		}
	}
	for (int off = 32; off > 0; off >>= 1) {
		mnx = fminf(mnx, __shfl_down(mnx, off, 64)); mny = fminf(mny, __shfl_down(mny, off, 64)); mnz = fminf(mnz, __shfl_down(mnz, off, 64));
		mxx = fmaxf(mxx, __shfl_down(mxx, off, 64)); mxy = fmaxf(mxy, __shfl_down(mxy, off, 64)); mxz = fmaxf(mxz, __shfl_down(mxz, off, 64));
	}
	__shared__ float red[6][TPB / 64];
	if ((threadIdx.x & 63) == 0) { const int wv = threadIdx.x >> 6; red[0][wv] = mnx; red[1][wv] = mny; red[2][wv] = mnz; red[3][wv] = mxx; red[4][wv] = mxy; red[5][wv] = mxz; }
	__syncthreads();
	if (threadIdx.x < 6) {
		float v = red[threadIdx.x][0];
		for (int k = 1; k < TPB / 64; ++k) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][k]) : fmaxf(v, red[threadIdx.x][k]);
		int* dst = d.bounds_acc + threadIdx.x;
		if (threadIdx.x < 3) { if (v < 2.9e38f) atomicMin(dst, float_to_ordered(v)); }
		else { if (v > -2.9e38f) atomicMax(dst, float_to_ordered(v)); }
	}
}

// Between steps (after adds / edits): refresh the device copy of the per-step scalars only.
__global__ void k_set_params(DV d, StepParams sp) { if (threadIdx.x == 0 && blockIdx.x == 0) *d.sp = sp; }

// Last launch of a step: the counters go straight into host-mapped pinned memory (no copy node).
__global__ void __launch_bounds__(TPB) k_step_end(DV d, StepCounters* host_mapped, EventCounters* host_events)
{
	const uint32_t* src = (const uint32_t*)d.ctr;
	uint32_t* dst = (uint32_t*)host_mapped;
	for (uint32_t i = threadIdx.x; i < sizeof(StepCounters) / 4; i += TPB) dst[i] = src[i];
	__syncthreads();
	if (threadIdx.x == 0 && d.ts_nt) { host_mapped->ts_error = d.ts_flags[0]; host_mapped->ts_all_adjacent = d.ts_flags[1]; }
	if (threadIdx.x < sizeof(EventCounters) / 4) ((uint32_t*)host_events)[threadIdx.x] = ((const uint32_t*)d.evc)[threadIdx.x];
}

__global__ void __launch_bounds__(TPB) k_fill_u64(uint64_t* p, uint64_t v, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (size_t)gridDim.x * TPB) p[i] = v;
}

// ---------------------------------------------------------------------------------------------------------------
// K2/K3: broad phase.  Small bodies are binned by AABB centre into a dense grid of cells whose edge is >= the largest
// small-body AABB (+ speculative margin), so overlapping bodies always sit in adjacent cells.  Bodies are counting-sorted
// into cell order together with a packed 32-byte AABB record; the pair kernel stages a 4x4x4-cell tile plus its halo in
// LDS and tests every body of the tile against the 27 neighbouring cells out of LDS.

// The grid of a step from the bounds k_step_begin accumulated: origin / dimensions; the cell edge grows until the dense table fits.  Every workgroup of
// k_bp_cell derives it for itself (a few dozen flops by one thread; it was a single-thread launch of its own), workgroup 0 also publishes it.
SGP_DEV BpGrid bp_grid_from_bounds(const DV& d)
{
	BpGrid g;
	g.min_x = d.bounds_acc[0]; g.min_y = d.bounds_acc[1]; g.min_z = d.bounds_acc[2]; g.max_x = d.bounds_acc[3]; g.max_y = d.bounds_acc[4]; g.max_z = d.bounds_acc[5];
	g.ox = g.oy = g.oz = 0.0f; g.nx = g.ny = g.nz = 1;
	float cell = d.sp->cell_size;
	if (g.min_x <= g.max_x) {
		const float x0 = ordered_to_float(g.min_x), y0 = ordered_to_float(g.min_y), z0 = ordered_to_float(g.min_z);
		const float x1 = ordered_to_float(g.max_x), y1 = ordered_to_float(g.max_y), z1 = ordered_to_float(g.max_z);
		for (int it = 0; it < 64; ++it) {
			const float inv = 1.0f / cell;
			const float fx = floorf((x1 - x0) * inv) + 1.0f, fy = floorf((y1 - y0) * inv) + 1.0f, fz = floorf((z1 - z0) * inv) + 1.0f;
			const float tx = floorf((fx + 3.0f) * 0.25f), ty = floorf((fy + 3.0f) * 0.25f), tz = floorf((fz + 3.0f) * 0.25f);      // tiles of 4 x 4 x 4 cells
			if (tx * ty * tz <= (float)d.tile_table_size && fx < 2.0e9f && fy < 2.0e9f && fz < 2.0e9f) { g.nx = (int)fx; g.ny = (int)fy; g.nz = (int)fz; break; }
			cell = cell * 1.5f;
			g.nx = g.ny = g.nz = 1;
		}
		g.ox = x0; g.oy = y0; g.oz = z0;
	}
	g.cell = cell; g.inv_cell = 1.0f / cell;
	g.n_cells = 1u;                                   // (a grid exists; how many cells it really has is 64 x the tiles k_bp_cell hands out)
	g.tnx = (g.nx + 3) >> 2; g.tny = (g.ny + 3) >> 2; g.tnz = (g.nz + 3) >> 2;
	return g;
}

// cell (x, y, z) of the paged grid (coordinates inside the bounding box): index into the cell arrays, or BP_TILE_NONE where the tile holds nobody
SGP_DEV uint32_t grid_cell(const DV& d, const BpGrid& g, int x, int y, int z)
{
	const uint32_t slot = d.tile_slot[((uint32_t)(z >> 2) * (uint32_t)g.tny + (uint32_t)(y >> 2)) * (uint32_t)g.tnx + (uint32_t)(x >> 2)];
	return slot >= BP_TILE_PENDING ? BP_TILE_NONE : slot * 64u + (uint32_t)((((z & 3) << 2) | (y & 3)) << 2 | (x & 3));
}
// fn(q0, q1): the cell-sorted records [q0, q1) of cells xa .. xb of row (y, z), tile by tile (the cells of a tile's row are neighbours in the arrays)
template <class F> SGP_DEV void grid_row_runs(const DV& d, const BpGrid& g, int xa, int xb, int y, int z, F fn)
{
	const uint32_t trow = ((uint32_t)(z >> 2) * (uint32_t)g.tny + (uint32_t)(y >> 2)) * (uint32_t)g.tnx;
	const uint32_t lrow = (uint32_t)((((z & 3) << 2) | (y & 3)) << 2);
	for (int x = xa; x <= xb; ) {
		const int xe = min(xb, x | 3);
		const uint32_t slot = d.tile_slot[trow + (uint32_t)(x >> 2)];
		if (slot < BP_TILE_PENDING) { const uint32_t c0 = slot * 64u + lrow + (uint32_t)(x & 3); fn(d.cell_start[c0], d.cell_start[c0 + (uint32_t)(xe - x) + 1u]); }
		x = xe + 1;
	}
}

__global__ void __launch_bounds__(TPB) k_bp_cell(DV d)
{
	__shared__ BpGrid sg;
	if (threadIdx.x == 0) {
		sg = bp_grid_from_bounds(d);
		if (blockIdx.x == 0) *d.grid = sg;
	}
	__syncthreads();
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	const bool in_range = i < d.sp->n_slots;
	const uint32_t f = in_range ? d.flags[i] : 0u;
	uint32_t h = 0xFFFFFFFFu, tile = 0, local = 0;
	bool binned = false;
	const BpGrid& g = sg;
	if ((f & BF_ALIVE) && !(f & BF_LARGE)) {
		const float4 mn = d.aabb_min[i], mx = d.aabb_max[i];
		int cx = (int)floorf(((mn.x + mx.x) * 0.5f - g.ox) * g.inv_cell);
		int cy = (int)floorf(((mn.y + mx.y) * 0.5f - g.oy) * g.inv_cell);
		int cz = (int)floorf(((mn.z + mx.z) * 0.5f - g.oz) * g.inv_cell);
		cx = min(max(cx, 0), g.nx - 1); cy = min(max(cy, 0), g.ny - 1); cz = min(max(cz, 0), g.nz - 1);
		tile = ((uint32_t)(cz >> 2) * (uint32_t)g.tny + (uint32_t)(cy >> 2)) * (uint32_t)g.tnx + (uint32_t)(cx >> 2);
		local = (uint32_t)((((cz & 3) << 2) | (cy & 3)) << 2 | (cx & 3));
		binned = true;
	}
	// The tile's slot: the first body to arrive fetches one (the order is whatever the atomics give: it decides where a tile's cells sit in the arrays and in
	// what order pairs come out, neither of which enters a result).  One lane per wave and tile talks to the page table -- neighbouring body ids often
	// share a tile, and atomics on one address queue (every body for itself: 53 us at 100k bodies) --, the others take its answer.  A lane that asks
	// may have to wait for another WAVE's lane to publish a slot it has claimed; it never waits for a lane of its own wave (their tiles differ).
	// (the wave's distinct tiles all at once: a lane leads its tile if no lower lane has the same one -- a row of a lattice spreads a wave's 64 bodies over
	// ten tiles, and one tile after the other was ten dependent round trips to the page table)
	const int lane = (int)(threadIdx.x & 63u);
	int leader = lane;
	for (int k = 0; k < 64; ++k) {
		const uint32_t tk = (uint32_t)__shfl((int)tile, k, 64);
		const int bk = __shfl((int)binned, k, 64);
		if (bk && binned && tk == tile && k < leader) leader = k;
	}
	uint32_t sl = BP_TILE_NONE;
	if (binned && leader == lane) {
		uint32_t* entry = &d.tile_slot[tile];
		sl = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (relaxed: the slot NUMBER is all that travels through the entry)
		for (int tries = 0; tries < (1 << 24) && sl >= BP_TILE_PENDING; ++tries) {
			const uint32_t old = atomicCAS(entry, BP_TILE_NONE, BP_TILE_PENDING);
			if (old == BP_TILE_NONE) {
				sl = atomicAdd(&d.ctr->n_tiles_used, 1u);
				d.tile_of_slot[sl] = tile;
				__hip_atomic_store(entry, sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			} else if (old != BP_TILE_PENDING) sl = old;
			else sl = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
	const uint32_t slot = (uint32_t)__shfl((int)sl, leader, 64);
	if (binned) {
		if (slot >= BP_TILE_PENDING) { h = 0xFFFFFFFFu; if (lane == leader) atomicAdd(&d.ctr->pairs_dropped, 1u); }      // (the bounded wait above ran out: not binned this step and counted, never an index)
		else { h = slot * 64u + local; atomicAdd(&d.cell_count[h], 1u); }
	}
	if (in_range) d.cell_hash[i] = h;
}

// exclusive scan of cell_count[0..n) -> cell_start, 3 passes, 1024 elements per block
__global__ void __launch_bounds__(TPB) k_scan_blocks(const uint32_t* in, uint32_t* out, uint32_t* block_sums, uint32_t n_cap, const uint32_t* n_live)
{
	__shared__ uint32_t wave_sums[TPB / 64];
	const uint32_t n = min(n_cap, *n_live * 64u + 1u);          // (n_live: the occupied tiles of this step's grid, 64 cells each, + the end sentinel)
	// (a fixed, small grid walking the blocks that hold live cells: the table has room for 64 cells per body, a step uses a fraction of it)
	for (uint32_t blk = blockIdx.x; blk * (TPB * 4u) < n; blk += gridDim.x) {
	__syncthreads();
	const uint32_t base = (blk * TPB + threadIdx.x) * 4;
	uint32_t v[4];
#pragma unroll
	for (int k = 0; k < 4; ++k) v[k] = (base + k < n) ? in[base + k] : 0u;
	const uint32_t tsum = v[0] + v[1] + v[2] + v[3];
	// wave inclusive scan
	uint32_t x = tsum;
	const int lane = threadIdx.x & 63;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
	const int wave = threadIdx.x >> 6;
	if (lane == 63) wave_sums[wave] = x;
	__syncthreads();
	uint32_t wbase = 0;
	for (int k = 0; k < wave; ++k) wbase += wave_sums[k];
	uint32_t excl = wbase + x - tsum;
#pragma unroll
	for (int k = 0; k < 4; ++k) { if (base + k < n) out[base + k] = excl; excl += v[k]; }
	if (threadIdx.x == TPB - 1) block_sums[blk] = wbase + x;
	}
}

__global__ void __launch_bounds__(1024) k_scan_sums(uint32_t* block_sums, uint32_t nb_cap, uint32_t n_cap, const uint32_t* n_live, uint32_t* cells_used_out)
{
	const uint32_t nb = min(nb_cap, (min(n_cap, *n_live * 64u + 1u) + TPB * 4u - 1u) / (TPB * 4u));      // (the blocks that hold cells of this step's grid)
	if (threadIdx.x == 0) *cells_used_out = min(n_cap, *n_live * 64u);      // what the next step's first launch resets
	__shared__ uint32_t wave_sums[16];
	__shared__ uint32_t carry;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (uint32_t start = 0; start < nb; start += 1024) {
		const uint32_t i = start + threadIdx.x;
		const uint32_t v = i < nb ? block_sums[i] : 0u;
		uint32_t x = v;
		const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
		if (lane == 63) wave_sums[wave] = x;
		__syncthreads();
		uint32_t wbase = carry;
		for (int k = 0; k < wave; ++k) wbase += wave_sums[k];
		if (i < nb) block_sums[i] = wbase + x - v;
		__syncthreads();
		if (threadIdx.x == 1023) carry = wbase + x;
		__syncthreads();
	}
}

__global__ void __launch_bounds__(TPB) k_scan_add(uint32_t* out, const uint32_t* block_sums, uint32_t n_cap, const uint32_t* n_live)
{
	const uint32_t n = min(n_cap, *n_live * 64u + 1u);
	for (uint32_t blk = blockIdx.x; blk * (TPB * 4u) < n; blk += gridDim.x) {
		const uint32_t base = (blk * TPB + threadIdx.x) * 4;
		const uint32_t add = block_sums[blk];
#pragma unroll
		for (int k = 0; k < 4; ++k) if (base + k < n) out[base + k] += add;
	}
}

SGP_DEV void bp_scatter_one(const DV& d, uint32_t i)
{
	if (i < 6u) d.bounds_acc[i] = i < 3u ? 0x7FFFFFFF : (int)0x80000000;      // (the grid has been derived: ready for the next step's -- or re-binning's -- bounds)
	if (i >= d.sp->n_slots) return;
	const uint32_t h = d.cell_hash[i];
	if (h == 0xFFFFFFFFu) return;
	const uint32_t slot = d.cell_start[h] + atomicAdd(&d.cell_fill[h], 1u);
	const float4 mn = d.aabb_min[i], mx = d.aabb_max[i];
	d.sorted_min[slot] = make_float4(mn.x, mn.y, mn.z, __uint_as_float(d.flags[i]));
	d.sorted_max[slot] = make_float4(mx.x, mx.y, mx.z, __uint_as_float(i));
}
__global__ void __launch_bounds__(TPB) k_bp_scatter(DV d) { bp_scatter_one(d, blockIdx.x * TPB + threadIdx.x); }      // (re-binning for queries between steps)

SGP_DEV bool pair_passes(const DV& d, uint32_t fi, float4 mni, float4 mxi, uint32_t j)
{
	const uint32_t fj = d.flags[j];
	if (!(fj & BF_ALIVE)) return false;
	if (f_motion(fi) != SGP_MOTION_DYNAMIC && f_motion(fj) != SGP_MOTION_DYNAMIC) return false;
	if (!layers_collide(f_layer(fi), f_layer(fj))) return false;
	const float4 mnj = d.aabb_min[j], mxj = d.aabb_max[j];
	const float s = d.st.speculative_contact_distance;
	if (mni.x - s > mxj.x || mnj.x - s > mxi.x) return false;
	if (mni.y - s > mxj.y || mnj.y - s > mxi.y) return false;
	if (mni.z - s > mxj.z || mnj.z - s > mxi.z) return false;
	return true;
}

// same predicate on two packed records (w of min = flags, w of max = id)
SGP_DEV bool rec_pair_passes(float spec, float4 mni, float4 mxi, float4 mnj, float4 mxj)
{
	const uint32_t fi = __float_as_uint(mni.w), fj = __float_as_uint(mnj.w);
	if (f_motion(fi) != SGP_MOTION_DYNAMIC && f_motion(fj) != SGP_MOTION_DYNAMIC) return false;
	if (!layers_collide(f_layer(fi), f_layer(fj))) return false;
	if (mni.x - spec > mxj.x || mnj.x - spec > mxi.x) return false;
	if (mni.y - spec > mxj.y || mnj.y - spec > mxi.y) return false;
	if (mni.z - spec > mxj.z || mnj.z - spec > mxi.z) return false;
	return true;
}

SGP_DEV uint32_t wave_alloc(uint32_t* counter);
SGP_DEV void push_pair(const DV& d, uint32_t i, uint32_t j)
{
	const uint32_t k = wave_alloc(&d.ctr->n_pairs);      // (one atomic per wave: the ground quad alone pairs with every body)
	if (k < d.cap_pairs) d.pairs[k] = make_uint2(i < j ? i : j, i < j ? j : i);
	else atomicAdd(&d.ctr->pairs_dropped, 1u);
}

// pair staged in LDS (falls back to the global list when the tile's buffer is full)
template <int PCAP> SGP_DEV void stage_pair(const DV& d, uint2* spairs, uint8_t* scls, uint32_t* lcount, uint32_t i, uint32_t j, uint32_t fi, uint32_t fj);

#define BP_TILE 4
#define BP_H 2
#define BP_HALO (BP_TILE + 2 * BP_H)
#define BP_HALO_CELLS (BP_HALO * BP_HALO * BP_HALO)
#define BP_INNER_CELLS (BP_TILE * BP_TILE * BP_TILE)
// LDS capacities of k_bp_pairs (records of a tile's halo, staged pairs): two instances.  The workgroups of this kernel spend two thirds of their
// cycles waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES = 0.67: staging loads and barriers), so how many of them a compute unit holds decides the
// launch: 70 KB of LDS = 2 workgroups per CU, 37 KB = 4 (config 3: 130 -> 83 us; its halos hold 500-640 records).  The small instance serves scenes whose halos hold at most
// BP_LDS_CAP_SMALL records (k_bp_pairs reports a larger one in StepCounters::bp_dense, the next step's plan then takes the large instance);
// a halo above the instance's capacity is read from global memory either way.
#define BP_LDS_CAP_SMALL 768
#define BP_PAIR_CAP_SMALL 1024
#define BP_LDS_CAP_LARGE 1536
#define BP_PAIR_CAP_LARGE 2048
#define BP_SPLIT 4

// The class of a pair = its two shape types: the staged pairs leave a workgroup sorted by class (flush_pairs), so that the narrow phase -- one thread
// per pair, one branch per pairing of shapes -- gets waves of one pairing instead of waves that walk through all six branches one after the other.
SGP_DEV uint32_t pair_class(uint32_t fa, uint32_t fb) { const uint32_t ta = f_shape(fa), tb = f_shape(fb); return (ta < tb ? ta : tb) * 8u + (ta < tb ? tb : ta); }
template <int PCAP> SGP_DEV void stage_pair(const DV& d, uint2* spairs, uint8_t* scls, uint32_t* lcount, uint32_t i, uint32_t j, uint32_t fi, uint32_t fj)
{
	const uint32_t k = atomicAdd(lcount, 1u);
	if (k < (uint32_t)PCAP) { spairs[k] = make_uint2(i < j ? i : j, i < j ? j : i); scls[k] = (uint8_t)pair_class(fi, fj); }
	else push_pair(d, i, j);
}
// the staged pairs to the global list, grouped by class (counting sort over 64 bins; the order inside a class is whatever the atomics gave: the
// list's order never enters a result).  Whole workgroup.
template <int PCAP> SGP_DEV void flush_pairs(const DV& d, const uint2* spairs, const uint8_t* scls, uint32_t lcount, uint32_t* gbase, uint32_t* bins)
{
	const uint32_t n_out = min(lcount, (uint32_t)PCAP);
	if (threadIdx.x < 64) bins[threadIdx.x] = 0;
	if (threadIdx.x == 0 && n_out) *gbase = atomicAdd(&d.ctr->n_pairs, n_out);
	__syncthreads();
	uint32_t rank[PCAP / TPB];
#pragma unroll
	for (int r = 0; r < PCAP / TPB; ++r) { const uint32_t k = threadIdx.x + (uint32_t)r * TPB; rank[r] = k < n_out ? atomicAdd(&bins[scls[k]], 1u) : 0u; }
	__syncthreads();
	if (threadIdx.x < 64) {
		// exclusive scan of the 64 bins by the first wave
		const uint32_t v = bins[threadIdx.x];
		uint32_t x = v;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if ((int)threadIdx.x >= off) x += y; }
		bins[threadIdx.x] = x - v;
	}
	__syncthreads();
#pragma unroll
	for (int r = 0; r < PCAP / TPB; ++r) {
		const uint32_t k = threadIdx.x + (uint32_t)r * TPB;
		if (k < n_out) {
			const uint32_t at = *gbase + bins[scls[k]] + rank[r];
			if (at < d.cap_pairs) d.pairs[at] = spairs[k];
			else atomicAdd(&d.ctr->pairs_dropped, 1u);
		}
	}
}

// exclusive scan of n <= 2*TPB values held in LDS (in place), result total returned to every thread
SGP_DEV uint32_t block_scan_512(uint32_t* a, int n, uint32_t* wave_tot)
{
	const int t = threadIdx.x;
	const uint32_t v0 = (2 * t < n) ? a[2 * t] : 0u, v1 = (2 * t + 1 < n) ? a[2 * t + 1] : 0u;
	uint32_t x = v0 + v1;
	const int lane = t & 63, wave = t >> 6;
	for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
	if (lane == 63) wave_tot[wave] = x;
	__syncthreads();
	uint32_t base = 0, total = 0;
	for (int k = 0; k < TPB / 64; ++k) { if (k < wave) base += wave_tot[k]; total += wave_tot[k]; }
	const uint32_t excl = base + x - (v0 + v1);
	__syncthreads();
	if (2 * t < n) a[2 * t] = excl;
	if (2 * t + 1 < n) a[2 * t + 1] = excl + v0;
	__syncthreads();
	return total;
}

// One workgroup per 4x4x4-cell tile.  Cell edge = R_max + margin (R_max = largest small-body bounding radius), so any
// partner of a body has its centre within 2 cells of the body's own AABB: the tile plus a 2-cell halo (8x8x8 cells = 64
// contiguous runs of the cell-sorted records) is staged in LDS and every active body of the tile scans only the cells its
// own AABB (+- R_max + margin) reaches.  A pair is emitted once: by the lower id when both are active, else by the active one.
template <int BP_LDS_CAP, int BP_PAIR_CAP> __global__ void __launch_bounds__(TPB) k_bp_pairs(DV d)
{
	__shared__ float4 smin[BP_LDS_CAP];
	__shared__ float4 smax[BP_LDS_CAP];
	__shared__ uint32_t cstart[BP_HALO_CELLS + 1];
	__shared__ uint32_t gstart[BP_HALO_CELLS];
	__shared__ uint32_t istart[BP_INNER_CELLS + 1];
	__shared__ uint32_t wave_tot[TPB / 64];
	__shared__ uint2 spairs[BP_PAIR_CAP];
	__shared__ uint8_t scls[BP_PAIR_CAP];
	__shared__ uint32_t pbins[64];
	__shared__ uint32_t lcount, gbase;
	__shared__ uint32_t nslot[27];
	const BpGrid g = *d.grid;
	// (round 4: the workgroups walk the OCCUPIED tiles of the paged grid -- slot by slot --, not every tile of the bounding box)
	const uint32_t n_tiles = d.ctr->n_tiles_used;
	const float spec = d.st.speculative_contact_distance;
	const float reach = d.sp->bp_rmax + spec;
	if (threadIdx.x == 0) lcount = 0;
	for (uint32_t slot = blockIdx.x; slot < n_tiles; slot += gridDim.x) {
		const uint32_t tile = d.tile_of_slot[slot];
		const int tx = (int)(tile % (uint32_t)g.tnx), ty = (int)((tile / (uint32_t)g.tnx) % (uint32_t)g.tny), tz = (int)(tile / ((uint32_t)g.tnx * (uint32_t)g.tny));
		const int x0 = tx * BP_TILE - BP_H, y0 = ty * BP_TILE - BP_H, z0 = tz * BP_TILE - BP_H;   // halo origin (cell coords)
		__syncthreads();
		// the tile's own cells first (64 neighbours in the cell arrays, in (z, y, x) order like the threads): a tile whose bodies have all gone is skipped without touching the halo
		if (threadIdx.x < BP_INNER_CELLS) {
			const uint32_t lin = slot * 64u + threadIdx.x;
			istart[threadIdx.x] = d.cell_start[lin + 1] - d.cell_start[lin];
		} else if (threadIdx.x < BP_INNER_CELLS + 27) {
			// the slots of the 27 tiles the halo reaches into, requested next to the counts (the halo's 512 cells then find them in LDS, not behind a page-table load each)
			const int k = (int)threadIdx.x - BP_INNER_CELLS;
			const int nx_ = tx + k % 3 - 1, ny_ = ty + (k / 3) % 3 - 1, nz_ = tz + k / 9 - 1;
			nslot[k] = (nx_ >= 0 && nx_ < g.tnx && ny_ >= 0 && ny_ < g.tny && nz_ >= 0 && nz_ < g.tnz) ? d.tile_slot[((uint32_t)nz_ * (uint32_t)g.tny + (uint32_t)ny_) * (uint32_t)g.tnx + (uint32_t)nx_] : BP_TILE_NONE;
		}
		__syncthreads();
		const uint32_t n_inner = block_scan_512(istart, BP_INNER_CELLS, wave_tot);
		if (threadIdx.x == 0) istart[BP_INNER_CELLS] = n_inner;
		if (n_inner == 0) continue;
		// per halo cell: global start and count
		for (int c = threadIdx.x; c < BP_HALO_CELLS; c += TPB) {
			const int hx = c % BP_HALO, hy = (c / BP_HALO) % BP_HALO, hz = c / (BP_HALO * BP_HALO);
			const int x = x0 + hx, y = y0 + hy, z = z0 + hz;
			uint32_t b = 0, cnt = 0;
			if (x >= 0 && x < g.nx && y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
				const uint32_t ns = nslot[(((z >> 2) - tz + 1) * 3 + ((y >> 2) - ty + 1)) * 3 + ((x >> 2) - tx + 1)];
				if (ns < BP_TILE_PENDING) { const uint32_t lin = ns * 64u + (uint32_t)((((z & 3) << 2) | (y & 3)) << 2 | (x & 3)); b = d.cell_start[lin]; cnt = d.cell_start[lin + 1] - b; }
			}
			gstart[c] = b;
			cstart[c] = cnt;
		}
		__syncthreads();
		const uint32_t total = block_scan_512(cstart, BP_HALO_CELLS, wave_tot);
		if (threadIdx.x == 0) cstart[BP_HALO_CELLS] = total;
		__syncthreads();
		const bool in_lds = total <= (uint32_t)BP_LDS_CAP;
		if (threadIdx.x == 0 && total > (uint32_t)BP_LDS_CAP_SMALL) d.ctr->bp_dense = 1u;      // (plain store of the same value from every such tile)
		if (in_lds) {
			for (uint32_t q = threadIdx.x; q < total; q += TPB) {
				int lo = 0, hi = BP_HALO_CELLS - 1;
				while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (cstart[mid] <= q) lo = mid; else hi = mid - 1; }
				const uint32_t src = gstart[lo] + (q - cstart[lo]);
				smin[q] = d.sorted_min[src]; smax[q] = d.sorted_max[src];
			}
		}
		__syncthreads();
		// BP_SPLIT threads share one body: each takes every BP_SPLIT-th (z, y) row of the cells the body reaches (a tile holds far
		// fewer bodies than the workgroup has threads, and the candidate loop is the long part)
		for (uint32_t tt = threadIdx.x; tt < n_inner * BP_SPLIT; tt += TPB) {
			const uint32_t t = tt / BP_SPLIT;
			const int sub = (int)(tt % BP_SPLIT);
			int lo = 0, hi = BP_INNER_CELLS - 1;
			while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (istart[mid] <= t) lo = mid; else hi = mid - 1; }
			const int ic = lo;
			const uint32_t k = t - istart[ic];
			const int ix = ic % BP_TILE, iy = (ic / BP_TILE) % BP_TILE, iz = ic / (BP_TILE * BP_TILE);
			const int hc = ((iz + BP_H) * BP_HALO + (iy + BP_H)) * BP_HALO + (ix + BP_H);
			const float4 mni = in_lds ? smin[cstart[hc] + k] : d.sorted_min[gstart[hc] + k];
			const float4 mxi = in_lds ? smax[cstart[hc] + k] : d.sorted_max[gstart[hc] + k];
			const uint32_t fi = __float_as_uint(mni.w), i = __float_as_uint(mxi.w);
			if (!f_active_for_pairs(fi)) continue;
			// cells (halo-local) whose bodies can touch this one
			int xl = (int)floorf((mni.x - reach - g.ox) * g.inv_cell) - x0, xh = (int)floorf((mxi.x + reach - g.ox) * g.inv_cell) - x0;
			int yl = (int)floorf((mni.y - reach - g.oy) * g.inv_cell) - y0, yh = (int)floorf((mxi.y + reach - g.oy) * g.inv_cell) - y0;
			int zl = (int)floorf((mni.z - reach - g.oz) * g.inv_cell) - z0, zh = (int)floorf((mxi.z + reach - g.oz) * g.inv_cell) - z0;
			// a body clamped into a border cell of the grid keeps scanning its full halo box
			xl = min(max(xl, 0), ix + BP_H); xh = max(min(xh, BP_HALO - 1), ix + BP_H);
			yl = min(max(yl, 0), iy + BP_H); yh = max(min(yh, BP_HALO - 1), iy + BP_H);
			zl = min(max(zl, 0), iz + BP_H); zh = max(min(zh, BP_HALO - 1), iz + BP_H);
			const int ny_rows = yh - yl + 1, n_rows = (zh - zl + 1) * ny_rows;
			for (int row = sub; row < n_rows; row += BP_SPLIT) {
				const int hz = zl + row / ny_rows, hy = yl + row % ny_rows;
				const int rb = (hz * BP_HALO + hy) * BP_HALO;
				if (in_lds) {
					const uint32_t q0 = cstart[rb + xl], q1 = cstart[rb + xh + 1];
					for (uint32_t q = q0; q < q1; ++q) {
						const float4 mnj = smin[q], mxj = smax[q];
						const uint32_t j = __float_as_uint(mxj.w);
						if (j == i) continue;
						if (f_active_for_pairs(__float_as_uint(mnj.w)) && j < i) continue;
						if (rec_pair_passes(spec, mni, mxi, mnj, mxj)) stage_pair<BP_PAIR_CAP>(d, spairs, scls, &lcount, i, j, fi, __float_as_uint(mnj.w));
					}
				} else {
					for (int c = xl; c <= xh; ++c) {
						const uint32_t gb = gstart[rb + c], gn = cstart[rb + c + 1] - cstart[rb + c];
						for (uint32_t q = 0; q < gn; ++q) {
							const float4 mnj = d.sorted_min[gb + q], mxj = d.sorted_max[gb + q];
							const uint32_t j = __float_as_uint(mxj.w);
							if (j == i) continue;
							if (f_active_for_pairs(__float_as_uint(mnj.w)) && j < i) continue;
							if (rec_pair_passes(spec, mni, mxi, mnj, mxj)) stage_pair<BP_PAIR_CAP>(d, spairs, scls, &lcount, i, j, fi, __float_as_uint(mnj.w));
						}
					}
				}
			}
		}
		// flush the staged pairs when the buffer is half full (the pairs of several tiles share one global atomic: atomics on the one pair
		// counter serialise, ~12 ns each), coalesced stores
		__syncthreads();
		if (lcount > BP_PAIR_CAP / 2) {
			flush_pairs<BP_PAIR_CAP>(d, spairs, scls, lcount, &gbase, pbins);
			__syncthreads();
			if (threadIdx.x == 0) lcount = 0;
		}
	}
	// what is left after the workgroup's last tile
	__syncthreads();
	flush_pairs<BP_PAIR_CAP>(d, spairs, scls, lcount, &gbase, pbins);
}

// ---- the static large bodies' grid (LargeGrid) ---------------------------------------------------------------------------------
// cell coordinate of x, clamped into the grid (the host sorts the bodies into cells with this very expression on the bounds it read back)
SGP_DEV int lg_cell(float x, float o, float inv, int n) { return min(max((int)floorf((x - o) * inv), 0), n - 1); }
// fn(body) for every body of the grid whose cells the box [lo, hi] touches -- each body ONCE: a body sits in several cells, and it is reported from
// the one that holds the lower corner of (box intersected with the body's bounds), a cell both ranges contain whenever the two overlap.
template <class F> SGP_DEV void large_grid_query(const DV& d, v3 lo, v3 hi, F fn)
{
	const LargeGrid g = *d.lgrid;
	if (!g.n_items) return;
	const int x0 = lg_cell(lo.x, g.ox, g.inv_cell, g.nx), x1 = lg_cell(hi.x, g.ox, g.inv_cell, g.nx);
	const int y0 = lg_cell(lo.y, g.oy, g.inv_cell, g.ny), y1 = lg_cell(hi.y, g.oy, g.inv_cell, g.ny);
	const int z0 = lg_cell(lo.z, g.oz, g.inv_cell, g.nz), z1 = lg_cell(hi.z, g.oz, g.inv_cell, g.nz);
	for (int z = z0; z <= z1; ++z) for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) {
		const uint32_t c = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx + (uint32_t)x;
		const uint32_t q0 = d.lg_start[c], q1 = d.lg_start[c + 1];
		for (uint32_t q = q0; q < q1; ++q) {
			const uint32_t b = d.lg_items[q];
			const float4 mn = d.aabb_min[b];
			if (lg_cell(fmaxf(lo.x, mn.x), g.ox, g.inv_cell, g.nx) != x || lg_cell(fmaxf(lo.y, mn.y), g.oy, g.inv_cell, g.ny) != y || lg_cell(fmaxf(lo.z, mn.z), g.oz, g.inv_cell, g.nz) != z) continue;
			fn(b);
		}
	}
}
// fn(body) for the bodies of the cells a ray (origin o, unit direction dir) crosses up to *max_t (which fn may shorten); a body may come more than once
template <class F> SGP_DEV void large_grid_ray(const DV& d, v3 o, v3 dir, const float* max_t, F fn)
{
	const LargeGrid g = *d.lgrid;
	if (!g.n_items) return;
	const float c = g.cell;
	const float oo[3] = { o.x, o.y, o.z }, dd[3] = { dir.x, dir.y, dir.z };
	const float bl[3] = { g.ox, g.oy, g.oz }, bh[3] = { g.ox + (float)g.nx * c, g.oy + (float)g.ny * c, g.oz + (float)g.nz * c };
	float t0 = 0.0f, t1 = *max_t;
	for (int a = 0; a < 3; ++a) {
		if (fabsf(dd[a]) < 1.0e-12f) { if (oo[a] < bl[a] - c || oo[a] > bh[a] + c) return; }
		else {
			float ta = (bl[a] - c - oo[a]) / dd[a], tb = (bh[a] + c - oo[a]) / dd[a];      // (one cell of slack: bounds outside the grid box sit in its border cells)
			if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
			t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
			if (t0 > t1) return;
		}
	}
	const v3 p0 = v3_add(o, v3_scale(dir, t0));
	int cx = (int)floorf((p0.x - g.ox) * g.inv_cell), cy = (int)floorf((p0.y - g.oy) * g.inv_cell), cz = (int)floorf((p0.z - g.oz) * g.inv_cell);
	cx = min(max(cx, -1), g.nx); cy = min(max(cy, -1), g.ny); cz = min(max(cz, -1), g.nz);
	const int sx = dir.x > 0.0f ? 1 : -1, sy = dir.y > 0.0f ? 1 : -1, sz = dir.z > 0.0f ? 1 : -1;
	const float inf = 3.0e38f;
	const float tdx = fabsf(dir.x) > 1.0e-12f ? c / fabsf(dir.x) : inf, tdy = fabsf(dir.y) > 1.0e-12f ? c / fabsf(dir.y) : inf, tdz = fabsf(dir.z) > 1.0e-12f ? c / fabsf(dir.z) : inf;
	float tmx = fabsf(dir.x) > 1.0e-12f ? ((g.ox + (float)(cx + (sx > 0 ? 1 : 0)) * c) - o.x) / dir.x : inf;
	float tmy = fabsf(dir.y) > 1.0e-12f ? ((g.oy + (float)(cy + (sy > 0 ? 1 : 0)) * c) - o.y) / dir.y : inf;
	float tmz = fabsf(dir.z) > 1.0e-12f ? ((g.oz + (float)(cz + (sz > 0 ? 1 : 0)) * c) - o.z) / dir.z : inf;
	float t_enter = t0;
	for (int iter = 0; iter < 100000; ++iter) {
		if (t_enter - c > *max_t) break;
		// (the cell and, against rounding at cell faces, nothing else: a body is in every cell its bounds touch; cells one step outside the box are its border cells)
		const int x = min(max(cx, 0), g.nx - 1), y = min(max(cy, 0), g.ny - 1), z = min(max(cz, 0), g.nz - 1);
		const uint32_t cell = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx + (uint32_t)x;
		const uint32_t q0 = d.lg_start[cell], q1 = d.lg_start[cell + 1];
		for (uint32_t q = q0; q < q1; ++q) fn(d.lg_items[q]);
		if (tmx <= tmy && tmx <= tmz) { t_enter = tmx; tmx += tdx; cx += sx; if (cx < -1 || cx > g.nx) break; }
		else if (tmy <= tmz) { t_enter = tmy; tmy += tdy; cy += sy; if (cy < -1 || cy > g.ny) break; }
		else { t_enter = tmz; tmz += tdz; cz += sz; if (cz < -1 || cz > g.nz) break; }
		if (t_enter > t1) break;
	}
}
__global__ void __launch_bounds__(TPB) k_gather_aabbs(DV d, const uint32_t* ids, uint32_t n, float4* out)
{
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= n) return;
	out[2 * (size_t)k] = d.aabb_min[ids[k]]; out[2 * (size_t)k + 1] = d.aabb_max[ids[k]];
}

// large bodies (ground quad, PhysicsWorld.cpp:1123) against every body
SGP_DEV uint32_t block_alloc(uint32_t* counter, bool want);
SGP_DEV void bp_large_one(const DV& d, uint32_t j)
{
	const uint32_t fj = j < d.sp->n_slots ? d.flags[j] : 0u;
	const bool live_j = (fj & BF_ALIVE) && !(fj & BF_ALIAS);      // (a mesh body's alias slots only carry manifolds: they never pair)
	float4 mnj = make_float4(0.0f, 0.0f, 0.0f, 0.0f), mxj = mnj;
	if (live_j) { mnj = d.aabb_min[j]; mxj = d.aabb_max[j]; }
	for (uint32_t l = 0; l < d.sp->n_large; ++l) {
		const uint32_t i = d.large_ids[l];
		bool pair = false;
		if (live_j && i != j) {
			const uint32_t fi = d.flags[i];
			pair = (fi & BF_ALIVE) && !((fj & BF_LARGE) && j < i)                 // large-large once
			       && (f_active_for_pairs(fi) || f_active_for_pairs(fj)) && pair_passes(d, fj, mnj, mxj, i);
		}
		// the ground quad alone pairs with every body: one atomic per workgroup on the pair counter, not one per wave
		const uint32_t k = block_alloc(&d.ctr->n_pairs, pair);
		if (pair) { if (k < d.cap_pairs) d.pairs[k] = make_uint2(i < j ? i : j, i < j ? j : i); else atomicAdd(&d.ctr->pairs_dropped, 1u); }
	}
	// the static large bodies in reach of an awake body: through their grid (a static body pairs with nothing that sleeps)
	if (live_j && f_active_for_pairs(fj)) {
		const float s = d.st.speculative_contact_distance;
		large_grid_query(d, V3(mnj.x - s, mnj.y - s, mnj.z - s), V3(mxj.x + s, mxj.y + s, mxj.z + s), [&](uint32_t i) {
			if (i == j || !pair_passes(d, fj, mnj, mxj, i)) return;
			if ((fj & BF_LARGE) && j < i) return;                // (a moving large body is on the list above: the grid body's own thread paired the two there when its id is the higher one)
			const uint32_t k = wave_alloc(&d.ctr->n_pairs);     // (one atomic for the lanes that found a pair in this turn: a terrain in the grid pairs with every body on it)
			if (k < d.cap_pairs) d.pairs[k] = make_uint2(i < j ? i : j, i < j ? j : i); else atomicAdd(&d.ctr->pairs_dropped, 1u);
		});
	}
}
__global__ void __launch_bounds__(TPB) k_bp_large(DV d) { bp_large_one(d, blockIdx.x * TPB + threadIdx.x); }
// In a step, one launch does both per-body jobs -- the body's record into its cell's run, then its pairs with the large bodies (neither reads what the
// other writes): a launch less on the step's chain (round 4).
__global__ void __launch_bounds__(TPB) k_bp_scatter_large(DV d) { const uint32_t i = blockIdx.x * TPB + threadIdx.x; bp_scatter_one(d, i); bp_large_one(d, i); }

// ---------------------------------------------------------------------------------------------------------------
// K4: narrow phase, one thread per candidate pair

SGP_DEV sgd_shape load_shape(const DV& d, uint32_t i, uint32_t f)
{
	sgd_shape s;
	s.pos = V3(d.pose[2 * (size_t)i]);
	s.R = quat_to_m33(Q4(d.pose[2 * (size_t)i + 1]));
	s.type = (int)f_shape(f);
	const float4 sh = d.prop[2 * (size_t)i + 1];
	s.p0 = sh.x; s.p1 = sh.y; s.p2 = sh.z;
	s.hull = s.type == SGP_SHAPE_HULL ? body_hull(d, sh) : (s.type == SGP_SHAPE_BOX ? &d.hulls[0] : nullptr);
	return s;
}

SGP_DEV uint32_t cache_find(const DV& d, uint64_t key);
// Colours a body's contacts may not take: a vehicle's rows are solved in the same launch as the first contact colour of every pass (they come first
// in the pass: non-contact constraints before contacts, as in PhysicsSystem's solve), so no contact of its chassis may sit in colour 0.
// Round 4: the same holds for a dynamic body under a wheel of an active vehicle -- the wheel rows act on it (DV::veh_claim of the current step).
SGP_DEV bool veh_body_claimed(const DV& d, uint32_t body) { return d.n_vehicles != 0u && (uint32_t)(d.veh_claim[body] >> 32) == *d.veh_epoch; }
SGP_DEV uint64_t chassis_colours(const DV& d, uint32_t body, uint32_t f) { return ((f & BF_CHASSIS) || veh_body_claimed(d, body)) ? 1ull : 0ull; }
#define MAN_PREV_LOOKUP 0xFFFFFFFFu
// Every lane that calls this (the lanes active at the call) gets its own index from *counter: one atomic per wave instead of one per lane
// (hundreds of thousands of atomics on ONE address serialise in L2: that, not the collision arithmetic, bounded the narrow phase).
SGP_DEV uint32_t wave_alloc(uint32_t* counter)
{
	const unsigned long long act = __ballot(1);
	const int lane = (int)(threadIdx.x & 63u), leader = __ffsll((long long)act) - 1;
	uint32_t base = 0;
	if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(act));
	base = __shfl(base, leader, 64);
	return base + (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
}
// prev: the pair's slot in the previous step's constraint buffer if the caller already looked it up (| MAN_PREV_REUSED for a manifold taken
// from the contact cache), MAN_PREV_LOOKUP to leave the look-up to k_colour_inherit
// safety net: a manifold without a direction (or with a NaN one) is dropped, never solved
SGP_DEV bool manifold_ok(const sgd_manifold& m) { return v3_len_sq(m.n) > 0.25f; }

// A sleeping dynamic body is touched by an awake one (or stands under a wheel): k_pre_solve wakes it, and k_wake_pairs wakes, in the same step, everything
// that fell asleep in the same island (the label's mark carries this step's epoch)
SGP_DEV void wake_body(const DV& d, uint32_t id)
{
	atomicOr(&d.flags[id], BF_WAKE);
	d.label_wake[d.sleep_label[id]] = *d.veh_epoch;
	d.ctr->wake_any = 1u;
}

// the manifold goes to slot `slot` of the step's manifold list (the caller allocated it)
SGP_DEV void emit_manifold_at(const DV& d, uint32_t slot, uint2 ab, uint32_t fa, uint32_t fb, const sgd_manifold& m, uint32_t prev)
{
	if (slot >= d.cap_manifolds) { atomicAdd(&d.ctr->manifolds_dropped, 1u); return; }
	d.man_ab[slot] = ab;
	const bool sensor = (fa | fb) & BF_SENSOR;
	// bit 8 = sensor pair (mIsSensor, PhysicsWorld.cpp:1235): reported in the contact events, kept in the contact list
	// (so that it is 'persisted' next step) but with zero points for the solver
	d.man_n[slot] = make_float4(m.n.x, m.n.y, m.n.z, __int_as_float(m.np | (sensor ? 0x100 : 0)));
	for (int k = 0; k < 4; ++k) if (k < m.np) { d.man_p1[k][slot] = F4(m.p1[k], 0.0f); d.man_p2[k][slot] = F4(m.p2[k], 0.0f); }
	d.man_prio[slot] = sgp_mix64(((uint64_t)ab.x << 32) | ab.y);
	d.man_prev[slot] = prev;          // (MAN_PREV_LOOKUP: k_colour_inherit -- a light kernel that hides the hash probe's latency -- resolves it)
	d.man_colour[slot] = -1;
	if (!sensor) {
		const bool actA = f_active_for_pairs(fa), actB = f_active_for_pairs(fb);
		// (a pair's other body is awake -- or, in the in-step activation round, neither was when the step began: one of the two has just been woken and the
		// contact wakes the other.  The broad phase makes no pair of two bodies that stay asleep, so "not awake and dynamic" says it all.)
		if (!actB && f_motion(fb) == SGP_MOTION_DYNAMIC) wake_body(d, ab.y);
		if (!actA && f_motion(fa) == SGP_MOTION_DYNAMIC) wake_body(d, ab.x);
	}
}
// ... with the slot taken here: one atomic per wave (the kernels with few manifolds per wave: hulls, meshes)
SGP_DEV void emit_manifold(const DV& d, uint2 ab, uint32_t fa, uint32_t fb, const sgd_manifold& m, uint32_t prev = MAN_PREV_LOOKUP)
{
	if (!manifold_ok(m)) return;
	emit_manifold_at(d, wave_alloc(&d.ctr->n_manifolds), ab, fa, fb, m, prev);
}

// pose of body 2 relative to body 1: centre of mass offset in body 1's frame, conj(q1) * q2
SGP_DEV void pair_relative_pose(v3 posA, quat qA, v3 posB, quat qB, v3* dpos, quat* drot)
{
	*dpos = m33_tmul(quat_to_m33(qA), v3_sub(posB, posA));
	quat ca; ca.x = -qA.x; ca.y = -qA.y; ca.z = -qA.z; ca.w = qA.w;
	*drot = quat_mul(ca, qB);
}

// The body-pair contact cache (ContactConstraintManager::GetContactsFromCache) for one pair of non-mesh bodies: true = *m is last step's
// manifold carried to the bodies' current poses -- the two bodies sit, relative to each other, where they sat when it was computed (within
// 1 mm and 2 degrees), so the collision test is skipped.  *prev: the pair's slot in last step's constraints (MAN_PREV_NONE if it had none),
// found with the one hash look-up every later kernel shares.
SGP_DEV bool reuse_cached_manifold(const DV& d, uint2 ab, uint32_t fa, uint32_t fb, sgd_manifold* m, uint32_t* prev)
{
	const uint32_t ps = cache_find(d, ((uint64_t)ab.x << 32) | ab.y);
	*prev = ps == 0xFFFFFFFFu ? MAN_PREV_NONE : ps;
	if (ps == 0xFFFFFFFFu || !d.st.use_body_pair_contact_cache || ((fa | fb) & (BF_CACHE_INVALID | BF_SENSOR))) return false;
	const v3 posA = V3(d.pose[2 * (size_t)ab.x]), posB = V3(d.pose[2 * (size_t)ab.y]);
	const quat qA = Q4(d.pose[2 * (size_t)ab.x + 1]), qB = Q4(d.pose[2 * (size_t)ab.y + 1]);
	v3 dpos; quat drot;
	pair_relative_pose(posA, qA, posB, qB, &dpos, &drot);
	const float4 cdp = PRV(d).cdp[ps], cdr = PRV(d).cdr[ps];
	if (!(v3_len_sq(v3_sub(dpos, V3(cdp))) <= d.st.body_pair_cache_max_delta_position_sq)) return false;
	const float dq = drot.x * cdr.x + drot.y * cdr.y + drot.z * cdr.z + drot.w * cdr.w;
	if (!(fabsf(dq) >= d.st.body_pair_cache_cos_max_delta_rotation_div2)) return false;
	const m33 RA = quat_to_m33(qA), RB = quat_to_m33(qB);
	const float2 cnl = PRV(d).cnl[ps];
	m->np = PRV(d).np_col[ps] & 0xFF;
	m->n = m33_mul(RB, V3(cdp.w, cnl.x, cnl.y));
	for (int i = 0; i < 4; ++i) if (i < m->np) { m->p1[i] = v3_add(posA, m33_mul(RA, V3(PRV(d).loc1[i][ps]))); m->p2[i] = v3_add(posB, m33_mul(RB, V3(PRV(d).loc2[i][ps]))); }
	*prev = ps | MAN_PREV_REUSED;
	return true;
}

// Every atomic on the one manifold counter costs ~12 ns however many lanes it serves (same-address atomics serialise in L2): with one per wave
// the 9k wave-iterations of config 3 spent 110 of the kernel's 230 us queueing for it (measured with parts of the output switched off: no output 108 us, slot allocated but nothing written 224 us, full 230 us).  The
// workgroup therefore allocates the slots of all its manifolds of an iteration with ONE atomic.
// (at least four waves per SIMD: the kernel waits for its gathers three cycles in four, and 128 instead of 157 registers per lane -- a few spills to
// scratch -- buy a third more waves to wait with: 138 -> 116 us at config 3; five waves: 143 us, six: 178 us)
// ROUND 0: the broad phase's pairs; ROUND 1: the pairs of the bodies this step wakes (k_wake_pairs)
// (HULLS = false: the world holds no convex hull -- the plan knows -- and the activation round's instance carries nothing of the sequential hull search: a launch of a
// handful of pairs lasts as long as its cold code takes to arrive)
template <int ROUND, bool HULLS = true> SGP_DEV void narrowphase_pairs(const DV& d)
{
	__shared__ uint32_t s_wave_cnt[TPB / 64];
	__shared__ uint32_t s_base;
	const uint32_t n = ROUND ? min(d.ctr->n_wake_pairs, d.cap_wake_pairs) : min(d.ctr->n_pairs, d.cap_pairs);
	const uint2* const pairs = ROUND ? d.wake_pairs : d.pairs;
	const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
	for (uint32_t p0 = blockIdx.x * TPB; p0 < n; p0 += gridDim.x * TPB) {
		const uint32_t p = p0 + threadIdx.x;
		bool have = false;
		uint2 ab = make_uint2(0u, 0u); uint32_t fa = 0, fb = 0;
		sgd_manifold m;
		uint32_t prev = MAN_PREV_LOOKUP;
		if (p < n) {
			ab = pairs[p];
			fa = d.flags[ab.x]; fb = d.flags[ab.y];
			if (f_shape(fa) == SGP_SHAPE_MESH || f_shape(fb) == SGP_SHAPE_MESH) {
				// (two meshes never collide: both are static or kinematic)
				const bool mesh_a = f_shape(fa) == SGP_SHAPE_MESH, mesh_b = f_shape(fb) == SGP_SHAPE_MESH;
				const uint32_t other = mesh_a ? f_shape(fb) : f_shape(fa);
#pragma unroll
				for (uint32_t t = 0; t < 4u; ++t) if (!(mesh_a && mesh_b) && other == t) {
					const uint32_t k = wave_alloc(&d.ctr->n_mesh_pairs[t]);
					if (k < d.cap_mesh_pairs) d.mesh_pairs[(size_t)t * d.cap_mesh_pairs + k] = ab; else atomicAdd(&d.ctr->pairs_dropped, 1u);
				}
			} else {
				// the contact cache is consulted for polytope pairs only (box / hull against box / hull): their separating-axis test and clipping
				// cost more than the gather of a cached manifold, and they are the pairs whose resting contacts a frozen manifold keeps from
				// jittering; a sphere or capsule contact is recomputed (a few dozen instructions, the same answer every step)
				const bool polytopes = (f_shape(fa) == SGP_SHAPE_BOX || f_shape(fa) == SGP_SHAPE_HULL) && (f_shape(fb) == SGP_SHAPE_BOX || f_shape(fb) == SGP_SHAPE_HULL);
				if (polytopes && reuse_cached_manifold(d, ab, fa, fb, &m, &prev)) have = true;
				else if ((HULLS || ROUND == 0) && (f_shape(fa) == SGP_SHAPE_HULL || f_shape(fb) == SGP_SHAPE_HULL)) {
					// the polytope paths (clip buffers in scratch, long loops) live in their own kernel so that they do not cost the
					// sphere / box / capsule pairs registers or scratch
					if constexpr (ROUND == 0) {
						const uint32_t k = wave_alloc(&d.ctr->n_hull_pairs);
						if (k < d.cap_hull_pairs) d.hull_pairs[k] = ab; else atomicAdd(&d.ctr->pairs_dropped, 1u);
					} else {
						// the in-step activation round has few pairs and is two launches shorter with the sequential form of the same search here
						// (every axis through the same device function, first maximum wins: the same manifold as the wave-parallel kernels')
						const sgd_shape sa = load_shape(d, ab.x, fa), sb = load_shape(d, ab.y, fb);
						have = sgd_collide_hull(&sa, &sb, d.st.speculative_contact_distance, &m) != 0;
					}
				} else {
					const sgd_shape sa = load_shape(d, ab.x, fa), sb = load_shape(d, ab.y, fb);
					have = sgd_collide(&sa, &sb, d.st.speculative_contact_distance, &m) != 0;
				}
				have = have && manifold_ok(m);
			}
		}
		const unsigned long long hm = __ballot(have);
		if (lane == 0) s_wave_cnt[wave] = (uint32_t)__popcll(hm);
		__syncthreads();
		if (threadIdx.x == 0) {
			uint32_t tot = 0;
			for (int k = 0; k < TPB / 64; ++k) tot += s_wave_cnt[k];
			s_base = tot ? atomicAdd(&d.ctr->n_manifolds, tot) : 0u;
		}
		__syncthreads();
		if (have) {
			uint32_t slot = s_base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
			for (int k = 0; k < wave; ++k) slot += s_wave_cnt[k];
			emit_manifold_at(d, slot, ab, fa, fb, m, prev);
		}
		__syncthreads();
	}
}
__global__ void __launch_bounds__(TPB, 4) k_narrowphase(DV d) { narrowphase_pairs<0>(d); }
template <bool HULLS> __global__ void __launch_bounds__(TPB, 4) k_narrowphase_wake(DV d) { narrowphase_pairs<1, HULLS>(d); }

// IN-STEP ACTIVATION (PhysicsSystem::JobFindCollisions keeps taking bodies from the active list while ProcessBodyPair appends the ones it wakes: a woken
// body collides in the step that woke it, and wakes what it touches in turn).  One extra round: a body the narrow phase or a wheel marked takes along
// everything that fell asleep in the same island (sleep_label / label_wake: sleeping bodies have not moved, so the contacts that made the island are the
// ones the cascade would follow), and every woken body is paired here with all that was not awake when the step began -- its pairs with awake
// bodies exist already.  The narrow-phase kernels then run once more over these pairs (the hull and mesh kernels from hull_base / mesh_base on).
// What those contacts wake in turn -- two islands that went to sleep apart and touch -- is woken too but meets its other contacts next step.
SGP_DEV bool body_woken(const DV& d, uint32_t j, uint32_t fj, uint32_t epoch)
{
	return (fj & (BF_ALIVE | BF_ACTIVE | BF_ALIAS)) == BF_ALIVE && f_motion(fj) == SGP_MOTION_DYNAMIC && d.label_wake[d.sleep_label[j]] == epoch;
}
__global__ void __launch_bounds__(TPB) k_wake_pairs(DV d)
{
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i == 0) {
		// (the narrow-phase launches that follow start behind the first round's pairs -- also in the usual step, in which nothing was woken)
		d.ctr->hull_base = min(d.ctr->n_hull_pairs, d.cap_hull_pairs);
		for (int t = 0; t < 4; ++t) d.ctr->mesh_base[t] = min(d.ctr->n_mesh_pairs[t], d.cap_mesh_pairs);
		d.ctr->mesh_big_base = min(d.ctr->n_mesh_big, d.cap_mesh_pairs);
	}
	if (!d.ctr->wake_any) return;                      // (uniform)
	const uint32_t epoch = *d.veh_epoch;
	const int lane = (int)(threadIdx.x & 63u);
	const uint32_t fi_own = i < d.sp->n_slots ? d.flags[i] : 0u;
	const bool woken = i < d.sp->n_slots && body_woken(d, i, fi_own, epoch);
	if (woken && !(fi_own & BF_WAKE)) d.flags[i] = fi_own | BF_WAKE;      // (nobody else writes this word during this launch; the bits others read of it do not change)
	// the woken bodies of a wave one after the other, each with all 64 lanes on its candidates (woken bodies are few among many: a lane
	// walking its body's cells alone would leave 63 idle)
	unsigned long long todo = __ballot(woken);
	if (lane == 0 && todo) atomicAdd(&d.ctr->n_woken, (uint32_t)__popcll(todo));
	const float sp = d.st.speculative_contact_distance;
	const BpGrid g = *d.grid;
	while (todo) {
		const int src = __ffsll((long long)todo) - 1;
		todo &= todo - 1ull;
		const uint32_t b = (i - (uint32_t)lane) + (uint32_t)src;
		const uint32_t fb = d.flags[b];
		const float4 mnb = d.aabb_min[b], mxb = d.aabb_max[b];
		auto candidate = [&](uint32_t j) {
			if (j == b) return;
			const uint32_t fj = d.flags[j];
			if (!(fj & BF_ALIVE) || (fj & BF_ALIAS) || f_active_for_pairs(fj)) return;
			if (j < b && body_woken(d, j, fj, epoch)) return;           // two woken bodies: the lower id makes the pair
			if (!pair_passes(d, fb, mnb, mxb, j)) return;
			const uint32_t k = wave_alloc(&d.ctr->n_wake_pairs);
			if (k < d.cap_wake_pairs) d.wake_pairs[k] = make_uint2(b < j ? b : j, b < j ? j : b); else atomicAdd(&d.ctr->pairs_dropped, 1u);
		};
		for (uint32_t l = (uint32_t)lane; l < d.sp->n_large; l += 64u) candidate(d.large_ids[l]);
		{
			uint32_t seen = 0;      // static large bodies around the body, dealt to the lanes in the order their grid yields them
			large_grid_query(d, V3(mnb.x - sp, mnb.y - sp, mnb.z - sp), V3(mxb.x + sp, mxb.y + sp, mxb.z + sp), [&](uint32_t j) { if ((int)(seen++ & 63u) == lane) candidate(j); });
		}
		if (g.n_cells > 0 && g.min_x <= g.max_x) {
			// (small bodies are binned by their centres into cells no smaller than the largest of them plus the margin: one cell of slack around the bounds)
			const int x0 = max((int)floorf((mnb.x - g.ox) * g.inv_cell) - 1, 0), x1 = min((int)floorf((mxb.x - g.ox) * g.inv_cell) + 1, g.nx - 1);
			const int y0 = max((int)floorf((mnb.y - g.oy) * g.inv_cell) - 1, 0), y1 = min((int)floorf((mxb.y - g.oy) * g.inv_cell) + 1, g.ny - 1);
			const int z0 = max((int)floorf((mnb.z - g.oz) * g.inv_cell) - 1, 0), z1 = min((int)floorf((mxb.z - g.oz) * g.inv_cell) + 1, g.nz - 1);
			// the (y, z) rows of cells side by side, four lanes to a row: a row is a chain of dependent loads (page table, cell range, record, the
			// candidate's flags and bounds), and a small body has up to sixteen of them
			const int ny = y1 - y0 + 1, nrows = x0 <= x1 ? (z1 - z0 + 1) * ny : 0;
			for (int r = lane >> 2; r < nrows; r += 16) {
				const int z = z0 + r / ny, y = y0 + r % ny;
				grid_row_runs(d, g, x0, x1, y, z, [&](uint32_t q0, uint32_t q1) { for (uint32_t q = q0 + (uint32_t)(lane & 3); q < q1; q += 4u) candidate(__float_as_uint(d.sorted_max[q].w)); });
			}
		}
	}
}

// ---- static triangle meshes ----------------------------------------------------------------------------------------
#define MESH_CAND_CAP 192

// Triangles of mesh body M (header mh, pose pos / R) whose tree leaves overlap the mesh-local box [llo, lhi]: indices (caller's order)
// into cand[], ascending.  Returns the count (capped; *overflow set).
SGP_DEV int mesh_candidates(const DV& d, const MeshHeader& mh, v3 llo, v3 lhi, uint32_t* cand, bool* overflow)
{
	int n = 0; *overflow = false;
	uint32_t stack[48]; int sp = 0;
	stack[sp++] = 0;
	while (sp > 0) {
		const MeshNode nd = d.mesh_nodes[mh.node_off + stack[--sp]];
		if (nd.mxx < llo.x || nd.mnx > lhi.x || nd.mxy < llo.y || nd.mny > lhi.y || nd.mxz < llo.z || nd.mnz > lhi.z) continue;
		if (nd.count == 0) { if (sp + 2 <= 48) { stack[sp++] = nd.left; stack[sp++] = nd.right; } else *overflow = true; continue; }
		for (uint32_t k = 0; k < nd.count; ++k) {
			if (n == MESH_CAND_CAP) { *overflow = true; break; }
			cand[n++] = nd.left + k;                     // position in the tree-ordered triangle array
		}
	}
	// order by the triangle's index in the caller's order (what the sequential reference walks): insertion sort on (orig, pos)
	for (int i = 1; i < n; ++i) {
		const uint32_t pos = cand[i]; const uint32_t key = MESH_TRI_INDEX(d.mesh_tris[mh.tri_off + pos].w);
		int j = i - 1;
		while (j >= 0 && MESH_TRI_INDEX(d.mesh_tris[mh.tri_off + cand[j]].w) > key) { cand[j + 1] = cand[j]; --j; }
		cand[j + 1] = pos;
	}
	return n;
}

// X against mesh body M: every triangle whose world bounds come within max_sep of [lo, hi], in index order, manifolds grouped by normal.
// Returns the number of groups (manifolds mesh -> X).  Sequential (one thread).
SGP_DEV int collide_with_mesh(const DV& d, uint32_t mbody, const sgd_shape& X, v3 lo, v3 hi, float max_sep, sgd_manifold* out, bool* dropped)      // (X: a capsule)
{
	const float4 msh = d.prop[2 * (size_t)mbody + 1];
	const MeshHeader mh = d.meshes[(uint32_t)msh.x];
	const v3 mpos = V3(d.pose[2 * (size_t)mbody]); const m33 R = quat_to_m33(Q4(d.pose[2 * (size_t)mbody + 1]));
	const v3 e = V3(max_sep, max_sep, max_sep);
	const v3 qlo = v3_sub(lo, e), qhi = v3_add(hi, e);
	// the query box in the mesh frame (bounds of its 8 corners), a little generous
	v3 llo = V3(3.4e38f, 3.4e38f, 3.4e38f), lhi = V3(-3.4e38f, -3.4e38f, -3.4e38f);
	for (int k = 0; k < 8; ++k) {
		const v3 c = V3((k & 1) ? qhi.x : qlo.x, (k & 2) ? qhi.y : qlo.y, (k & 4) ? qhi.z : qlo.z);
		const v3 l = m33_tmul(R, v3_sub(c, mpos));
		llo = V3(fminf(llo.x, l.x), fminf(llo.y, l.y), fminf(llo.z, l.z)); lhi = V3(fmaxf(lhi.x, l.x), fmaxf(lhi.y, l.y), fmaxf(lhi.z, l.z));
	}
	const float pad = 1.0e-4f * (1.0f + fabsf(llo.x) + fabsf(llo.y) + fabsf(llo.z) + fabsf(lhi.x) + fabsf(lhi.y) + fabsf(lhi.z));
	llo = v3_sub(llo, V3(pad, pad, pad)); lhi = v3_add(lhi, V3(pad, pad, pad));
	uint32_t cand[MESH_CAND_CAP];
	const int nc = mesh_candidates(d, mh, llo, lhi, cand, dropped);
	sgd_mesh_contacts mc; mc.ng = 0;
	for (int k = 0; k < nc; ++k) {
		const uint4 tri = d.mesh_tris[mh.tri_off + cand[k]];
		const v3 a = V3(d.mesh_verts[mh.vert_off + tri.x]), b = V3(d.mesh_verts[mh.vert_off + tri.y]), c = V3(d.mesh_verts[mh.vert_off + tri.z]);
		const v3 wa = v3_add(mpos, m33_mul(R, a)), wb = v3_add(mpos, m33_mul(R, b)), wc = v3_add(mpos, m33_mul(R, c));
		const v3 tmin = V3(fminf(fminf(wa.x, wb.x), wc.x), fminf(fminf(wa.y, wb.y), wc.y), fminf(fminf(wa.z, wb.z), wc.z));
		const v3 tmax = V3(fmaxf(fmaxf(wa.x, wb.x), wc.x), fmaxf(fmaxf(wa.y, wb.y), wc.y), fmaxf(fmaxf(wa.z, wb.z), wc.z));
		if (tmax.x < qlo.x || tmin.x > qhi.x || tmax.y < qlo.y || tmin.y > qhi.y || tmax.z < qlo.z || tmin.z > qhi.z) continue;
		sgd_tri_hull_t th; v3 cen, n;
		sgd_tri_hull(a, b, c, &th, &cen, &n);
		sgd_tri_view T; T.pos = v3_add(mpos, m33_mul(R, cen)); T.R = R; T.scale = V3(1.0f, 1.0f, 1.0f); T.h = &th;
		sgd_manifold m;
		if (sgd_collide_tri<4>(&X, &T, m33_mul(R, n), max_sep, &m, 7u, V3(0.0f, 0.0f, 0.0f))) sgd_mesh_add(&mc, &m);      // (a shape query: no active-edge fixing; X is the character's capsule)
	}
	return sgd_mesh_finish(&mc, out);
}

// Pairs with a static mesh: EIGHT LANES PER PAIR, eight pairs per wave.  The sequential statement walks a pair's candidate triangles in
// index order, tests each against the body and merges the triangle's manifold into <= 3 groups by normal -- the tests are independent and
// are the cost (a thin-hull SAT with clipping for a box or hull), the merge depends on the order and is cheap.  So: lane 0 of the group
// walks the mesh's tree and drops the candidates into LDS, the eight lanes order them by triangle index (rank sort: keys are unique), then
// round after round each lane tests one of the next eight candidates and the hits are merged one lane at a time, in candidate order, into
// the group table in LDS -- the sequence of sgd_mesh_add calls of the sequential walk.  The <= 3 groups are reduced and emitted by three lanes.
// (Eight, not 64: a body on a terrain or floor mesh touches a handful of triangles, and a wave per pair would idle 56 lanes.  A chassis across 150
// triangles of a detailed mesh is another matter -- a box - triangle test is ~40 us of one lane's instructions, 24 rounds of them ~2 ms --: a pair
// with more than MESH_BIG_MIN candidates is passed on to a second launch of the same kernel with all 64 lanes on one pair.)
#define MESH_BIG_MIN 32      // pairs with more candidates than this go to the wave-per-pair launch ...
#define MESH_BIG_CAP 1024    // ... which holds this many (a body across more triangles than that loses the rest: counted in manifolds_dropped)
// (the tables of a pair in LDS; G = 8: a pair that fills more than MESH_BIG_MIN entries is passed on, so 64 entries do, and no copy of the polytope)
struct MeshNoHull {};
// (HULL: room for a copy of the body's polytope -- a wave per pair, and the eight-lane instance for convex hulls: a hull against a triangle walks the hull's
// corners twice per axis, ~130 axes per triangle)
template <int G, bool HULL = (G == 64)> struct MeshPairLds {
	static constexpr int CAP = G == 64 ? MESH_BIG_CAP : 64;
	uint32_t found[CAP]; uint32_t key[CAP]; uint32_t cand[CAP]; uint32_t n_front[2], n_found, redo;
	sgd_mesh_contacts mc;
	typename std::conditional<HULL, sgd_hull, MeshNoHull>::type hull;      // the body's polytope (cube template / convex hull) next to the lanes that walk its vertices once per axis
};
#define MESH_LDS_T(G, KINDS) MeshPairLds<G, (G) == 64 || (KINDS) == 8>

// every triangle whose leaf box overlaps [llo, lhi] (mesh frame): positions in the tree-ordered triangle array and the triangles' indices in the
// caller's order, as found (unsorted)
SGP_DEV int mesh_candidates_found(const DV& d, const MeshHeader& mh, v3 llo, v3 lhi, uint32_t* found, uint32_t* key, bool* overflow, int stop_after = MESH_CAND_CAP, int cap = MESH_CAND_CAP)
{      // (stop_after: the caller only wants to know that there are more than this many)
	int n = 0; *overflow = false;
	uint32_t stack[48]; int sp = 0;
	stack[sp++] = 0;
	while (sp > 0) {
		const MeshNode nd = d.mesh_nodes[mh.node_off + stack[--sp]];
		if (nd.mxx < llo.x || nd.mnx > lhi.x || nd.mxy < llo.y || nd.mny > lhi.y || nd.mxz < llo.z || nd.mnz > lhi.z) continue;
		if (nd.count == 0) { if (sp + 2 <= 48) { stack[sp++] = nd.left; stack[sp++] = nd.right; } else *overflow = true; continue; }
		for (uint32_t k = 0; k < nd.count; ++k) {
			if (n == cap) { *overflow = true; break; }
			found[n] = nd.left + k; key[n] = MESH_TRI_INDEX(d.mesh_tris[mh.tri_off + nd.left + k].w); ++n;
		}
		if (n > stop_after) return n;
	}
	return n;
}

// sgd_mesh_add by the lanes of a pair together (whole wave: every lane calls this; `mine`: this lane's manifold m is the one its pair merges in this turn -- at
// most one lane per pair).  The sequential function compares every point of the manifold with every point its group already holds, one lane at work while the
// others wait: ~4 us per manifold, and a wave-per-pair round merges dozens of them.  Here the manifold is handed to all lanes of the pair, which look for its
// group (the same reads, the same answer) and share the comparisons; lane 0 of the pair appends.  The same decisions in the same order: the same groups.
template <int G> SGP_DEV void mesh_add_coop(sgd_mesh_contacts& mc, bool mine, const sgd_manifold& m, int grp, int sub)
{
	const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
	const unsigned long long who = (__ballot(mine) >> (G == 64 ? 0 : grp * G)) & gmask;
	const bool any = who != 0ull;
	const int src = (any ? __ffsll((long long)who) - 1 : 0) + (G == 64 ? 0 : grp * G);
	const v3 n = V3(__shfl(m.n.x, src, 64), __shfl(m.n.y, src, 64), __shfl(m.n.z, src, 64));
	const int np = any ? __shfl(m.np, src, 64) : 0;
	int gi = -1; bool open_new = false;
	if (any) {
		const int ng = mc.ng;
		for (int k = 0; k < ng; ++k) if (v3_dot(mc.g[k].n, n) >= SGD_MESH_GROUP_COS) { gi = k; break; }
		if (gi < 0 && ng < SGD_MESH_MAX_GROUPS) { gi = ng; open_new = true; }
	}
	__syncthreads();
	if (open_new && sub == 0) { mc.g[gi].n = n; mc.g[gi].np = 0; mc.ng = gi + 1; }
	__syncthreads();
#pragma unroll
	for (int i = 0; i < 4; ++i) {      // (a triangle's manifold: at most four points)
		const v3 p1 = V3(__shfl(m.p1[i].x, src, 64), __shfl(m.p1[i].y, src, 64), __shfl(m.p1[i].z, src, 64));
		const v3 p2 = V3(__shfl(m.p2[i].x, src, 64), __shfl(m.p2[i].y, src, 64), __shfl(m.p2[i].z, src, 64));
		const bool act = any && gi >= 0 && i < np;
		bool dup = false; int gnp = 0;
		if (act) {
			gnp = mc.g[gi].np;
			// the same point reached through two triangles that share it (an edge or a vertex of the mesh) counts once
			for (int j = sub; j < gnp; j += G) if (v3_len_sq(v3_sub(mc.g[gi].p_body[j], p2)) < 1.0e-8f) dup = true;
		}
		const bool pair_dup = ((__ballot(dup) >> (G == 64 ? 0 : grp * G)) & gmask) != 0ull;
		if (act && sub == 0 && gnp < SGD_HULL_CLIP_CAP && !pair_dup) { mc.g[gi].p_mesh[gnp] = p1; mc.g[gi].p_body[gnp] = p2; mc.g[gi].np = gnp + 1; }
		__syncthreads();
	}
}

// The groups of one (body X, mesh body mid) pair by the lanes of its group (whole workgroup: every lane calls this; lanes of a group pass the same
// pair): what lies between "here is the pair" and "here are its <= 3 groups in L.mc".  valid: false for a group without a pair, and false on return
// when the pair was handed to the wave-per-pair launch (pair = its index in mesh_pairs; G = 8 only).  [qlo, qhi]: X's bounds grown by max_sep.
// KINDS: what X can be (bits of SGD_SHAPE_*, sgd_collide_tri): an instance for the primitives carries nothing of the general hull search, one for hulls nothing of the box's.
template <int MESH_GROUP, int KINDS = SGD_KINDS_ALL> SGP_DEV void mesh_pair_groups(const DV& d, MESH_LDS_T(MESH_GROUP, KINDS)& L, bool& valid, sgd_shape& X, uint32_t mid, v3 qlo, v3 qhi, float max_sep, int grp, int sub, uint32_t pair, bool& dropped, v3 movement, bool active_edges = true, float* lpoly = nullptr)
{
	MeshHeader mh; v3 mpos = V3(0.0f, 0.0f, 0.0f); m33 R = quat_to_m33(Q4(make_float4(0.0f, 0.0f, 0.0f, 1.0f)));
	int nc = 0;
	if (valid) {
		if ((MESH_GROUP == 64 || KINDS == 8) && X.hull) {
			// a triangle test walks the polytope's vertices twice per candidate axis: out of LDS, not out of a pointer into global memory (worth the copy
			// where dozens of tests follow: 2.50 -> 2.13 ms on the car-sized boxes, but 0.93 -> 1.09 ms on the small bodies of tools/experiments/mesh_terrain_bench.py when
			// every eight-lane pair did it; the eight-lane instance for convex hulls does -- ~130 axes per triangle, each a walk over the hull's corners: 0.80 -> 0.78 ms for 3.5k hulls)
			const uint32_t* src = (const uint32_t*)X.hull; uint32_t* dst = (uint32_t*)&L.hull;
			for (uint32_t i = (uint32_t)sub; i < sizeof(sgd_hull) / 4; i += MESH_GROUP) dst[i] = src[i];
			X.hull = (const sgd_hull*)(const void*)&L.hull;
		}
		mh = d.meshes[(uint32_t)d.prop[2 * (size_t)mid + 1].x];
		mpos = V3(d.pose[2 * (size_t)mid]); R = quat_to_m33(Q4(d.pose[2 * (size_t)mid + 1]));
	}
	// the query box in the mesh frame: bounds of the box's 8 corners, a little generous (every lane of the group: the same operands, the same box)
	v3 llo = V3(3.4e38f, 3.4e38f, 3.4e38f), lhi = V3(-3.4e38f, -3.4e38f, -3.4e38f);
	if (valid) {
		for (int k = 0; k < 8; ++k) {
			const v3 c = V3((k & 1) ? qhi.x : qlo.x, (k & 2) ? qhi.y : qlo.y, (k & 4) ? qhi.z : qlo.z);
			const v3 l = m33_tmul(R, v3_sub(c, mpos));
			llo = V3(fminf(llo.x, l.x), fminf(llo.y, l.y), fminf(llo.z, l.z)); lhi = V3(fmaxf(lhi.x, l.x), fmaxf(lhi.y, l.y), fmaxf(lhi.z, l.z));
		}
		const float pad = 1.0e-4f * (1.0f + fabsf(llo.x) + fabsf(llo.y) + fabsf(llo.z) + fabsf(lhi.x) + fabsf(lhi.y) + fabsf(lhi.z));
		llo = v3_sub(llo, V3(pad, pad, pad)); lhi = v3_add(lhi, V3(pad, pad, pad));
	}
	// The candidates: the tree level by level, the lanes of the group taking the nodes of a level MESH_GROUP at a time -- a level is one fetch deep whatever
	// it holds, where the depth-first walk of one lane is a chain of every node it visits (a small body: ~70 dependent fetches, 45 us; a car-sized box on a
	// fine mesh: hundreds); the two frontiers live in the arrays the sort uses afterwards.  The SET found is that of the depth-first walk unless a table
	// overflows -- then the answer depends on the order of the walk, and lane 0 repeats it depth-first (eight lanes per pair: it then gives up once it holds
	// more than MESH_BIG_MIN -- the pair is passed on to the wave-per-pair launch; a table of 64 that overflowed says as much).
	if (sub == 0) { L.n_front[0] = valid ? 1u : 0u; L.n_front[1] = 0u; L.n_found = 0u; L.redo = 0u; L.key[0] = 0u; L.mc.ng = 0; }
	__syncthreads();
	{
		for (int level = 0; level < 64; ++level) {
			uint32_t* cur = (level & 1) ? L.cand : L.key; uint32_t* nxt = (level & 1) ? L.key : L.cand;
			// (a table overflowed: what the frontiers hold no longer matters; eight lanes that hold more than they will keep: the pair is passed on as soon as that is known)
			const uint32_t ncur = (L.redo || (MESH_GROUP != 64 && L.n_found > (uint32_t)MESH_BIG_MIN)) ? 0u : L.n_front[level & 1];
			if (!__any(ncur != 0u)) break;
			for (uint32_t i = (uint32_t)sub; i < ncur; i += MESH_GROUP) {
				const MeshNode nd = d.mesh_nodes[mh.node_off + cur[i]];
				if (nd.mxx < llo.x || nd.mnx > lhi.x || nd.mxy < llo.y || nd.mny > lhi.y || nd.mxz < llo.z || nd.mnz > lhi.z) continue;
				if (nd.count == 0) {
					const uint32_t at = atomicAdd(&L.n_front[(level & 1) ^ 1], 2u);
					if (at + 2u <= (uint32_t)MESH_LDS_T(MESH_GROUP, KINDS)::CAP) { nxt[at] = nd.left; nxt[at + 1] = nd.right; } else L.redo = 1u;
				} else {
					const uint32_t at = atomicAdd(&L.n_found, nd.count);
					if (at + nd.count <= (uint32_t)MESH_LDS_T(MESH_GROUP, KINDS)::CAP) { for (uint32_t k = 0; k < nd.count; ++k) L.found[at + k] = nd.left + k; } else L.redo = 1u;
				}
			}
			__syncthreads();
			if (sub == 0) L.n_front[level & 1] = 0u;
			__syncthreads();
		}
	}
	if (valid && L.redo && !(MESH_GROUP != 64 && L.n_found > (uint32_t)MESH_BIG_MIN)) {
		if (sub == 0) L.n_found = (uint32_t)mesh_candidates_found(d, mh, llo, lhi, L.found, L.key, &dropped, MESH_GROUP == 64 ? MESH_BIG_CAP : MESH_BIG_MIN, MESH_LDS_T(MESH_GROUP, KINDS)::CAP);
	}
	__syncthreads();
	nc = valid ? (int)min(L.n_found, (uint32_t)MESH_LDS_T(MESH_GROUP, KINDS)::CAP) : 0;
	if (MESH_GROUP != 64 && nc > MESH_BIG_MIN) {
		// too many triangles for eight lanes: the wave-per-pair launch takes the pair (and finds its candidates again)
		if (sub == 0) { const uint32_t kb = atomicAdd(&d.ctr->n_mesh_big, 1u); if (kb < d.cap_mesh_pairs) d.mesh_big[kb] = pair; else atomicAdd(&d.ctr->pairs_dropped, 1u); }      // (four lists feed this one: bounded like them, the excess is counted)
		valid = false; nc = 0; dropped = false;
	}
	for (int i = sub; i < nc; i += MESH_GROUP) L.key[i] = MESH_TRI_INDEX(d.mesh_tris[mh.tri_off + L.found[i]].w);
	__syncthreads();
	// candidates in the order of the caller's triangle indices: the rank of a key is the number of smaller keys
	for (int i = sub; i < nc; i += MESH_GROUP) {
		const uint32_t ki = L.key[i];
		int rank = 0;
		for (int j = 0; j < nc; ++j) rank += L.key[j] < ki ? 1 : 0;
		L.cand[rank] = L.found[i];
	}
	__syncthreads();
	// A round lasts as long as its slowest lane, and a lane whose triangle's own bounds miss the body's is done at once while its neighbour clips polygons: the
	// candidates of a tree leaf are mostly such misses (a 0.4 m box on a terrain: 8 - 32 candidates, 2 - 6 of them near it -- four rounds of which three held one
	// real test among their 64 lanes).  So the cheap bounds test runs first, over all candidates, and the survivors are packed (order kept: the merge below
	// depends on it, the set of hits does not) -- the rounds then hold real tests only.  L.key is free after the sort and takes the packed list.
	{
		int pre = (nc + MESH_GROUP - 1) / MESH_GROUP;
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) pre = max(pre, __shfl_xor(pre, off, 64));
		int kept = 0;
		for (int rd = 0; rd < pre; ++rd) {
			const int k = rd * MESH_GROUP + sub;
			bool keep = false; uint32_t cand_k = 0u;
			if (valid && k < nc) {
				cand_k = L.cand[k];
				const uint4 tri = d.mesh_tris[mh.tri_off + cand_k];
				const v3 a = V3(d.mesh_verts[mh.vert_off + tri.x]), b = V3(d.mesh_verts[mh.vert_off + tri.y]), c = V3(d.mesh_verts[mh.vert_off + tri.z]);
				const v3 wa = v3_add(mpos, m33_mul(R, a)), wb = v3_add(mpos, m33_mul(R, b)), wc = v3_add(mpos, m33_mul(R, c));
				const v3 tmin = V3(fminf(fminf(wa.x, wb.x), wc.x), fminf(fminf(wa.y, wb.y), wc.y), fminf(fminf(wa.z, wb.z), wc.z));
				const v3 tmax = V3(fmaxf(fmaxf(wa.x, wb.x), wc.x), fmaxf(fmaxf(wa.y, wb.y), wc.y), fmaxf(fmaxf(wa.z, wb.z), wc.z));
				keep = !(tmax.x < qlo.x || tmin.x > qhi.x || tmax.y < qlo.y || tmin.y > qhi.y || tmax.z < qlo.z || tmin.z > qhi.z);
			}
			const unsigned long long all = __ballot(keep);
			const unsigned long long mine_m = MESH_GROUP == 64 ? all : ((all >> (grp * MESH_GROUP)) & ((1ull << (MESH_GROUP & 63)) - 1ull));
			if (keep) L.key[kept + __popcll(mine_m & ((1ull << sub) - 1ull))] = cand_k;
			kept += __popcll(mine_m);
		}
		__syncthreads();
		nc = valid ? kept : 0;
	}
	int rounds = (nc + MESH_GROUP - 1) / MESH_GROUP;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) rounds = max(rounds, __shfl_xor(rounds, off, 64));
	// a box: the order and corners of the cube template's edges, read once (the closed-form separating-axis search of sgd_tri_box_sat)
	sgd_box_code box_code; box_code.bits = 0ull;
	if ((KINDS & 2) && valid && X.type == SGD_SHAPE_BOX) box_code = sgd_box_code_of(&d.hulls[0]);
	for (int rd = 0; rd < rounds; ++rd) {
		const int k = rd * MESH_GROUP + sub;
		bool hit = false; sgd_manifold m; m.np = 0;
		if (valid && k < nc) {
			const uint4 tri = d.mesh_tris[mh.tri_off + L.key[k]];      // (the packed list: every entry passed the bounds test)
			const v3 a = V3(d.mesh_verts[mh.vert_off + tri.x]), b = V3(d.mesh_verts[mh.vert_off + tri.y]), c = V3(d.mesh_verts[mh.vert_off + tri.z]);
			{
				sgd_tri_hull_t th; v3 cen, nrm;
				sgd_tri_hull(a, b, c, &th, &cen, &nrm);
				sgd_tri_view T; T.pos = v3_add(mpos, m33_mul(R, cen)); T.R = R; T.scale = V3(1.0f, 1.0f, 1.0f); T.h = &th;
				hit = sgd_collide_tri<KINDS>(&X, &T, m33_mul(R, nrm), max_sep, &m, active_edges ? MESH_TRI_EDGES(tri.w) : 7u, movement, (KINDS & 2) ? &box_code : nullptr, lpoly) != 0;
			}
		}
		// the hits of this round into the pair's groups, in candidate order: in turn r every group of the wave merges its r-th hit (as many turns as the group
		// with the most hits has hits -- one turn per lane POSITION with a hit somewhere in the wave was up to eight turns for two or three hits per group)
		const unsigned long long hits = __ballot(hit);
		const unsigned long long mine_h = MESH_GROUP == 64 ? hits : ((hits >> (grp * MESH_GROUP)) & ((1ull << (MESH_GROUP & 63)) - 1ull));
		const int my_rank = __popcll(mine_h & ((1ull << sub) - 1ull));
		int n_turns = __popcll(mine_h);
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) n_turns = max(n_turns, __shfl_xor(n_turns, off, 64));
		// (eight lanes per pair: the lanes share a merge's comparisons; a wave per pair: dozens of turns per round, and a turn of the shared form -- thirty
		// lane-to-lane moves of the manifold and six barriers -- measured longer than one lane's walk through the group: 0.35 -> 0.40 ms on the car-sized boxes)
		for (int t = 0; t < n_turns; ++t) {
			if (MESH_GROUP == 8) mesh_add_coop<MESH_GROUP>(L.mc, hit && my_rank == t, m, grp, sub);
			else { if (hit && my_rank == t) sgd_mesh_add(&L.mc, &m); __syncthreads(); }
		}
	}
}

// KINDS: the shapes of the other body this instance serves (bits of SGD_SHAPE_*).  Two instances per group size: the primitives (spheres, boxes with the
// closed-form separating-axis search, capsules -- no general hull code, a fraction of the registers and scratch) and the convex hulls; the second one is
// launched only in worlds that have hulls.  G = 8 takes the lists of its kinds; G = 64 walks the one list of big pairs and skips the other instance's.
template <int MESH_GROUP, int KINDS> __global__ void __launch_bounds__(64) k_narrowphase_mesh(DV d)
{
	constexpr int MESH_PAIRS_PER_WAVE = 64 / MESH_GROUP;
	__shared__ MESH_LDS_T(MESH_GROUP, KINDS) lds[MESH_PAIRS_PER_WAVE];
	__shared__ float s_lpoly[((KINDS & 2) && !(KINDS & 8)) ? 3 * SGD_LPOLY_FLOATS : 1];      // a box's clip polygons, a column per lane (sgd_tri_box_manifold)
	const int grp = (int)(threadIdx.x / MESH_GROUP), sub = (int)(threadIdx.x % MESH_GROUP);
	MESH_LDS_T(MESH_GROUP, KINDS)& L = lds[grp];
	const float max_sep = d.st.speculative_contact_distance;
	// G = 8: a wave's eight pairs hold the same kind of body (one list per kind); G = 64: the list of the big pairs.  The lists of this instance's kinds are
	// ONE sequence of work items (an item = the next eight pairs of a list) dealt to the workgroups: with the lists taken one after the other by
	// "workgroup b takes pairs 8 b .. of every list" the first few hundred workgroups walked through a chain of spheres, THEN one of boxes, THEN one of capsules
	// while the others had nothing to do -- the launch lasted the sum of the three chains instead of the longest.
	uint32_t it_first[5], seg_base[4], seg_end[4];
	it_first[0] = 0u;
#pragma unroll
	for (uint32_t seg = 0; seg < 4u; ++seg) {
		uint32_t items = 0u; seg_base[seg] = 0u; seg_end[seg] = 0u;
		if (MESH_GROUP == 64 ? seg == 0u : ((KINDS >> seg) & 1) != 0) {
			const uint32_t seg0 = MESH_GROUP == 64 ? 0u : seg * d.cap_mesh_pairs;
			seg_end[seg] = seg0 + (MESH_GROUP == 64 ? min(d.ctr->n_mesh_big, d.cap_mesh_pairs) : min(d.ctr->n_mesh_pairs[seg], d.cap_mesh_pairs));
			seg_base[seg] = seg0 + (MESH_GROUP == 64 ? d.ctr->mesh_big_base : d.ctr->mesh_base[seg]);      // (0, or where the in-step activation round's pairs begin)
			if (seg_end[seg] > seg_base[seg]) items = (seg_end[seg] - seg_base[seg] + (uint32_t)MESH_PAIRS_PER_WAVE - 1u) / (uint32_t)MESH_PAIRS_PER_WAVE;
		}
		it_first[seg + 1] = it_first[seg] + items;
	}
	{
	for (uint32_t it = blockIdx.x; it < it_first[4]; it += gridDim.x) {
		const uint32_t seg = it >= it_first[3] ? 3u : (it >= it_first[2] ? 2u : (it >= it_first[1] ? 1u : 0u));
		const uint32_t s_first = seg == 3u ? it_first[3] : (seg == 2u ? it_first[2] : (seg == 1u ? it_first[1] : it_first[0]));
		const uint32_t n = seg == 3u ? seg_end[3] : (seg == 2u ? seg_end[2] : (seg == 1u ? seg_end[1] : seg_end[0]));
		const uint32_t p0 = (seg == 3u ? seg_base[3] : (seg == 2u ? seg_base[2] : (seg == 1u ? seg_base[1] : seg_base[0]))) + (it - s_first) * (uint32_t)MESH_PAIRS_PER_WAVE;
		const uint32_t p = p0 + (uint32_t)grp;
		bool valid = p < n;
		uint32_t mid = 0, xid = 0, fx = 0, pair = 0;
		if (valid) {
			pair = MESH_GROUP == 64 ? d.mesh_big[p] : p;
			const uint2 ab = d.mesh_pairs[pair];
			const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
			const bool mesh_a = f_shape(fa) == SGP_SHAPE_MESH, mesh_b = f_shape(fb) == SGP_SHAPE_MESH;
			if (mesh_a && mesh_b) valid = false;
			mid = mesh_a ? ab.x : ab.y; xid = mesh_a ? ab.y : ab.x; fx = mesh_a ? fb : fa;
			if (!((KINDS >> f_shape(fx)) & 1)) valid = false;      // (the other instance's pair: only the list of big pairs mixes the kinds)
		}
		sgd_shape X; v3 qlo = V3(0.0f, 0.0f, 0.0f), qhi = qlo; bool dropped = false;
		if (valid) {
			X = load_shape(d, xid, fx);
			const v3 e = V3(max_sep, max_sep, max_sep);
			qlo = v3_sub(V3(d.aabb_min[xid]), e); qhi = v3_add(V3(d.aabb_max[xid]), e);
		}
		// the movement hint of the active-edge rule (PhysicsSystem::ProcessBodyPair: mActiveEdgeMovementDirection = v1 - v2, after ApplyGravity): the forces
		// of this step are not applied yet at this point (k_pre_solve follows the narrow phase), so gravity is added here -- the same expression as the CPU statement's
		v3 movement = V3(0.0f, 0.0f, 0.0f);
		if (valid) {
			const v3 vx = v3_add(V3(d.vel[2 * (size_t)xid]), v3_scale(v3_scale(V3(d.gx, d.gy, d.gz), d.dyn[xid].z), d.sp->dt));
			movement = v3_sub(vx, f_motion(d.flags[mid]) == SGP_MOTION_STATIC ? V3(0.0f, 0.0f, 0.0f) : V3(d.vel[2 * (size_t)mid]));
		}
		mesh_pair_groups<MESH_GROUP, KINDS>(d, L, valid, X, mid, qlo, qhi, max_sep, grp, sub, pair, dropped, movement, true, s_lpoly + threadIdx.x);
		// the groups as manifolds (mesh -> body), each pruned to <= 4 points; the constraint runs lower id -> higher id, with the mesh's g-th slot
		const int ng = valid ? L.mc.ng : 0;
		if (sub < ng) {
			const sgd_mesh_group& g = L.mc.g[sub];
			sgd_manifold mm;
			sgd_hull_reduce(g.n, g.p_mesh, g.p_body, g.np, &mm);
			const uint32_t alias = mid + (uint32_t)sub;
			uint2 key;
			if (alias < xid) key = make_uint2(alias, xid); else { key = make_uint2(xid, alias); sgd_flip_manifold(&mm); }
			emit_manifold(d, key, d.flags[key.x], d.flags[key.y], mm);
		}
		if (valid && sub == 0 && dropped) atomicAdd(&d.ctr->manifolds_dropped, 1u);
		__syncthreads();          // (the tables are reused by the next eight pairs)
	}
	}
}

// The separating-axis search of one hull pair spread over the 64 lanes of a wave: lane l takes axes l, l + 64, ... of the
// flattened list [faces of A | faces of B | edge pairs]; every axis is evaluated by the same device function the sequential
// search uses, and the reduction takes (largest separation, lowest axis index) -- exactly the sequential "first maximum wins".
SGP_DEV int hull_sat_search_wave(const sgd_hview* A, const sgd_hview* B, float max_sep, sgd_hull_sat* r)
{
	const int lane = (int)(threadIdx.x & 63u);
	const int nfA = A->h->nf, nfB = B->h->nf, neA = A->h->ne, neB = B->h->ne;
	const int total = nfA + nfB + neA * neB;
	const v3 T = v3_sub(B->pos, A->pos);
	float sA = -3.4e38f, sB = -3.4e38f, sE = -3.4e38f; int iA = 0x7FFFFFFF, iB = 0x7FFFFFFF, iE = 0x7FFFFFFF;
	bool separated = false;
	// the face axes first: most candidate pairs that do not touch are told apart by one of them, and the edge pairs (nine tenths of the axes) are then never looked at
	for (int t = lane; t < nfA + nfB; t += 64) {
		if (t < nfA) {
			const float s = sgd_hull_axis_face(A, B, t);
			if (s > max_sep) separated = true;
			if (s > sA) { sA = s; iA = t; }
		} else {
			const int f = t - nfA;
			const float s = sgd_hull_axis_face(B, A, f);
			if (s > max_sep) separated = true;
			if (s > sB) { sB = s; iB = f; }
		}
	}
	if (__any(separated)) return 0;
	for (int e = lane; e < total - nfA - nfB; e += 64) {
		v3 ax; float s; int sup;
		if (sgd_hull_axis_edge(A, B, e / neB, e % neB, T, &ax, &s, &sup)) {
			if (s > max_sep) separated = true;
			if (s > sE && sup) { sE = s; iE = e; }
		}
	}
	if (__any(separated)) return 0;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		float os = __shfl_xor(sA, off); int oi = __shfl_xor(iA, off);
		if (os > sA || (os == sA && oi < iA)) { sA = os; iA = oi; }
		os = __shfl_xor(sB, off); oi = __shfl_xor(iB, off);
		if (os > sB || (os == sB && oi < iB)) { sB = os; iB = oi; }
		os = __shfl_xor(sE, off); oi = __shfl_xor(iE, off);
		if (os > sE || (os == sE && oi < iE)) { sE = os; iE = oi; }
	}
	r->sA = sA; r->fA = iA == 0x7FFFFFFF ? 0 : iA; r->sB = sB; r->fB = iB == 0x7FFFFFFF ? 0 : iB;
	r->sE = sE; r->eA = -1; r->eB = -1; r->nE = V3(0.0f, 0.0f, 0.0f);
	if (iE != 0x7FFFFFFF) {
		r->eA = iE / neB; r->eB = iE % neB;
		float s; int sup;
		sgd_hull_axis_edge(A, B, r->eA, r->eB, T, &r->nE, &s, &sup);      // the axis of the winning pair (same arithmetic as above)
	}
	return 1;
}

// pairs with a convex hull (hull - hull / box / sphere / capsule), in two launches:
//   k_narrowphase_hull       one WAVE per pair: polytope pairs search their separating axes in parallel (faces of both hulls and all edge pairs
//                            across the 64 lanes); a pair that survives, and every hull - sphere / capsule pair, becomes a work item;
//   k_narrowphase_hull_manifold  one THREAD per work item: the manifold (reference / incident face, clipping, reduction to <= 4 points), the
//                            sphere / capsule cases.  These are sequential by nature; with a thread per item 64 of them share a wave instead of
//                            each idling 63 lanes of its own.
struct HullWork { uint2 ab; sgd_hull_sat r; uint32_t round_other; };

__global__ void __launch_bounds__(64, 3) k_narrowphase_hull(DV d)
{
	const uint32_t n = min(d.ctr->n_hull_pairs, d.cap_hull_pairs);
	for (uint32_t p = d.ctr->hull_base + blockIdx.x; p < n; p += gridDim.x) {      // (hull_base: 0, or where the in-step activation round's pairs begin)
		const uint2 ab = d.hull_pairs[p];
		const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
		const sgd_shape sa = load_shape(d, ab.x, fa), sb = load_shape(d, ab.y, fb);
		const float max_sep = d.st.speculative_contact_distance;
		HullWork wk; wk.ab = ab;
		wk.round_other = (sa.type == SGP_SHAPE_SPHERE || sa.type == SGP_SHAPE_CAPSULE || sb.type == SGP_SHAPE_SPHERE || sb.type == SGP_SHAPE_CAPSULE) ? 1u : 0u;
		int hit = 1;
		if (!wk.round_other) {
			// canonical order (box < hull; hull - hull keeps its order), as sgd_collide_hull
			const bool flip = sa.type > sb.type;
			const sgd_shape* x = flip ? &sb : &sa; const sgd_shape* y = flip ? &sa : &sb;
			const sgd_hview hx = sgd_hull_view(x), hy = sgd_hull_view(y);
			hit = hull_sat_search_wave(&hx, &hy, max_sep, &wk.r);
		} else memset(&wk.r, 0, sizeof(wk.r));
		// work item p (no list to append to: one counter shared by ten thousand waves would cost more than the search)
		if (!hit) wk.round_other = 2u;
		if (threadIdx.x == 0) d.hull_work[p] = wk;
	}
}

__global__ void __launch_bounds__(64, 3) k_narrowphase_hull_manifold(DV d)
{
	const uint32_t n = min(d.ctr->n_hull_pairs, d.cap_hull_pairs);
	for (uint32_t k = d.ctr->hull_base + blockIdx.x * 64 + threadIdx.x; k < n; k += gridDim.x * 64) {
		const HullWork wk = d.hull_work[k];
		if (wk.round_other == 2u) continue;            // separated: nothing to do
		const uint2 ab = wk.ab;
		const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
		const sgd_shape sa = load_shape(d, ab.x, fa), sb = load_shape(d, ab.y, fb);
		const float max_sep = d.st.speculative_contact_distance;
		sgd_manifold m;
		int hit;
		if (wk.round_other) hit = sgd_collide_hull(&sa, &sb, max_sep, &m);
		else {
			const bool flip = sa.type > sb.type;
			const sgd_shape* x = flip ? &sb : &sa; const sgd_shape* y = flip ? &sa : &sb;
			const sgd_hview hx = sgd_hull_view(x), hy = sgd_hull_view(y);
			hit = sgd_hull_manifold(&hx, &hy, max_sep, &wk.r, &m);
			if (hit && flip) sgd_flip_manifold(&m);
		}
		if (hit) emit_manifold(d, ab, fa, fb, m);
	}
}

// THE BODY-ARRAY SWEEP, part 1 of 3 (k_pre_solve, then k_integrate_pose, then k_finalize).  Per body: wake it if an active body touched it this
// step; apply gravity / forces / damping / velocity clamps if it was movable when the step began (Jolt applies gravity before it finds
// collisions, so a body woken during this step gets none); leave the result in the body's velocity record together with the EFFECTIVE inverse
// mass of this step (0 unless dynamic and awake) -- the record the velocity iterations gather.  Nothing else is copied: the world-space inverse
// inertia and the material are derived by the kernels that need them (k_setup, the warm start) from the pose and property records they gather
// anyway.  A body that is asleep or static costs its 4 flag bytes: its velocity record already says (0, 0, 0 | 0) (k_sleep_apply, creation).
// Traffic per awake body: flags 4 + velocity record 32 + dyn 16 read, velocity record 32 + component scratch 8 written = 92 B
// (round 2: 205 B, of which 64 B were the per-step solver record this layout no longer has).
__global__ void __launch_bounds__(TPB) k_pre_solve(DV d)
{
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= d.cap_bodies) return;
	// everything an awake body needs is requested at once, next to the flags that say whether it is needed (one memory round trip instead of two)
	const uint32_t f0 = d.flags[i];
	const float4 lv4 = d.vel[2 * (size_t)i], av4 = d.vel[2 * (size_t)i + 1];
	const float4 dy = d.dyn[i];                                        // linear damping, angular damping, gravity factor, inverse mass
	const float dt = d.sp->dt;
	if (i >= d.sp->n_slots) return;
	uint32_t f = f0;
	if (!(f & BF_ALIVE)) return;
	const bool was_movable = f_movable(f);
	// sleeping bodies touched by an active body wake up (Jolt activates them while finding collisions)
	if (f & BF_WAKE) {
		f &= ~BF_WAKE;
		if (!(f & BF_ACTIVE)) { f |= BF_ACTIVE; push_event(d.ev_activated, &d.evc->n_activated, d.cap_bodies, i); }
		reset_sleep(d, i, f_shape(f), d.prop[2 * (size_t)i + 1], V3(d.pose[2 * (size_t)i]), Q4(d.pose[2 * (size_t)i + 1]));
	}
	if ((f & BF_ACTIVE) && f_motion(f) != SGP_MOTION_STATIC) {
		v3 lv = V3(lv4), av = V3(av4);
		float im = 0.0f;
		if (f_movable(f)) {
			im = dy.w;
			if (was_movable) {
				// K8a: forces, gravity, damping, velocity clamps (JobApplyGravity)
				v3 F = V3(0.0f, 0.0f, 0.0f), T = F;
				sym33 Iw = sym33_zero();
				if (f & BF_HAS_FORCE) {                                      // the accumulators hold something: read them, clear them
					const float4 F4v = d.force[i], T4 = d.torque[i];
					F = V3(F4v); T = V3(T4);
					Iw = world_inv_inertia(quat_to_m33(Q4(d.pose[2 * (size_t)i + 1])), V3(d.prop[2 * (size_t)i]));
					d.force[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
					d.torque[i] = make_float4(0.0f, 0.0f, 0.0f, T4.w);
					f &= ~BF_HAS_FORCE;
				}
				const v3 g = V3(d.gx, d.gy, d.gz);
				lv = v3_add(lv, v3_scale(v3_add(v3_scale(g, dy.z), v3_scale(F, im)), dt));
				av = v3_add(av, v3_scale(sym33_mul(Iw, T), dt));
				lv = v3_scale(lv, fmaxf(0.0f, 1.0f - dy.x * dt));
				av = v3_scale(av, fmaxf(0.0f, 1.0f - dy.y * dt));
				const float l2 = v3_len_sq(lv), ml = d.st.max_linear_velocity;
				if (l2 > ml * ml) lv = v3_scale(lv, ml / sqrtf(l2));
				const float a2 = v3_len_sq(av), ma = d.st.max_angular_velocity;
				if (a2 > ma * ma) av = v3_scale(av, ma / sqrtf(a2));
			}
		}
		d.vel[2 * (size_t)i] = F4(lv, im);
		d.vel[2 * (size_t)i + 1] = F4(av, 0.0f);
	}
	d.hc_root[i] = i; d.hc_count[i] = 0u;      // every body a component of its own (k_hc_hook joins them along the high-colour constraints)
	// remember whether the body was movable when the previous step coloured its constraints (colour inheritance)
	uint32_t nf = f & ~(BF_MOVABLE_PREV | BF_MOVABLE_CUR | BF_CACHE_INVALID);      // (the narrow phase of this step has seen the flag)
	if (f & BF_MOVABLE_CUR) nf |= BF_MOVABLE_PREV;
	if (f_movable(f)) nf |= BF_MOVABLE_CUR;
	if (nf != f0) d.flags[i] = nf;
}

// ---------------------------------------------------------------------------------------------------------------
// K6: deterministic round-based greedy colouring.  priority = mix64(pair key); per round each uncoloured manifold
// claims its movable bodies with atomicMin; the manifold that holds both claims takes the lowest colour free on
// both bodies.  The result depends only on the SET of manifolds (spec: DESIGN.md "Colouring").

SGP_DEV uint32_t cache_find(const DV& d, uint64_t key);

// Colour inheritance through the contact cache: a persisted manifold keeps last step's colour when both of its movable
// bodies were already movable when that colour was chosen (last step's proper colouring then guarantees that no two
// inheritors sharing a movable body carry the same colour).  Only the remaining manifolds go through the rounds.
__global__ void __launch_bounds__(TPB) k_colour_inherit(DV d)
{
	const uint32_t n = min(d.ctr->n_manifolds, d.cap_manifolds);
	for (uint32_t m = blockIdx.x * TPB + threadIdx.x; m < n; m += gridDim.x * TPB) {
		const uint2 ab = d.man_ab[m];
		// the one hash probe per manifold (unless the narrow phase already made it for a contact-cache attempt): every later kernel reads man_prev
		uint32_t mp = d.man_prev[m];
		if (mp == MAN_PREV_LOOKUP) { const uint32_t f = cache_find(d, ((uint64_t)ab.x << 32) | ab.y); mp = f == 0xFFFFFFFFu ? MAN_PREV_NONE : f; d.man_prev[m] = mp; }
		if (d.man_colour[m] != -1) continue;      // (debug bit 2: no colour inheritance -- every manifold goes through the rounds)
		const uint32_t ps = mp & ~MAN_PREV_REUSED;
		if (ps == MAN_PREV_NONE) continue;
		const int pc = (PRV(d).np_col[ps] >> 8) & 0xFF;
		if (pc >= SGP_OVERFLOW_COLOUR) continue;
		const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
		const bool ma = fa & BF_MOVABLE_CUR, mb = fb & BF_MOVABLE_CUR;
		if ((ma && !(fa & BF_MOVABLE_PREV)) || (mb && !(fb & BF_MOVABLE_PREV))) continue;
		if (pc == 0 && ((ma && chassis_colours(d, ab.x, fa)) || (mb && chassis_colours(d, ab.y, fb)))) continue;      // (the body became a chassis since: colour 0 is the vehicle's)
		d.man_colour[m] = pc;
		if (ma) atomicOr((unsigned long long*)&d.colour_mask[ab.x], 1ull << pc);
		if (mb) atomicOr((unsigned long long*)&d.colour_mask[ab.y], 1ull << pc);
	}
}

// Round 0 walks every manifold; later rounds walk the compacted worklist of still-uncoloured manifolds that the previous
// commit produced, so the work per round shrinks with the remaining set.
__global__ void __launch_bounds__(TPB) k_colour_claim(DV d, uint32_t round)
{
	const uint32_t par = round & 1;
	const uint32_t n = round == 0 ? min(d.ctr->n_manifolds, d.cap_manifolds) : d.ctr->ucount[par];
	if (blockIdx.x == 0 && threadIdx.x == 0) { d.ctr->ucount[par ^ 1] = 0; if (round >= 1 && round < 32) d.ctr->round_n[round] = n; }      // commit(round) appends to the other list
	const uint32_t* list = d.ulist[par];
	unsigned long long* claim = (unsigned long long*)d.claim[par];
	bool saw = false;
	for (uint32_t idx = blockIdx.x * TPB + threadIdx.x; idx < n; idx += gridDim.x * TPB) {
		const uint32_t m = round == 0 ? idx : list[idx];
		if (round == 0 && d.man_colour[m] != -1) continue;
		saw = true;
		const uint2 ab = d.man_ab[m];
		const unsigned long long pr = d.man_prio[m];
		if (f_movable(d.flags[ab.x])) atomicMin(&claim[ab.x], pr);
		if (f_movable(d.flags[ab.y])) atomicMin(&claim[ab.y], pr);
	}
	// a round counts when it found an uncoloured manifold
	const unsigned long long any = __ballot(saw);
	if (any && (threadIdx.x & 63) == 0) atomicMax(&d.ctr->rounds_used, round + 1);
}

__global__ void __launch_bounds__(TPB) k_colour_commit(DV d, uint32_t round)
{
	const uint32_t par = round & 1;
	const uint32_t n = round == 0 ? min(d.ctr->n_manifolds, d.cap_manifolds) : d.ctr->ucount[par];
	const uint32_t* list = d.ulist[par];
	uint32_t* out = d.ulist[par ^ 1];
	const uint64_t* claim = d.claim[par];
	uint64_t* next = d.claim[par ^ 1];
	const int lane = threadIdx.x & 63;
	for (uint32_t base = blockIdx.x * TPB; base < n; base += gridDim.x * TPB) {
		const uint32_t idx = base + threadIdx.x;
		bool lose = false; uint32_t m = 0;
		if (idx < n) {
			m = round == 0 ? idx : list[idx];
			if (!(round == 0 && d.man_colour[m] != -1)) {
				const uint2 ab = d.man_ab[m];
				const uint64_t pr = d.man_prio[m];
				const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
				const bool ma = f_movable(fa), mb = f_movable(fb);
				const bool win = (!ma || claim[ab.x] == pr) && (!mb || claim[ab.y] == pr);
				if (win) {
					const uint64_t used = (ma ? d.colour_mask[ab.x] | chassis_colours(d, ab.x, fa) : 0ull) | (mb ? d.colour_mask[ab.y] | chassis_colours(d, ab.y, fb) : 0ull);
					int col = __ffsll((long long)~used) - 1;
					if (col < 0 || col > SGP_OVERFLOW_COLOUR) col = SGP_OVERFLOW_COLOUR;
					d.man_colour[m] = col;
					if (col < SGP_OVERFLOW_COLOUR) {
						if (ma) d.colour_mask[ab.x] = d.colour_mask[ab.x] | (1ull << col);
						if (mb) d.colour_mask[ab.y] = d.colour_mask[ab.y] | (1ull << col);
					}
				} else lose = true;
				next[ab.x] = ~0ull;
				next[ab.y] = ~0ull;
			}
		}
		// losers go to the next round's worklist: one atomic per wave
		const unsigned long long mask = __ballot(lose);
		if (mask) {
			uint32_t wbase = 0;
			if (lane == 0) wbase = atomicAdd(&d.ctr->ucount[par ^ 1], (uint32_t)__popcll(mask));
			wbase = __shfl(wbase, 0, 64);
			if (lose) out[wbase + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = m;
		}
	}
}

SGP_DEV void colour_scan_block(const DV& d);
__global__ void __launch_bounds__(TPB) k_colour_count(DV d, int scan_too)
{
	__shared__ uint32_t hist[SGP_MAX_COLOURS + 3];
	__shared__ uint32_t hist4[SGP_MAX_COLOURS * 4];
	if (threadIdx.x < SGP_MAX_COLOURS + 3) hist[threadIdx.x] = 0;
	hist4[threadIdx.x] = 0;                                   // (TPB = 256 = 64 colours x 4 classes)
	__syncthreads();
	const uint32_t n = min(d.ctr->n_manifolds, d.cap_manifolds);
	uint32_t my_points = 0, my_cons = 0, my_cached = 0;          // the totals are summed per thread and reduced per wave: one LDS atomic per wave, not per manifold
	for (uint32_t m = blockIdx.x * TPB + threadIdx.x; m < n; m += gridDim.x * TPB) {
		const int c = d.man_colour[m];
		if (c < 0) continue;
		atomicAdd(&hist[c], 1u);
		{ const int npb = __float_as_int(d.man_n[m].w); const uint32_t np = (npb & 0x100) ? 0u : (uint32_t)(npb & 0xFF); my_points += np; atomicAdd(&hist4[c * 4 + (np <= 1u ? 3u : 4u - np)], 1u); }      // (class 0 = four points ... class 3 = at most one: the long manifolds get the first slots, so their waves start first)
		my_cons += 1u;
		my_cached += (d.man_prev[m] & MAN_PREV_REUSED) ? 1u : 0u;      // (statistics: manifolds taken from the body-pair contact cache)
	}
	for (int off = 32; off > 0; off >>= 1) { my_points += __shfl_down(my_points, off, 64); my_cons += __shfl_down(my_cons, off, 64); my_cached += __shfl_down(my_cached, off, 64); }
	if ((threadIdx.x & 63) == 0) { if (my_points) atomicAdd(&hist[SGP_MAX_COLOURS], my_points); if (my_cons) atomicAdd(&hist[SGP_MAX_COLOURS + 1], my_cons); if (my_cached) atomicAdd(&hist[SGP_MAX_COLOURS + 2], my_cached); }
	__syncthreads();
	if (hist4[threadIdx.x]) atomicAdd(&d.ctr->cnp_count[threadIdx.x], hist4[threadIdx.x]);
	if (threadIdx.x < SGP_MAX_COLOURS) { if (hist[threadIdx.x]) atomicAdd(&d.ctr->colour_count[threadIdx.x], hist[threadIdx.x]); }
	else if (threadIdx.x == SGP_MAX_COLOURS) { if (hist[SGP_MAX_COLOURS]) atomicAdd(&d.ctr->n_points, hist[SGP_MAX_COLOURS]); }
	else if (threadIdx.x == SGP_MAX_COLOURS + 1) { if (hist[SGP_MAX_COLOURS + 1]) atomicAdd(&d.ctr->n_constraints, hist[SGP_MAX_COLOURS + 1]); }
	else if (threadIdx.x == SGP_MAX_COLOURS + 2) { if (hist[SGP_MAX_COLOURS + 2]) atomicAdd(&d.ctr->n_cached, hist[SGP_MAX_COLOURS + 2]); }
	// the scan of the histogram (first slot of every colour and point-count class) by whoever finishes last: it was a launch of its own
	if (scan_too && last_block(&d.ctr->tickets[0])) colour_scan_block(d);
}

// Catch-all: if the planned number of rounds left manifolds uncoloured, ONE workgroup finishes the job with workgroup
// barriers between the phases (same algorithm, same result; only reached when the plan from the previous step was short).
__global__ void __launch_bounds__(1024) k_colour_finish(DV d, uint32_t first_round, int build_list)
{
	// small worlds skip the per-round launches altogether: this workgroup collects the manifolds that did not inherit a colour and runs
	// every round itself (same algorithm, same colours: the outcome of a round does not depend on the order of the worklist)
	if (build_list) {
		if (threadIdx.x == 0) d.ctr->ucount[first_round & 1] = 0;
		__syncthreads();
		const uint32_t nm = min(d.ctr->n_manifolds, d.cap_manifolds);
		for (uint32_t m = threadIdx.x; m < nm; m += 1024) if (d.man_colour[m] == -1) d.ulist[first_round & 1][atomicAdd(&d.ctr->ucount[first_round & 1], 1u)] = m;
		__threadfence();
		__syncthreads();
	}
	for (uint32_t round = first_round; round < first_round + 4096u; ++round) {
		const uint32_t par = round & 1;
		const uint32_t n = d.ctr->ucount[par];
		__syncthreads();
		if (n == 0) return;
		if (threadIdx.x == 0) { d.ctr->ucount[par ^ 1] = 0; d.ctr->rounds_used = round + 1; if (round < 32) d.ctr->round_n[round] = n; }
		const uint32_t* list = d.ulist[par];
		uint32_t* out = d.ulist[par ^ 1];
		unsigned long long* claim = (unsigned long long*)d.claim[par];
		uint64_t* next = d.claim[par ^ 1];
		for (uint32_t idx = threadIdx.x; idx < n; idx += 1024) {
			const uint32_t m = list[idx];
			const uint2 ab = d.man_ab[m];
			const unsigned long long pr = d.man_prio[m];
			if (f_movable(d.flags[ab.x])) atomicMin(&claim[ab.x], pr);
			if (f_movable(d.flags[ab.y])) atomicMin(&claim[ab.y], pr);
		}
		__threadfence();
		__syncthreads();
		for (uint32_t idx = threadIdx.x; idx < n; idx += 1024) {
			const uint32_t m = list[idx];
			const uint2 ab = d.man_ab[m];
			const uint64_t pr = d.man_prio[m];
			const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
			const bool ma = f_movable(fa), mb = f_movable(fb);
			const bool win = (!ma || claim[ab.x] == pr) && (!mb || claim[ab.y] == pr);
			if (win) {
				const uint64_t used = (ma ? d.colour_mask[ab.x] | chassis_colours(d, ab.x, fa) : 0ull) | (mb ? d.colour_mask[ab.y] | chassis_colours(d, ab.y, fb) : 0ull);
				int col = __ffsll((long long)~used) - 1;
				if (col < 0 || col > SGP_OVERFLOW_COLOUR) col = SGP_OVERFLOW_COLOUR;
				d.man_colour[m] = col;
				if (col < SGP_OVERFLOW_COLOUR) {
					if (ma) d.colour_mask[ab.x] = d.colour_mask[ab.x] | (1ull << col);
					if (mb) d.colour_mask[ab.y] = d.colour_mask[ab.y] | (1ull << col);
				}
			} else out[atomicAdd(&d.ctr->ucount[par ^ 1], 1u)] = m;
		}
		__threadfence();
		__syncthreads();
		// reset the claim words this round used (the next round's buffer was reset by the previous commit)
		for (uint32_t idx = threadIdx.x; idx < n; idx += 1024) {
			const uint2 ab = d.man_ab[list[idx]];
			claim[ab.x] = ~0ull; claim[ab.y] = ~0ull;
			next[ab.x] = ~0ull; next[ab.y] = ~0ull;
		}
		__threadfence();
		__syncthreads();
	}
}

// exclusive scan of the colour histogram -> first slot of every colour, on the device (no host round trip)
SGP_DEV void colour_scan_block(const DV& d)
{
	// buckets in (colour, point-count class) order: the start of a colour is the start of its first class
	__shared__ uint32_t wsum[4];
	const int b = threadIdx.x, lane = b & 63, wave = b >> 6;
	const uint32_t v = d.ctr->cnp_count[b];
	uint32_t x = v;
	for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
	if (lane == 63) wsum[wave] = x;
	__syncthreads();
	uint32_t base = 0;
	for (int k = 0; k < wave; ++k) base += wsum[k];
	const uint32_t start = base + x - v;
	d.ctr->cnp_start[b] = start;
	if ((b & 3) == 0) d.cstarts[b >> 2] = start;
	if (b == 255) d.cstarts[SGP_MAX_COLOURS] = start + v;
	if (b < 64) {
		const uint32_t cv = d.ctr->colour_count[b];
		const unsigned long long used = __ballot(cv != 0 && b < SGP_OVERFLOW_COLOUR);
		if (b == 0) d.ctr->n_colours = used ? 64u - (uint32_t)__clzll(used) : 0u;
	}
}
__global__ void __launch_bounds__(256) k_colour_scan(DV d) { colour_scan_block(d); }

// ---------------------------------------------------------------------------------------------------------------
// K5: contact constraint setup (Jolt ContactConstraintManager::TemplatedAddContactConstraint): contact-cache match
// for warm starting, restitution / speculative bias, effective masses.  Constraints are written colour-sorted.

SGP_DEV float axis_eff_mass(float im1, const sym33& I1, v3 r1, float im2, const sym33& I2, v3 r2, v3 axis)
{
	float inv = 0.0f;
	if (im1 > 0.0f) { const v3 c = v3_cross(r1, axis); inv = im1 + v3_dot(sym33_mul(I1, c), c); }
	if (im2 > 0.0f) { const v3 c = v3_cross(r2, axis); inv = inv + (im2 + v3_dot(sym33_mul(I2, c), c)); }
	return inv > 0.0f ? 1.0f / inv : 0.0f;
}

// rows of one (point, axis) for the velocity iterations: the two lever-arm cross products and their inverse-inertia images
SGP_DEV float4* axis_rows(const DV& d, uint32_t slot, int point, int axis) { return d.rows + (size_t)((point * 3 + axis) * 4) * d.cap_manifolds + slot; }
// (w2, w3: spare lanes of the two inverse-inertia rows; point 0 carries the first tangent there, see k_setup)
SGP_DEV void write_axis_rows(const DV& d, uint32_t slot, int point, int axis, v3 r1, v3 r2, v3 a, const sym33& I1, const sym33& I2, float w0, float w1, float w2 = 0.0f, float w3 = 0.0f)
{
	if (d.sp->compact_rows == 2u) return;            // (rows-free layout: the lanes rebuild everything from r1b / r2e / efft, half_load_rows)
	const v3 c1 = v3_cross(r1, a), c2 = v3_cross(r2, a);
	float4* p = axis_rows(d, slot, point, axis);
	const size_t st = d.cap_manifolds;
	p[0] = F4(c1, w0);
	p[st] = F4(c2, w1);
	if (d.sp->compact_rows) return;                  // (a million-body world streams its rows from HBM in every pass: half the bytes, a few flops more)
	p[2 * st] = F4(sym33_mul(I1, c1), w2);
	p[3 * st] = F4(sym33_mul(I2, c2), w3);
}

SGP_DEV uint32_t ht_hash(uint64_t key, uint32_t mask) { return (uint32_t)(sgp_mix64(key) >> 20) & mask; }

SGP_DEV uint32_t cache_find(const DV& d, uint64_t key)
{
	const uint32_t size = *d.ht_cur;          // (the part of the table the last rebuild used: k_cache_clear)
	const uint32_t mask = size - 1;
	uint32_t h = ht_hash(key, mask);
	for (uint32_t probe = 0; probe < size; ++probe) {
		const uint64_t k = d.ht_keys[h];
		if (k == key) return d.ht_vals[h];
		if (k == ~0ull) return 0xFFFFFFFFu;
		h = (h + 1) & mask;
	}
	return 0xFFFFFFFFu;
}

// Constraint slot of every manifold (colour-sorted layout): a light kernel of its own, each workgroup taking SLOTS_PER_THREAD x TPB manifolds
// per global atomic and colour -- the set-up kernel proper is heavy (218 VGPRs) and would otherwise queue for the per-colour fill counters
// once per 256 manifolds.
#define SLOTS_PER_THREAD 8
__global__ void __launch_bounds__(TPB) k_setup_slots(DV d)
{
	__shared__ uint32_t hist[SGP_MAX_COLOURS * 4];
	__shared__ uint32_t base[SGP_MAX_COLOURS * 4];
	const uint32_t n = min(d.ctr->n_manifolds, d.cap_manifolds);
	for (uint32_t c0 = blockIdx.x * TPB * SLOTS_PER_THREAD; c0 < n; c0 += gridDim.x * TPB * SLOTS_PER_THREAD) {
		hist[threadIdx.x] = 0;                                 // (TPB = 256 buckets: colour x point-count class)
		__syncthreads();
		int bins[SLOTS_PER_THREAD]; uint32_t ranks[SLOTS_PER_THREAD];
#pragma unroll
		for (int j = 0; j < SLOTS_PER_THREAD; ++j) {
			const uint32_t m = c0 + (uint32_t)j * TPB + threadIdx.x;
			const int col = m < n ? d.man_colour[m] : -1;
			bins[j] = -1;
			if (col >= 0) {
				const int npb = __float_as_int(d.man_n[m].w);
				const uint32_t np = (npb & 0x100) ? 0u : (uint32_t)(npb & 0xFF);
				bins[j] = col * 4 + (int)(np <= 1u ? 3u : 4u - np);
			}
			ranks[j] = bins[j] >= 0 ? atomicAdd(&hist[bins[j]], 1u) : 0u;
		}
		__syncthreads();
		if (hist[threadIdx.x]) base[threadIdx.x] = atomicAdd(&d.ctr->cnp_fill[threadIdx.x], hist[threadIdx.x]);
		__syncthreads();
#pragma unroll
		for (int j = 0; j < SLOTS_PER_THREAD; ++j) {
			const uint32_t m = c0 + (uint32_t)j * TPB + threadIdx.x;
			if (bins[j] >= 0) d.man_slot[m] = d.ctr->cnp_start[bins[j]] + base[bins[j]] + ranks[j];
		}
		__syncthreads();
	}
}

__global__ void __launch_bounds__(TPB) k_setup(DV d)
{
	const float dt = d.sp->dt;
	const uint32_t n = min(d.ctr->n_manifolds, d.cap_manifolds);
	for (uint32_t m = blockIdx.x * TPB + threadIdx.x; m < n; m += gridDim.x * TPB) {
		const int col = d.man_colour[m];
		if (col < 0) continue;
		const uint32_t slot = d.man_slot[m];
		const uint2 ab = d.man_ab[m];
		const uint64_t key = ((uint64_t)ab.x << 32) | ab.y;
		const float4 n4 = d.man_n[m];
		const int npb = __float_as_int(n4.w);
		const int np = (npb & 0x100) ? 0 : (npb & 0xFF);          // sensor pairs carry no points
		const v3 nrm = V3(n4);
		// per body: pose record, velocity record (velocities after gravity + the effective inverse mass: k_pre_solve), property record
		const float4 pa4 = d.pose[2 * (size_t)ab.x], qa4 = d.pose[2 * (size_t)ab.x + 1], pb4 = d.pose[2 * (size_t)ab.y], qb4 = d.pose[2 * (size_t)ab.y + 1];
		const float4 va4 = d.vel[2 * (size_t)ab.x], wa4 = d.vel[2 * (size_t)ab.x + 1], vb4 = d.vel[2 * (size_t)ab.y], wb4 = d.vel[2 * (size_t)ab.y + 1];
		const float4 ia4 = d.prop[2 * (size_t)ab.x], sa4 = d.prop[2 * (size_t)ab.x + 1], ib4 = d.prop[2 * (size_t)ab.y], sb4 = d.prop[2 * (size_t)ab.y + 1];
		const v3 posA = V3(pa4), posB = V3(pb4);
		const m33 RA = quat_to_m33(Q4(qa4)), RB = quat_to_m33(Q4(qb4));
		const float im1 = va4.w, im2 = vb4.w;
		// world-space inverse inertia of the bodies that can move (the others never use theirs)
		const sym33 I1 = im1 > 0.0f ? world_inv_inertia(RA, V3(ia4)) : sym33_zero(), I2 = im2 > 0.0f ? world_inv_inertia(RB, V3(ib4)) : sym33_zero();
		const float friction = sqrtf(sa4.w * sb4.w);
		const float restitution = fmaxf(ia4.w, ib4.w);
		const v3 t1 = v3_normalized_perpendicular(nrm);
		const v3 t2 = v3_cross(nrm, t1);
		const v3 lvA = V3(va4), avA = V3(wa4), lvB = V3(vb4), avB = V3(wb4);
		const uint32_t mprev = d.man_prev[m];
		const bool reused = mprev & MAN_PREV_REUSED;                 // the manifold came from the body-pair contact cache
		const uint32_t fslot = (mprev & ~MAN_PREV_REUSED) == MAN_PREV_NONE ? 0xFFFFFFFFu : (mprev & ~MAN_PREV_REUSED);
		const uint32_t pslot = d.st.warm_start ? fslot : 0xFFFFFFFFu;
		int pnp = 0;
		if (pslot != 0xFFFFFFFFu) pnp = PRV(d).np_col[pslot] & 0xFF;
		const v3 g = V3(d.gx, d.gy, d.gz);
		CUR(d).ab[slot] = ab;
		// (body, colour) -> constraint: a proper colouring gives every movable body at most one constraint per colour, so this
		// table needs no clearing -- its valid entries are exactly the bits of colour_mask[body] (read by k_warm_bodies)
		if (col < SGP_OVERFLOW_COLOUR) {
			if (im1 > 0.0f) d.body_con[(size_t)ab.x * SGP_MAX_COLOURS + col] = slot * 2u;
			if (im2 > 0.0f) d.body_con[(size_t)ab.y * SGP_MAX_COLOURS + col] = slot * 2u + 1u;
		}
		CUR(d).n_fric[slot] = F4(nrm, friction);
		CUR(d).key[slot] = key;
		CUR(d).np_col[slot] = np | (col << 8) | ((fslot != 0xFFFFFFFFu ? 1 : 0) << 16);
		// body-pair contact cache: a fresh manifold records where the bodies are relative to each other now; a reused one keeps the record of
		// the step its points were computed in (slow drift then ends the reuse)
		if (reused) { CUR(d).cdp[slot] = PRV(d).cdp[fslot]; CUR(d).cdr[slot] = PRV(d).cdr[fslot]; CUR(d).cnl[slot] = PRV(d).cnl[fslot]; }
		else {
			v3 dpos; quat drot;
			pair_relative_pose(posA, Q4(qa4), posB, Q4(qb4), &dpos, &drot);
			const v3 nl = m33_tmul(RB, nrm);
			CUR(d).cdp[slot] = make_float4(dpos.x, dpos.y, dpos.z, nl.x);
			CUR(d).cdr[slot] = make_float4(drot.x, drot.y, drot.z, drot.w);
			CUR(d).cnl[slot] = make_float2(nl.y, nl.z);
		}
		for (int i = 0; i < 4; ++i) {
			if (i >= np) break;
			const v3 p1 = V3(d.man_p1[i][m]), p2 = V3(d.man_p2[i][m]);
			v3 local1 = m33_tmul(RA, v3_sub(p1, posA));
			v3 local2 = m33_tmul(RB, v3_sub(p2, posB));
			if (reused) { local1 = V3(PRV(d).loc1[i][fslot]); local2 = V3(PRV(d).loc2[i][fslot]); }      // the cached body-space points themselves: no drift from re-deriving them
			float lam_n = 0.0f, lam_t1 = 0.0f, lam_t2 = 0.0f;
			for (int j = 0; j < 4; ++j) {
				if (j >= pnp) break;
				const v3 c1 = V3(PRV(d).loc1[j][pslot]), c2 = V3(PRV(d).loc2[j][pslot]);
				if (v3_len_sq(v3_sub(local1, c1)) < d.st.contact_point_preserve_lambda_max_dist_sq &&
				    v3_len_sq(v3_sub(local2, c2)) < d.st.contact_point_preserve_lambda_max_dist_sq) {
					const float4 pl = PRV(d).lam[j][pslot];
					lam_n = pl.x; lam_t1 = pl.y; lam_t2 = pl.z;
					break;
				}
			}
			const v3 mid = v3_scale(v3_add(p1, p2), 0.5f);
			const v3 r1 = v3_sub(mid, posA), r2 = v3_sub(mid, posB);
			const v3 va = v3_add(lvA, v3_cross(avA, r1));
			const v3 vb = v3_add(lvB, v3_cross(avB, r2));
			const float normal_velocity = v3_dot(v3_sub(vb, va), nrm);
			const float penetration = v3_dot(v3_sub(p1, p2), nrm);
			const float spec_bias = fmaxf(0.0f, -penetration / dt);
			float bias = spec_bias;
			if (restitution > 0.0f && normal_velocity < -d.st.min_velocity_for_restitution) {
				if (normal_velocity < -spec_bias) {
					v3 rel_acc = V3(0.0f, 0.0f, 0.0f);
					if (im2 > 0.0f) rel_acc = v3_add(rel_acc, v3_scale(g, d.dyn[ab.y].z));      // gravity factors: only bouncing contacts get here
					if (im1 > 0.0f) rel_acc = v3_sub(rel_acc, v3_scale(g, d.dyn[ab.x].z));
					const float force_dv = fminf(0.0f, v3_dot(rel_acc, nrm)) * dt;
					bias = restitution * (normal_velocity - force_dv);
				}
			}
			const float eff_n = axis_eff_mass(im1, I1, r1, im2, I2, r2, nrm);
			const float eff_t1 = axis_eff_mass(im1, I1, r1, im2, I2, r2, t1);
			const float eff_t2 = axis_eff_mass(im1, I1, r1, im2, I2, r2, t2);
			// what every velocity iteration would otherwise recompute per axis (Jolt's AxisConstraintPart keeps the same products)
			// point 0 also carries the first tangent (spare lanes of its rows): the velocity iterations then need no square root and no division
			// to rebuild the friction basis from the normal -- same function, same input, computed once instead of ten times
			write_axis_rows(d, slot, i, 0, r1, r2, nrm, I1, I2, bias, eff_n, i == 0 ? t1.x : 0.0f, i == 0 ? t1.y : 0.0f);
			write_axis_rows(d, slot, i, 1, r1, r2, t1, I1, I2, 0.0f, eff_t1, i == 0 ? t1.z : 0.0f);
			write_axis_rows(d, slot, i, 2, r1, r2, t2, I1, I2, 0.0f, eff_t2);
			CUR(d).r1b[i][slot] = F4(r1, bias);
			CUR(d).r2e[i][slot] = F4(r2, eff_n);
			CUR(d).lam[i][slot] = make_float4(lam_n, lam_t1, lam_t2, 0.0f);
			CUR(d).efft[i][slot] = make_float2(eff_t1, eff_t2);
			CUR(d).loc1[i][slot] = F4(local1, 0.0f);
			CUR(d).loc2[i][slot] = F4(local2, 0.0f);
		}
	}
}

// ---------------------------------------------------------------------------------------------------------------
// K7: sequential impulses.  One launch per colour: constraints of a colour share no movable body.

struct BodyVel { v3 lv, av; };

SGP_DEV void apply_impulse(BodyVel& A, BodyVel& B, float im1, const sym33& I1, float im2, const sym33& I2, v3 r1, v3 r2, v3 axis, float lambda)
{
	if (im1 > 0.0f) {
		A.lv = v3_sub(A.lv, v3_scale(axis, lambda * im1));
		A.av = v3_sub(A.av, v3_scale(sym33_mul(I1, v3_cross(r1, axis)), lambda));
	}
	if (im2 > 0.0f) {
		B.lv = v3_add(B.lv, v3_scale(axis, lambda * im2));
		B.av = v3_add(B.av, v3_scale(sym33_mul(I2, v3_cross(r2, axis)), lambda));
	}
}

SGP_DEV float axis_jv(const BodyVel& A, const BodyVel& B, v3 r1, v3 r2, v3 axis)
{
	return (v3_dot(axis, A.lv) + v3_dot(v3_cross(r1, axis), A.av)) - (v3_dot(axis, B.lv) + v3_dot(v3_cross(r2, axis), B.av));
}

struct PairCtx { uint2 ab; float im1, im2; sym33 I1, I2; BodyVel A, B; v3 n, t1, t2; float friction; int np; };

// `vel` / VS: where the velocity records live -- the global array (vel = d.vel) or a workgroup's copy in LDS; VS = float4 per record (2).
// The world-space inverse inertia is derived here from the pose and property records (read-only during the solve).
SGP_DEV sym33 body_world_inv_inertia(const DV& d, uint32_t body)
{
	return world_inv_inertia(quat_to_m33(Q4(d.pose[2 * (size_t)body + 1])), V3(d.prop[2 * (size_t)body]));
}
template <int VS> SGP_DEV void load_pair(const DV& d, uint32_t slot, PairCtx& c, const float4* vel)
{
	c.ab = CUR(d).ab[slot];
	const float4 nf = CUR(d).n_fric[slot];
	c.n = V3(nf); c.friction = nf.w;
	c.np = CUR(d).np_col[slot] & 0xFF;
	const float4 va = vel[VS * (size_t)c.ab.x], wa = vel[VS * (size_t)c.ab.x + 1];
	const float4 vb = vel[VS * (size_t)c.ab.y], wb = vel[VS * (size_t)c.ab.y + 1];
	c.im1 = va.w; c.im2 = vb.w;
	c.I1 = c.im1 > 0.0f ? body_world_inv_inertia(d, c.ab.x) : sym33_zero();
	c.I2 = c.im2 > 0.0f ? body_world_inv_inertia(d, c.ab.y) : sym33_zero();
	c.A.lv = V3(va); c.A.av = V3(wa);
	c.B.lv = V3(vb); c.B.av = V3(wb);
}

template <int VS> SGP_DEV void store_pair_vel(const PairCtx& c, float4* vel)
{
	if (c.im1 > 0.0f) { vel[VS * (size_t)c.ab.x] = F4(c.A.lv, c.im1); vel[VS * (size_t)c.ab.x + 1] = F4(c.A.av, 0.0f); }
	if (c.im2 > 0.0f) { vel[VS * (size_t)c.ab.y] = F4(c.B.lv, c.im2); vel[VS * (size_t)c.ab.y + 1] = F4(c.B.av, 0.0f); }
}

template <int VS> SGP_DEV void warm_start_one_t(const DV& d, uint32_t slot, float4* vel)
{
	PairCtx c;
	load_pair<VS>(d, slot, c, vel);
	c.t1 = v3_normalized_perpendicular(c.n);
	c.t2 = v3_cross(c.n, c.t1);
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (i < c.np) {
			const v3 r1 = V3(CUR(d).r1b[i][slot]), r2 = V3(CUR(d).r2e[i][slot]);
			const float4 l = CUR(d).lam[i][slot];
			if (c.friction > 0.0f) {
				apply_impulse(c.A, c.B, c.im1, c.I1, c.im2, c.I2, r1, r2, c.t1, l.y);
				apply_impulse(c.A, c.B, c.im1, c.I1, c.im2, c.I2, r1, r2, c.t2, l.z);
			}
			apply_impulse(c.A, c.B, c.im1, c.I1, c.im2, c.I2, r1, r2, c.n, l.x);
		}
	}
	store_pair_vel<VS>(c, vel);
}
SGP_DEV void warm_start_one(const DV& d, uint32_t slot) { warm_start_one_t<2>(d, slot, d.vel); }

// Warm start, one thread per BODY instead of one launch per colour.  A warm-start impulse depends only on its own constraint (cached
// lambdas, axes, lever arms) and on the inverse mass / inertia of the body it is applied to -- not on any velocity -- so what the
// colour-by-colour order does to one body is a fixed sequence of additions: its constraints in ascending colour (a body has at most one
// per colour), each contributing friction t1, friction t2, normal per point exactly as warm_start_one_t applies them.  This kernel
// replays that sequence per body from the (body, colour) table written by k_setup; same operations in the same order, hence the same
// bits, in one launch.  Constraints of the overflow colour come last in the order and are still applied serially by k_solve_tail.
SGP_DEV void warm_body_one(const DV& d, uint32_t i)
{
	if (i >= d.sp->n_slots) return;
	uint64_t mask = d.colour_mask[i] & ~(1ull << SGP_OVERFLOW_COLOUR);
	if (!mask) return;
	float4* rec = d.vel + 2 * (size_t)i;
	const float4 v4 = rec[0], w4 = rec[1];
	const float im = v4.w;
	if (!(im > 0.0f)) return;
	const sym33 I = body_world_inv_inertia(d, i);
	v3 lv = V3(v4), av = V3(w4);
	while (mask) {
		const int col = __ffsll((long long)mask) - 1;
		mask &= mask - 1;
		const uint32_t e = d.body_con[(size_t)i * SGP_MAX_COLOURS + col];
		const uint32_t slot = e >> 1;
		const bool second = e & 1u;
		const float4 nf = CUR(d).n_fric[slot];
		const int np = CUR(d).np_col[slot] & 0xFF;
		const v3 n = V3(nf);
		const v3 t1 = v3_normalized_perpendicular(n);
		const v3 t2 = v3_cross(n, t1);
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			if (k < np) {
				const v3 r = second ? V3(CUR(d).r2e[k][slot]) : V3(CUR(d).r1b[k][slot]);
				const float4 l = CUR(d).lam[k][slot];
				// apply_impulse, one side: body 1 subtracts, body 2 adds
				if (nf.w > 0.0f) {
					if (second) { lv = v3_add(lv, v3_scale(t1, l.y * im)); av = v3_add(av, v3_scale(sym33_mul(I, v3_cross(r, t1)), l.y)); }
					else        { lv = v3_sub(lv, v3_scale(t1, l.y * im)); av = v3_sub(av, v3_scale(sym33_mul(I, v3_cross(r, t1)), l.y)); }
					if (second) { lv = v3_add(lv, v3_scale(t2, l.z * im)); av = v3_add(av, v3_scale(sym33_mul(I, v3_cross(r, t2)), l.z)); }
					else        { lv = v3_sub(lv, v3_scale(t2, l.z * im)); av = v3_sub(av, v3_scale(sym33_mul(I, v3_cross(r, t2)), l.z)); }
				}
				if (second) { lv = v3_add(lv, v3_scale(n, l.x * im)); av = v3_add(av, v3_scale(sym33_mul(I, v3_cross(r, n)), l.x)); }
				else        { lv = v3_sub(lv, v3_scale(n, l.x * im)); av = v3_sub(av, v3_scale(sym33_mul(I, v3_cross(r, n)), l.x)); }
			}
		}
	}
	rec[0] = F4(lv, im);
	rec[1] = F4(av, 0.0f);
}
SGP_DEV void warm_start_one(const DV& d, uint32_t k);
SGP_DEV uint32_t overflow_next(const DV& d, uint32_t first, uint32_t count, uint64_t& last, bool& have_last);
__global__ void __launch_bounds__(TPB) k_warm_bodies(DV d)
{
	warm_body_one(d, blockIdx.x * TPB + threadIdx.x);
	// the overflow colour comes last for every body: its constraints one after the other, in priority order, once every workgroup is through
	// (round 4: it was a launch of its own that found nothing to do in almost every step)
	const uint32_t first = d.cstarts[SGP_OVERFLOW_COLOUR], count = d.cstarts[SGP_OVERFLOW_COLOUR + 1] - first;
	if (count == 0u) return;                                    // (uniform over the grid)
	if (!last_block(&d.ctr->tickets[1]) || threadIdx.x != 0) return;
	uint64_t last = 0; bool have_last = false;
	for (uint32_t it = 0; it < count; ++it) warm_start_one(d, overflow_next(d, first, count, last, have_last));
}

// One contact manifold, one velocity iteration (ContactConstraintManager::SolveVelocityConstraints): friction rows of
// every point first (they use the normal impulse of the previous iteration), then the non-penetration rows.
// The lever-arm products of every (point, axis) come precomputed from k_setup (axis_rows); they are the very values the expressions
// cross(r, axis) and I (r x axis) would yield here, so the arithmetic -- and every bit of the result -- is that of the plain
// formulation (apply_impulse / axis_jv above, which the warm start and the oracle use), at less than half the instructions.
// The value the neighbouring lane (lane ^ 1) holds: a DPP quad permutation [1, 0, 3, 2] -- a register move modifier, where __shfl_xor compiles to
// ds_bpermute_b32, a round trip through the LDS crossbar that sat on the dependent chain of every row of every constraint.
SGP_DEV float lane_swap1(float x)
{
	return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));
}
struct AxisRows { float4 c1, c2, i1, i2; };      // r1 x axis (w: bias), r2 x axis (w: effective mass), I1 (r1 x axis), I2 (r2 x axis)

SGP_DEV AxisRows load_axis_rows(const DV& d, uint32_t slot, int point, int axis)
{
	const float4* p = axis_rows(d, slot, point, axis);
	const size_t st = d.cap_manifolds;
	AxisRows r; r.c1 = p[0]; r.c2 = p[st]; r.i1 = p[2 * st]; r.i2 = p[3 * st];
	return r;
}

SGP_DEV float rows_jv(const BodyVel& A, const BodyVel& B, v3 axis, const AxisRows& r)
{
	return (v3_dot(axis, A.lv) + v3_dot(V3(r.c1), A.av)) - (v3_dot(axis, B.lv) + v3_dot(V3(r.c2), B.av));      // each body's share, then the difference
}

SGP_DEV void rows_apply(BodyVel& A, BodyVel& B, float im1, float im2, v3 axis, const AxisRows& r, float lambda)
{
	if (im1 > 0.0f) {
		A.lv = v3_sub(A.lv, v3_scale(axis, lambda * im1));
		A.av = v3_sub(A.av, v3_scale(V3(r.i1), lambda));
	}
	if (im2 > 0.0f) {
		B.lv = v3_add(B.lv, v3_scale(axis, lambda * im2));
		B.av = v3_add(B.av, v3_scale(V3(r.i2), lambda));
	}
}

// VELOCITY ITERATIONS: TWO LANES PER CONSTRAINT.  The arithmetic of one constraint is a dependent chain (every row reads the velocities the
// row before it wrote), and a launch -- or a colour phase of the single-workgroup kernels -- lasts as long as that chain in its slowest wave.
// Lane `side` (0 / 1 = body 1 / body 2 of the constraint; the two lanes are neighbours) holds its own body's velocities and its own
// body's half of every row; per row it computes its body's share of J v, swaps shares with its neighbour (one cross-lane move), computes
// the same impulse as its neighbour from the same operands, and applies it to its own body: a little over half the instructions per lane,
// and half the registers.  (The summation order of J v -- each body's share first, then the difference -- is that of the oracle's axis_jv.)
//
// A constraint half in registers: loaded once (half_load), iterated any number of times (half_solve: only this lane's body's velocities
// are gathered and scattered), lambdas written back at the end by lane 0 (half_store).
struct ConHalf {
	uint32_t body;          // this lane's body
	float4 nf; int np_col;
	v3     c[4][3];         // r x axis of this lane's body for (point, axis n / t1 / t2)
	v3     iv[4][3];        // I (r x axis) of this lane's body
	float  eff[4][3];       // effective mass of the row (both lanes)
	float  bias[4];         // of the normal row (both lanes)
	v3     t1;              // first friction direction (both lanes)
	v3     lam[4];          // accumulated impulses n, t1, t2
};

// ROWS: the row layout as a compile-time fact (0 full, 1 compact) where the launch knows it -- the colour launches: with both layouts behind a run-time
// branch the velocity kernel spilled ten registers --, -1 = read StepParams::compact_rows
template <int ROWS = -1> SGP_DEV void half_load_rows(const DV& d, uint32_t slot, int side, ConHalf& h)
{
	const int np = h.np_col & 0xFF;
	const size_t st = d.cap_manifolds;
	if (ROWS < 0 ? d.sp->compact_rows == 2u : ROWS == 2) {
		// no rows at all: the lever arm of this lane's body (r1 | bias, r2 | effective mass of the normal row: what the warm start reads anyway) and the
		// friction rows' effective masses; r x axis and I (r x axis) are rebuilt here -- the expressions k_setup evaluates for the full rows on the
		// same operands, hence the same bits.  40 bytes per point and lane where the full rows are 112: for worlds whose passes stream from HBM.
		const sym33 I = body_world_inv_inertia(d, h.body);
		const v3 n = V3(h.nf);
		h.t1 = v3_normalized_perpendicular(n);
		const v3 t2 = v3_cross(n, h.t1);
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if (i == 0 || i < np) {
				const float4 r4 = side ? CUR(d).r2e[i][slot] : CUR(d).r1b[i][slot];
				const float2 et = CUR(d).efft[i][slot];
				const v3 r = V3(r4);
				h.c[i][0] = v3_cross(r, n); h.c[i][1] = v3_cross(r, h.t1); h.c[i][2] = v3_cross(r, t2);
#pragma unroll
				for (int a = 0; a < 3; ++a) h.iv[i][a] = sym33_mul(I, h.c[i][a]);
				const float ow = lane_swap1(r4.w);
				h.eff[i][0] = side ? r4.w : ow; h.bias[i] = side ? ow : r4.w;
				h.eff[i][1] = et.x; h.eff[i][2] = et.y;
				h.lam[i] = V3(CUR(d).lam[i][slot]);
			}
		}
		return;
	}
	if (ROWS < 0 ? d.sp->compact_rows != 0u : ROWS != 0) {
		// compact rows: r x axis only; this lane rebuilds I (r x axis) from its body's pose and inertia records -- the same function of the same
		// operands k_setup evaluates for the full rows, hence the same bits
		const sym33 I = body_world_inv_inertia(d, h.body);
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if (i == 0 || i < np) {
#pragma unroll
				for (int a = 0; a < 3; ++a) {
					const float4 c4 = axis_rows(d, slot, i, a)[(size_t)side * st];
					h.c[i][a] = V3(c4); h.iv[i][a] = sym33_mul(I, V3(c4));
					const float ow = lane_swap1(c4.w);
					h.eff[i][a] = side ? c4.w : ow;
					if (a == 0) h.bias[i] = side ? ow : c4.w;
				}
				h.lam[i] = V3(CUR(d).lam[i][slot]);
			}
		}
		h.t1 = v3_normalized_perpendicular(V3(h.nf));
		return;
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		// (point 0 is read without waiting for the point count -- the slot's rows exist whatever they hold, and only a sensor pair has none
		// to use: one dependent load level less for the nine constraints in ten that have a single point)
		if (i == 0 || i < np) {
#pragma unroll
			for (int a = 0; a < 3; ++a) {
				const float4* p = axis_rows(d, slot, i, a);
				const float4 c4 = p[(size_t)side * st];          // lane 0: r1 x axis (w: bias of the normal row); lane 1: r2 x axis (w: effective mass)
				const float4 i4 = p[(size_t)(2 + side) * st];    // I1 (r1 x axis) / I2 (r2 x axis) (w of point 0: the stored tangent, see k_setup)
				h.c[i][a] = V3(c4); h.iv[i][a] = V3(i4);
				// what the other lane holds in its .w components: effective masses (lane 1), the bias (lane 0), the tangent (x, z: lane 0; y: lane 1)
				const float ow = lane_swap1(c4.w);
				h.eff[i][a] = side ? c4.w : ow;
				if (a == 0) h.bias[i] = side ? ow : c4.w;
				if (i == 0 && a < 2) {
					const float oi = lane_swap1(i4.w);
					if (a == 0) { h.t1.x = side ? oi : i4.w; h.t1.y = side ? i4.w : oi; } else h.t1.z = side ? oi : i4.w;
				}
			}
			const float4 l4 = CUR(d).lam[i][slot];
			h.lam[i] = V3(l4);
		}
	}
}

template <int ROWS = -1> SGP_DEV void half_load_known(const DV& d, uint32_t slot, int side, int np_col, uint32_t body, ConHalf& h)      // (header already known: nothing here waits for it)
{
	h.body = body;
	h.nf = CUR(d).n_fric[slot];
	h.np_col = np_col;
	half_load_rows<ROWS>(d, slot, side, h);
}
template <int ROWS = -1> SGP_DEV void half_load(const DV& d, uint32_t slot, int side, ConHalf& h)
{
	const uint2 ab = CUR(d).ab[slot];
	half_load_known<ROWS>(d, slot, side, CUR(d).np_col[slot], side ? ab.y : ab.x, h);
}

SGP_DEV void half_store(const DV& d, uint32_t slot, int side, const ConHalf& h)
{
	if (side) return;
	const int np = h.np_col & 0xFF;
#pragma unroll
	for (int i = 0; i < 4; ++i) { if (i < np) CUR(d).lam[i][slot] = F4(h.lam[i], 0.0f); }
}

// this body's share of J v for one row, the neighbour's share, their difference (share of body 1 minus share of body 2: identical on both lanes)
SGP_DEV float half_jv(v3 lv, v3 av, v3 axis, v3 c, int side)
{
	const float mine = v3_dot(axis, lv) + v3_dot(c, av);
	const float other = lane_swap1(mine);
	return side ? other - mine : mine - other;
}
SGP_DEV void half_apply(v3& lv, v3& av, float im, v3 axis, v3 iv, float lambda, int side)
{
	if (!(im > 0.0f)) return;
	if (side) { lv = v3_add(lv, v3_scale(axis, lambda * im)); av = v3_add(av, v3_scale(iv, lambda)); }
	else      { lv = v3_sub(lv, v3_scale(axis, lambda * im)); av = v3_sub(av, v3_scale(iv, lambda)); }
}

// One contact manifold, one velocity iteration (ContactConstraintManager::SolveVelocityConstraints): friction rows of every point first (they
// use the normal impulse of the previous iteration), then the non-penetration rows.  Both lanes of the constraint must call this together.
// `vel` / VS: where the velocity records live (the global array d.vel or an LDS copy; VS = 2 float4 per record).
// The arithmetic of half_solve on values: this lane's body's velocity record in (v4: linear velocity + effective inverse mass, w4: angular
// velocity), the updated record out.  Returns whether the body can move (whether the record changed).
SGP_DEV bool half_solve_core(ConHalf& h, int side, float4& v4, float4& w4, uint32_t dbg)
{
	const int np = h.np_col & 0xFF;
	const float im = v4.w, friction = h.nf.w;
	v3 lv = V3(v4), av = V3(w4);
	const v3 n = V3(h.nf);
	const v3 t1 = (dbg & 2u) ? v3_normalized_perpendicular(n) : h.t1;      // = v3_normalized_perpendicular(n), stored by k_setup
	const v3 t2 = v3_cross(n, t1);
	if (friction > 0.0f) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if (i < np && !(h.eff[i][1] <= 0.0f && h.eff[i][2] <= 0.0f)) {
				float l1 = h.lam[i].y + h.eff[i][1] * half_jv(lv, av, t1, h.c[i][1], side);
				float l2 = h.lam[i].z + h.eff[i][2] * half_jv(lv, av, t2, h.c[i][2], side);
				const float max_f = friction * h.lam[i].x;
				const float tot_sq = l1 * l1 + l2 * l2;
				if (tot_sq > max_f * max_f) { const float sc = max_f / sqrtf(tot_sq); l1 = l1 * sc; l2 = l2 * sc; }
				half_apply(lv, av, im, t1, h.iv[i][1], l1 - h.lam[i].y, side); h.lam[i].y = l1;
				half_apply(lv, av, im, t2, h.iv[i][2], l2 - h.lam[i].z, side); h.lam[i].z = l2;
			}
		}
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (i < np && h.eff[i][0] > 0.0f) {
			const float jv = half_jv(lv, av, n, h.c[i][0], side);
			const float lambda = h.eff[i][0] * (jv - h.bias[i]);
			const float nl = max0f(h.lam[i].x + lambda);
			half_apply(lv, av, im, n, h.iv[i][0], nl - h.lam[i].x, side);
			h.lam[i].x = nl;
		}
	}
	v4 = F4(lv, im); w4 = F4(av, 0.0f);
	return im > 0.0f;
}
template <int VS> SGP_DEV void half_solve(ConHalf& h, int side, float4* vel, uint32_t dbg = 0)
{
	if ((h.np_col & 0xFF) == 0) return;                    // a sensor pair: kept in the contact list, nothing to solve
	float4 v4 = vel[VS * (size_t)h.body], w4 = vel[VS * (size_t)h.body + 1];
	if (half_solve_core(h, side, v4, w4, dbg)) { vel[VS * (size_t)h.body] = v4; vel[VS * (size_t)h.body + 1] = w4; }
}

// load + one iteration + store: what a colour launch does per constraint (lanes 2k and 2k + 1 of a wave call it with the same slot)
template <int VS, int ROWS = -1> SGP_DEV void solve_velocity_pair_t(const DV& d, uint32_t slot, int side, float4* vel)
{
	ConHalf h;
	half_load<ROWS>(d, slot, side, h);
	half_solve<VS>(h, side, vel, d.dbg_flags);
	half_store(d, slot, side, h);
}

SGP_DEV void solve_position_one(const DV& d, uint32_t slot)
{
	const uint2 ab = CUR(d).ab[slot];
	const float4 nf = CUR(d).n_fric[slot];
	const v3 nrm = V3(nf);
	const int np = CUR(d).np_col[slot] & 0xFF;
	// the pose records themselves (k_integrate_pose advanced them; the corrections are made in place) + the local inverse inertia
	float4* ra = d.pose + 2 * (size_t)ab.x;
	float4* rb = d.pose + 2 * (size_t)ab.y;
	const float4 pa = ra[0], pb = rb[0];
	const float im1 = pa.w, im2 = pb.w;                 // 0 unless dynamic (and a dynamic body in a constraint is awake: touched sleepers are woken by k_pre_solve)
	quat qa = Q4(ra[1]), qb = Q4(rb[1]);
	const v3 iiA = V3(d.prop[2 * (size_t)ab.x]), iiB = V3(d.prop[2 * (size_t)ab.y]);
	v3 posA = V3(pa), posB = V3(pb);
	bool moved = false;
	m33 RA = quat_to_m33(qa), RB = quat_to_m33(qb);          // recomputed below only after a correction turned a body (same values as computing them per point)
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (i >= np) continue;
		const v3 p1 = v3_add(posA, m33_mul(RA, V3(CUR(d).loc1[i][slot])));
		const v3 p2 = v3_add(posB, m33_mul(RB, V3(CUR(d).loc2[i][slot])));
		float sep = v3_dot(v3_sub(p2, p1), nrm) + d.st.penetration_slop;
		if (sep < 0.0f) {
			sep = fmaxf(sep, -d.st.max_penetration_distance);
			const v3 mid = v3_scale(v3_add(p1, p2), 0.5f);
			const v3 r1 = v3_sub(mid, posA), r2 = v3_sub(mid, posB);
			sym33 I1 = sym33_zero(), I2 = sym33_zero();
			if (im1 > 0.0f) I1 = world_inv_inertia(RA, iiA);
			if (im2 > 0.0f) I2 = world_inv_inertia(RB, iiB);
			const float eff = axis_eff_mass(im1, I1, r1, im2, I2, r2, nrm);
			if (eff <= 0.0f) continue;
			const float lambda = -eff * d.st.baumgarte * sep;
			if (im1 > 0.0f) {
				posA = v3_sub(posA, v3_scale(nrm, lambda * im1));
				qa = quat_add_rotation_step(qa, v3_scale(sym33_mul(I1, v3_cross(r1, nrm)), -lambda));
				RA = quat_to_m33(qa);
			}
			if (im2 > 0.0f) {
				posB = v3_add(posB, v3_scale(nrm, lambda * im2));
				qb = quat_add_rotation_step(qb, v3_scale(sym33_mul(I2, v3_cross(r2, nrm)), lambda));
				RB = quat_to_m33(qb);
			}
			moved = true;
		}
	}
	if (moved) {
		if (im1 > 0.0f) { ra[0] = F4(posA, pa.w); ra[1] = make_float4(qa.x, qa.y, qa.z, qa.w); }
		if (im2 > 0.0f) { rb[0] = F4(posB, pb.w); rb[1] = make_float4(qb.x, qb.y, qb.z, qb.w); }
	}
}

// The position iteration of one manifold on TWO LANES (side 0 / 1 = body 1 / body 2, neighbouring lanes, both must call it): each lane carries
// its own body's pose, computes its own contact point and its own share of the effective mass, swaps them with its neighbour, and corrects
// its own body.  Same operands, same operations as solve_position_one (the effective mass is share of body 1 + share of body 2 there too),
// hence the same bits -- at about half the instructions per lane, which is what a position launch is made of (4700 of them per manifold).
SGP_DEV void solve_position_pair_at(const DV& d, uint32_t slot, int side, float4* rec, v3 ii);
SGP_DEV void solve_position_pair(const DV& d, uint32_t slot, int side)
{
	const uint2 ab = CUR(d).ab[slot];
	const uint32_t body = side ? ab.y : ab.x;
	solve_position_pair_at(d, slot, side, d.pose + 2 * (size_t)body, V3(d.prop[2 * (size_t)body]));      // this lane's body's pose record + its local inverse inertia
}
// What a position iteration reads of the constraint itself (this lane's side): loaded once, iterated any number of times.
struct PosHalf { float4 nf; int np; v3 loc[4]; };
SGP_DEV void pos_half_load(const DV& d, uint32_t slot, int side, int np_col, PosHalf& ph)
{
	ph.nf = CUR(d).n_fric[slot];
	ph.np = np_col & 0xFF;
#pragma unroll
	for (int i = 0; i < 4; ++i) if (i == 0 || i < ph.np) ph.loc[i] = V3(side ? CUR(d).loc2[i][slot] : CUR(d).loc1[i][slot]);      // (point 0: without waiting for the count)
}
// (rec: where this lane's body's pose record lives -- the global one, or a workgroup's copy in LDS; ii: its local inverse inertia diagonal)
SGP_DEV void pos_half_solve(const DV& d, const PosHalf& ph, int side, float4* rec, v3 ii)
{
	const v3 nrm = V3(ph.nf);
	const int np = ph.np;
	const float4 p4 = rec[0];
	const float im = p4.w;                                          // 0 unless dynamic (see solve_position_one)
	quat q = Q4(rec[1]);
	v3 pos = V3(p4);
	bool moved = false;
	m33 R = quat_to_m33(q);
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (i >= np) continue;
		const v3 mine = v3_add(pos, m33_mul(R, ph.loc[i]));
		const v3 other = V3(lane_swap1(mine.x), lane_swap1(mine.y), lane_swap1(mine.z));
		const v3 p1 = side ? other : mine, p2 = side ? mine : other;
		float sep = v3_dot(v3_sub(p2, p1), nrm) + d.st.penetration_slop;
		if (sep < 0.0f) {
			sep = fmaxf(sep, -d.st.max_penetration_distance);
			const v3 mid = v3_scale(v3_add(p1, p2), 0.5f);
			const v3 r = v3_sub(mid, pos);
			// this body's share of the inverse effective mass (axis_eff_mass: im + (I (r x n)) . (r x n), 0 for a body that cannot move)
			v3 Ic = V3(0.0f, 0.0f, 0.0f);
			float share = 0.0f;
			if (im > 0.0f) { const v3 c = v3_cross(r, nrm); Ic = sym33_mul(world_inv_inertia(R, ii), c); share = im + v3_dot(Ic, c); }
			const float oshare = lane_swap1(share);
			const float s1 = side ? oshare : share, s2 = side ? share : oshare;
			// (axis_eff_mass adds body 2's share to body 1's only when body 2 can move, and starts from it when body 1 cannot: x + 0 and 0 + x are exact)
			const float inv = s1 + s2;
			const float eff = inv > 0.0f ? 1.0f / inv : 0.0f;
			if (eff <= 0.0f) continue;
			const float lambda = -eff * d.st.baumgarte * sep;
			if (im > 0.0f) {
				if (side) { pos = v3_add(pos, v3_scale(nrm, lambda * im)); q = quat_add_rotation_step(q, v3_scale(Ic, lambda)); }
				else      { pos = v3_sub(pos, v3_scale(nrm, lambda * im)); q = quat_add_rotation_step(q, v3_scale(Ic, -lambda)); }
				R = quat_to_m33(q);
			}
			moved = true;
		}
	}
	if (moved && im > 0.0f) { rec[0] = F4(pos, im); rec[1] = make_float4(q.x, q.y, q.z, q.w); }
}
SGP_DEV void solve_position_pair_at(const DV& d, uint32_t slot, int side, float4* rec, v3 ii)
{
	PosHalf ph;
	pos_half_load(d, slot, side, CUR(d).np_col[slot], ph);
	pos_half_solve(d, ph, side, rec, ii);
}

// One launch = one colour of one pass.  The slot range comes from the device-side colour table, so the host never has
// to know the counts of the current step; the grid is sized from the previous step and the loop strides over the rest.
#define SOLVE_TPB 64      // one wave per workgroup: a colour of ~17k constraints then spreads over all 256 CUs instead of 67 of them
// (velocity and position iterations: two neighbouring lanes per constraint; workgroups of four waves = 128 constraints.  Measured on config 3:
// one-wave workgroups dispatch ~1 us longer per launch than two-wave ones, two-wave ones another 0.15 us longer than four-wave ones; eight
// waves are as fast for the velocity launches and slower for the position launches)
#define SOLVE_VEL_TPB 256
#define SOLVE_XCD_CHUNKS 0x100      // flag in the colour argument: XCD-contiguous chunks (launch_solve_colour sets it for colours that fit the L2s)
template <int MODE, int ROWS = -1> __global__ void __launch_bounds__(MODE != 0 ? SOLVE_VEL_TPB : SOLVE_TPB) k_solve_colour(DV d, int colour_arg)
{
	const int colour = colour_arg & 0xFF;
	const uint32_t first = d.cstarts[colour], end = d.cstarts[colour + 1];
	if (MODE != 0) {
		// velocity and position iterations: two neighbouring lanes per constraint
		const int side = (int)(threadIdx.x & 1u);
		// Workgroups are dealt to the eight XCDs in turn, each with an L2 of its own: workgroup b takes chunk (b % 8) * (n / 8) + b / 8 of the colour's
		// slots, so that one XCD works through a CONTIGUOUS eighth of them -- neighbouring slots are neighbouring manifolds, which share bodies'
		// cache lines, and the same XCD meets the same rows again in the next pass.  (The grid is a multiple of eight: launch_solve_colour.)
		const uint32_t bx = (colour_arg & SOLVE_XCD_CHUNKS) ? xcd_block() : blockIdx.x;      // (a colour of 200k constraints -- config 4 -- streams from HBM whatever the order, and lost 17 % with the chunks)
		for (uint32_t k = first + ((bx * SOLVE_VEL_TPB + threadIdx.x) >> 1); k < end; k += gridDim.x * (SOLVE_VEL_TPB / 2)) {
			if (MODE == 1) solve_velocity_pair_t<2, ROWS>(d, k, side, d.vel); else solve_position_pair(d, k, side);
		}
		return;
	}
	for (uint32_t k = first + blockIdx.x * SOLVE_TPB + threadIdx.x; k < end; k += gridDim.x * SOLVE_TPB) warm_start_one(d, k);
}

#ifdef SGP_EXPERIMENTS
#include "experiments/sgp_solver_probe.inc"
#else
void launch_solve_probe(const DV&, int, int, uint32_t, hipStream_t) {}
#endif

// Tail colours (few constraints each) share ONE launch: a single workgroup walks colours first_colour..62 in
// order with a workgroup barrier in between (ordered exactly like separate launches), then solves the overflow
// colour 63 (a body with > 63 contacts; Jolt's non-parallel split) serially in ascending priority.  Because it covers
// every colour from first_colour on, it is also the catch-all when this step uses more colours than the plan expected.
// k_solve_tail: warm start of the overflow colour (mode 0) and position iterations (mode 2), one thread per constraint;
// k_solve_tail_vel: velocity iterations, two lanes per constraint (768 threads = 384 constraints per phase).
SGP_DEV uint32_t overflow_next(const DV& d, uint32_t first, uint32_t count, uint64_t& last, bool& have_last)
{
	// the overflow constraint with the lowest priority above `last` (selection by scanning: the overflow colour is rare and short)
	uint64_t best = ~0ull; uint32_t bslot = first;
	for (uint32_t k = 0; k < count; ++k) {
		const uint64_t pr = sgp_mix64(CUR(d).key[first + k]);
		if ((!have_last || pr > last) && pr <= best) { best = pr; bslot = first + k; }
	}
	last = best; have_last = true;
	return bslot;
}

__global__ void __launch_bounds__(512) k_solve_tail(DV d, int first_colour, int mode)
{
	// the colour table in LDS: one coalesced load instead of a dependent global load per (mostly empty) colour
	__shared__ uint32_t cs[SGP_MAX_COLOURS + 1];
	if (threadIdx.x <= SGP_MAX_COLOURS) cs[threadIdx.x] = d.cstarts[threadIdx.x];
	__syncthreads();
	if (cs[first_colour] == cs[SGP_MAX_COLOURS]) return;          // nothing from first_colour on (incl. the overflow colour)
	for (int c = first_colour; c < SGP_OVERFLOW_COLOUR; ++c) {
		const uint32_t b = cs[c], e = cs[c + 1];
		if (b == e) continue;
		for (uint32_t k = b + threadIdx.x; k < e; k += 512) {
			if (mode == 0) warm_start_one(d, k); else solve_position_one(d, k);
		}
		__syncthreads();      // workgroup scope is enough: all waves of the workgroup share one CU (one L1)
	}
	const uint32_t first = cs[SGP_OVERFLOW_COLOUR], count = cs[SGP_OVERFLOW_COLOUR + 1] - first;
	if (count == 0 || threadIdx.x != 0) return;
	uint64_t last = 0; bool have_last = false;
	for (uint32_t it = 0; it < count; ++it) {
		const uint32_t bslot = overflow_next(d, first, count, last, have_last);
		if (mode == 0) warm_start_one(d, bslot); else solve_position_one(d, bslot);
	}
}

#define TAIL_VEL_TPB 768    // 384 constraints per phase; 3 waves per SIMD (a constraint half needs ~150 registers)
template <int ROWS> __global__ void __launch_bounds__(TAIL_VEL_TPB) k_solve_tail_vel(DV d, int first_colour)
{
	__shared__ uint32_t cs[SGP_MAX_COLOURS + 1];
	if (threadIdx.x <= SGP_MAX_COLOURS) cs[threadIdx.x] = d.cstarts[threadIdx.x];
	__syncthreads();
	if (cs[first_colour] == cs[SGP_MAX_COLOURS]) return;
	const int side = (int)(threadIdx.x & 1u);
	const uint32_t pair = threadIdx.x >> 1;
	const uint32_t tail_n = cs[SGP_OVERFLOW_COLOUR] - cs[first_colour];
	if (tail_n <= TAIL_VEL_TPB / 2 && !(d.dbg_flags & 1u)) {
		// one constraint per lane pair, read once up front (all loads in flight together); a colour phase is then only the velocity gather,
		// the arithmetic and the scatter.  Phases and their order are those of the loop below.
		const uint32_t slot = cs[first_colour] + pair;
		const bool mine = pair < tail_n;
		ConHalf h; int my_col = -1;
		if (mine) { half_load<ROWS>(d, slot, side, h); my_col = (h.np_col >> 8) & 0xFF; }
		for (int c = first_colour; c < SGP_OVERFLOW_COLOUR; ++c) {
			if (cs[c] == cs[c + 1]) continue;
			if (my_col == c) half_solve<2>(h, side, d.vel, d.dbg_flags);
			__syncthreads();
		}
		if (mine) half_store(d, slot, side, h);
	} else
	for (int c = first_colour; c < SGP_OVERFLOW_COLOUR; ++c) {
		const uint32_t b = cs[c], e = cs[c + 1];
		if (b == e) continue;
		for (uint32_t k = b + pair; k < e; k += TAIL_VEL_TPB / 2) solve_velocity_pair_t<2, ROWS>(d, k, side, d.vel);
		__syncthreads();
	}
	const uint32_t first = cs[SGP_OVERFLOW_COLOUR], count = cs[SGP_OVERFLOW_COLOUR + 1] - first;
	if (count == 0 || threadIdx.x >= 2) return;                   // lanes 0 and 1: the two sides of one constraint at a time
	uint64_t last = 0; bool have_last = false;
	for (uint32_t it = 0; it < count; ++it) {
		const uint32_t bslot = overflow_next(d, first, count, last, have_last);
		solve_velocity_pair_t<2, ROWS>(d, bslot, side, d.vel);
	}
}

// ---- high colours by connected component ----------------------------------------------------------------------------------------
// The colour histogram of a pile is geometric: the first few colours hold almost every constraint, colour after colour the rest halves
// (config 3: 47k, 42k, ... 9k, 5.6k, 3.4k, 2k, 1.1k ...), yet every one of them costs a launch per pass.  The constraints of colours
// >= K form a sparse sub-graph of the contact graph that falls apart into thousands of small connected components (config 3, K = 10:
// 18k constraints in 6k components of at most 61 constraints).  Components share no body that can move, so each can be solved on its
// own, its constraints in colour order, while the other components run -- which is exactly the order of operations per body that the
// colour-by-colour launches produce.  So: label the components once per step (union-find over the bodies with an inverse mass), lay
// them out in size classes (1, 2, 4 .. 128 constraints, aligned so that a workgroup of 128 lane pairs holds whole components), and
// replace the launches of ALL colours >= K of a pass by ONE launch in which a workgroup walks those colours over the constraints in
// its registers.  Bit-identical for every K (K is a launch-plan knob, chosen on the host from the previous step's histogram);
// a component of more than 128 constraints and the overflow colour go through the serial catch-all at the end of the same launch.
SGP_DEV uint32_t uf_prio(uint32_t x);
SGP_DEV uint32_t uf_find(const uint32_t* parent, uint32_t x);
#define HC_CLASSES 9                  // component size classes: 1 << class constraints
#define HC_WG_PAIRS 256               // lane pairs (= constraints) per workgroup (8 waves: 2 per SIMD, a constraint half keeps its ~150 registers)
#define HC_TPB (2 * HC_WG_PAIRS)
#define HC_BIG 0xFFFFFFFFu
#define HC_NONE 0xFFFFFFFFu
#define NPCOL_CATCH_ALL (1 << 17)     // np_col: the constraint's component is too large for a workgroup
#define HC_BIG_LIST (4 * HC_WG_PAIRS)  // the catch-all's own list of such constraints (up to four per lane pair; more: it searches the colours for the flag)

SGP_DEV bool hc_can_move(const DV& d, uint32_t body) { return d.vel[2 * (size_t)body].w > 0.0f; }      // effective inverse mass of the step (k_pre_solve)

// (1) a constraint between two bodies that can move joins their components (k_pre_solve made every body a component of its own);
//     the slot list is cleared to "no constraint"
__global__ void __launch_bounds__(TPB) k_hc_hook(DV d, int first_colour)
{
	const uint32_t b = d.cstarts[first_colour], e = d.cstarts[SGP_OVERFLOW_COLOUR];
	const uint32_t tid = blockIdx.x * TPB + threadIdx.x, stride = gridDim.x * TPB;
	const uint32_t lim = min(2u * (e - b) + HC_CLASSES * HC_WG_PAIRS, d.cap_hc_list);
	for (uint32_t i = tid; i < lim; i += stride) d.hc_list[i] = HC_NONE;
	for (uint32_t k = b + tid; k < e; k += stride) {
		const uint2 ab = CUR(d).ab[k];
		if (!hc_can_move(d, ab.x) || !hc_can_move(d, ab.y)) continue;
		uint32_t ra = uf_find(d.hc_root, ab.x), rb = uf_find(d.hc_root, ab.y);
		while (ra != rb) {
			const bool a_hi = uf_prio(ra) > uf_prio(rb);
			const uint32_t hi = a_hi ? ra : rb, lo = a_hi ? rb : ra;
			const uint32_t old = atomicCAS(&d.hc_root[hi], hi, lo);
			if (old == hi) break;
			ra = uf_find(d.hc_root, old); rb = uf_find(d.hc_root, lo);
		}
	}
}
SGP_DEV uint32_t hc_root_of(const DV& d, uint2 ab)
{
	const uint32_t x = hc_can_move(d, ab.x) ? ab.x : (hc_can_move(d, ab.y) ? ab.y : HC_NONE);
	return x == HC_NONE ? HC_NONE : uf_find(d.hc_root, x);
}
// (2) size of every component, and each constraint's rank within its component
__global__ void __launch_bounds__(TPB) k_hc_count(DV d, int first_colour)
{
	const uint32_t b = d.cstarts[first_colour], e = d.cstarts[SGP_OVERFLOW_COLOUR];
	for (uint32_t k = b + blockIdx.x * TPB + threadIdx.x; k < e; k += gridDim.x * TPB) {
		const uint32_t r = hc_root_of(d, CUR(d).ab[k]);
		d.hc_rank[k] = r == HC_NONE ? 0u : atomicAdd(&d.hc_count[r], 1u);
	}
}
// (3) the first constraint of a component takes the component's place in its size class (one atomic per wave and class)
__global__ void __launch_bounds__(TPB) k_hc_alloc(DV d, int first_colour)
{
	const uint32_t b = d.cstarts[first_colour], e = d.cstarts[SGP_OVERFLOW_COLOUR];
	const int lane = (int)(threadIdx.x & 63u);
	for (uint32_t k0 = b + blockIdx.x * TPB; k0 < e; k0 += gridDim.x * TPB) {          // (uniform per workgroup: the ballots below need whole waves)
		const uint32_t k = k0 + threadIdx.x;
		uint32_t r = HC_NONE, size = 0;
		if (k < e && d.hc_rank[k] == 0u) { r = hc_root_of(d, CUR(d).ab[k]); if (r != HC_NONE) size = d.hc_count[r]; }
		const bool lead = r != HC_NONE;
		int cls = -1;
		if (lead) {
			if (size > (uint32_t)HC_WG_PAIRS) d.hc_base[r] = HC_BIG;
			else cls = size <= 1u ? 0 : 32 - __clz((int)(size - 1u));
		}
		// one atomic per (wave, class), all of a wave's classes in flight together: the first lane of each class asks for its class
		unsigned long long mine_m = 0ull;
#pragma unroll
		for (int c = 0; c < HC_CLASSES; ++c) { const unsigned long long m = __ballot(cls == c); if (cls == c) mine_m = m; }
		const int leader = mine_m ? __ffsll((long long)mine_m) - 1 : lane;
		uint32_t base = 0;
		if (cls >= 0 && lane == leader) base = atomicAdd(&d.ctr->hc_class[cls], (uint32_t)__popcll(mine_m));
		base = __shfl(base, leader, 64);
		if (cls >= 0) d.hc_base[r] = ((uint32_t)cls << 28) | (base + (uint32_t)__popcll(mine_m & ((1ull << lane) - 1ull)));
	}
}
// first list entry of a size class: the classes follow each other, each padded to whole workgroups
SGP_DEV uint32_t hc_class_first(const DV& d, int cls)
{
	uint32_t first = 0;
	for (int c = 0; c < cls; ++c) first += ((d.ctr->hc_class[c] << c) + (HC_WG_PAIRS - 1)) & ~(uint32_t)(HC_WG_PAIRS - 1);
	return first;
}
// (4) every constraint goes to its component's place in the list (entry = lane pair of the solve launch), or is marked for the catch-all
__global__ void __launch_bounds__(TPB) k_hc_scatter(DV d, int first_colour)
{
	const uint32_t b = d.cstarts[first_colour], e = d.cstarts[SGP_OVERFLOW_COLOUR];
	if (blockIdx.x == 0 && threadIdx.x == 0) { d.ctr->hc_entries = hc_class_first(d, HC_CLASSES); d.ctr->hc_n = e - b; }
	for (uint32_t k = b + blockIdx.x * TPB + threadIdx.x; k < e; k += gridDim.x * TPB) {
		const uint32_t r = hc_root_of(d, CUR(d).ab[k]);
		const uint32_t place = r == HC_NONE ? HC_BIG : d.hc_base[r];
		uint32_t at = HC_NONE;
		if (place != HC_BIG) {
			const int cls = (int)(place >> 28);
			at = hc_class_first(d, cls) + ((place & 0x0FFFFFFFu) << cls) + d.hc_rank[k];
		}
		if (at < d.cap_hc_list) d.hc_list[at] = k;          // (the list has room for every constraint rounded up to its class: at is always inside)
		else {
			CUR(d).np_col[k] |= NPCOL_CATCH_ALL;
			const uint32_t bi = wave_alloc(&d.ctr->hc_n_big);
			if (bi < HC_BIG_LIST) d.hc_big_list[bi] = k;          // (the catch-all walks this list instead of searching the colours for the flag)
		}
	}
}

// One pass over every constraint of colours >= first_colour (and the overflow colour).  MODE 1: velocity iteration, 2: position iteration.
// Probe (launch plan, every so often): would the colours >= first_colour -- one more than the plan uses now -- still fall apart into components
// a workgroup can hold?  Counts the constraints that would not (hc_probe_big); k_hc_init then resets the union-find for the real build.
__global__ void __launch_bounds__(TPB) k_hc_init(DV d, int first_colour)
{
	const uint32_t b = d.cstarts[first_colour], e = d.cstarts[SGP_OVERFLOW_COLOUR];
	for (uint32_t k = b + blockIdx.x * TPB + threadIdx.x; k < e; k += gridDim.x * TPB) {
		const uint2 ab = CUR(d).ab[k];
		if (hc_can_move(d, ab.x)) { d.hc_root[ab.x] = ab.x; d.hc_count[ab.x] = 0u; }
		if (hc_can_move(d, ab.y)) { d.hc_root[ab.y] = ab.y; d.hc_count[ab.y] = 0u; }
	}
}
__global__ void __launch_bounds__(TPB) k_hc_probe(DV d, int first_colour)
{
	const uint32_t b = d.cstarts[first_colour], e = d.cstarts[SGP_OVERFLOW_COLOUR];
	for (uint32_t k = b + blockIdx.x * TPB + threadIdx.x; k < e; k += gridDim.x * TPB) {
		if (d.hc_rank[k] != 0u) continue;
		const uint32_t r = hc_root_of(d, CUR(d).ab[k]);
		if (r == HC_NONE) continue;
		const uint32_t size = d.hc_count[r];
		if (size > (uint32_t)HC_WG_PAIRS) atomicAdd(&d.ctr->hc_probe_big, size);
	}
}

// (5) within a workgroup's share of the list (which constraint sits on which lane pair is free), order the constraints by colour: a wave then
//     holds one or two colours and runs one or two phases of the pass, instead of every wave running every phase for a few lanes each
__global__ void __launch_bounds__(HC_WG_PAIRS) k_hc_sort(DV d)
{
	__shared__ uint32_t s_cnt[SGP_MAX_COLOURS], s_first[SGP_MAX_COLOURS];
	__shared__ uint4 s_slot[HC_WG_PAIRS];
	const uint32_t entries = d.ctr->hc_entries;
	for (uint32_t e0 = blockIdx.x * HC_WG_PAIRS; e0 < entries; e0 += gridDim.x * HC_WG_PAIRS) {
		if (threadIdx.x < SGP_MAX_COLOURS) s_cnt[threadIdx.x] = 0u;
		__syncthreads();
		const uint32_t slot = d.hc_list[e0 + threadIdx.x];
		const int npc = slot != HC_NONE ? CUR(d).np_col[slot] : 0;
		const uint2 ab = slot != HC_NONE ? CUR(d).ab[slot] : make_uint2(0u, 0u);
		const int col = slot != HC_NONE ? ((npc >> 8) & 0xFF) : SGP_MAX_COLOURS - 1;      // (unused lane pairs last)
		const uint32_t rank = atomicAdd(&s_cnt[col], 1u);
		__syncthreads();
		if (threadIdx.x < 64) {
			const uint32_t v = s_cnt[threadIdx.x];
			uint32_t x = v;
			for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if ((int)threadIdx.x >= off) x += y; }
			s_first[threadIdx.x] = x - v;
		}
		__syncthreads();
		s_slot[s_first[col] + rank] = make_uint4(slot, (uint32_t)npc, ab.x, ab.y);
		__syncthreads();
		d.hc_entry[e0 + threadIdx.x] = s_slot[threadIdx.x];      // what the solve launches read: slot, its point count and colour, its two bodies
		__syncthreads();
	}
}

// A workgroup's components own their movable bodies, so their solver records live in LDS for the whole pass (read once, written once; a
// colour phase is an LDS gather, the arithmetic and an LDS scatter): HC_TABLE hash slots keyed by body id, filled by the lanes themselves.
#define HC_TABLE 1024                 // >= 2 x the bodies a workgroup can meet (one per lane)
template <int MODE, int ROWS = -1> __global__ void __launch_bounds__(HC_TPB) k_solve_hc(DV d, int first_colour)
{
	constexpr int RS = MODE == 1 ? 2 : 3;      // float4 per body: velocity half (lin + inverse mass, ang) / pose half (pos + inverse mass, rot, inertia)
	__shared__ float4 s_rec[HC_TABLE * RS];
	__shared__ uint32_t s_key[HC_TABLE];
	__shared__ unsigned long long s_present;
	__shared__ uint32_t s_ticket;
	const int side = (int)(threadIdx.x & 1u);
	const uint32_t pair = threadIdx.x >> 1;
	const uint32_t entries = d.ctr->hc_entries;
	const int n_colours = (int)d.ctr->n_colours;
	for (uint32_t e0 = blockIdx.x * HC_WG_PAIRS; e0 < entries; e0 += gridDim.x * HC_WG_PAIRS) {
		__syncthreads();      // (everyone is done with the previous round's table and mask)
		for (uint32_t i = threadIdx.x; i < HC_TABLE; i += HC_TPB) s_key[i] = HC_NONE;
		if (threadIdx.x == 0) s_present = 0ull;
		__syncthreads();
		const uint4 entry = d.hc_entry[e0 + pair];
		const uint32_t slot = entry.x;
		const bool mine = slot != HC_NONE;
		ConHalf h; PosHalf ph; int my_col = -1;
		uint32_t body = HC_NONE, at = 0; bool owner = false;
		if (mine) {
			my_col = ((int)entry.y >> 8) & 0xFF;
			body = side ? entry.w : entry.z;
			if (MODE == 1) half_load_known<ROWS>(d, slot, side, (int)entry.y, body, h); else pos_half_load(d, slot, side, (int)entry.y, ph);
			if (side == 0) atomicOr(&s_present, 1ull << my_col);
			// this body's LDS slot; the lane that claims it brings the record in
			at = uf_prio(body) & (HC_TABLE - 1);
			for (;;) {
				const uint32_t old = atomicCAS(&s_key[at], HC_NONE, body);
				if (old == HC_NONE) { owner = true; break; }
				if (old == body) break;
				at = (at + 1) & (HC_TABLE - 1);
			}
			if (owner) {
				const float4* g = (MODE == 1 ? d.vel : d.pose) + 2 * (size_t)body;
				s_rec[RS * at] = g[0]; s_rec[RS * at + 1] = g[1];
				if (MODE != 1) s_rec[RS * at + 2] = d.prop[2 * (size_t)body];
			}
			if (MODE == 1) h.body = at;
		}
		__syncthreads();
		const unsigned long long present = s_present;
		for (int c = first_colour; c < n_colours; ++c) {
			if (!((present >> c) & 1ull)) continue;
			if (my_col == c) { if (MODE == 1) half_solve<2>(h, side, s_rec, d.dbg_flags); else pos_half_solve(d, ph, side, s_rec + RS * at, V3(s_rec[RS * at + 2])); }
			__syncthreads();
		}
		if (MODE == 1 && mine) half_store(d, slot, side, h);
		if (owner && s_rec[RS * at].w > 0.0f) {
			float4* g = (MODE == 1 ? d.vel : d.pose) + 2 * (size_t)body;
			g[0] = s_rec[RS * at]; g[1] = s_rec[RS * at + 1];
		}
	}
	// catch-all: components too large for a workgroup and the overflow colour, by the last workgroup to finish (nothing to do: no ticket either)
	const uint32_t n_big = d.ctr->hc_n_big;
	const uint32_t ofirst = d.cstarts[SGP_OVERFLOW_COLOUR], ocount = d.cstarts[SGP_OVERFLOW_COLOUR + 1] - ofirst;
	if (n_big == 0u && ocount == 0u) return;
	__syncthreads();
	if (threadIdx.x == 0) { __threadfence(); s_ticket = atomicAdd(&d.ctr->hc_done, 1u); }
	__syncthreads();
	if (s_ticket != gridDim.x - 1u) return;
	if (threadIdx.x == 0) d.ctr->hc_done = 0u;      // for the next launch
	__threadfence();          // what the other workgroups wrote (and this compute unit may still hold older copies of)
	if (n_big != 0u && n_big <= (uint32_t)HC_BIG_LIST) {
		// the constraints of the oversized components from their list, up to four per lane pair, colour by colour (constraints of one colour share no
		// movable body: any order).  Searching every colour's whole range for the flag instead cost 140 us per pass for 288 constraints -- a step of
		// 3.7 instead of 2.0 ms whenever one component of the pile outgrew a workgroup.
		uint32_t mine[4]; int mcol[4]; int cnt = 0;
		for (uint32_t e = pair; e < n_big; e += HC_WG_PAIRS) { const uint32_t k = d.hc_big_list[e]; mine[cnt] = k; mcol[cnt] = (int)((CUR(d).np_col[k] >> 8) & 0xFF); ++cnt; }
		for (int c = first_colour; c < n_colours; ++c) {
#pragma unroll
			for (int j = 0; j < 4; ++j) if (j < cnt && mcol[j] == c) { if (MODE == 1) solve_velocity_pair_t<2, ROWS>(d, mine[j], side, d.vel); else solve_position_pair(d, mine[j], side); }
			__syncthreads();
		}
	} else
	for (int c = first_colour; c < n_colours && n_big != 0u; ++c) {
		const uint32_t cb = d.cstarts[c], ce = d.cstarts[c + 1];
		for (uint32_t k = cb + pair; k < ce; k += HC_WG_PAIRS) {
			if (!(CUR(d).np_col[k] & NPCOL_CATCH_ALL)) continue;
			if (MODE == 1) solve_velocity_pair_t<2, ROWS>(d, k, side, d.vel); else solve_position_pair(d, k, side);
		}
		__syncthreads();
	}
	if (ocount == 0u || threadIdx.x >= 2u) return;               // lanes 0 and 1: the two sides of one constraint at a time
	uint64_t last = 0; bool have_last = false;
	for (uint32_t it = 0; it < ocount; ++it) {
		const uint32_t bslot = overflow_next(d, ofirst, ocount, last, have_last);
		if (MODE == 1) solve_velocity_pair_t<2, ROWS>(d, bslot, side, d.vel); else solve_position_pair(d, bslot, side);
	}
}

// Small worlds (every colour in the tail, <= SMALL_LDS_BODIES body slots, no vehicles -- i.e. a typical Substrata scene of a
// few hundred awake bodies): the warm start and ALL velocity iterations in ONE launch of one workgroup.  The velocity half of
// the solver records lives in LDS for the whole solve (a phase then costs an LDS gather instead of a dependent global one);
// phases are ordered exactly like the separate launches they replace (colour by colour, workgroup barrier in between, overflow
// colour serially by priority), so the result is bit-identical.
#define SMALL_LDS_BODIES 2048
#define SMALL_TPB 768       // velocity iterations take two lanes per constraint: 384 constraints per phase (12 waves: 3 per SIMD leaves a constraint half its ~150 registers)
__global__ void __launch_bounds__(SMALL_TPB) k_solve_small(DV d, int warm_start, int iterations)
{
	__shared__ float4 sv[2 * SMALL_LDS_BODIES];        // 64 KB: [lin vel, effective inverse mass][ang vel, -] per body slot
	__shared__ uint32_t cs[SGP_MAX_COLOURS + 1];
	const uint32_t n = min(d.sp->n_slots, (uint32_t)SMALL_LDS_BODIES);
	if (threadIdx.x <= SGP_MAX_COLOURS) cs[threadIdx.x] = d.cstarts[threadIdx.x];
	for (uint32_t i = threadIdx.x; i < 2 * n; i += SMALL_TPB) sv[i] = d.vel[i];
	__syncthreads();
	const int side = (int)(threadIdx.x & 1u);
	const uint32_t pair = threadIdx.x >> 1;
	const uint32_t all_n = cs[SGP_OVERFLOW_COLOUR];
	if (all_n != 0 && all_n <= SMALL_TPB / 2 && cs[SGP_OVERFLOW_COLOUR] == cs[SGP_MAX_COLOURS]) {
		// at most one constraint per lane pair and no overflow colour: the constraint lives in registers for the whole solve (read once,
		// lambdas written once), the velocities in LDS; a phase is an LDS gather, the arithmetic and an LDS scatter.  Same phases in
		// the same order as the general path below.
		const uint32_t slot = pair;
		const bool mine = slot < all_n;
		if (warm_start) {
			for (int c = 0; c < SGP_OVERFLOW_COLOUR; ++c) {
				const uint32_t b = cs[c], e = cs[c + 1];
				if (b == e) continue;
				if (side == 0 && slot >= b && slot < e) warm_start_one_t<2>(d, slot, sv);      // (the warm start is one thread per constraint)
				__syncthreads();
			}
		}
		ConHalf h; int my_col = -1;
		if (mine) { half_load<0>(d, slot, side, h); my_col = (h.np_col >> 8) & 0xFF; }
		for (int pass = 0; pass < iterations; ++pass) {
			for (int c = 0; c < SGP_OVERFLOW_COLOUR; ++c) {
				if (cs[c] == cs[c + 1]) continue;
				if (my_col == c) half_solve<2>(h, side, sv, d.dbg_flags);
				__syncthreads();
			}
		}
		if (mine) half_store(d, slot, side, h);
	} else
	if (cs[0] != cs[SGP_MAX_COLOURS]) {
		for (int pass = warm_start ? -1 : 0; pass < iterations; ++pass) {
			for (int c = 0; c < SGP_OVERFLOW_COLOUR; ++c) {
				const uint32_t b = cs[c], e = cs[c + 1];
				if (b == e) continue;
				if (pass < 0) { for (uint32_t k = b + threadIdx.x; k < e; k += SMALL_TPB) warm_start_one_t<2>(d, k, sv); }
				else { for (uint32_t k = b + pair; k < e; k += SMALL_TPB / 2) solve_velocity_pair_t<2, 0>(d, k, side, sv); }
				__syncthreads();
			}
			const uint32_t first = cs[SGP_OVERFLOW_COLOUR], count = cs[SGP_OVERFLOW_COLOUR + 1] - first;
			if (count != 0) {
				if (threadIdx.x < 2) {                     // lanes 0 and 1: the two sides of one constraint at a time (the warm start: lane 0 alone)
					uint64_t last = 0; bool have_last = false;
					for (uint32_t it = 0; it < count; ++it) {
						const uint32_t bslot = overflow_next(d, first, count, last, have_last);
						if (pass < 0) { if (side == 0) warm_start_one_t<2>(d, bslot, sv); } else solve_velocity_pair_t<2, 0>(d, bslot, side, sv);
					}
				}
				__syncthreads();
			}
		}
	}
	for (uint32_t i = threadIdx.x; i < 2 * n; i += SMALL_TPB) d.vel[i] = sv[i];
}

// The small-world solve with ONE THREAD PER CONSTRAINT (512 threads, the constraint's ~240 registers in one lane): for worlds of 385..512
// constraints, which the lane-pair kernel above cannot keep in registers (768 threads x 2 lanes = 384 constraints) and would re-read from
// memory in every phase (measured on 427 constraints: 0.36 ms against 0.26 ms for this kernel; below 385 the lane pairs win by 4-8 %).
// Same phases, same operands, same operations: the two kernels produce the same bits.
// A constraint held in registers: loaded once (con_load), iterated any number of times (con_solve_velocity: only the two bodies'
// velocities are gathered and scattered), lambdas written back at the end (con_store).  solve_velocity_one_t is the three in a row; the
// single-workgroup kernels (tail colours, small worlds) keep the record across their colour phases / iterations instead of re-reading it.
struct ConReg { uint2 ab; float4 nf; int np_col; AxisRows rn[4], rt1[4], rt2[4]; float4 lam[4]; };

SGP_DEV void con_load(const DV& d, uint32_t slot, ConReg& r)
{
	r.ab = CUR(d).ab[slot];
	r.nf = CUR(d).n_fric[slot];
	r.np_col = CUR(d).np_col[slot];
	const int np = r.np_col & 0xFF;
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (i < np) {
			r.rn[i] = load_axis_rows(d, slot, i, 0); r.rt1[i] = load_axis_rows(d, slot, i, 1); r.rt2[i] = load_axis_rows(d, slot, i, 2);
			r.lam[i] = CUR(d).lam[i][slot];
		}
	}
}

SGP_DEV void con_store(const DV& d, uint32_t slot, const ConReg& r)
{
	const int np = r.np_col & 0xFF;
#pragma unroll
	for (int i = 0; i < 4; ++i) { if (i < np) CUR(d).lam[i][slot] = r.lam[i]; }
}

template <int VS> SGP_DEV void con_solve_velocity(ConReg& r, float4* vel, uint32_t dbg = 0)
{
	const uint2 ab = r.ab;
	const int np = r.np_col & 0xFF;
	if (np == 0) return;                    // a sensor pair: kept in the contact list, nothing to solve
	const float4 va = vel[VS * (size_t)ab.x], wa = vel[VS * (size_t)ab.x + 1];
	const float4 vb = vel[VS * (size_t)ab.y], wb = vel[VS * (size_t)ab.y + 1];
	const float im1 = va.w, im2 = vb.w, friction = r.nf.w;
	BodyVel A, B;
	A.lv = V3(va); A.av = V3(wa); B.lv = V3(vb); B.av = V3(wb);
	const v3 n = V3(r.nf);
	const v3 t1 = (dbg & 2u) ? v3_normalized_perpendicular(n) : V3(r.rn[0].i1.w, r.rn[0].i2.w, r.rt1[0].i1.w);      // = v3_normalized_perpendicular(n), stored by k_setup (np >= 1 here)
	const v3 t2 = v3_cross(n, t1);
	if (friction > 0.0f) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if (i < np && !(r.rt1[i].c2.w <= 0.0f && r.rt2[i].c2.w <= 0.0f)) {
				float l1 = r.lam[i].y + r.rt1[i].c2.w * rows_jv(A, B, t1, r.rt1[i]);
				float l2 = r.lam[i].z + r.rt2[i].c2.w * rows_jv(A, B, t2, r.rt2[i]);
				const float max_f = friction * r.lam[i].x;
				const float tot_sq = l1 * l1 + l2 * l2;
				if (tot_sq > max_f * max_f) { const float sc = max_f / sqrtf(tot_sq); l1 = l1 * sc; l2 = l2 * sc; }
				rows_apply(A, B, im1, im2, t1, r.rt1[i], l1 - r.lam[i].y); r.lam[i].y = l1;
				rows_apply(A, B, im1, im2, t2, r.rt2[i], l2 - r.lam[i].z); r.lam[i].z = l2;
			}
		}
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (i < np && r.rn[i].c2.w > 0.0f) {
			const float jv = rows_jv(A, B, n, r.rn[i]);
			const float lambda = r.rn[i].c2.w * (jv - r.rn[i].c1.w);
			const float nl = max0f(r.lam[i].x + lambda);
			rows_apply(A, B, im1, im2, n, r.rn[i], nl - r.lam[i].x);
			r.lam[i].x = nl;
		}
	}
	if (im1 > 0.0f) { vel[VS * (size_t)ab.x] = F4(A.lv, im1); vel[VS * (size_t)ab.x + 1] = F4(A.av, 0.0f); }
	if (im2 > 0.0f) { vel[VS * (size_t)ab.y] = F4(B.lv, im2); vel[VS * (size_t)ab.y + 1] = F4(B.av, 0.0f); }
}

template <int VS> SGP_DEV void solve_velocity_one_t(const DV& d, uint32_t slot, float4* vel)
{
	ConReg r;
	con_load(d, slot, r);
	con_solve_velocity<VS>(r, vel, d.dbg_flags);
	con_store(d, slot, r);
}
__global__ void __launch_bounds__(512) k_solve_small_single(DV d, int warm_start, int iterations)
{
	__shared__ float4 sv[2 * SMALL_LDS_BODIES];        // 64 KB: [lin vel, effective inverse mass][ang vel, -] per body slot
	__shared__ uint32_t cs[SGP_MAX_COLOURS + 1];
	const uint32_t n = min(d.sp->n_slots, (uint32_t)SMALL_LDS_BODIES);
	if (threadIdx.x <= SGP_MAX_COLOURS) cs[threadIdx.x] = d.cstarts[threadIdx.x];
	for (uint32_t i = threadIdx.x; i < 2 * n; i += 512) sv[i] = d.vel[i];
	__syncthreads();
	const uint32_t all_n = cs[SGP_OVERFLOW_COLOUR];
	if (all_n != 0 && all_n <= 512u && cs[SGP_OVERFLOW_COLOUR] == cs[SGP_MAX_COLOURS]) {
		// at most one constraint per thread and no overflow colour: the constraint lives in registers for the whole solve (read once,
		// lambdas written once), the velocities in LDS; a phase is an LDS gather, the arithmetic and an LDS scatter.  Same phases in
		// the same order as the general path below.
		const uint32_t slot = threadIdx.x;
		const bool mine = slot < all_n;
		if (warm_start) {
			for (int c = 0; c < SGP_OVERFLOW_COLOUR; ++c) {
				const uint32_t b = cs[c], e = cs[c + 1];
				if (b == e) continue;
				if (slot >= b && slot < e) warm_start_one_t<2>(d, slot, sv);
				__syncthreads();
			}
		}
		ConReg r; int my_col = -1;
		if (mine) { con_load(d, slot, r); my_col = (r.np_col >> 8) & 0xFF; }
		for (int pass = 0; pass < iterations; ++pass) {
			for (int c = 0; c < SGP_OVERFLOW_COLOUR; ++c) {
				if (cs[c] == cs[c + 1]) continue;
				if (my_col == c) con_solve_velocity<2>(r, sv, d.dbg_flags);
				__syncthreads();
			}
		}
		if (mine) con_store(d, slot, r);
	} else
	if (cs[0] != cs[SGP_MAX_COLOURS]) {
		for (int pass = warm_start ? -1 : 0; pass < iterations; ++pass) {
			for (int c = 0; c < SGP_OVERFLOW_COLOUR; ++c) {
				const uint32_t b = cs[c], e = cs[c + 1];
				if (b == e) continue;
				for (uint32_t k = b + threadIdx.x; k < e; k += 512) {
					if (pass < 0) warm_start_one_t<2>(d, k, sv); else solve_velocity_one_t<2>(d, k, sv);
				}
				__syncthreads();
			}
			const uint32_t first = cs[SGP_OVERFLOW_COLOUR], count = cs[SGP_OVERFLOW_COLOUR + 1] - first;
			if (count != 0) {
				if (threadIdx.x == 0) {
					uint64_t last = 0; bool have_last = false;
					for (uint32_t it = 0; it < count; ++it) {
						uint64_t best = ~0ull; uint32_t bslot = first;
						for (uint32_t k = 0; k < count; ++k) {
							const uint64_t pr = sgp_mix64(CUR(d).key[first + k]);
							if ((!have_last || pr > last) && pr <= best) { best = pr; bslot = first + k; }
						}
						last = best; have_last = true;
						if (pass < 0) warm_start_one_t<2>(d, bslot, sv); else solve_velocity_one_t<2>(d, bslot, sv);
					}
				}
				__syncthreads();
			}
		}
	}
	for (uint32_t i = threadIdx.x; i < 2 * n; i += 512) d.vel[i] = sv[i];
}

// ---------------------------------------------------------------------------------------------------------------
// K8b: THE BODY-ARRAY SWEEP.  x += v dt, q <- normalize(rot(w dt) * q) for every active non-static body.

__global__ void __launch_bounds__(TPB) k_integrate_pose(DV d)
{
	// part 2 of 3 of the body-array sweep: the pose of every active non-static body advances by its solved velocities, in place; the position
	// iterations then correct the pose records directly.  Traffic per body: flags 4 + velocity record 32 + pose record 32 read, pose record 32
	// written = 100 B (round 2: 213 B -- it also copied the velocities back to their arrays and built a 48 B pose record per body).
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= d.cap_bodies) return;
	const uint32_t f = d.flags[i];                                      // (flags and records requested together: one memory round trip)
	const float4 v4 = d.vel[2 * (size_t)i], w4 = d.vel[2 * (size_t)i + 1];
	float4 p = d.pose[2 * (size_t)i], r4 = d.pose[2 * (size_t)i + 1];
	const float dt = d.sp->dt;
	if (i >= d.sp->n_slots) return;
	if ((f & (BF_ALIVE | BF_ACTIVE)) != (BF_ALIVE | BF_ACTIVE) || f_motion(f) == SGP_MOTION_STATIC) return;
	v3 lv = V3(v4), av = V3(w4);
	if (f_motion(f) == SGP_MOTION_DYNAMIC) {
		const float l2 = v3_len_sq(lv), ml = d.st.max_linear_velocity;
		const float a2 = v3_len_sq(av), ma = d.st.max_angular_velocity;
		const bool cl = l2 > ml * ml, ca = a2 > ma * ma;
		if (cl) lv = v3_scale(lv, ml / sqrtf(l2));
		if (ca) av = v3_scale(av, ma / sqrtf(a2));
		if (cl) d.vel[2 * (size_t)i] = F4(lv, v4.w);                       // (only a clamped velocity changes)
		if (ca) d.vel[2 * (size_t)i + 1] = F4(av, w4.w);
	}
	const v3 np = v3_add(V3(p), v3_scale(lv, dt));
	const quat q = quat_add_rotation_step(Q4(r4), v3_scale(av, dt));
	d.pose[2 * (size_t)i] = F4(np, p.w);
	d.pose[2 * (size_t)i + 1] = make_float4(q.x, q.y, q.z, q.w);
	if (f_shape(f) == SGP_SHAPE_MESH) {
		// a kinematic mesh body (a scripted platform): the two alias slots behind it -- second / third contact manifold of a pair -- share its pose
		for (uint32_t k = 1; k <= 2; ++k) { d.pose[2 * (size_t)(i + k)] = F4(np, p.w); d.pose[2 * (size_t)(i + k) + 1] = make_float4(q.x, q.y, q.z, q.w); }
	}
}

// ---------------------------------------------------------------------------------------------------------------
// K1 + K9: AABB refresh, sleep test spheres (Body::UpdateSleepStateInternal), island bookkeeping

// part 3 of 3 of the body-array sweep.  Traffic per awake body: flags 4 + pose record 32 + property record 32 + three sleep spheres 48 + timer 4
// read, AABB 32 + timer 4 + island scratch 6 written (+ a sphere that grew, + the flags when the sleep verdict changed) = 162 B
// (round 2: 269 B -- it also wrote the pose back from the solver record and rewrote every sphere and the flags every step).
__global__ void __launch_bounds__(TPB) k_finalize(DV d)
{
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= d.cap_bodies) return;
	uint32_t f = d.flags[i];                                            // (flags and records requested together: one memory round trip)
	const float4 sh = d.prop[2 * (size_t)i + 1];
	const float4 p4 = d.pose[2 * (size_t)i], r4 = d.pose[2 * (size_t)i + 1];      // (the position iterations corrected the pose records in place)
	float4 s[3];
	for (int k = 0; k < 3; ++k) s[k] = d.sleep_s[k][i];
	const float timer = d.sleep_timer[i];
	const float dt = d.sp->dt;
	if (i >= d.sp->n_slots) return;
	d.island[i] = i;
	d.island_awake[i] = 0;
	d.awake_mark[i] = 0;
	if ((f & (BF_ALIVE | BF_ACTIVE)) != (BF_ALIVE | BF_ACTIVE)) return;
	const uint32_t type = f_shape(f);
	const v3 pos = V3(p4);
	const quat q = Q4(r4);
	v3 mn, mx;
	compute_aabb(d, type, sh, pos, q, mn, mx);
	d.aabb_min[i] = F4(mn, 0.0f);
	d.aabb_max[i] = F4(mx, 0.0f);
	if (!f_movable(f)) return;
	bool can_sleep;
	if (!(f & BF_ALLOW_SLEEP) || !d.st.allow_sleeping) can_sleep = false;
	else {
		const float max_movement = d.st.point_velocity_sleep_threshold * d.st.time_before_sleep;
		v3 pts[3];
		sleep_points(d, type, sh, pos, q, pts);
		bool reset = false;
		bool grew[3];
		for (int k = 0; k < 3; ++k) {
			const v3 dd = v3_sub(pts[k], V3(s[k]));
			const float d2 = v3_len_sq(dd);
			grew[k] = d2 > s[k].w * s[k].w;
			if (grew[k]) {
				const float dl = sqrtf(d2);
				const float nr = 0.5f * (s[k].w + dl);
				const v3 c = v3_add(V3(s[k]), v3_scale(dd, (nr - s[k].w) / dl));
				s[k] = F4(c, nr);
			}
			if (s[k].w > max_movement) reset = true;
		}
		if (reset) {
			for (int k = 0; k < 3; ++k) d.sleep_s[k][i] = F4(pts[k], 0.0f);
			d.sleep_timer[i] = 0.0f;
			can_sleep = false;
		} else {
			for (int k = 0; k < 3; ++k) if (grew[k]) d.sleep_s[k][i] = s[k];      // (a test point still inside its sphere leaves the sphere as it is)
			const float t = timer + dt;
			d.sleep_timer[i] = t;
			can_sleep = t >= d.st.time_before_sleep;
		}
	}
	const uint32_t nf = can_sleep ? (f | BF_CAN_SLEEP) : (f & ~BF_CAN_SLEEP);
	if (nf != f) d.flags[i] = nf;
}

// Union-find over the sleepy bodies, linked by a random priority (a bijective hash of the body id) instead of by id: the
// body ids of a lattice-like pile are spatially ordered, and "smaller id wins" then builds parent chains as long as a row of
// the pile; with random priorities the expected depth is logarithmic.  Which member ends up as the root of a component is
// irrelevant (only the per-component awake flag is read), so this does not change any result.
SGP_DEV uint32_t uf_prio(uint32_t x) { uint32_t h = x * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; return h; }
SGP_DEV uint32_t uf_find(const uint32_t* parent, uint32_t x)
{
	uint32_t p = parent[x];
	while (p != x) { x = p; p = parent[x]; }
	return x;
}
// Marking rounds before the union-find.  A sleepy body that touches a movable body which failed the sleep test, or a sleepy body
// already marked, is certainly awake: k_island_mark propagates that along the constraints for a few rounds (plain stores of 1; a
// round also sees marks made earlier in the same launch, so a mark usually travels several hops per round).  Every mark is a true
// "stays awake", so the exact union-find below only has to process what is still unmarked -- in a jittering pile, where awake
// bodies are spread everywhere, that is almost nothing, instead of one giant component of a hundred thousand sleepy bodies; an island
// that really is about to sleep, or one whose only awake member is many hops away, still goes through the union-find, which
// yields the same set of sleepers as before.
// (the vehicles' row export, defined with the vehicle kernels below)
#define VEH_HEAD_F4 5
#define VEH_CHUNK_NORMAL 0
SGP_DEV size_t veh_chunk_at(const DV& d, uint32_t k, int i, int c) { return (size_t)c * (4u * (size_t)d.veh_cap) + 4u * (size_t)k + (size_t)i; }
// Edge k of the island graph: the contact constraints, then one link per wheel of an active vehicle that stands on a dynamic body (chassis, that
// body) -- VehicleConstraint::BuildIslands links them, so a car and the loose box under its wheel fall asleep together or not at all.
SGP_DEV uint32_t island_edges(const DV& d) { return d.ctr->n_constraints + 4u * d.n_vehicles; }
SGP_DEV bool island_edge(const DV& d, uint32_t k, uint32_t n_con, uint2& ab)
{
	if (k < n_con) { ab = CUR(d).ab[k]; return true; }
	const uint32_t e = k - n_con, v = e >> 2, i = e & 3u;
	const float4 h0 = d.veh_head[(size_t)v * VEH_HEAD_F4];
	if (!(__float_as_uint(h0.y) & 1u) || i >= __float_as_uint(h0.z)) return false;
	const uint32_t wbits = __float_as_uint(d.veh_rows[veh_chunk_at(d, v, (int)i, VEH_CHUNK_NORMAL)].w);
	if (!(wbits >> 5)) return false;
	ab = make_uint2(__float_as_uint(h0.x), (wbits >> 5) - 1u);
	return true;
}

SGP_DEV uint32_t cache_table_size(const DV& d);
__global__ void __launch_bounds__(TPB) k_island_mark(DV d, int clear_cache)
{
	// The first of the marking launches also empties the contact-cache table for this step's rebuild (nothing reads the old table after the set-up;
	// the stores ride along with a launch that waits for its gathers: k_cache_clear was a launch of its own, 6 us on the step's chain)
	if (clear_cache) {
		const uint32_t size = cache_table_size(d);
		for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < size; i += gridDim.x * TPB) d.ht_keys[i] = ~0ull;
		if (blockIdx.x == 0 && threadIdx.x == 0) *d.ht_cur = size;
	}
	// (measured, round 4: the three rounds inside ONE launch -- agent-scope loads so that marks cross the XCDs' L2s -- cost 52 us against 36 us for three
	// launches with plain accesses: the kernel boundary is the cheaper way to make the marks of a round visible everywhere)
	const uint32_t n_con = d.ctr->n_constraints, n_edges = island_edges(d);
	for (uint32_t k = blockIdx.x * TPB + threadIdx.x; k < n_edges; k += gridDim.x * TPB) {
		uint2 ab; if (!island_edge(d, k, n_con, ab)) continue;
		const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
		if (!f_movable(fa) || !f_movable(fb)) continue;
		const bool ka = !(fa & BF_CAN_SLEEP) || d.awake_mark[ab.x], kb = !(fb & BF_CAN_SLEEP) || d.awake_mark[ab.y];
		if (ka == kb) continue;
		d.awake_mark[ka ? ab.y : ab.x] = 1;
	}
}

// Island sleeping without building every island: an island sleeps iff all its members pass the sleep test.  Only
// bodies that pass it ("sleepy") are united (union by smaller root id, ECL-CC style hooking); a sleepy component is kept
// awake iff one of its members touches a movable body that failed the test.  Same result as uniting whole islands, but
// an active pile (few sleepy bodies) does almost no union work.
__global__ void __launch_bounds__(TPB) k_island_hook(DV d)
{
	const uint32_t n_con = d.ctr->n_constraints, n_edges = island_edges(d);
	for (uint32_t k = blockIdx.x * TPB + threadIdx.x; k < n_edges; k += gridDim.x * TPB) {
	uint2 ab; if (!island_edge(d, k, n_con, ab)) continue;
	const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
	if (!f_movable(fa) || !f_movable(fb)) continue;
	if (!(fa & BF_CAN_SLEEP) || !(fb & BF_CAN_SLEEP)) continue;
	if (d.awake_mark[ab.x] || d.awake_mark[ab.y]) continue;      // a marked body is known to stay awake; flag pass handles the edge
	uint32_t ra = uf_find(d.island, ab.x), rb = uf_find(d.island, ab.y);
	while (ra != rb) {
		const bool a_hi = uf_prio(ra) > uf_prio(rb);
		const uint32_t hi = a_hi ? ra : rb, lo = a_hi ? rb : ra;
		const uint32_t old = atomicCAS(&d.island[hi], hi, lo);
		if (old == hi) break;
		ra = uf_find(d.island, old); rb = uf_find(d.island, lo);
	}
	}
}

__global__ void __launch_bounds__(TPB) k_island_flag(DV d)
{
	const uint32_t n_con = d.ctr->n_constraints, n_edges = island_edges(d);
	for (uint32_t k = blockIdx.x * TPB + threadIdx.x; k < n_edges; k += gridDim.x * TPB) {
		uint2 ab; if (!island_edge(d, k, n_con, ab)) continue;
		const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
		if (!f_movable(fa) || !f_movable(fb)) continue;
		// "undecided" = sleepy and not marked awake by k_island_mark; an undecided body next to a decided-awake one keeps its component up
		const bool sa = (fa & BF_CAN_SLEEP) && !d.awake_mark[ab.x], sb = (fb & BF_CAN_SLEEP) && !d.awake_mark[ab.y];
		if (sa == sb) continue;
		d.island_awake[uf_find(d.island, sa ? ab.x : ab.y)] = 1;
	}
}

// Workgroup-wide allocation from ONE counter with one atomic: returns this thread's index if `want` (every thread of the workgroup must call it).
// Same-address atomics serialise at ~12 ns each in L2, so a per-wave atomic (1.5k of them for 100k bodies) is already ~20 us.
SGP_DEV uint32_t block_alloc(uint32_t* counter, bool want)
{
	__shared__ uint32_t s_cnt[TPB / 64];
	__shared__ uint32_t s_b;
	const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
	const unsigned long long m = __ballot(want);
	if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t tot = 0; for (int k = 0; k < TPB / 64; ++k) tot += s_cnt[k]; s_b = tot ? atomicAdd(counter, tot) : 0u; }
	__syncthreads();
	uint32_t idx = s_b + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
	for (int k = 0; k < wave; ++k) idx += s_cnt[k];
	__syncthreads();
	return idx;
}

SGP_DEV void sleep_apply_one(const DV& d, uint32_t i, bool& active);
__global__ void __launch_bounds__(TPB) k_sleep_apply(DV d)
{
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	bool active = false;
	if (i < d.sp->n_slots) sleep_apply_one(d, i, active);
	// one atomic per workgroup for the active-body count
	block_alloc(&d.ctr->n_active, active);
}

SGP_DEV void sleep_apply_one(const DV& d, uint32_t i, bool& active)
{
	uint32_t f = d.flags[i];
	if (!(f & BF_ALIVE)) return;
	if (f_movable(f)) {
		uint32_t root = i;
		if ((f & BF_CAN_SLEEP) && !d.awake_mark[i] && d.island_awake[root = uf_find(d.island, i)] == 0) {
			f &= ~(BF_ACTIVE | BF_CAN_SLEEP);
			d.flags[i] = f;
			d.sleep_label[i] = root;        // the island goes to sleep as a whole and is remembered by its root: what wakes a member wakes them all (k_wake_pairs)
			// (the record of a body that is not awake reads (0, 0, 0 | effective inverse mass 0): k_pre_solve then has nothing to write for it)
			d.vel[2 * (size_t)i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			d.vel[2 * (size_t)i + 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			push_event(d.ev_deactivated, &d.evc->n_deactivated, d.cap_bodies, i);
		}
	} else if (f_motion(f) == SGP_MOTION_KINEMATIC && (f & BF_ACTIVE)) {
		const v3 lv = V3(d.vel[2 * (size_t)i]), av = V3(d.vel[2 * (size_t)i + 1]);
		if (v3_len_sq(lv) == 0.0f && v3_len_sq(av) == 0.0f) {
			f &= ~BF_ACTIVE;
			d.flags[i] = f;
			push_event(d.ev_deactivated, &d.evc->n_deactivated, d.cap_bodies, i);
		}
	}
	active = (f & BF_ACTIVE) != 0;
}

// ---------------------------------------------------------------------------------------------------------------
// A3: water buoyancy sweep, PhysicsWorld.cpp:1367-1442 (Body::GetSubmergedVolume + Body::ApplyBuoyancyImpulse)

SGP_DEV void box_submerged(v3 h, m33 R, float posz, float wz, float* vol_out, v3* centroid_out)
{
	const v3 n = m33_tmul(R, V3(0.0f, 0.0f, 1.0f));
	const float dpl = wz - posz;
	float vol = 0.0f; v3 cen = V3(0.0f, 0.0f, 0.0f);
	v3 cap[24]; int ncap = 0;
	for (int ax = 0; ax < 3; ++ax) for (int sg = -1; sg <= 1; sg += 2) {
		const int u = (ax + 1) % 3, v = (ax + 2) % 3;
		v3 q[4];
		const float su[4] = { 1, -1, -1, 1 }, sv[4] = { 1, 1, -1, -1 };
		for (int k = 0; k < 4; ++k) {
			v3 p = V3(0.0f, 0.0f, 0.0f);
			v3_set(p, ax, (float)sg * v3_get(h, ax));
			const int kk = sg > 0 ? k : 3 - k;
			v3_set(p, u, su[kk] * v3_get(h, u)); v3_set(p, v, sv[kk] * v3_get(h, v));
			q[k] = p;
		}
		v3 poly[8]; int np = 0;
		for (int k = 0; k < 4; ++k) {
			const v3 a = q[k], c = q[(k + 1) % 4];
			const float da = v3_dot(n, a) - dpl, dc = v3_dot(n, c) - dpl;
			if (da <= 0.0f) poly[np++] = a;
			if ((da <= 0.0f) != (dc <= 0.0f)) {
				const float t = da / (da - dc);
				const v3 x = v3_add(a, v3_scale(v3_sub(c, a), t));
				poly[np++] = x;
				if (ncap < 24) cap[ncap++] = x;
			}
		}
		for (int k = 1; k + 1 < np; ++k) {
			const float tv = v3_dot(poly[0], v3_cross(poly[k], poly[k + 1])) / 6.0f;
			vol += tv;
			cen = v3_add(cen, v3_scale(v3_add(v3_add(poly[0], poly[k]), poly[k + 1]), tv * 0.25f));
		}
	}
	if (ncap >= 3) {
		v3 mean = V3(0.0f, 0.0f, 0.0f);
		for (int k = 0; k < ncap; ++k) mean = v3_add(mean, cap[k]);
		mean = v3_scale(mean, 1.0f / (float)ncap);
		const v3 e1 = v3_normalized_perpendicular(n), e2 = v3_cross(n, e1);
		float ang[24];
		for (int k = 0; k < ncap; ++k) {
			const v3 r = v3_sub(cap[k], mean);
			const float dx = v3_dot(r, e1), dy = v3_dot(r, e2);
			const float den = fabsf(dx) + fabsf(dy);
			const float pa = den > 0.0f ? 1.0f - dx / den : 0.0f;
			ang[k] = dy < 0.0f ? -pa : pa;
		}
		for (int i = 1; i < ncap; ++i) { const float a = ang[i]; const v3 p = cap[i]; int j = i - 1; while (j >= 0 && ang[j] > a) { ang[j + 1] = ang[j]; cap[j + 1] = cap[j]; --j; } ang[j + 1] = a; cap[j + 1] = p; }
		for (int k = 0; k < ncap; ++k) {
			const v3 a = cap[k], c = cap[(k + 1) % ncap];
			const float tv = v3_dot(mean, v3_cross(a, c)) / 6.0f;
			vol += tv;
			cen = v3_add(cen, v3_scale(v3_add(v3_add(mean, a), c), tv * 0.25f));
		}
	}
	*vol_out = vol;
	*centroid_out = vol > 1.0e-12f ? m33_mul(R, v3_scale(cen, 1.0f / vol)) : V3(0.0f, 0.0f, 0.0f);
}

// ConvexHullShape::GetSubmergedVolume: the exact part of the polyhedron under the plane -- every face polygon clipped to the half space and fanned
// into tetrahedra whose apex lies in the plane, so that the cut surface contributes nothing (hull frame = body frame, origin = centre of mass)
SGP_DEV void hull_submerged(const sgd_hull* hl, m33 R, float posz, float wz, float* vol_out, v3* centroid_out)
{
	const v3 n = m33_tmul(R, V3(0.0f, 0.0f, 1.0f));
	const float dpl = wz - posz;
	float lo = 3.4e38f, hi = -3.4e38f;
	for (int i = 0; i < hl->nv; ++i) { const float t = v3_dot(n, hl->verts[i]); lo = fminf(lo, t); hi = fmaxf(hi, t); }
	if (lo >= dpl) { *vol_out = 0.0f; *centroid_out = V3(0.0f, 0.0f, 0.0f); return; }
	if (hi <= dpl) { *vol_out = hl->volume; *centroid_out = V3(0.0f, 0.0f, 0.0f); return; }
	const v3 apex = v3_scale(n, dpl);
	float vol = 0.0f; v3 cen = V3(0.0f, 0.0f, 0.0f);
	for (int f = 0; f < hl->nf; ++f) {
		const int b0 = hl->face_start[f], cnt = hl->face_start[f + 1] - b0;
		// (fan from the first kept point: no polygon buffer, the face's points stream by)
		v3 p0 = V3(0.0f, 0.0f, 0.0f), prev = p0; int np = 0;
		for (int k = 0; k < cnt; ++k) {
			const v3 a = hl->verts[hl->face_idx[b0 + k]], c = hl->verts[hl->face_idx[b0 + (k + 1 == cnt ? 0 : k + 1)]];
			const float da = v3_dot(n, a) - dpl, dc = v3_dot(n, c) - dpl;
			for (int which = 0; which < 2; ++which) {
				v3 q;
				if (which == 0) { if (!(da <= 0.0f)) continue; q = v3_sub(a, apex); }
				else { if ((da <= 0.0f) == (dc <= 0.0f)) continue; const float t = da / (da - dc); q = v3_sub(v3_add(a, v3_scale(v3_sub(c, a), t)), apex); }
				if (np == 0) p0 = q;
				else if (np >= 2) {
					const float tv = v3_dot(p0, v3_cross(prev, q)) / 6.0f;
					vol += tv;
					cen = v3_add(cen, v3_scale(v3_add(v3_add(p0, prev), q), tv * 0.25f));
				}
				prev = q; ++np;
			}
		}
	}
	*vol_out = vol;
	*centroid_out = vol > 1.0e-12f ? m33_mul(R, v3_add(apex, v3_scale(cen, 1.0f / vol))) : V3(0.0f, 0.0f, 0.0f);
}

__global__ void __launch_bounds__(TPB) k_buoyancy(DV d)
{
	const float dt = d.sp->dt;
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= d.sp->n_slots) return;
	uint32_t f = d.flags[i];
	if (!f_movable(f)) return;                                                       // :1377
	const float4 mn = d.aabb_min[i], mx = d.aabb_max[i];
	if (mn.z < d.sp->water_z) {                                                          // :1379
		const float fluid_density = 1020.0f;                                         // :1381
		const uint32_t type = f_shape(f);
		const float4 sh = d.prop[2 * (size_t)i + 1];
		const float4 pim = d.pose[2 * (size_t)i];
		const v3 pos = V3(pim);
		const m33 R = quat_to_m33(Q4(d.pose[2 * (size_t)i + 1]));
		// Shape::GetSubmergedVolume as Jolt's shapes implement it: box and hull exactly, sphere by the cap formula, the capsule through
		// ConvexShape's stand-in -- its local bounding box (total = the box's volume, submerged = the box's part under the plane)
		const float real_volume = shape_volume(d, type, sh);
		float total = real_volume;
		float sub; v3 rc;
		if (type == SGP_SHAPE_BOX) box_submerged(V3(sh.x, sh.y, sh.z), R, pos.z, d.sp->water_z, &sub, &rc);
		else if (type == SGP_SHAPE_HULL) hull_submerged(body_hull(d, sh), R, pos.z, d.sp->water_z, &sub, &rc);
		else if (type == SGP_SHAPE_CAPSULE) {
			const v3 hb = V3(sh.x, sh.x, sh.y + sh.x);
			total = 8.0f * hb.x * hb.y * hb.z;
			box_submerged(hb, R, pos.z, d.sp->water_z, &sub, &rc);
		}
		else if (type == SGP_SHAPE_SPHERE) {
			const float r = sh.x;
			const float h = clampf((d.sp->water_z - pos.z) + r, 0.0f, 2.0f * r);
			const float pi = 3.14159265358979323846f;
			sub = pi * h * h * (3.0f * r - h) / 3.0f;
			float cz = 0.0f;
			if (h > 0.0f) { const float k = 2.0f * r - h; cz = -(3.0f * k * k) / (4.0f * (3.0f * r - h)); }
			rc = V3(0.0f, 0.0f, cz);
		} else {
			const float fr = clampf((d.sp->water_z - mn.z) / (mx.z - mn.z), 0.0f, 1.0f);
			sub = total * fr;
			rc = V3(0.0f, 0.0f, (mn.z + 0.5f * fr * (mx.z - mn.z)) - pos.z);
		}
		const float mass = d.torque[i].w;
		const float buoyancy = fluid_density * real_volume / mass;                   // :1387 (Shape::GetVolume)
		bool applied = false;
		if (sub > 0.0f) {
			const float inv_mass = pim.w;
			const float rho = buoyancy / (total * inv_mass);
			const v3 g = V3(0.0f, 0.0f, -9.81f);                                      // :1407
			const float gf = d.dyn[i].z;
			const v3 buoy_imp = v3_scale(g, -rho * sub * gf * dt);
			float4 lv4 = d.vel[2 * (size_t)i], av4 = d.vel[2 * (size_t)i + 1];
			const v3 lv = V3(lv4), av = V3(av4);
			const v3 cob_vel = v3_add(lv, v3_cross(av, rc));
			const v3 rel = v3_neg(cob_vel);
			const float lin_drag = (f & BF_ZERO_LIN_DRAG) ? 0.0f : 0.1f;             // :1404
			const v3 size = v3_scale(shape_local_half(d, type, sh), 2.0f);
			const v3 lrel = m33_tmul(R, rel);
			const float rl2 = v3_len_sq(lrel);
			v3 drag_imp = V3(0.0f, 0.0f, 0.0f);
			if (rl2 > 1.0e-12f) {
				const float rl = sqrtf(rl2);
				const v3 dirl = v3_scale(v3_abs(lrel), 1.0f / rl);
				const float area = (sub / total) * (dirl.x * size.y * size.z + dirl.y * size.x * size.z + dirl.z * size.x * size.y);
				float dv = 0.5f * rho * rl2 * lin_drag * area * dt * inv_mass;
				if (dv > rl) dv = rl;
				drag_imp = v3_scale(rel, dv / (rl * inv_mass));
			}
			const v3 dlin = v3_scale(v3_add(drag_imp, buoy_imp), inv_mass);
			const float l = (size.x + size.y + size.z) / 3.0f;
			const float ang_drag = 3.0f;                                             // :1405
			const v3 drag_ang_imp = v3_scale(av, -ang_drag * sub / total * dt * (l * l) / inv_mass);
			const sym33 Iw = world_inv_inertia(R, V3(d.prop[2 * (size_t)i]));
			v3 ddrag = sym33_mul(Iw, drag_ang_imp);
			if (v3_len_sq(ddrag) > v3_len_sq(av)) ddrag = v3_neg(av);
			const v3 dang = v3_add(ddrag, sym33_mul(Iw, v3_cross(rc, v3_add(buoy_imp, drag_imp))));
			d.vel[2 * (size_t)i] = F4(v3_add(lv, dlin), lv4.w);
			d.vel[2 * (size_t)i + 1] = F4(v3_add(av, dang), av4.w);
			applied = true;
		}
		if (applied) {
			if (!(f & BF_UNDERWATER)) { push_event(d.ev_water, &d.evc->n_water, d.cap_bodies, i); f |= BF_UNDERWATER; d.flags[i] = f; }
			d.submerged[i] = sub;
		} else { if (f & BF_UNDERWATER) d.flags[i] = f & ~BF_UNDERWATER; d.submerged[i] = 0.0f; }
	} else if (f & BF_UNDERWATER) { d.flags[i] = f & ~BF_UNDERWATER; d.submerged[i] = 0.0f; }
}

// ---------------------------------------------------------------------------------------------------------------
// contact cache (pair key -> constraint slot) for the next step's warm start; contact events

// The table is allocated for the world's manifold capacity, but a step uses (and clears) only the power of two that holds four times its
// constraints: 8 MB instead of 33 MB per step at config 3, and a table that stays in the caches between the rebuild and the next step's probes.
SGP_DEV uint32_t cache_table_size(const DV& d)
{
	const uint32_t want = 4u * max(d.ctr->n_constraints, 256u);
	uint32_t size = 1024u;
	while (size < want && size < d.ht_size) size <<= 1;
	return min(size, d.ht_size);
}
__global__ void __launch_bounds__(TPB) k_cache_clear(DV d)
{
	const uint32_t size = cache_table_size(d);
	for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < size; i += gridDim.x * TPB) d.ht_keys[i] = ~0ull;
	if (blockIdx.x == 0 && threadIdx.x == 0) *d.ht_cur = size;
}

SGP_DEV void step_end_block(const DV& d, StepCounters* host_mapped, EventCounters* host_events)
{
	const uint32_t* src = (const uint32_t*)d.ctr;
	uint32_t* dst = (uint32_t*)host_mapped;
	for (uint32_t i = threadIdx.x; i < sizeof(StepCounters) / 4; i += TPB) dst[i] = src[i];
	__syncthreads();
	if (threadIdx.x == 0 && d.ts_nt) { host_mapped->ts_error = d.ts_flags[0]; host_mapped->ts_all_adjacent = d.ts_flags[1]; }
	if (threadIdx.x < sizeof(EventCounters) / 4) ((uint32_t*)host_events)[threadIdx.x] = ((const uint32_t*)d.evc)[threadIdx.x];
}

// (round 4: the step's counters go to the host from workgroup 0 of this, the step's last, launch: k_step_end was a launch of its own)
__global__ void __launch_bounds__(TPB) k_cache_build(DV d, StepCounters* host_mapped, EventCounters* host_events)
{
	const uint32_t n_con = d.ctr->n_constraints;
	const uint32_t size = *d.ht_cur;
	const uint32_t mask = size - 1;
	for (uint32_t k = blockIdx.x * TPB + threadIdx.x; k < n_con; k += gridDim.x * TPB) {
		const uint64_t key = CUR(d).key[k];
		uint32_t h = ht_hash(key, mask);
		for (uint32_t probe = 0; probe < size; ++probe) {
			const unsigned long long old = atomicCAS((unsigned long long*)&d.ht_keys[h], ~0ull, (unsigned long long)key);
			if (old == ~0ull || old == key) { d.ht_vals[h] = k; break; }
			h = (h + 1) & mask;
		}
	}
	// the step's counters are final before this launch starts and nothing here touches them: workgroup 0 sends them to the host, no waiting for the others
	// (measured: a ticket + fence per workgroup after the hash-table inserts cost 42 us -- the fence writes back every dirty line of the XCD's L2)
	if (host_mapped && blockIdx.x == 0) step_end_block(d, host_mapped, host_events);
}

__global__ void __launch_bounds__(TPB) k_contact_events(DV d)
{
	const uint32_t n = min(d.ctr->n_manifolds, d.cap_manifolds);
	for (uint32_t m = blockIdx.x * TPB + threadIdx.x; m < n; m += gridDim.x * TPB) {
		const uint2 ab = d.man_ab[m];
		const bool persisted = (d.man_prev[m] & ~MAN_PREV_REUSED) != MAN_PREV_NONE;
		// one atomic per wave and list (the lanes here are the loop's active lanes; wave_alloc serves those that call it together)
		uint32_t k;
		if (persisted) k = wave_alloc(&d.evc->n_contact_persisted); else k = wave_alloc(&d.evc->n_contact_added);
		if (k >= d.cap_contact_events) continue;
		sgp_contact_event e;
		e.id1 = ab.x; e.id2 = ab.y; e.userdata1 = 0; e.userdata2 = 0;
		const float4 la = d.vel[2 * (size_t)ab.x], lb = d.vel[2 * (size_t)ab.y];      // velocities after gravity, before the solve (k_pre_solve)
		e.lin_vel1[0] = la.x; e.lin_vel1[1] = la.y; e.lin_vel1[2] = la.z;
		e.lin_vel2[0] = lb.x; e.lin_vel2[1] = lb.y; e.lin_vel2[2] = lb.z;
		const float4 n4 = d.man_n[m];
		const int np = __float_as_int(n4.w) & 0xFF;
		const v3 nrm = V3(n4);
		const v3 base = V3(d.man_p1[0][m]);
		e.base_offset[0] = base.x; e.base_offset[1] = base.y; e.base_offset[2] = base.z;
		e.normal[0] = nrm.x; e.normal[1] = nrm.y; e.normal[2] = nrm.z;
		e.num_points = (uint32_t)np;
		float pen = -3.4e38f;
		for (int i = 0; i < 4; ++i) {
			v3 r = V3(0.0f, 0.0f, 0.0f);
			if (i < np) {
				const v3 p1 = V3(d.man_p1[i][m]), p2 = V3(d.man_p2[i][m]);
				r = v3_sub(p1, base);
				pen = fmaxf(pen, v3_dot(v3_sub(p1, p2), nrm));
			}
			e.rel_points_on1[i][0] = r.x; e.rel_points_on1[i][1] = r.y; e.rel_points_on1[i][2] = r.z;
		}
		e.penetration = pen;
		(persisted ? d.ev_contacts_persisted : d.ev_contacts_added)[k] = e;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// host edits: one thread per body, its commands applied in submission order

SGP_DEV void refresh_aabb(const DV& d, uint32_t i, uint32_t f)
{
	v3 mn, mx;
	compute_aabb(d, f_shape(f), d.prop[2 * (size_t)i + 1], V3(d.pose[2 * (size_t)i]), Q4(d.pose[2 * (size_t)i + 1]), mn, mx);
	d.aabb_min[i] = F4(mn, 0.0f); d.aabb_max[i] = F4(mx, 0.0f);
}

SGP_DEV uint32_t activate_body(const DV& d, uint32_t i, uint32_t f)
{
	if (!(f & BF_ALIVE) || f_motion(f) == SGP_MOTION_STATIC) return f;
	if (!(f & BF_ACTIVE)) { f |= BF_ACTIVE; push_event(d.ev_activated, &d.evc->n_activated, d.cap_bodies, i); }
	reset_sleep(d, i, f_shape(f), d.prop[2 * (size_t)i + 1], V3(d.pose[2 * (size_t)i]), Q4(d.pose[2 * (size_t)i + 1]));
	return f;
}

// The ghosts of a tile are refreshed every step with the poses their owners exported: same effect, in the same order, as the
// SET_POS | SET_ROT | SET_VEL | ACTIVATE command of k_apply_cmds, without the 136-byte command record and the run detection.
__global__ void __launch_bounds__(TPB) k_ghost_refresh(DV d, const GhostRefresh* recs, uint32_t n)
{
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= n) return;
	const GhostRefresh c = recs[k];
	const uint32_t i = c.id;
	uint32_t f = d.flags[i];
	if (!(f & BF_ALIVE)) return;
	d.pose[2 * (size_t)i] = make_float4(c.pos[0], c.pos[1], c.pos[2], d.pose[2 * (size_t)i].w);
	d.pose[2 * (size_t)i + 1] = make_float4(c.rot[0], c.rot[1], c.rot[2], c.rot[3]);
	if (f_motion(f) != SGP_MOTION_STATIC) {
		d.vel[2 * (size_t)i] = make_float4(c.linv[0], c.linv[1], c.linv[2], d.vel[2 * (size_t)i].w);
		d.vel[2 * (size_t)i + 1] = make_float4(c.angv[0], c.angv[1], c.angv[2], d.vel[2 * (size_t)i + 1].w);
	}
	refresh_aabb(d, i, f);
	f = activate_body(d, i, f);
	d.flags[i] = f;
}

__global__ void __launch_bounds__(TPB) k_apply_cmds(DV d, const BodyCmd* cmds, const uint32_t* run_start, uint32_t n_runs)
{
	const uint32_t r = blockIdx.x * TPB + threadIdx.x;
	if (r >= n_runs) return;
	const uint32_t b = run_start[r], e = run_start[r + 1];
	const uint32_t i = cmds[b].id;
	uint32_t f = d.flags[i];
	for (uint32_t k = b; k < e; ++k) {
		const BodyCmd& c = cmds[k];
		if (c.ops & CMD_CREATE) {
			f = c.flags | BF_CACHE_INVALID;
			d.pose[2 * (size_t)i] = make_float4(c.pos[0], c.pos[1], c.pos[2], c.inv_mass);
			d.pose[2 * (size_t)i + 1] = make_float4(c.rot[0], c.rot[1], c.rot[2], c.rot[3]);
			d.vel[2 * (size_t)i] = make_float4(c.linv[0], c.linv[1], c.linv[2], 0.0f);        // (effective inverse mass: set by k_pre_solve once the body is awake)
			d.vel[2 * (size_t)i + 1] = make_float4(c.angv[0], c.angv[1], c.angv[2], 0.0f);
			d.dyn[i] = make_float4(c.lin_damp, c.ang_damp, c.gravity_factor, c.inv_mass);
			d.force[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			d.torque[i] = make_float4(0.0f, 0.0f, 0.0f, c.mass);
			d.prop[2 * (size_t)i] = make_float4(c.inv_inertia[0], c.inv_inertia[1], c.inv_inertia[2], c.restitution);
			d.prop[2 * (size_t)i + 1] = make_float4(c.shape[0], c.shape[1], c.shape[2], c.friction);
			d.submerged[i] = 0.0f;
			d.userdata[i] = c.userdata;
			d.sleep_label[i] = i;
			refresh_aabb(d, i, f);
			reset_sleep(d, i, f_shape(f), d.prop[2 * (size_t)i + 1], V3(d.pose[2 * (size_t)i]), Q4(d.pose[2 * (size_t)i + 1]));
			continue;
		}
		if (c.ops & CMD_REMOVE) { f = 0; continue; }
		if (!(f & BF_ALIVE)) continue;
		if (c.ops & CMD_SET_CHASSIS) { f = (f & ~BF_CHASSIS) | (c.flags & BF_CHASSIS); continue; }
		if (c.ops & CMD_SET_LAYER) f = (f & ~BF_LAYER_MASK) | ((c.flags & 0x3u) << BF_LAYER_SHIFT);
		if (c.ops & CMD_MOVE_KINEMATIC) {
			// MotionProperties::MoveKinematic: velocities that reach the target in dt
			if (f_motion(f) == SGP_MOTION_KINEMATIC && c.dt > 0.0f) {
				const v3 pos = V3(d.pose[2 * (size_t)i]);
				const quat q = Q4(d.pose[2 * (size_t)i + 1]);
				const v3 lv = v3_scale(v3_sub(V3(c.pos[0], c.pos[1], c.pos[2]), pos), 1.0f / c.dt);
				quat t; t.x = c.rot[0]; t.y = c.rot[1]; t.z = c.rot[2]; t.w = c.rot[3];
				quat cj; cj.x = -q.x; cj.y = -q.y; cj.z = -q.z; cj.w = q.w;
				quat dq = quat_mul(t, cj);
				if (dq.w < 0.0f) { dq.x = -dq.x; dq.y = -dq.y; dq.z = -dq.z; dq.w = -dq.w; }
				const float sl = sqrtf(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z);
				v3 av = V3(0.0f, 0.0f, 0.0f);
				if (sl > 1.0e-12f) { const float angle = sgd_quat_angle(sl, dq.w); av = v3_scale(V3(dq.x / sl, dq.y / sl, dq.z / sl), angle / c.dt); }
				d.vel[2 * (size_t)i] = F4(lv, d.vel[2 * (size_t)i].w);
				d.vel[2 * (size_t)i + 1] = F4(av, d.vel[2 * (size_t)i + 1].w);
				if (!(f & BF_ALIAS)) f = activate_body(d, i, f);      // (a mesh body's alias slots follow its pose and velocities, they are never awake themselves)
			}
			continue;
		}
		bool pose = false;
		if (c.ops & CMD_SET_POS) { d.pose[2 * (size_t)i] = make_float4(c.pos[0], c.pos[1], c.pos[2], d.pose[2 * (size_t)i].w); pose = true; }
		if (c.ops & CMD_SET_ROT) { d.pose[2 * (size_t)i + 1] = make_float4(c.rot[0], c.rot[1], c.rot[2], c.rot[3]); pose = true; }
		if (c.ops & CMD_SET_SHAPE) {
			d.prop[2 * (size_t)i + 1] = make_float4(c.shape[0], c.shape[1], c.shape[2], d.prop[2 * (size_t)i + 1].w); pose = true;
			f = ((f & ~BF_LARGE) | (c.flags & BF_LARGE)) | BF_CACHE_INVALID;      // a new scale can move the body across the broad phase's large-body radius (host: note_radius)
		}
		if ((c.ops & CMD_SET_VEL) && f_motion(f) != SGP_MOTION_STATIC) {
			d.vel[2 * (size_t)i] = make_float4(c.linv[0], c.linv[1], c.linv[2], d.vel[2 * (size_t)i].w);
			d.vel[2 * (size_t)i + 1] = make_float4(c.angv[0], c.angv[1], c.angv[2], d.vel[2 * (size_t)i + 1].w);
		}
		if (pose) refresh_aabb(d, i, f);
		if (f_motion(f) == SGP_MOTION_DYNAMIC) {
			if (c.ops & CMD_ADD_FORCE) {
				const float4 F = d.force[i];
				d.force[i] = F4(v3_add(V3(F), V3(c.linv[0], c.linv[1], c.linv[2])), F.w);
				f = activate_body(d, i, f) | BF_HAS_FORCE;
			}
			if (c.ops & CMD_ADD_TORQUE) {
				const float4 T = d.torque[i];
				d.torque[i] = F4(v3_add(V3(T), V3(c.angv[0], c.angv[1], c.angv[2])), T.w);
				f = activate_body(d, i, f) | BF_HAS_FORCE;
			}
			if (c.ops & CMD_ADD_FORCE_AT) {
				const v3 Fv = V3(c.linv[0], c.linv[1], c.linv[2]);
				const float4 F = d.force[i], T = d.torque[i];
				d.force[i] = F4(v3_add(V3(F), Fv), F.w);
				d.torque[i] = F4(v3_add(V3(T), v3_cross(v3_sub(V3(c.pos[0], c.pos[1], c.pos[2]), V3(d.pose[2 * (size_t)i])), Fv)), T.w);
				f = activate_body(d, i, f) | BF_HAS_FORCE;
			}
		}
		if (c.ops & CMD_ACTIVATE) f = activate_body(d, i, f);
	}
	d.flags[i] = f;
}

// ---------------------------------------------------------------------------------------------------------------
// read-back

SGP_DEV void fill_state(const DV& d, uint32_t i, sgp_body_state* s)
{
	const float4 p = d.pose[2 * (size_t)i], q = d.pose[2 * (size_t)i + 1], lv = d.vel[2 * (size_t)i], av = d.vel[2 * (size_t)i + 1];
	const uint32_t f = d.flags[i];
	s->pos[0] = p.x; s->pos[1] = p.y; s->pos[2] = p.z;
	s->rot[0] = q.x; s->rot[1] = q.y; s->rot[2] = q.z; s->rot[3] = q.w;
	s->lin_vel[0] = lv.x; s->lin_vel[1] = lv.y; s->lin_vel[2] = lv.z;
	s->ang_vel[0] = av.x; s->ang_vel[1] = av.y; s->ang_vel[2] = av.z;
	s->active = (f & BF_ACTIVE) ? 1u : 0u;
	s->underwater = (f & BF_UNDERWATER) ? 1u : 0u;
	s->submerged_volume = d.submerged[i];
	s->id = (f & BF_ALIVE) ? i : SGP_INVALID_ID;
}

__global__ void __launch_bounds__(TPB) k_gather_states(DV d, const uint32_t* ids, uint32_t first, uint32_t n, sgp_body_state* out)
{
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= n) return;
	const uint32_t i = ids ? ids[k] : first + k;
	if (i < d.cap_bodies) fill_state(d, i, &out[k]);
}

__global__ void __launch_bounds__(TPB) k_gather_active(DV d, sgp_body_state* out, uint32_t cap)
{
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	const uint32_t f = i < d.sp->n_slots ? d.flags[i] : 0u;
	const bool want = (f & (BF_ALIVE | BF_ACTIVE)) == (BF_ALIVE | BF_ACTIVE);
	const uint32_t k = block_alloc(&d.ctr->n_read_active, want);      // one atomic per workgroup
	if (want && k < cap) fill_state(d, i, &out[k]);
}

// poses only (two float4 per body: position + id, rotation): what the caller's per-frame loop reads
__global__ void __launch_bounds__(TPB) k_gather_active_poses(DV d, float4* out, uint32_t cap)
{
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	const uint32_t f = i < d.sp->n_slots ? d.flags[i] : 0u;
	const bool want = (f & (BF_ALIVE | BF_ACTIVE)) == (BF_ALIVE | BF_ACTIVE);
	const uint32_t k = block_alloc(&d.ctr->n_read_active, want);      // one atomic per workgroup
	if (want && k < cap) {
		const float4 p = d.pose[2 * (size_t)i];
		out[2 * (size_t)k] = make_float4(p.x, p.y, p.z, __uint_as_float(i));
		out[2 * (size_t)k + 1] = d.pose[2 * (size_t)i + 1];
	}
}

struct ConstraintDumpRec { uint32_t a, b; int32_t colour; int32_t np; float n[3]; float lam_n[4]; float lam_t1[4]; float lam_t2[4]; float bias[4]; };

__global__ void __launch_bounds__(TPB) k_dump_constraints(DV d, uint32_t which, uint32_t n_con, ConstraintDumpRec* out, uint32_t cap)
{
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= n_con || k >= cap) return;
	const ConstraintArrays& ca = d.ca[which & 1];
	ConstraintDumpRec r;
	const uint2 ab = ca.ab[k];
	const int nc = ca.np_col[k];
	const float4 nf = ca.n_fric[k];
	r.a = ab.x; r.b = ab.y; r.colour = (nc >> 8) & 0xFF; r.np = nc & 0xFF;
	r.n[0] = nf.x; r.n[1] = nf.y; r.n[2] = nf.z;
	for (int i = 0; i < 4; ++i) {
		if (i < r.np) { const float4 l = ca.lam[i][k]; r.lam_n[i] = l.x; r.lam_t1[i] = l.y; r.lam_t2[i] = l.z; r.bias[i] = ca.r1b[i][k].w; }
		else { r.lam_n[i] = 0; r.lam_t1[i] = 0; r.lam_t2[i] = 0; r.bias[i] = 0; }
	}
	out[k] = r;
}

// ---------------------------------------------------------------------------------------------------------------
// ray queries (traceRay, PhysicsWorld.cpp:1668-1725), one thread per ray, brute force over bodies with an AABB slab test

struct RaySub { uint32_t tri, mat; float u, v; };      // which triangle of a mesh a ray hit, its user data, barycentrics

SGP_DEV float ray_body(const DV& d, uint32_t type, float4 sh, v3 pos, quat q, v3 o, v3 dir, float max_t, v3* n_out, RaySub* sub)
{
	const m33 R = quat_to_m33(q);
	const v3 ol = m33_tmul(R, v3_sub(o, pos)), dl = m33_tmul(R, dir);
	sub->tri = SGP_INVALID_ID; sub->mat = 0; sub->u = 0.0f; sub->v = 0.0f;
	if (type == SGP_SHAPE_MESH) {
		// closest front-facing triangle; on equal distance the lower triangle index (caller's order) wins
		const MeshHeader mh = d.meshes[(uint32_t)sh.x];
		float best = max_t; uint32_t best_idx = 0xFFFFFFFFu; v3 bn = V3(0.0f, 0.0f, 0.0f);
		const v3 inv = V3(fabsf(dl.x) > 1.0e-12f ? 1.0f / dl.x : 3.0e38f, fabsf(dl.y) > 1.0e-12f ? 1.0f / dl.y : 3.0e38f, fabsf(dl.z) > 1.0e-12f ? 1.0f / dl.z : 3.0e38f);
		uint32_t stack[48]; int sp = 0;
		stack[sp++] = 0;
		while (sp > 0) {
			const MeshNode nd = d.mesh_nodes[mh.node_off + stack[--sp]];
			// slab test against the node box grown a little (never rejects a triangle the exact test would accept)
			const float g = 1.0e-4f * (1.0f + fabsf(nd.mxx) + fabsf(nd.mxy) + fabsf(nd.mxz) + fabsf(nd.mnx) + fabsf(nd.mny) + fabsf(nd.mnz));
			float t0 = 0.0f, t1 = best; bool miss = false;
			const float lo3[3] = { nd.mnx - g, nd.mny - g, nd.mnz - g }, hi3[3] = { nd.mxx + g, nd.mxy + g, nd.mxz + g };
			const float o3[3] = { ol.x, ol.y, ol.z }, d3[3] = { dl.x, dl.y, dl.z }, i3[3] = { inv.x, inv.y, inv.z };
			for (int a = 0; a < 3 && !miss; ++a) {
				if (fabsf(d3[a]) <= 1.0e-12f) { if (o3[a] < lo3[a] || o3[a] > hi3[a]) miss = true; }
				else { float ta = (lo3[a] - o3[a]) * i3[a], tb = (hi3[a] - o3[a]) * i3[a]; if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; } t0 = fmaxf(t0, ta - g); t1 = fminf(t1, tb + g); if (t0 > t1) miss = true; }
			}
			if (miss) continue;
			if (nd.count == 0) { if (sp + 2 <= 48) { stack[sp++] = nd.left; stack[sp++] = nd.right; } continue; }
			for (uint32_t k = 0; k < nd.count; ++k) {
				const uint4 tri = d.mesh_tris[mh.tri_off + nd.left + k];
				const v3 pa = V3(d.mesh_verts[mh.vert_off + tri.x]), pb = V3(d.mesh_verts[mh.vert_off + tri.y]), pc = V3(d.mesh_verts[mh.vert_off + tri.z]);
				float uv[2];
				const float tt = sgd_ray_tri_uv(ol, dl, pa, pb, pc, best, uv);
				if (tt >= 0.0f && (tt < best || best_idx == 0xFFFFFFFFu || (tt == best && MESH_TRI_INDEX(tri.w) < best_idx))) {
					best = tt; best_idx = MESH_TRI_INDEX(tri.w);
					const v3 nn = v3_cross(v3_sub(pb, pa), v3_sub(pc, pa)); bn = v3_scale(nn, 1.0f / v3_len(nn));
					sub->tri = MESH_TRI_INDEX(tri.w); sub->mat = d.mesh_tri_mat[mh.tri_off + nd.left + k]; sub->u = uv[0]; sub->v = uv[1];
				}
			}
		}
		if (best_idx == 0xFFFFFFFFu) return -1.0f;
		*n_out = m33_mul(R, bn);
		return best;
	}
	if (type == SGP_SHAPE_HULL) {
		v3 nl;
		const float t = sgd_ray_hull(body_hull(d, sh), ol, dl, max_t, 0.0f, &nl);
		if (t < 0.0f) return -1.0f;
		*n_out = m33_mul(R, nl);
		return t;
	}
	if (type == SGP_SHAPE_SPHERE) {
		const float r = sh.x;
		const float B = v3_dot(ol, dl), C = v3_len_sq(ol) - r * r;
		if (C <= 0.0f) { *n_out = v3_neg(dir); return 0.0f; }
		const float disc = B * B - C;
		if (disc < 0.0f) return -1.0f;
		const float t = -B - sqrtf(disc);
		if (t < 0.0f || t > max_t) return -1.0f;
		*n_out = m33_mul(R, v3_scale(v3_add(ol, v3_scale(dl, t)), 1.0f / r));
		return t;
	}
	if (type == SGP_SHAPE_BOX) {
		const v3 h = V3(sh.x, sh.y, sh.z);
		float t0 = 0.0f, t1 = max_t; int ax = -1; float sg = 0.0f;
		for (int k = 0; k < 3; ++k) {
			const float ok = v3_get(ol, k), dk = v3_get(dl, k), hk = v3_get(h, k);
			if (fabsf(dk) < 1.0e-12f) { if (ok < -hk || ok > hk) return -1.0f; continue; }
			float ta = (-hk - ok) / dk, tb = (hk - ok) / dk; float s = -1.0f;
			if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; s = 1.0f; }
			if (ta > t0) { t0 = ta; ax = k; sg = s; }
			if (tb < t1) t1 = tb;
			if (t0 > t1) return -1.0f;
		}
		if (ax < 0) { *n_out = v3_neg(dir); return 0.0f; }
		v3 nl = V3(0.0f, 0.0f, 0.0f); v3_set(nl, ax, sg);
		*n_out = m33_mul(R, nl);
		return t0;
	}
	{
		const float r = sh.x, hh = sh.y;
		{      // starting inside comes first (else an interior cap-sphere entry can win, depending on max_t; see sgd_ray_capsule_z)
			const v3 qq = sgd_closest_on_segment(V3(0.0f, 0.0f, -hh), V3(0.0f, 0.0f, hh), ol);
			if (v3_len_sq(v3_sub(ol, qq)) <= r * r) { *n_out = v3_neg(dir); return 0.0f; }
		}
		float best = -1.0f; v3 bn = V3(0.0f, 0.0f, 0.0f);
		const float a = dl.x * dl.x + dl.y * dl.y;
		const float bq = ol.x * dl.x + ol.y * dl.y, c = ol.x * ol.x + ol.y * ol.y - r * r;
		if (a > 1.0e-12f) {
			const float disc = bq * bq - a * c;
			if (disc >= 0.0f) {
				const float t = (-bq - sqrtf(disc)) / a;
				const float z = ol.z + dl.z * t;
				if (t >= 0.0f && t <= max_t && fabsf(z) <= hh) { best = t; bn = V3((ol.x + dl.x * t) / r, (ol.y + dl.y * t) / r, 0.0f); }
			}
		}
		for (int sgn = -1; sgn <= 1; sgn += 2) {
			const v3 oc = V3(ol.x, ol.y, ol.z - (float)sgn * hh);
			const float B = v3_dot(oc, dl), C = v3_len_sq(oc) - r * r;
			const float disc = B * B - C;
			if (disc < 0.0f) continue;
			const float t = -B - sqrtf(disc);
			if (t < 0.0f || t > max_t) continue;
			if (best < 0.0f || t < best) { best = t; bn = v3_scale(v3_add(oc, v3_scale(dl, t)), 1.0f / r); }
		}
		if (best < 0.0f) return -1.0f;
		*n_out = m33_mul(R, bn);
		return best;
	}
}

SGP_DEV bool ray_aabb(v3 o, v3 dir, float4 mn, float4 mx, float tmax)
{
	float t0 = 0.0f, t1 = tmax;
	const float oo[3] = { o.x, o.y, o.z }, dd[3] = { dir.x, dir.y, dir.z };
	const float lo[3] = { mn.x, mn.y, mn.z }, hi[3] = { mx.x, mx.y, mx.z };
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		if (fabsf(dd[a]) < 1.0e-12f) { if (oo[a] < lo[a] - 1.0e-4f || oo[a] > hi[a] + 1.0e-4f) return false; }
		else {
			float ta = (lo[a] - 1.0e-4f - oo[a]) / dd[a], tb = (hi[a] + 1.0e-4f - oo[a]) / dd[a];
			if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
			t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
			if (t0 > t1) return false;
		}
	}
	return true;
}

struct RayBest { float t; uint32_t id; v3 n; RaySub sub; };

SGP_DEV void ray_test_body(const DV& d, const sgp_ray& ry, v3 o, v3 dir, uint32_t i, RayBest& best)
{
	if (i == ry.ignore_id) return;
	const uint32_t f = d.flags[i];
	if (!(f & BF_ALIVE) || (f & BF_ALIAS)) return;
	const uint32_t layer = f_layer(f);
	if (ry.collidable_only && !(layer == SGP_LAYER_NON_MOVING || layer == SGP_LAYER_MOVING)) return;
	if (!ray_aabb(o, dir, d.aabb_min[i], d.aabb_max[i], best.t)) return;
	v3 nn; RaySub sub;
	const float t = ray_body(d, f_shape(f), d.prop[2 * (size_t)i + 1], V3(d.pose[2 * (size_t)i]), Q4(d.pose[2 * (size_t)i + 1]), o, dir, best.t, &nn, &sub);
	// closest hit; ties go to the lower body id so the result does not depend on the traversal order
	if (t >= 0.0f && t <= best.t && (t < best.t || best.id == SGP_INVALID_ID || i < best.id)) { best.t = t; best.id = i; best.n = nn; best.sub = sub; }
}

// traceRay (PhysicsWorld.cpp:1668-1725), batched: one thread per ray.  Large bodies (ground quad ...) are tested directly;
// small bodies through a 3D-DDA walk of the broad-phase cell grid (bodies are binned by centre and reach at most one cell
// beyond it, so every visited cell also looks at its 26 neighbours), stopping once the cell entry distance passes the best hit.
__global__ void __launch_bounds__(64) k_raycast(DV d, const sgp_ray* rays, uint32_t n, sgp_hit* hits)
{
	const uint32_t k = blockIdx.x * 64 + threadIdx.x;
	if (k >= n) return;
	const sgp_ray ry = rays[k];
	const v3 o = V3(ry.origin[0], ry.origin[1], ry.origin[2]), dir = V3(ry.dir[0], ry.dir[1], ry.dir[2]);
	RayBest best; best.t = ry.max_t; best.id = SGP_INVALID_ID; best.n = V3(0.0f, 0.0f, 0.0f);
	best.sub.tri = SGP_INVALID_ID; best.sub.mat = 0; best.sub.u = best.sub.v = 0.0f;
	for (uint32_t l = 0; l < d.sp->n_large; ++l) ray_test_body(d, ry, o, dir, d.large_ids[l], best);
	large_grid_ray(d, o, dir, &best.t, [&](uint32_t i) { ray_test_body(d, ry, o, dir, i, best); });
	const BpGrid g = *d.grid;
	if (g.n_cells > 0 && g.min_x <= g.max_x) {
		// clip the ray to the grid box inflated by one cell (bodies reach one cell beyond their centre cell)
		const float c = g.cell;
		const v3 lo = V3(g.ox - c, g.oy - c, g.oz - c);
		const v3 hi = V3(g.ox + ((float)g.nx + 1.0f) * c, g.oy + ((float)g.ny + 1.0f) * c, g.oz + ((float)g.nz + 1.0f) * c);
		float t0 = 0.0f, t1 = best.t; bool miss = false;
		const float oo[3] = { o.x, o.y, o.z }, dd[3] = { dir.x, dir.y, dir.z };
		const float bl[3] = { lo.x, lo.y, lo.z }, bh[3] = { hi.x, hi.y, hi.z };
		for (int a = 0; a < 3 && !miss; ++a) {
			if (fabsf(dd[a]) < 1.0e-12f) { if (oo[a] < bl[a] || oo[a] > bh[a]) miss = true; }
			else {
				float ta = (bl[a] - oo[a]) / dd[a], tb = (bh[a] - oo[a]) / dd[a];
				if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
				t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
				if (t0 > t1) miss = true;
			}
		}
		if (!miss) {
			// DDA over cells (cell coordinates may run one cell outside the grid on each side)
			const v3 p0 = v3_add(o, v3_scale(dir, t0));
			int cx = (int)floorf((p0.x - g.ox) * g.inv_cell), cy = (int)floorf((p0.y - g.oy) * g.inv_cell), cz = (int)floorf((p0.z - g.oz) * g.inv_cell);
			cx = min(max(cx, -1), g.nx); cy = min(max(cy, -1), g.ny); cz = min(max(cz, -1), g.nz);
			const int sx = dir.x > 0.0f ? 1 : -1, sy = dir.y > 0.0f ? 1 : -1, sz = dir.z > 0.0f ? 1 : -1;
			const float inf = 3.0e38f;
			const float tdx = fabsf(dir.x) > 1.0e-12f ? c / fabsf(dir.x) : inf, tdy = fabsf(dir.y) > 1.0e-12f ? c / fabsf(dir.y) : inf, tdz = fabsf(dir.z) > 1.0e-12f ? c / fabsf(dir.z) : inf;
			float tmx = fabsf(dir.x) > 1.0e-12f ? ((g.ox + (float)(cx + (sx > 0 ? 1 : 0)) * c) - o.x) / dir.x : inf;
			float tmy = fabsf(dir.y) > 1.0e-12f ? ((g.oy + (float)(cy + (sy > 0 ? 1 : 0)) * c) - o.y) / dir.y : inf;
			float tmz = fabsf(dir.z) > 1.0e-12f ? ((g.oz + (float)(cz + (sz > 0 ? 1 : 0)) * c) - o.z) / dir.z : inf;
			float t_enter = t0;
			for (int iter = 0; iter < 100000; ++iter) {
				if (t_enter - 2.0f * c > best.t) break;           // nothing nearer can come from cells this far along the ray
				for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) {
					const int y = cy + dy, z = cz + dz;
					if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
					const int xa = max(cx - 1, 0), xb = min(cx + 1, g.nx - 1);
					if (xa > xb) continue;
					grid_row_runs(d, g, xa, xb, y, z, [&](uint32_t q0, uint32_t q1) { for (uint32_t q = q0; q < q1; ++q) ray_test_body(d, ry, o, dir, __float_as_uint(d.sorted_max[q].w), best); });
				}
				// next cell
				if (tmx <= tmy && tmx <= tmz) { t_enter = tmx; tmx += tdx; cx += sx; if (cx < -1 || cx > g.nx) break; }
				else if (tmy <= tmz) { t_enter = tmy; tmy += tdy; cy += sy; if (cy < -1 || cy > g.ny) break; }
				else { t_enter = tmz; tmz += tdz; cz += sz; if (cz < -1 || cz > g.nz) break; }
				if (t_enter > t1) break;
			}
		}
	}
	sgp_hit h;
	h.id = best.id; h.t = best.id == SGP_INVALID_ID ? 0.0f : best.t;
	h.normal[0] = best.n.x; h.normal[1] = best.n.y; h.normal[2] = best.n.z;
	h.triangle = best.sub.tri; h.material = best.sub.mat; h.bary[0] = best.sub.u; h.bary[1] = best.sub.v; h.sub_shape = 0;
	h.userdata = 0;
	hits[k] = h;
}

// ---------------------------------------------------------------------------------------------------------------
// Wheeled vehicles (sgp_device_vehicle.h): one thread per vehicle.  Vehicles never share a chassis and apply no impulse to
// the body under a wheel, so each phase is race free without colouring; it runs as its own launch before the contact colours
// of the same pass (PhysicsSystem solves non-contact constraints first).

// swept sphere against mesh body j: closest front-side touch; on equal distance the lower triangle index (caller's order) wins
SGP_DEV float cast_sphere_mesh(const DV& d, uint32_t j, v3 o, v3 dir, float max_t, float rs, v3* n_out, v3* p_out)
{
	const MeshHeader mh = d.meshes[(uint32_t)d.prop[2 * (size_t)j + 1].x];
	const v3 mpos = V3(d.pose[2 * (size_t)j]); const m33 R = quat_to_m33(Q4(d.pose[2 * (size_t)j + 1]));
	const v3 ol = m33_tmul(R, v3_sub(o, mpos)), dl = m33_tmul(R, dir);
	float best = max_t; uint32_t best_idx = 0xFFFFFFFFu; v3 bn = V3(0.0f, 0.0f, 0.0f);
	uint32_t stack[48]; int sp = 0;
	stack[sp++] = 0;
	while (sp > 0) {
		const MeshNode nd = d.mesh_nodes[mh.node_off + stack[--sp]];
		// slab test of the centre's path against the node box grown by the sphere radius (+ a little)
		const float g = rs + 1.0e-4f * (1.0f + fabsf(nd.mxx) + fabsf(nd.mxy) + fabsf(nd.mxz) + fabsf(nd.mnx) + fabsf(nd.mny) + fabsf(nd.mnz));
		float t0 = 0.0f, t1 = best; bool miss = false;
		const float lo3[3] = { nd.mnx - g, nd.mny - g, nd.mnz - g }, hi3[3] = { nd.mxx + g, nd.mxy + g, nd.mxz + g };
		const float o3[3] = { ol.x, ol.y, ol.z }, d3[3] = { dl.x, dl.y, dl.z };
		for (int a = 0; a < 3 && !miss; ++a) {
			if (fabsf(d3[a]) <= 1.0e-12f) { if (o3[a] < lo3[a] || o3[a] > hi3[a]) miss = true; }
			else { float ta = (lo3[a] - o3[a]) / d3[a], tb = (hi3[a] - o3[a]) / d3[a]; if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; } t0 = fmaxf(t0, ta - 1.0e-4f); t1 = fminf(t1, tb + 1.0e-4f); if (t0 > t1) miss = true; }
		}
		if (miss) continue;
		if (nd.count == 0) { if (sp + 2 <= 48) { stack[sp++] = nd.left; stack[sp++] = nd.right; } continue; }
		for (uint32_t k = 0; k < nd.count; ++k) {
			const uint4 tri = d.mesh_tris[mh.tri_off + nd.left + k];
			const v3 pa = V3(d.mesh_verts[mh.vert_off + tri.x]), pb = V3(d.mesh_verts[mh.vert_off + tri.y]), pc = V3(d.mesh_verts[mh.vert_off + tri.z]);
			v3 nn;
			const float tt = sgd_cast_sphere_tri(ol, dl, pa, pb, pc, best, rs, &nn);
			if (tt >= 0.0f && (tt < best || best_idx == 0xFFFFFFFFu || (tt == best && MESH_TRI_INDEX(tri.w) < best_idx))) { best = tt; best_idx = MESH_TRI_INDEX(tri.w); bn = nn; }
		}
	}
	if (best_idx == 0xFFFFFFFFu) return -1.0f;
	const v3 n = m33_mul(R, bn);
	*n_out = n;
	*p_out = v3_sub(v3_add(o, v3_scale(dir, best)), v3_scale(n, rs));
	return best;
}

SGP_DEV void veh_cast_test(const DV& d, const sgd_vehicle* v, v3 o, v3 dir, float rs, float cast_len, uint32_t j, float& best, uint32_t& bid, v3& bn, v3& bp)
{
	if (j == v->body) return;
	const uint32_t f = d.flags[j];
	if (!(f & BF_ALIVE) || (f & (BF_SENSOR | BF_ALIAS))) return;
	const uint32_t layer = f_layer(f);
	if (!(layer == SGP_LAYER_NON_MOVING || layer == SGP_LAYER_MOVING)) return;            // tester object layer MOVING, CarPhysics.cpp:62
	const float4 mn = d.aabb_min[j], mx = d.aabb_max[j];
	const float e = rs + 1.0e-3f;
	// the bounds filter uses the full cast length, not the best hit so far: the planes-only swept-sphere test of boxes and hulls can report a
	// touch just outside the inflated bounds (it is generous at corners), and the answer must not depend on the order of the candidates
	if (!ray_aabb(o, dir, make_float4(mn.x - e, mn.y - e, mn.z - e, 0.0f), make_float4(mx.x + e, mx.y + e, mx.z + e, 0.0f), cast_len)) return;
	const float4 sh = d.prop[2 * (size_t)j + 1];
	const float prm[3] = { sh.x, sh.y, sh.z };
	v3 n, p;
	const float t = f_shape(f) == SGP_SHAPE_MESH ? cast_sphere_mesh(d, j, o, dir, best, rs, &n, &p)
	              : sgd_cast_sphere_body((int)f_shape(f), prm, f_shape(f) == SGP_SHAPE_HULL ? body_hull(d, sh) : nullptr, V3(d.pose[2 * (size_t)j]), quat_to_m33(Q4(d.pose[2 * (size_t)j + 1])), o, dir, best, rs, &n, &p);
	if (t < 0.0f || n.z < v->cos_max_slope) return;
	// closest accepted hit; on equal distance the lower body id wins (the oracle visits ids in ascending order)
	if (t < best || bid == SGP_INVALID_ID || (t == best && j < bid)) { best = t; bid = j; bn = n; bp = p; }
}

SGP_DEV sgd_chassis veh_chassis_pose_vel(const DV& d, uint32_t b)
{
	sgd_chassis c;
	const float4 p = d.pose[2 * (size_t)b];
	c.pos = V3(p); c.rot = Q4(d.pose[2 * (size_t)b + 1]); c.v = V3(d.vel[2 * (size_t)b]); c.w = V3(d.vel[2 * (size_t)b + 1]);
	c.im = p.w; c.inv_inertia_local = V3(d.prop[2 * (size_t)b]);
	c.I = world_inv_inertia(quat_to_m33(c.rot), c.inv_inertia_local);
	return c;
}

// One wave owns one vehicle: the 64 lanes stage the vehicle record (~2.3 KB) between HBM/L2 and LDS in a few coalesced
// bursts, and the sequential per-vehicle arithmetic then runs out of LDS instead of paying a global round trip per field.
SGP_DEV void veh_stage_in(sgd_vehicle* sv, const sgd_vehicle* gv)
{
	const uint32_t* src = (const uint32_t*)gv; uint32_t* dst = (uint32_t*)sv;
	for (uint32_t i = threadIdx.x; i < sizeof(sgd_vehicle) / 4; i += 64) dst[i] = src[i];
	__syncthreads();
}
SGP_DEV void veh_stage_out(sgd_vehicle* gv, const sgd_vehicle* sv)
{
	__syncthreads();
	const uint32_t* src = (const uint32_t*)sv; uint32_t* dst = (uint32_t*)gv;
	for (uint32_t i = threadIdx.x; i < sizeof(sgd_vehicle) / 4; i += 64) dst[i] = src[i];
}

// VehicleConstraint::OnStep for every vehicle whose chassis is awake: runs after this step's broad-phase grid is built (the
// wheel casts walk it) and before the forces are applied.  Two launches so that no vehicle reads a chassis velocity another
// vehicle is updating: (A) k_vehicle_cast -- wheel casts, read-only on the bodies; (B) k_vehicle_controller -- tyres,
// drivetrain, row setup, anti-roll impulses on the own chassis.
// Cast: 16 lanes per wheel share the candidate list (large bodies + the grid cells under the swept sphere); each lane keeps its
// closest accepted hit and a butterfly reduction takes the lexicographic (distance, body id) minimum, which does not depend
// on how the candidates were dealt to the lanes.
__global__ void __launch_bounds__(64) k_vehicle_cast(DV d)
{
	__shared__ sgd_vehicle sv;
	const uint32_t k = blockIdx.x;
	sgd_vehicle* gv = &d.vehicles[k];
	if ((k & 31u) == 0u && threadIdx.x == 0) d.veh_defer_bits[k >> 5] = 0u;      // (k_vehicle_controller, the next launch, sets the bits of this step)
	if (!gv->alive) return;
	veh_stage_in(&sv, gv);
	if (threadIdx.x == 0) {
		const sgp_vehicle_input in = d.vehicle_inputs[k];
		sv.in_forward = in.forward; sv.in_right = in.right; sv.in_brake = in.brake; sv.in_handbrake = in.hand_brake;
		const uint32_t b = sv.body;
		sv.active = (b < d.sp->n_slots && f_movable(d.flags[b])) ? 1 : 0;
	}
	__syncthreads();
	if (sv.active) {
		{ const sgd_chassis c = veh_chassis_pose_vel(d, sv.body); sgd_vehicle_precast_lanes(&sv, &c, d.sp->dt, (int)threadIdx.x); }
		__syncthreads();
		const int wi = (int)(threadIdx.x >> 4); const uint32_t sub = threadIdx.x & 15u;
		float best = 0.0f; uint32_t bid = SGP_INVALID_ID; v3 bn = V3(0.0f, 0.0f, 0.0f), bp = bn;
		if (wi < sv.num_wheels) {
			const sgd_wheel* wh = &sv.wheels[wi];
			const v3 o = wh->cast_origin, dir = wh->cast_dir;
			const float rs = sv.cast_radius;
			best = wh->cast_len;
			for (uint32_t l = sub; l < d.sp->n_large; l += 16) veh_cast_test(d, &sv, o, dir, rs, wh->cast_len, d.large_ids[l], best, bid, bn, bp);
			{
				// static large bodies under the swept sphere's bounds, dealt to the wheel's 16 lanes in the order the grid yields them
				const v3 e2 = v3_add(o, v3_scale(dir, wh->cast_len));
				const float m2 = rs + 2.0e-3f;
				uint32_t seen = 0;
				large_grid_query(d, V3(fminf(o.x, e2.x) - m2, fminf(o.y, e2.y) - m2, fminf(o.z, e2.z) - m2), V3(fmaxf(o.x, e2.x) + m2, fmaxf(o.y, e2.y) + m2, fmaxf(o.z, e2.z) + m2),
				                 [&](uint32_t i) { if ((seen++ & 15u) == sub) veh_cast_test(d, &sv, o, dir, rs, wh->cast_len, i, best, bid, bn, bp); });
			}
			const BpGrid g = *d.grid;
			if (g.n_cells > 0 && g.min_x <= g.max_x) {
				// cells overlapped by the swept sphere's box, one more cell each side (bodies are binned by centre and reach at most one cell beyond it)
				const v3 e = v3_add(o, v3_scale(dir, wh->cast_len));
				const float m = rs + 1.0e-3f;
				const int x0 = max((int)floorf((fminf(o.x, e.x) - m - g.ox) * g.inv_cell) - 1, 0), x1 = min((int)floorf((fmaxf(o.x, e.x) + m - g.ox) * g.inv_cell) + 1, g.nx - 1);
				const int y0 = max((int)floorf((fminf(o.y, e.y) - m - g.oy) * g.inv_cell) - 1, 0), y1 = min((int)floorf((fmaxf(o.y, e.y) + m - g.oy) * g.inv_cell) + 1, g.ny - 1);
				const int z0 = max((int)floorf((fminf(o.z, e.z) - m - g.oz) * g.inv_cell) - 1, 0), z1 = min((int)floorf((fmaxf(o.z, e.z) + m - g.oz) * g.inv_cell) + 1, g.nz - 1);
				if (x0 <= x1) for (int z = z0; z <= z1; ++z) for (int y = y0; y <= y1; ++y) {
					grid_row_runs(d, g, x0, x1, y, z, [&](uint32_t q0, uint32_t q1) { for (uint32_t q = q0 + sub; q < q1; q += 16) veh_cast_test(d, &sv, o, dir, rs, wh->cast_len, __float_as_uint(d.sorted_max[q].w), best, bid, bn, bp); });
				}
			}
		}
		// (distance, id) minimum over the 16 lanes of the wheel; lanes without a hit carry id = invalid
#pragma unroll
		for (int off = 8; off >= 1; off >>= 1) {
			const float ot = __shfl_xor(best, off, 16); const uint32_t oid = __shfl_xor(bid, off, 16);
			const float onx = __shfl_xor(bn.x, off, 16), ony = __shfl_xor(bn.y, off, 16), onz = __shfl_xor(bn.z, off, 16);
			const float opx = __shfl_xor(bp.x, off, 16), opy = __shfl_xor(bp.y, off, 16), opz = __shfl_xor(bp.z, off, 16);
			const bool take = oid != SGP_INVALID_ID && (bid == SGP_INVALID_ID || ot < best || (ot == best && oid < bid));
			if (take) { best = ot; bid = oid; bn = V3(onx, ony, onz); bp = V3(opx, opy, opz); }
		}
		if (wi < sv.num_wheels && sub == 0 && bid != SGP_INVALID_ID) {
			const uint32_t fo = d.flags[bid];
			v3 gvel = V3(0.0f, 0.0f, 0.0f);
			if (f_motion(fo) != SGP_MOTION_STATIC) gvel = v3_add(V3(d.vel[2 * (size_t)bid]), v3_cross(V3(d.vel[2 * (size_t)bid + 1]), v3_sub(bp, V3(d.pose[2 * (size_t)bid]))));
			sgd_vehicle_set_hit(&sv, wi, bid, best, bn, bp, gvel, d.prop[2 * (size_t)bid + 1].w);
			if (f_motion(fo) == SGP_MOTION_DYNAMIC) {
				// the rows act on a dynamic body under the wheel (VehicleConstraint::SetupVelocityConstraint, body 2): it wakes up if it sleeps
				// (VehicleConstraint::BuildIslands; k_pre_solve does it, like for a body an active one touches) and this vehicle claims it
				sv.wheels[wi].ground_dynamic = 1;
				if (!(fo & BF_ACTIVE)) wake_body(d, bid);
				atomicMax((unsigned long long*)&d.veh_claim[bid], ((unsigned long long)*d.veh_epoch << 32) | (unsigned long long)(0xFFFFFFFFu - k));
			}
		}
		if (threadIdx.x == 0) atomicMax((unsigned long long*)&d.veh_claim[sv.body], ((unsigned long long)*d.veh_epoch << 32) | (unsigned long long)(0xFFFFFFFFu - k));
	}
	veh_stage_out(gv, &sv);
}

SGP_DEV void veh_export(const DV& d, uint32_t k, const sgd_vehicle& v, int deferred);
__global__ void __launch_bounds__(64) k_vehicle_controller(DV d)
{
	__shared__ sgd_vehicle sv;
	int sv_deferred = 0;          // (uniform over the wave)
	sgd_vehicle* gv = &d.vehicles[blockIdx.x];
	if (!gv->alive || !gv->active) { if (threadIdx.x == 0) d.veh_head[(size_t)blockIdx.x * VEH_HEAD_F4] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); return; }      // (no rows this step)
	veh_stage_in(&sv, gv);
	{
		// every lane holds the chassis state (one broadcast load); lane i works on wheel i, lane 0 on what couples the wheels
		const uint32_t b = sv.body;
		sgd_chassis c = veh_chassis_pose_vel(d, b);
		const float lvw = d.vel[2 * (size_t)b].w, avw = d.vel[2 * (size_t)b + 1].w;
		// lane i: the dynamic body under wheel i, as the row set-up needs it (inverse mass of the body, not of the step: a sleeper is woken by this step's k_pre_solve)
		sgd_ground g; g.dyn = 0; g.pos = V3(0.0f, 0.0f, 0.0f); g.im = 0.0f; g.I = sym33_zero();
		bool lost = false;
		const unsigned long long my_claim = ((unsigned long long)*d.veh_epoch << 32) | (unsigned long long)(0xFFFFFFFFu - blockIdx.x);
		if ((int)threadIdx.x < sv.num_wheels && sv.wheels[threadIdx.x].has_contact && sv.wheels[threadIdx.x].ground_dynamic) {
			const uint32_t gb = sv.wheels[threadIdx.x].contact_body;
			const float4 gp = d.pose[2 * (size_t)gb];
			g.dyn = 1; g.pos = V3(gp); g.im = gp.w;
			g.I = world_inv_inertia(quat_to_m33(Q4(d.pose[2 * (size_t)gb + 1])), V3(d.prop[2 * (size_t)gb]));
			lost = d.veh_claim[gb] != my_claim;
		}
		if (threadIdx.x == 0 && d.veh_claim[b] != my_claim) lost = true;
		// a vehicle that shares a movable body with one of lower index waits for it in every pass (veh_block_solve)
		if (__any(lost)) { sv_deferred = 1; if (threadIdx.x == 0) { atomicOr(&d.veh_defer_bits[blockIdx.x >> 5], 1u << (blockIdx.x & 31u)); atomicAdd(&d.ctr->veh_deferred, 1u); } }
		const int spinning = sgd_vehicle_controller_lanes(&sv, &c, &g, d.sp->dt, (int)threadIdx.x);
		if (threadIdx.x == 0) {
			if (spinning) d.sleep_timer[b] = 0.0f;
			d.vel[2 * (size_t)b] = F4(c.v, lvw); d.vel[2 * (size_t)b + 1] = F4(c.w, avw);      // (the anti-roll impulses)
		}
	}
	__syncthreads();
	veh_export(d, blockIdx.x, sv, sv_deferred);          // the rows of this step, lane-major, for the solver passes
	veh_stage_out(gv, &sv);
}

// ---- the vehicle rows inside the solver passes: four lanes per vehicle -------------------------------------------------------
// A solver pass visits a vehicle's rows in a fixed order (VehicleConstraint::SolveVelocityConstraint, then the controller's
// longitudinal and lateral rows: suspension + upper stop of wheel 0..3, longitudinal 0..3, lateral 0..3, the motorcycle's lean
// spring), every row reading the chassis velocity the previous one left: a chain, not a reduction.  What the lanes buy is the
// memory side: lane i of a quad holds wheel i's rows in registers -- sixteen 16-byte chunks per wheel that k_vehicle_controller
// exported in a lane-major layout (DV::veh_rows: chunk c of wheel i of vehicle k at [c][4 k + i], so one load instruction of a
// wave fetches 1 KB contiguous) --, all four lanes carry a copy of the chassis state, the lane whose turn it is advances it and a
// DPP quad broadcast (a register move modifier, no LDS round trip) hands it to the other three.  No LDS, no workgroup barrier:
// the quads of sixteen vehicles share a wave, and the same code runs as the first workgroups of a contact-colour launch
// (k_solve_colour_veh below).  Row impulses and wheel spin also go back to the vehicle record, which stays the one the host
// reads and the next step's cast / controller kernels start from.
//      VEH_CHUNK_NORMAL 0      // contact normal | bits: 1 has contact, 2 suspension row, 4 upper-stop row, 8 longitudinal row, 16 lateral row; from bit 5: id + 1 of the DYNAMIC body under the wheel (0: none, the ground does not move under the rows)
#define VEH_CHUNK_LONG 1        // longitudinal direction | combined longitudinal friction
#define VEH_CHUNK_LAT 2         // lateral direction | combined lateral friction
#define VEH_CHUNK_GVEL 3        // velocity of the ground at the contact point | brake impulse
#define VEH_CHUNK_CPOS 4        // contact position | wheel angular velocity (state)
#define VEH_CHUNK_MISC 5        // radius, inertia, suspension softness, suspension bias
#define VEH_CHUNK_ROW 6         // + 2 r: r1 x axis | effective mass;  + 2 r + 1: I^-1 (r1 x axis) | accumulated impulse (state); r = 0 suspension, 1 upper stop, 2 longitudinal, 3 lateral
#define VEH_CHUNK_WPOS 14       // wheel position (chassis space) | minimum suspension length
#define VEH_CHUNK_SDIR 15       // suspension direction (chassis space) | axle plane constant
// VEH_HEAD_F4 = 5 float4 per vehicle: (body, bits: 1 active 2 lean spring on 4 deferred (solved after the others, in index order), wheels, integrated lean error), forward | K, up | D, target lean | Ki, (decay, applied lean impulse, -, -)

template <int I> SGP_DEV float quad_bcast(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), I * 0x55, 0xF, 0xF, true)); }      // quad_perm [I, I, I, I]
template <int I> SGP_DEV v3 quad_bcast(v3 a) { return V3(quad_bcast<I>(a.x), quad_bcast<I>(a.y), quad_bcast<I>(a.z)); }

// the chunk (c) of wheel (i) of the vehicle record in LDS: what k_vehicle_controller's 64 lanes write out, one chunk each
SGP_DEV float4 veh_export_chunk(const sgd_vehicle& v, int i, int c)
{
	const sgd_wheel& w = v.wheels[i];
	const sgd_axis_part* part[4] = { &w.suspension, &w.max_up, &w.longitudinal, &w.lateral };
	if (i >= v.num_wheels) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	if (c == VEH_CHUNK_NORMAL) return F4(w.contact_normal, __uint_as_float((w.has_contact ? 1u : 0u) | (w.suspension.active ? 2u : 0u) | (w.max_up.active ? 4u : 0u) | (w.longitudinal.active ? 8u : 0u) | (w.lateral.active ? 16u : 0u)
	                                                                       | ((w.has_contact && w.ground_dynamic) ? (w.contact_body + 1u) << 5 : 0u)));
	if (c == VEH_CHUNK_LONG) return F4(w.contact_long, w.comb_long_fric);
	if (c == VEH_CHUNK_LAT) return F4(w.contact_lat, w.comb_lat_fric);
	if (c == VEH_CHUNK_GVEL) return F4(w.contact_point_vel, w.brake_impulse);
	if (c == VEH_CHUNK_CPOS) return F4(w.contact_pos, w.angular_velocity);
	if (c == VEH_CHUNK_MISC) return make_float4(w.radius, w.inertia, w.suspension.softness, w.suspension.bias);
	if (c == VEH_CHUNK_WPOS) return F4(w.position, w.sus_min);
	if (c == VEH_CHUNK_SDIR) return F4(w.suspension_dir, w.axle_plane_constant);
	const sgd_axis_part& p = *part[(c - VEH_CHUNK_ROW) >> 1];
	return ((c - VEH_CHUNK_ROW) & 1) ? F4(p.iI_r1xa, p.lambda) : F4(p.r1xa, p.eff);
}
SGP_DEV void veh_export(const DV& d, uint32_t k, const sgd_vehicle& v, int deferred)
{
	const int i = (int)(threadIdx.x & 3u), c = (int)(threadIdx.x >> 2);
	d.veh_rows[veh_chunk_at(d, k, i, c)] = veh_export_chunk(v, i, c);
	if (threadIdx.x < VEH_HEAD_F4) {
		const uint32_t bits = (v.active ? 1u : 0u) | ((v.is_motorcycle && v.lean_enabled) ? 2u : 0u) | (deferred ? 4u : 0u);
		float4 h;
		if (threadIdx.x == 0) h = make_float4(__uint_as_float(v.body), __uint_as_float(bits), __uint_as_float((uint32_t)v.num_wheels), v.lean_integrated_delta);
		else if (threadIdx.x == 1) h = F4(v.forward, v.lean_spring_constant);
		else if (threadIdx.x == 2) h = F4(v.up, v.lean_spring_damping);
		else if (threadIdx.x == 3) h = F4(v.target_lean, v.lean_integration_coefficient);
		else h = make_float4(v.lean_integration_decay, v.lean_applied_impulse, 0.0f, 0.0f);
		d.veh_head[(size_t)k * VEH_HEAD_F4 + threadIdx.x] = h;
	}
}

// the chassis as the rows see it, one copy per lane of the quad
struct VehBody { v3 v, w; float im; sym33 I; };
// the dynamic body under this lane's wheel (id == SGP_INVALID_ID: none): its velocity is read and written by the rows like the chassis'
struct VehGround { uint32_t id; v3 v, w; float im; sym33 I; v3 r2; };
// one row between the chassis and the ground: AxisConstraintPart::SolveVelocityConstraint with the spring's softness and bias.  A ground that is
// not dynamic contributes the contact point velocity sampled at cast time (gvel); a dynamic one is read live and takes the reaction.
SGP_DEV void veh_row_solve(VehBody& c, VehGround& g, float4 ra, float4& ri, float softness, float bias, v3 ground_vel, v3 axis, float lo, float hi)
{
	const v3 r1xa = V3(ra), iI = V3(ri);
	const bool two = g.id != SGP_INVALID_ID;
	v3 r2xa = V3(0.0f, 0.0f, 0.0f), iI2 = r2xa;
	float jv;
	if (two) {
		r2xa = v3_cross(g.r2, axis);
		iI2 = sym33_mul(g.I, r2xa);
		jv = (v3_dot(axis, v3_sub(c.v, g.v)) + v3_dot(r1xa, c.w)) - v3_dot(r2xa, g.w);
	} else jv = v3_dot(axis, v3_sub(c.v, ground_vel)) + v3_dot(r1xa, c.w);
	const float lambda = ra.w * (jv - (softness * ri.w + bias));
	const float nl = clampf(ri.w + lambda, lo, hi);
	const float dl = nl - ri.w;
	c.v = v3_sub(c.v, v3_scale(axis, dl * c.im));
	c.w = v3_sub(c.w, v3_scale(iI, dl));
	if (two) {
		g.v = v3_add(g.v, v3_scale(axis, dl * g.im));
		g.w = v3_add(g.w, v3_scale(iI2, dl));
	}
	ri.w = nl;
}
SGP_DEV void veh_row_apply(VehBody& c, VehGround& g, float4 ri, v3 axis)
{
	c.v = v3_sub(c.v, v3_scale(axis, ri.w * c.im));
	c.w = v3_sub(c.w, v3_scale(V3(ri), ri.w));
	if (g.id != SGP_INVALID_ID) {
		g.v = v3_add(g.v, v3_scale(axis, ri.w * g.im));
		g.w = v3_add(g.w, v3_scale(sym33_mul(g.I, v3_cross(g.r2, axis)), ri.w));
	}
}
template <int I> SGP_DEV uint32_t quad_bcast_u(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, I * 0x55, 0xF, 0xF, true); }
// after lane I's turn: its chassis velocity becomes everybody's, and its ground's velocity that of every lane whose wheel stands on the same body
#define VEH_TURN(WI, ...) { if (L == WI) { __VA_ARGS__ } c.v = quad_bcast<WI>(c.v); c.w = quad_bcast<WI>(c.w); \
	{ const uint32_t og = quad_bcast_u<WI>(g.id); const v3 ov = quad_bcast<WI>(g.v), ow = quad_bcast<WI>(g.w); if (g.id != SGP_INVALID_ID && og == g.id) { g.v = ov; g.w = ow; } } }
#define VEH_TURNS(...) VEH_TURN(0, __VA_ARGS__) VEH_TURN(1, __VA_ARGS__) VEH_TURN(2, __VA_ARGS__) VEH_TURN(3, __VA_ARGS__)

// k = vehicle slot, L = lane of its quad; every lane of a quad takes the same branches up to the turns.  DEFERRED: this call is the catch-all's
// (vehicles that wait for one of lower index); the regular call skips those.
template <int MODE> SGP_DEV void veh_quad_solve(const DV& d, uint32_t k, int L, bool deferred_pass)
{
	if (k >= d.n_vehicles) return;
	const float4 h0 = d.veh_head[(size_t)k * VEH_HEAD_F4];
	const uint32_t hbits = __float_as_uint(h0.y);
	if (!(hbits & 1u)) return;                       // not alive, or the chassis was asleep when this step's pre-step ran
	if (((hbits & 4u) != 0u) != deferred_pass) return;
	const uint32_t b = __float_as_uint(h0.x);
	const int nw = (int)__float_as_uint(h0.z);
	sgd_vehicle* gv = &d.vehicles[k];
	sgd_wheel* gw = &gv->wheels[L];
	const float4 cn = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_NORMAL)];
	const uint32_t wbits = L < nw ? __float_as_uint(cn.w) : 0u;
	const bool contact = wbits & 1u;
	const uint32_t gid = (wbits >> 5) ? (wbits >> 5) - 1u : SGP_INVALID_ID;
	const v3 neg_n = v3_neg(V3(cn));
	if (MODE == 2) {
		// VehicleConstraint::SolvePositionConstraint: the axle at minimum suspension length stays on the outer side of the plane through the
		// axle position at cast time; wheel after wheel on the poses the previous one left (the chassis', and that of a dynamic body under the wheel)
		const float4 cp = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_CPOS)], wp = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_WPOS)], sd = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_SDIR)];
		const float4 p4 = d.pose[2 * (size_t)b], q4 = d.pose[2 * (size_t)b + 1];
		const v3 iil = V3(d.prop[2 * (size_t)b]);
		v3 pos = V3(p4); quat rot = Q4(q4);
		const float im = p4.w, baumgarte = d.st.baumgarte;
		float4 gp4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), gq4 = make_float4(0.0f, 0.0f, 0.0f, 1.0f); v3 giil = V3(0.0f, 0.0f, 0.0f);
		if (gid != SGP_INVALID_ID) { gp4 = d.pose[2 * (size_t)gid]; gq4 = d.pose[2 * (size_t)gid + 1]; giil = V3(d.prop[2 * (size_t)gid]); }
		v3 gpos = V3(gp4); quat grot = Q4(gq4);
		bool gmoved = false;
#define VEH_POS_TURN(WI) { \
		if (L == WI && contact) { \
			const m33 R = quat_to_m33(rot); \
			const v3 ws_dir = m33_mul(R, V3(sd)); \
			const v3 ws_pos = v3_add(pos, m33_mul(R, V3(wp))); \
			const v3 min_pos = v3_add(ws_pos, v3_scale(ws_dir, wp.w)); \
			const float err = v3_dot(V3(cn), min_pos) - sd.w; \
			if (err < 0.0f) { \
				const v3 r1 = v3_sub(V3(cp), pos); \
				const sym33 I = world_inv_inertia(R, iil); \
				const v3 r1xa = v3_cross(r1, neg_n); \
				const v3 iI = sym33_mul(I, r1xa); \
				float inv_eff = im + v3_dot(r1xa, iI); \
				v3 iI2 = V3(0.0f, 0.0f, 0.0f); \
				if (gid != SGP_INVALID_ID) { \
					const v3 r2xa = v3_cross(v3_sub(V3(cp), gpos), neg_n); \
					iI2 = sym33_mul(world_inv_inertia(quat_to_m33(grot), giil), r2xa); \
					inv_eff = inv_eff + (gp4.w + v3_dot(r2xa, iI2)); \
				} \
				if (inv_eff > 0.0f) { \
					const float lambda = -(1.0f / inv_eff) * baumgarte * err; \
					pos = v3_sub(pos, v3_scale(neg_n, lambda * im)); \
					rot = quat_add_rotation_step(rot, v3_scale(iI, -lambda)); \
					if (gid != SGP_INVALID_ID) { \
						gpos = v3_add(gpos, v3_scale(neg_n, lambda * gp4.w)); \
						grot = quat_add_rotation_step(grot, v3_scale(iI2, lambda)); \
						gmoved = true; \
					} \
				} \
			} \
		} \
		pos = quad_bcast<WI>(pos); rot.x = quad_bcast<WI>(rot.x); rot.y = quad_bcast<WI>(rot.y); rot.z = quad_bcast<WI>(rot.z); rot.w = quad_bcast<WI>(rot.w); \
		{ const uint32_t og = quad_bcast_u<WI>(gid); const v3 op = quad_bcast<WI>(gpos); const float ox = quad_bcast<WI>(grot.x), oy = quad_bcast<WI>(grot.y), oz = quad_bcast<WI>(grot.z), ow = quad_bcast<WI>(grot.w); \
		  if (L != WI && gid != SGP_INVALID_ID && og == gid) { gpos = op; grot.x = ox; grot.y = oy; grot.z = oz; grot.w = ow; } } }
		VEH_POS_TURN(0) VEH_POS_TURN(1) VEH_POS_TURN(2) VEH_POS_TURN(3)
#undef VEH_POS_TURN
		if (L == 0) { d.pose[2 * (size_t)b] = F4(pos, p4.w); d.pose[2 * (size_t)b + 1] = make_float4(rot.x, rot.y, rot.z, rot.w); }
		if (gmoved) { d.pose[2 * (size_t)gid] = F4(gpos, gp4.w); d.pose[2 * (size_t)gid + 1] = make_float4(grot.x, grot.y, grot.z, grot.w); }      // (the lane that moved it: a later lane on the same body started from this pose)
		return;
	}
	// the rows of this lane's wheel
	float4 ra[4], ri[4];
#pragma unroll
	for (int r = 0; r < 4; ++r) { ra[r] = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_ROW + 2 * r)]; ri[r] = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_ROW + 2 * r + 1)]; }
	const float4 s0 = d.vel[2 * (size_t)b], s1 = d.vel[2 * (size_t)b + 1];
	VehBody c;
	c.v = V3(s0); c.im = s0.w; c.w = V3(s1);                  // (s0.w: the effective inverse mass of this step, k_pre_solve)
	const quat crot = Q4(d.pose[2 * (size_t)b + 1]);
	c.I = c.im > 0.0f ? world_inv_inertia(quat_to_m33(crot), V3(d.prop[2 * (size_t)b])) : sym33_zero();
	float4 cp = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_CPOS)];
	// the dynamic body under the wheel: velocity record (live), inverse mass, world inverse inertia, lever arm
	VehGround g; g.id = gid; g.v = V3(0.0f, 0.0f, 0.0f); g.w = g.v; g.im = 0.0f; g.I = sym33_zero(); g.r2 = g.v;
	float4 g0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), g1 = g0;
	if (gid != SGP_INVALID_ID) {
		g0 = d.vel[2 * (size_t)gid]; g1 = d.vel[2 * (size_t)gid + 1];
		const float4 gp4 = d.pose[2 * (size_t)gid];
		g.v = V3(g0); g.w = V3(g1); g.im = gp4.w;
		g.I = world_inv_inertia(quat_to_m33(Q4(d.pose[2 * (size_t)gid + 1])), V3(d.prop[2 * (size_t)gid]));
		g.r2 = v3_sub(V3(cp), V3(gp4));
	}
	if (MODE == 0) {
		// VehicleConstraint::WarmStartVelocityConstraint: suspension, upper stop, lateral (the longitudinal row starts every step from zero)
		const v3 neg_lat = v3_neg(V3(d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_LAT)]));
		VEH_TURNS(if (contact) { if (wbits & 2u) veh_row_apply(c, g, ri[0], neg_n); if (wbits & 4u) veh_row_apply(c, g, ri[1], neg_n); if (wbits & 16u) veh_row_apply(c, g, ri[3], neg_lat); })
		if (L == 0) { d.vel[2 * (size_t)b] = F4(c.v, s0.w); d.vel[2 * (size_t)b + 1] = F4(c.w, s1.w); }
		if (gid != SGP_INVALID_ID) { d.vel[2 * (size_t)gid] = F4(g.v, g0.w); d.vel[2 * (size_t)gid + 1] = F4(g.w, g1.w); }
		return;
	}
	const float4 cl = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_LONG)], ct = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_LAT)], cg = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_GVEL)];
	const float4 cm = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_MISC)];
	const v3 cpos = V3(d.pose[2 * (size_t)b]);
	const v3 gvel = V3(cg);
	// 1. suspension spring and upper stop: push, never pull
	VEH_TURNS(if (contact) { if (wbits & 2u) veh_row_solve(c, g, ra[0], ri[0], cm.z, cm.w, gvel, neg_n, 0.0f, 3.0e38f); if (wbits & 4u) veh_row_solve(c, g, ra[1], ri[1], 0.0f, 0.0f, gvel, neg_n, 0.0f, 3.0e38f); })
	// 2. longitudinal: the brake within the friction limit, or the impulse that brings the contact patch to the wheel's rolling speed in this step
	//    (relative to the contact point velocity sampled at cast time, like WheeledVehicleController::SolveLongitudinalAndLateralConstraints)
	const float sus_lambda = ri[0].w + ri[1].w;
	const float max_long = cl.w * sus_lambda, max_lat = contact ? ct.w * sus_lambda : 0.0f;
	VEH_TURNS(if (contact && (wbits & 8u)) {
		const v3 rel = v3_sub(v3_add(c.v, v3_cross(c.w, v3_sub(V3(cp), cpos))), gvel);
		const float rel_long = v3_dot(rel, V3(cl));
		if (cg.w != 0.0f) {
			const float bi = fminf(cg.w, max_long);
			float lo, hi;
			if (rel_long >= 0.0f) { lo = -bi; hi = 0.0f; } else { lo = 0.0f; hi = bi; }
			veh_row_solve(c, g, ra[2], ri[2], 0.0f, 0.0f, gvel, v3_neg(V3(cl)), lo, hi);
		} else {
			const float desired_w = rel_long / cm.x;
			const float lin_imp = (cp.w - desired_w) * cm.y / cm.x;
			const float prev = ri[2].w;
			const float lim = clampf(prev + lin_imp, -max_long, max_long);
			veh_row_solve(c, g, ra[2], ri[2], 0.0f, 0.0f, gvel, v3_neg(V3(cl)), lim, lim);
			cp.w = cp.w - (ri[2].w - prev) * cm.x / cm.y;
		}
	})
	// 3. lateral
	VEH_TURNS(if (contact && (wbits & 16u)) veh_row_solve(c, g, ra[3], ri[3], 0.0f, 0.0f, gvel, v3_neg(V3(ct)), -max_lat, max_lat);)
	// 4. MotorcycleController: the lean spring (a PID on the angle to the target lean), only with every wheel loaded; the matching linear impulse keeps
	//    the contact patches from being swept sideways.  Every lane of the quad computes it (the sums run over the wheels in order 0..3).
	float lean_integrated = h0.w, lean_applied = 0.0f;
	if (hbits & 2u) {
		const float4 h1 = d.veh_head[(size_t)k * VEH_HEAD_F4 + 1], h2 = d.veh_head[(size_t)k * VEH_HEAD_F4 + 2], h3 = d.veh_head[(size_t)k * VEH_HEAD_F4 + 3], h4 = d.veh_head[(size_t)k * VEH_HEAD_F4 + 4];
		lean_applied = h4.y;
		const float lam = ri[0].w + ri[1].w;
		const v3 arm = v3_sub(V3(cp), cpos);
		const float lam_w[4] = { quad_bcast<0>(lam), quad_bcast<1>(lam), quad_bcast<2>(lam), quad_bcast<3>(lam) };
		const v3 arm_w[4] = { quad_bcast<0>(arm), quad_bcast<1>(arm), quad_bcast<2>(arm), quad_bcast<3>(arm) };
		const uint32_t con = contact ? 1u : 0u;
		const uint32_t con_w[4] = { quad_bcast_u<0>(con), quad_bcast_u<1>(con), quad_bcast_u<2>(con), quad_bcast_u<3>(con) };
		bool all_in_contact = true;
#pragma unroll
		for (int i = 0; i < 4; ++i) if (i < nw && (!con_w[i] || !(lam_w[i] > 0.0f))) all_in_contact = false;
		const float dt = d.sp->dt;
		if (all_in_contact) {
			const m33 R = quat_to_m33(crot);
			const v3 forward = m33_mul(R, V3(h1)), up = m33_mul(R, V3(h2));
			const v3 target = V3(h3);
			const float d_angle = -sgd_signf(v3_dot(v3_cross(target, up), forward)) * sgd_acos11(clampf(v3_dot(target, up), -1.0f, 1.0f));
			const float ddt_angle = v3_dot(c.w, forward);
			// (the fixed point of Jolt's per-iteration re-evaluation, solved for directly: DESIGN.md 4b)
			const v3 If = sym33_mul(c.I, forward);
			const float iff = v3_dot(forward, If);
			const float wf0 = ddt_angle - iff * lean_applied;
			const float total = (h1.w * d_angle - h2.w * wf0 + h3.w * lean_integrated) * dt / (1.0f + h2.w * dt * iff);
			const v3 old_w = c.w;
			c.w = v3_add(c.w, v3_scale(If, total - lean_applied));
			lean_applied = total;
			const v3 dw = v3_sub(c.w, old_w);
			v3 lin_acc = V3(0.0f, 0.0f, 0.0f); float total_lambda = 0.0f;
#pragma unroll
			for (int i = 0; i < 4; ++i) if (i < nw) { total_lambda = total_lambda + lam_w[i]; lin_acc = v3_add(lin_acc, v3_scale(v3_cross(dw, arm_w[i]), lam_w[i])); }
			c.v = v3_sub(c.v, v3_scale(lin_acc, 1.0f / total_lambda));
		} else lean_integrated = lean_integrated * fmaxf(0.0f, 1.0f - h4.x * dt);
	}
	// state back: the row chunks (next pass), the vehicle record (host reads, next step's pre-step), the velocities of the chassis and of the bodies under the wheels
	if (L < nw) {
#pragma unroll
		for (int r = 0; r < 4; ++r) d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_ROW + 2 * r + 1)] = ri[r];
		d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_CPOS)] = cp;
		gw->suspension.lambda = ri[0].w; gw->max_up.lambda = ri[1].w; gw->longitudinal.lambda = ri[2].w; gw->lateral.lambda = ri[3].w;
		gw->angular_velocity = cp.w;
	}
	if (gid != SGP_INVALID_ID) { d.vel[2 * (size_t)gid] = F4(g.v, g0.w); d.vel[2 * (size_t)gid + 1] = F4(g.w, g1.w); }      // (lanes on the same body hold the same values)
	if (L == 0) {
		d.vel[2 * (size_t)b] = F4(c.v, s0.w); d.vel[2 * (size_t)b + 1] = F4(c.w, s1.w);
		if (hbits & 2u) {
			d.veh_head[(size_t)k * VEH_HEAD_F4] = make_float4(h0.x, h0.y, h0.z, lean_integrated);
			float4* h4p = &d.veh_head[(size_t)k * VEH_HEAD_F4 + 4];
			*h4p = make_float4(h4p->x, lean_applied, 0.0f, 0.0f);
			gv->lean_integrated_delta = lean_integrated; gv->lean_applied_impulse = lean_applied;
		}
	}
}
#undef VEH_TURNS
#undef VEH_TURN

// The vehicle rows of one pass by the workgroups [0, n_blocks) of a launch: every vehicle that shares no movable body with one of lower index by
// its quad; the others (StepCounters::veh_deferred: two cars with a wheel on the same loose box, a car standing on another) afterwards, one after
// the other in index order, by the first quad of the workgroup that finishes last -- the order a sequential solve visits them in.
template <int MODE> SGP_DEV void veh_block_solve(const DV& d, uint32_t block, uint32_t n_blocks)
{
	__shared__ uint32_t s_veh_ticket;
	const uint32_t t = block * blockDim.x + threadIdx.x;
	veh_quad_solve<MODE>(d, t >> 2, (int)(t & 3u), false);
	if (d.ctr->veh_deferred == 0u) return;
	__syncthreads();
	if (threadIdx.x == 0) { __threadfence(); s_veh_ticket = atomicAdd(&d.ctr->veh_done, 1u); }
	__syncthreads();
	if (s_veh_ticket != n_blocks - 1u) return;
	if (threadIdx.x == 0) d.ctr->veh_done = 0u;      // for the next launch
	__threadfence();          // what the other workgroups wrote (and this compute unit may still hold older copies of)
	if (threadIdx.x >= 4u) return;
	const uint32_t n_words = (d.n_vehicles + 31u) >> 5;
	for (uint32_t wd = 0; wd < n_words; ++wd) {
		uint32_t bits = d.veh_defer_bits[wd];
		while (bits) {
			const uint32_t k = (wd << 5) + (uint32_t)__ffs((int)bits) - 1u;
			bits &= bits - 1u;
			veh_quad_solve<MODE>(d, k, (int)threadIdx.x, true);
			__threadfence_block();      // the next vehicle may read what this one wrote
		}
	}
}

#define VEH_SOLVE_TPB SOLVE_VEL_TPB      // 64 vehicles per workgroup
// MODE 0 warm start, 1 velocity iteration, 2 position iteration (the chassis pose)
template <int MODE> __global__ void __launch_bounds__(VEH_SOLVE_TPB) k_vehicle_solve(DV d)
{
	veh_block_solve<MODE>(d, blockIdx.x, gridDim.x);
}

// The first contact colour of a velocity / position pass with the vehicles' rows in the same launch: no contact of a chassis sits in colour 0
// (chassis_colours), so the two touch disjoint bodies and the pass order "vehicles, then the contact colours" holds without a launch of
// its own.  The vehicle workgroups come first in the grid: they are the longer chains.
template <int MODE, int ROWS = -1> __global__ void __launch_bounds__(SOLVE_VEL_TPB) k_solve_colour_veh(DV d, int colour_arg, uint32_t veh_blocks)
{
	const int colour = colour_arg & 0xFF;
	if (blockIdx.x < veh_blocks) {
		veh_block_solve<MODE>(d, blockIdx.x, veh_blocks);
		return;
	}
	const uint32_t first = d.cstarts[colour], end = d.cstarts[colour + 1];
	const int side = (int)(threadIdx.x & 1u);
	const uint32_t cb = blockIdx.x - veh_blocks, cg = gridDim.x - veh_blocks;      // (the colour's workgroups: XCD-contiguous chunks as in k_solve_colour; cg is a multiple of eight)
	const uint32_t bx = (colour_arg & SOLVE_XCD_CHUNKS) ? (cb & 7u) * (cg >> 3) + (cb >> 3) : cb;
	for (uint32_t k = first + ((bx * SOLVE_VEL_TPB + threadIdx.x) >> 1); k < end; k += cg * (SOLVE_VEL_TPB / 2)) {
		if (MODE == 1) solve_velocity_pair_t<2, ROWS>(d, k, side, d.vel); else solve_position_pair(d, k, side);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Shape queries of the character controller (JPH::CharacterVirtual: CollideShape with a maximum separation, swept test).

// the points of one manifold (normal: body -> capsule) as contacts of query k with body j
SGP_DEV void capsule_emit(const DV& d, uint32_t k, uint32_t j, uint32_t f, int g, const sgd_manifold& m, sgp_query_contact* out, uint32_t cap, uint32_t* count)
{
	for (int i = 0; i < m.np; ++i) {
		const uint32_t slot = atomicAdd(count, 1u);
		if (slot >= cap) continue;
		sgp_query_contact c;
		c.query = k; c.body = j; c.sub_shape = (uint32_t)(4 * g + i);      // point index for the host's sort; the host then stores the compound child index here
		c.point[0] = m.p1[i].x; c.point[1] = m.p1[i].y; c.point[2] = m.p1[i].z;
		c.normal[0] = m.n.x; c.normal[1] = m.n.y; c.normal[2] = m.n.z;
		c.distance = v3_dot(v3_sub(m.p2[i], m.p1[i]), m.n);
		v3 pv = V3(0.0f, 0.0f, 0.0f);
		if (f_motion(f) != SGP_MOTION_STATIC) pv = v3_add(V3(d.vel[2 * (size_t)j]), v3_cross(V3(d.vel[2 * (size_t)j + 1]), v3_sub(m.p1[i], V3(d.pose[2 * (size_t)j]))));
		c.point_velocity[0] = pv.x; c.point_velocity[1] = pv.y; c.point_velocity[2] = pv.z;
		c.motion_type = f_motion(f); c.is_sensor = (f & BF_SENSOR) ? 1u : 0u; c.inv_mass = d.pose[2 * (size_t)j].w; c.userdata = 0;
		out[slot] = c;
	}
}

// One candidate body of a query, by one lane: the filters, then the collision test -- except for mesh bodies, which go on the wave's list (their
// triangles are the whole wave's work).
#define QUERY_MESH_LIST 32
SGP_DEV void capsule_query_body(const DV& d, const sgp_capsule_query& q, uint32_t k, const sgd_shape& sc, v3 lo, v3 hi, uint32_t j, sgp_query_contact* out, uint32_t cap, uint32_t* count, uint32_t* mesh_list, uint32_t* n_mesh)
{
	if (j == q.ignore_id) return;
	const uint32_t f = d.flags[j];
	if (!(f & BF_ALIVE) || (f & BF_ALIAS)) return;
	const uint32_t layer = f_layer(f);
	if (q.collidable_only && !(layer == SGP_LAYER_NON_MOVING || layer == SGP_LAYER_MOVING)) return;
	const float4 mn = d.aabb_min[j], mx = d.aabb_max[j];
	if (mx.x < lo.x || mn.x > hi.x || mx.y < lo.y || mn.y > hi.y || mx.z < lo.z || mn.z > hi.z) return;
	const sgd_shape sb = load_shape(d, j, f);
	sgd_manifold mm[SGD_MESH_MAX_GROUPS]; int ng; bool dropped = false;
	if (sb.type == SGP_SHAPE_MESH) {
		const uint32_t at = atomicAdd(n_mesh, 1u);
		if (at < QUERY_MESH_LIST) { mesh_list[at] = j; return; }
		ng = collide_with_mesh(d, j, sc, lo, hi, q.max_separation, mm, &dropped);      // (more meshes around one capsule than the list holds: this lane walks the rest)
	}
	else ng = (sb.type == SGP_SHAPE_HULL ? sgd_collide_hull(&sb, &sc, q.max_separation, &mm[0]) : sgd_collide(&sb, &sc, q.max_separation, &mm[0])) ? 1 : 0;   // normal: body -> capsule
	for (int g = 0; g < ng; ++g) capsule_emit(d, k, j, f, g, mm[g], out, cap, count);
}

// ONE WAVE PER QUERY CAPSULE (the character controller asks for one or a few per update, and waits for the answer): the candidate bodies -- the
// large ones, and those of the broad-phase cells its bounds reach -- dealt to the 64 lanes; the mesh bodies among them (a player stands on one and
// next to others all the time) are then taken one after the other by the whole wave, 64 candidate triangles per round (mesh_pair_groups).
__global__ void __launch_bounds__(64) k_collide_capsules(DV d, const sgp_capsule_query* qs, uint32_t n, sgp_query_contact* out, uint32_t cap, uint32_t* count)
{
	__shared__ MeshPairLds<64> L;
	__shared__ uint32_t mesh_list[QUERY_MESH_LIST];
	__shared__ uint32_t n_mesh;
	const uint32_t k = blockIdx.x;
	if (k >= n) return;
	const uint32_t lane = threadIdx.x;
	const sgp_capsule_query q = qs[k];
	sgd_shape sc;
	sc.pos = V3(q.pos[0], q.pos[1], q.pos[2]);
	quat qq; qq.x = q.rot[0]; qq.y = q.rot[1]; qq.z = q.rot[2]; qq.w = q.rot[3];
	sc.R = quat_to_m33(qq); sc.type = SGP_SHAPE_CAPSULE; sc.p0 = q.radius; sc.p1 = q.half_height; sc.p2 = 0.0f; sc.hull = nullptr;
	const v3 ax = v3_scale(sc.R.c2, q.half_height);
	const float e = q.radius + q.max_separation;
	const v3 ext = V3(fabsf(ax.x) + e, fabsf(ax.y) + e, fabsf(ax.z) + e);
	const v3 lo = v3_sub(sc.pos, ext), hi = v3_add(sc.pos, ext);
	if (lane == 0) n_mesh = 0;
	__syncthreads();
	for (uint32_t l = lane; l < d.sp->n_large; l += 64) capsule_query_body(d, q, k, sc, lo, hi, d.large_ids[l], out, cap, count, mesh_list, &n_mesh);
	{
		uint32_t seen = 0;      // static large bodies around the capsule, dealt to the lanes in the order the grid yields them
		large_grid_query(d, lo, hi, [&](uint32_t i) { if ((seen++ & 63u) == lane) capsule_query_body(d, q, k, sc, lo, hi, i, out, cap, count, mesh_list, &n_mesh); });
	}
	const BpGrid g = *d.grid;
	if (g.n_cells > 0 && g.min_x <= g.max_x) {
		const int x0 = max((int)floorf((lo.x - g.ox) * g.inv_cell) - 1, 0), x1 = min((int)floorf((hi.x - g.ox) * g.inv_cell) + 1, g.nx - 1);
		const int y0 = max((int)floorf((lo.y - g.oy) * g.inv_cell) - 1, 0), y1 = min((int)floorf((hi.y - g.oy) * g.inv_cell) + 1, g.ny - 1);
		const int z0 = max((int)floorf((lo.z - g.oz) * g.inv_cell) - 1, 0), z1 = min((int)floorf((hi.z - g.oz) * g.inv_cell) + 1, g.nz - 1);
		if (x0 <= x1) for (int z = z0; z <= z1; ++z) for (int y = y0; y <= y1; ++y) {
			grid_row_runs(d, g, x0, x1, y, z, [&](uint32_t c0, uint32_t c1) { for (uint32_t c = c0 + lane; c < c1; c += 64) capsule_query_body(d, q, k, sc, lo, hi, __float_as_uint(d.sorted_max[c].w), out, cap, count, mesh_list, &n_mesh); });
		}
	}
	__syncthreads();
	const uint32_t nm = min(n_mesh, (uint32_t)QUERY_MESH_LIST);
	const v3 es = V3(q.max_separation, q.max_separation, q.max_separation);
	for (uint32_t mi = 0; mi < nm; ++mi) {
		const uint32_t mid = mesh_list[mi];
		bool valid = true, dropped = false;
		sgd_shape X = sc;
		mesh_pair_groups<64, 4>(d, L, valid, X, mid, v3_sub(lo, es), v3_add(hi, es), q.max_separation, 0, (int)lane, 0u, dropped, V3(q.movement[0], q.movement[1], q.movement[2]), q.active_edges != 0u);      // (CharacterVirtual::GetContactsAtPosition: CollideOnlyWithActive + its direction of travel; 0: every edge with its own normal)
		if ((int)lane < L.mc.ng) {
			const sgd_mesh_group& grp = L.mc.g[lane];
			sgd_manifold mm;
			sgd_hull_reduce(grp.n, grp.p_mesh, grp.p_body, grp.np, &mm);
			capsule_emit(d, k, mid, d.flags[mid], (int)lane, mm, out, cap, count);
		}
		__syncthreads();
	}
}

SGP_DEV void spherecast_body(const DV& d, const sgp_ray& ry, float rs, v3 o, v3 dir, uint32_t j, RayBest& best)
{
	if (j == ry.ignore_id) return;
	const uint32_t f = d.flags[j];
	if (!(f & BF_ALIVE) || (f & (BF_SENSOR | BF_ALIAS))) return;
	const uint32_t layer = f_layer(f);
	if (ry.collidable_only && !(layer == SGP_LAYER_NON_MOVING || layer == SGP_LAYER_MOVING)) return;
	const float4 mn = d.aabb_min[j], mx = d.aabb_max[j];
	const float e = rs + 1.0e-3f;
	if (!ray_aabb(o, dir, make_float4(mn.x - e, mn.y - e, mn.z - e, 0.0f), make_float4(mx.x + e, mx.y + e, mx.z + e, 0.0f), ry.max_t)) return;      // full length: see veh_cast_test
	const float4 sh = d.prop[2 * (size_t)j + 1];
	const float prm[3] = { sh.x, sh.y, sh.z };
	v3 n, p;
	const float t = f_shape(f) == SGP_SHAPE_MESH ? cast_sphere_mesh(d, j, o, dir, best.t, rs, &n, &p)
	              : sgd_cast_sphere_body((int)f_shape(f), prm, f_shape(f) == SGP_SHAPE_HULL ? body_hull(d, sh) : nullptr, V3(d.pose[2 * (size_t)j]), quat_to_m33(Q4(d.pose[2 * (size_t)j + 1])), o, dir, best.t, rs, &n, &p);
	if (t >= 0.0f && t <= best.t && (t < best.t || best.id == SGP_INVALID_ID || j < best.id)) { best.t = t; best.id = j; best.n = n; }
}

// one thread per cast; cells under the swept sphere's bounds (casts are short: a character's step)
__global__ void __launch_bounds__(64) k_spherecast(DV d, const sgp_ray* rays, const float* radii, uint32_t n, sgp_hit* hits)
{
	const uint32_t k = blockIdx.x * 64 + threadIdx.x;
	if (k >= n) return;
	const sgp_ray ry = rays[k];
	const float rs = radii[k];
	const v3 o = V3(ry.origin[0], ry.origin[1], ry.origin[2]), dir = V3(ry.dir[0], ry.dir[1], ry.dir[2]);
	RayBest best; best.t = ry.max_t; best.id = SGP_INVALID_ID; best.n = V3(0.0f, 0.0f, 0.0f);
	for (uint32_t l = 0; l < d.sp->n_large; ++l) spherecast_body(d, ry, rs, o, dir, d.large_ids[l], best);
	{
		// the static large bodies under the swept sphere's bounds (casts are short)
		const v3 e = v3_add(o, v3_scale(dir, ry.max_t));
		const float m = rs + 2.0e-3f;
		large_grid_query(d, V3(fminf(o.x, e.x) - m, fminf(o.y, e.y) - m, fminf(o.z, e.z) - m), V3(fmaxf(o.x, e.x) + m, fmaxf(o.y, e.y) + m, fmaxf(o.z, e.z) + m),
		                 [&](uint32_t i) { spherecast_body(d, ry, rs, o, dir, i, best); });
	}
	const BpGrid g = *d.grid;
	if (g.n_cells > 0 && g.min_x <= g.max_x) {
		const v3 e = v3_add(o, v3_scale(dir, ry.max_t));
		const float m = rs + 1.0e-3f;
		const int x0 = max((int)floorf((fminf(o.x, e.x) - m - g.ox) * g.inv_cell) - 1, 0), x1 = min((int)floorf((fmaxf(o.x, e.x) + m - g.ox) * g.inv_cell) + 1, g.nx - 1);
		const int y0 = max((int)floorf((fminf(o.y, e.y) - m - g.oy) * g.inv_cell) - 1, 0), y1 = min((int)floorf((fmaxf(o.y, e.y) + m - g.oy) * g.inv_cell) + 1, g.ny - 1);
		const int z0 = max((int)floorf((fminf(o.z, e.z) - m - g.oz) * g.inv_cell) - 1, 0), z1 = min((int)floorf((fmaxf(o.z, e.z) + m - g.oz) * g.inv_cell) + 1, g.nz - 1);
		if (x0 <= x1) for (int z = z0; z <= z1; ++z) for (int y = y0; y <= y1; ++y) {
			grid_row_runs(d, g, x0, x1, y, z, [&](uint32_t c0, uint32_t c1) { for (uint32_t c = c0; c < c1; ++c) spherecast_body(d, ry, rs, o, dir, __float_as_uint(d.sorted_max[c].w), best); });
		}
	}
	sgp_hit h;
	h.id = best.id; h.t = best.id == SGP_INVALID_ID ? 0.0f : best.t;
	h.normal[0] = best.n.x; h.normal[1] = best.n.y; h.normal[2] = best.n.z;
	h.triangle = SGP_INVALID_ID; h.material = 0; h.bary[0] = h.bary[1] = 0.0f; h.sub_shape = 0;
	h.userdata = 0;
	hits[k] = h;
}

// multi-GPU tiles: bodies owned by this tile whose inflated AABB pokes outside [lo,hi)
// Export in ASCENDING BODY ID without a sort: pass 1 counts the qualifying bodies of every 256-body block, pass 2 gives each block the sum
// of the counts before it and each qualifying thread its rank inside the block (wave ballots), so record k of the output is the k-th
// qualifying body.  (The exchange wants a deterministic order; the host used to sort a few thousand 96-byte records every step.)
SGP_DEV bool export_qualifies(const DV& d, uint32_t i, float3 lo, float3 hi, float margin, uint32_t& f_out)
{
	if (i >= d.sp->n_slots) return false;
	const uint32_t f = d.flags[i];
	f_out = f;
	if (!(f & BF_ALIVE) || (f & (BF_GHOST | BF_LARGE)) || f_motion(f) == SGP_MOTION_STATIC) return false;
	const float4 mn = d.aabb_min[i], mx = d.aabb_max[i];
	return mn.x - margin < lo.x || mn.y - margin < lo.y || mn.z - margin < lo.z ||
	       mx.x + margin >= hi.x || mx.y + margin >= hi.y || mx.z + margin >= hi.z;
}

// the part of the record only an ownership migration reads: user data, layer + flags, damping, gravity factor
SGP_DEV void fill_ghost_desc(const DV& d, uint32_t i, uint32_t f, sgp_ghost_record& r)
{
	r.userdata = d.userdata[i];
	const float4 dy = d.dyn[i];
	r.gravity_factor = dy.z; r.linear_damping = dy.x; r.angular_damping = dy.y;
	r.flags = f_layer(f) | ((f & BF_SENSOR) ? SGP_GHOST_FLAG_SENSOR : 0u) | ((f & BF_ALLOW_SLEEP) ? SGP_GHOST_FLAG_ALLOW_SLEEP : 0u) | ((f & BF_ZERO_LIN_DRAG) ? SGP_GHOST_FLAG_ZERO_DRAG : 0u) | ((f & BF_CHASSIS) ? SGP_GHOST_FLAG_CHASSIS : 0u);
	r._pad[0] = 0; r._pad[1] = 0;
}

__global__ void __launch_bounds__(TPB) k_export_count(DV d, float3 lo, float3 hi, float margin)
{
	__shared__ uint32_t wsum[TPB / 64];
	uint32_t f;
	const bool q = export_qualifies(d, blockIdx.x * TPB + threadIdx.x, lo, hi, margin, f);
	const unsigned long long m = __ballot(q);
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < TPB / 64; ++k) t += wsum[k]; d.export_counts[blockIdx.x] = t; }
}

__global__ void __launch_bounds__(TPB) k_export_boundary(DV d, float3 lo, float3 hi, float margin, sgp_ghost_record* out, uint32_t cap, uint32_t* count)
{
	__shared__ uint32_t part[TPB];
	__shared__ uint32_t wsum[TPB / 64];
	// blocks before this one
	uint32_t acc = 0;
	for (uint32_t b = threadIdx.x; b < blockIdx.x; b += TPB) acc += d.export_counts[b];
	part[threadIdx.x] = acc;
	__syncthreads();
	for (int off = TPB / 2; off > 0; off >>= 1) { if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off]; __syncthreads(); }
	const uint32_t base = part[0];
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	uint32_t f = 0;
	const bool q = export_qualifies(d, i, lo, hi, margin, f);
	const unsigned long long m = __ballot(q);
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (lane == 0) wsum[wv] = (uint32_t)__popcll(m);
	__syncthreads();
	uint32_t wbase = 0, total = 0;
	for (int k = 0; k < TPB / 64; ++k) { if (k < wv) wbase += wsum[k]; total += wsum[k]; }
	if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *count = base + total;      // the last block knows the grand total
	if (!q) return;
	const uint32_t k = base + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
	if (k >= cap) return;
	sgp_ghost_record r;
	const float4 p = d.pose[2 * (size_t)i], qq = d.pose[2 * (size_t)i + 1], lv = d.vel[2 * (size_t)i], av = d.vel[2 * (size_t)i + 1], sh = d.prop[2 * (size_t)i + 1];
	r.pos[0] = p.x; r.pos[1] = p.y; r.pos[2] = p.z;
	r.rot[0] = qq.x; r.rot[1] = qq.y; r.rot[2] = qq.z; r.rot[3] = qq.w;
	r.lin_vel[0] = lv.x; r.lin_vel[1] = lv.y; r.lin_vel[2] = lv.z;
	r.ang_vel[0] = av.x; r.ang_vel[1] = av.y; r.ang_vel[2] = av.z;
	r.shape_type = (int32_t)f_shape(f);
	r.shape[0] = sh.x; r.shape[1] = sh.y; r.shape[2] = sh.z; r.shape[3] = 0.0f;
	r.mass = d.torque[i].w; r.friction = sh.w; r.restitution = d.prop[2 * (size_t)i].w;
	r.motion_type = f_motion(f);
	r.global_id = i;
	fill_ghost_desc(d, i, f, r);
	out[k] = r;
}

// ---- export with the routing done on the device: one record per (qualifying body, destination tile) --------------------------------
// Same rules as the host statement (sgp_tiles_route): a qualifying body goes to every OTHER tile whose region grown by `pad` contains its
// centre; an owned dynamic body whose centre has left this tile's region emigrates (flagged, listed).  The send buffer is segmented by
// destination (rank order) and ascending in body id inside a segment: counts per (block, destination) -> scan -> write, no sort, no atomics.
SGP_DEV bool tile_in_box(float4 p, const float* lo, const float* hi, float pad)
{
	return p.x >= lo[0] - pad && p.x < hi[0] + pad && p.y >= lo[1] - pad && p.y < hi[1] + pad && p.z >= lo[2] - pad && p.z < hi[2] + pad;
}
SGP_DEV unsigned long long route_mask(const DV& d, uint32_t i, const TileRoute& t, bool& emigrates, uint32_t& f)
{
	emigrates = false;
	const float* mylo = t.boxes + 6 * t.my_rank; const float* myhi = mylo + 3;
	if (!export_qualifies(d, i, make_float3(mylo[0], mylo[1], mylo[2]), make_float3(myhi[0], myhi[1], myhi[2]), t.margin, f)) return 0ull;
	const float4 p = d.pose[2 * (size_t)i];
	// an owned dynamic body emigrates only when another tile's own (unpadded) region contains its centre: where the caller's boxes leave a gap
	// nobody would accept the body, so it stays with its current owner instead of vanishing
	// (a vehicle's chassis stays with the tile that holds the vehicle record: SGP_GHOST_FLAG_CHASSIS)
	const bool left = t.n_tiles > 1 && f_motion(f) == SGP_MOTION_DYNAMIC && !(f & BF_CHASSIS) && !tile_in_box(p, mylo, myhi, 0.0f);
	bool taker = false;
	unsigned long long m = 0ull;
	for (uint32_t r = 0; r < t.n_tiles; ++r) {
		if (r == t.my_rank) continue;
		const float* lo = t.boxes + 6 * r;
		if (tile_in_box(p, lo, lo + 3, t.pad)) m |= 1ull << r;
		if (left && tile_in_box(p, lo, lo + 3, 0.0f)) taker = true;
	}
	emigrates = left && taker;
	return m;
}

// Where the owned bodies are, for sgp_tiles_rebalance (a few times a second at most: plain global atomics)
__global__ void __launch_bounds__(TPB) k_tiles_hist(DV d, TilePlanes tp, int level, uint32_t* out)
{
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= d.sp->n_slots) return;
	const uint32_t f = d.flags[i];
	if (!(f & BF_ALIVE) || (f & BF_ALIAS) || f_motion(f) != SGP_MOTION_DYNAMIC) return;      // (ghosts are kinematic here: owned bodies only)
	const float4 p = d.pose[2 * (size_t)i];
	if (level == 0) {
		int* o = (int*)out;
		atomicMin(&o[0], float_to_ordered(p.x)); atomicMin(&o[1], float_to_ordered(p.y)); atomicMin(&o[2], float_to_ordered(p.z));
		atomicMax(&o[3], float_to_ordered(p.x)); atomicMax(&o[4], float_to_ordered(p.y)); atomicMax(&o[5], float_to_ordered(p.z));
		return;
	}
	uint32_t ix = 0, iy = 0;
	for (uint32_t k = 0; k + 1 < tp.gx; ++k) if (p.x >= tp.xp[k]) ix = k + 1;
	if (level == 3) for (uint32_t k = 0; k + 1 < tp.gy; ++k) if (p.y >= tp.yp[4 * ix + k]) iy = k + 1;
	const int a = level - 1;
	const float c = a == 0 ? p.x : (a == 1 ? p.y : p.z);
	const float w = tp.ghi[a] - tp.glo[a];
	int bin = w > 0.0f ? (int)floorf((c - tp.glo[a]) / w * (float)SGP_TILE_HIST_BINS) : 0;
	bin = min(max(bin, 0), SGP_TILE_HIST_BINS - 1);
	const uint32_t group = level == 1 ? 0u : (level == 2 ? ix : ix + tp.gx * iy);
	// what a body weighs: 1, or (by_contacts) 1 + the contact constraints it was in last step (the colours in its mask): the work of a tile is its
	// constraints more than its bodies, and a pile has them at the bottom
	const uint32_t wgt = tp.by_contacts ? 1u + (uint32_t)__popcll(d.colour_mask[i]) : 1u;
	atomicAdd(&out[group * SGP_TILE_HIST_BINS + (uint32_t)bin], wgt);
}

__global__ void __launch_bounds__(TPB) k_route_count(DV d, TileRoute t, uint32_t* block_counts)
{
	__shared__ uint32_t cnt[SGP_MAX_TILES + 1];
	if (threadIdx.x <= SGP_MAX_TILES) cnt[threadIdx.x] = 0;
	__syncthreads();
	bool emig; uint32_t f;
	const unsigned long long mask = route_mask(d, blockIdx.x * TPB + threadIdx.x, t, emig, f);
	const int lane = threadIdx.x & 63;
	for (uint32_t r = 0; r < t.n_tiles; ++r) { const unsigned long long b = __ballot((mask >> r) & 1ull); if (b && lane == 0) atomicAdd(&cnt[r], (uint32_t)__popcll(b)); }
	{ const unsigned long long b = __ballot(emig); if (b && lane == 0) atomicAdd(&cnt[t.n_tiles], (uint32_t)__popcll(b)); }
	__syncthreads();
	if (threadIdx.x <= t.n_tiles) block_counts[(size_t)blockIdx.x * (t.n_tiles + 1) + threadIdx.x] = cnt[threadIdx.x];
}

// one workgroup: per column (destination, or emigrants) the exclusive scan of the block counts; then the segment starts
__global__ void __launch_bounds__(1024) k_route_scan(const uint32_t* block_counts, uint32_t* block_offsets, uint32_t n_blocks, uint32_t n_tiles, RouteHeader* header)
{
	__shared__ uint32_t wave_sums[16];
	__shared__ uint32_t carry;
	__shared__ uint32_t totals[SGP_MAX_TILES + 1];
	const uint32_t cols = n_tiles + 1;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (uint32_t c = 0; c < cols; ++c) {
		if (threadIdx.x == 0) carry = 0;
		__syncthreads();
		for (uint32_t start = 0; start < n_blocks; start += 1024) {
			const uint32_t b = start + threadIdx.x;
			const uint32_t v = b < n_blocks ? block_counts[(size_t)b * cols + c] : 0u;
			uint32_t x = v;
			for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
			if (lane == 63) wave_sums[wave] = x;
			__syncthreads();
			uint32_t wbase = carry;
			for (int k = 0; k < wave; ++k) wbase += wave_sums[k];
			if (b < n_blocks) block_offsets[(size_t)b * cols + c] = wbase + x - v;
			__syncthreads();
			if (threadIdx.x == 1023) carry = wbase + x;
			__syncthreads();
		}
		if (threadIdx.x == 0) totals[c] = carry;
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		uint32_t acc = 0;
		for (uint32_t r = 0; r < SGP_MAX_TILES; ++r) { const uint32_t n = r < n_tiles ? totals[r] : 0u; header->seg_count[r] = n; header->seg_start[r] = acc; acc += n; }
		header->total = acc; header->n_emigrants = totals[n_tiles]; header->pad[0] = header->pad[1] = 0;
	}
}

__global__ void __launch_bounds__(TPB) k_route_write(DV d, TileRoute t, const uint32_t* block_offsets, const RouteHeader* header, sgp_ghost_record* out, uint32_t cap,
                                                     uint32_t* emigrant_ids, uint32_t emigrant_cap)
{
	__shared__ uint32_t wcnt[TPB / 64][SGP_MAX_TILES + 1];
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	bool emig; uint32_t f = 0;
	const unsigned long long mask = route_mask(d, i, t, emig, f);
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const unsigned long long below = (1ull << lane) - 1ull;
	// per wave and column: how many of this wave's lanes write to it
	for (uint32_t r = 0; r <= t.n_tiles; ++r) {
		const unsigned long long b = __ballot(r < t.n_tiles ? ((mask >> r) & 1ull) != 0ull : emig);
		if (lane == 0) wcnt[wv][r] = (uint32_t)__popcll(b);
	}
	__syncthreads();
	if (!mask && !emig) return;
	const uint32_t cols = t.n_tiles + 1;
	sgp_ghost_record r;
	bool built = false;
	for (uint32_t dst = 0; dst < t.n_tiles; ++dst) {
		const unsigned long long b = __ballot(((mask >> dst) & 1ull) != 0ull);      // (every lane that reached this point takes part: the loop bounds are wave-uniform)
		if (!((mask >> dst) & 1ull)) continue;
		uint32_t wbase = 0;
		for (int k = 0; k < wv; ++k) wbase += wcnt[k][dst];
		const uint32_t k = header->seg_start[dst] + block_offsets[(size_t)blockIdx.x * cols + dst] + wbase + (uint32_t)__popcll(b & below);
		if (k >= cap) continue;
		if (!built) {
			const float4 p = d.pose[2 * (size_t)i], qq = d.pose[2 * (size_t)i + 1], lv = d.vel[2 * (size_t)i], av = d.vel[2 * (size_t)i + 1], sh = d.prop[2 * (size_t)i + 1];
			r.pos[0] = p.x; r.pos[1] = p.y; r.pos[2] = p.z;
			r.rot[0] = qq.x; r.rot[1] = qq.y; r.rot[2] = qq.z; r.rot[3] = qq.w;
			r.lin_vel[0] = lv.x; r.lin_vel[1] = lv.y; r.lin_vel[2] = lv.z;
			r.ang_vel[0] = av.x; r.ang_vel[1] = av.y; r.ang_vel[2] = av.z;
			r.shape_type = (int32_t)f_shape(f);
			r.shape[0] = sh.x; r.shape[1] = sh.y; r.shape[2] = sh.z; r.shape[3] = 0.0f;
			r.mass = d.torque[i].w; r.friction = sh.w; r.restitution = d.prop[2 * (size_t)i].w;
			r.motion_type = emig ? (SGP_MOTION_DYNAMIC | SGP_GHOST_TAKE_OWNERSHIP) : f_motion(f);
			r.global_id = (uint64_t)i | ((uint64_t)t.my_rank << 40);
			fill_ghost_desc(d, i, f, r);
			built = true;
		}
		out[k] = r;
	}
	if (emig) {
		uint32_t wbase = 0;
		for (int k = 0; k < wv; ++k) wbase += wcnt[k][t.n_tiles];
		const unsigned long long b = __ballot(emig);
		const uint32_t k = block_offsets[(size_t)blockIdx.x * cols + t.n_tiles] + wbase + (uint32_t)__popcll(b & below);
		if (k < emigrant_cap) emigrant_ids[k] = i;
	}
}

__global__ void __launch_bounds__(TPB) k_ghost_refresh_records(DV d, const sgp_ghost_record* recs, const uint32_t* ids, uint32_t n)
{
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= n) return;
	const uint32_t i = ids[k];
	if (i == SGP_INVALID_ID) return;      // (a record that is no ghost here: an immigrant of the same exchange, a rejected one)
	uint32_t f = d.flags[i];
	if (!(f & BF_ALIVE)) return;
	const sgp_ghost_record& c = recs[k];
	d.pose[2 * (size_t)i] = make_float4(c.pos[0], c.pos[1], c.pos[2], d.pose[2 * (size_t)i].w);
	d.pose[2 * (size_t)i + 1] = make_float4(c.rot[0], c.rot[1], c.rot[2], c.rot[3]);
	if (f_motion(f) != SGP_MOTION_STATIC) {
		d.vel[2 * (size_t)i] = make_float4(c.lin_vel[0], c.lin_vel[1], c.lin_vel[2], d.vel[2 * (size_t)i].w);
		d.vel[2 * (size_t)i + 1] = make_float4(c.ang_vel[0], c.ang_vel[1], c.ang_vel[2], d.vel[2 * (size_t)i + 1].w);
	}
	refresh_aabb(d, i, f);
	f = activate_body(d, i, f);
	d.flags[i] = f;
}

#ifdef SGP_EXPERIMENTS
#define SGP_TILE_SOLVER_INCLUDED 1
#endif
// What the host needs of a received record to decide whether the ghost set changed: its global id and whether it asks for a change of ownership
// (16 B instead of the 128 B record).
__global__ void __launch_bounds__(TPB) k_pack_ghost_keys(const sgp_ghost_record* recs, uint32_t n, uint4* out)
{
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= n) return;
	const uint64_t g = recs[k].global_id;
	out[k] = make_uint4((uint32_t)g, (uint32_t)(g >> 32), recs[k].motion_type, 0u);
}

// ---------------------------------------------------------------------------------------------------------------
// launch wrappers

static inline uint32_t blocks_for(uint32_t n) { return n ? (n + TPB - 1) / TPB : 1; }
static inline uint32_t stride_grid(uint32_t estimate) { uint32_t b = blocks_for(estimate); if (b < 64) b = 64; if (b > 4096) b = 4096; return (b + 7u) & ~7u; }      // (a multiple of eight: xcd_block)
#ifdef SGP_EXPERIMENTS
#include "experiments/sgp_tile_solver.inc"
#else
// (the resident tile solver of round 3 is an experiment: built only with -DSGP_EXPERIMENTS; a plain build never plans it, sgp_world.hip)
void launch_ts_label(const DV&, uint32_t, hipStream_t) {}
void launch_colour_count_ts(const DV&, uint32_t, hipStream_t) {}
void launch_setup_ts(const DV&, uint32_t, hipStream_t) {}
void launch_ts_solve(const DV&, int, int, hipStream_t) {}
#endif

void launch_step_begin(const DV& d, const StepParams& sp, uint32_t nb, bool reset_step_scratch, hipStream_t s)
{
	const uint32_t work = std::max(d.table_size + 4, reset_step_scratch ? nb : 0u);
	uint32_t blocks = (work + TPB * 4 - 1) / (TPB * 4);
	if (blocks < 1) blocks = 1; if (blocks > 1024) blocks = 1024;
	hipLaunchKernelGGL(k_step_begin, dim3(blocks), dim3(TPB), 0, s, d, sp, nb, reset_step_scratch ? 1 : 0, std::min(blocks, std::min(blocks_for(nb), 128u)));
}
void launch_set_params(const DV& d, const StepParams& sp, hipStream_t s) { hipLaunchKernelGGL(k_set_params, dim3(1), dim3(64), 0, s, d, sp); }
void launch_step_end(const DV& d, StepCounters* host_mapped, EventCounters* host_events, hipStream_t s) { hipLaunchKernelGGL(k_step_end, dim3(1), dim3(TPB), 0, s, d, host_mapped, host_events); }
void launch_fill_u64(uint64_t* p, uint64_t v, size_t n, hipStream_t s)
{
	size_t blocks = (n + TPB * 8 - 1) / (TPB * 8);
	if (blocks < 1) blocks = 1; if (blocks > 2048) blocks = 2048;
	hipLaunchKernelGGL(k_fill_u64, dim3((uint32_t)blocks), dim3(TPB), 0, s, p, v, n);
}
void launch_pre_solve(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_pre_solve, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_bp_bounds(const DV&, uint32_t, hipStream_t) {}      // (round 4: inside launch_step_begin; the grid parameters inside launch_bp_cell)
void launch_bp_cell(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_bp_cell, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_bp_scan(const DV& d, hipStream_t s)
{
	const uint32_t n = d.table_size + 1;
	const uint32_t nb = (n + 1023) / 1024;
	const uint32_t* n_tiles = &d.ctr->n_tiles_used;
	const uint32_t grid = std::min(nb, 1024u);
	hipLaunchKernelGGL(k_scan_blocks, dim3(grid), dim3(TPB), 0, s, d.cell_count, d.cell_start, d.scan_block_sums, n, n_tiles);
	hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, s, d.scan_block_sums, nb, n, n_tiles, d.grid_cells_used);
	hipLaunchKernelGGL(k_scan_add, dim3(grid), dim3(TPB), 0, s, d.cell_start, d.scan_block_sums, n, n_tiles);
}
void launch_bp_scatter(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_bp_scatter, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_bp_pairs(const DV& d, int small_lds, hipStream_t s)
{
	if (small_lds) hipLaunchKernelGGL((k_bp_pairs<BP_LDS_CAP_SMALL, BP_PAIR_CAP_SMALL>), dim3(4096), dim3(TPB), 0, s, d);
	else hipLaunchKernelGGL((k_bp_pairs<BP_LDS_CAP_LARGE, BP_PAIR_CAP_LARGE>), dim3(4096), dim3(TPB), 0, s, d);
}      // (fewer workgroups walking several tiles each were slower: 512 -> 159 us against 132 us, the tiles are uneven)
void launch_bp_large(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_bp_large, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_bp_scatter_large(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_bp_scatter_large, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_narrowphase(const DV& d, uint32_t est, hipStream_t s)
{
	hipLaunchKernelGGL(k_narrowphase, dim3(stride_grid(est)), dim3(TPB), 0, s, d);
}
void launch_wake_round(const DV& d, uint32_t nb, bool has_hulls, bool has_meshes, hipStream_t s)
{
	hipLaunchKernelGGL(k_wake_pairs, dim3(blocks_for(nb)), dim3(TPB), 0, s, d);
	if (has_hulls) hipLaunchKernelGGL(k_narrowphase_wake<true>, dim3(32), dim3(TPB), 0, s, d);      // (few pairs, 1.7 KB of scratch per lane: a small grid starts faster)
	else hipLaunchKernelGGL(k_narrowphase_wake<false>, dim3(32), dim3(TPB), 0, s, d);
	// (hull pairs of this round are collided by k_narrowphase_wake itself; hull - mesh pairs by the hull instances of the mesh kernels)
	if (has_meshes) {
		hipLaunchKernelGGL((k_narrowphase_mesh<8, SGD_KINDS_PRIMITIVES>), dim3(256), dim3(64), 0, s, d);
		if (has_hulls) hipLaunchKernelGGL((k_narrowphase_mesh<8, 8>), dim3(256), dim3(64), 0, s, d);
		hipLaunchKernelGGL((k_narrowphase_mesh<64, SGD_KINDS_PRIMITIVES>), dim3(256), dim3(64), 0, s, d);
		if (has_hulls) hipLaunchKernelGGL((k_narrowphase_mesh<64, 8>), dim3(256), dim3(64), 0, s, d);
	}
}
void launch_narrowphase_hull(const DV& d, hipStream_t s)
{
	hipLaunchKernelGGL(k_narrowphase_hull, dim3(4096), dim3(64), 0, s, d);
	hipLaunchKernelGGL(k_narrowphase_hull_manifold, dim3(1024), dim3(64), 0, s, d);
}
void launch_narrowphase_mesh(const DV& d, bool has_hulls, hipStream_t s)
{
	// eight lanes per pair: the primitives, then (worlds with hulls) the hulls; both pass the pairs with many candidate triangles on to ...
	hipLaunchKernelGGL((k_narrowphase_mesh<8, SGD_KINDS_PRIMITIVES>), dim3(2048), dim3(64), 0, s, d);
	if (has_hulls) hipLaunchKernelGGL((k_narrowphase_mesh<8, 8>), dim3(2048), dim3(64), 0, s, d);
	// ... a wave per pair
	hipLaunchKernelGGL((k_narrowphase_mesh<64, SGD_KINDS_PRIMITIVES>), dim3(2048), dim3(64), 0, s, d);
	if (has_hulls) hipLaunchKernelGGL((k_narrowphase_mesh<64, 8>), dim3(2048), dim3(64), 0, s, d);
}
void launch_colour_inherit(const DV& d, uint32_t est, hipStream_t s) { hipLaunchKernelGGL(k_colour_inherit, dim3(stride_grid(est)), dim3(TPB), 0, s, d); }
void launch_colour_claim(const DV& d, uint32_t est, uint32_t round, hipStream_t s)
{
	hipLaunchKernelGGL(k_colour_claim, dim3(stride_grid(est)), dim3(TPB), 0, s, d, round);
}
void launch_colour_commit(const DV& d, uint32_t est, uint32_t round, hipStream_t s) { hipLaunchKernelGGL(k_colour_commit, dim3(stride_grid(est)), dim3(TPB), 0, s, d, round); }
void launch_colour_count(const DV& d, uint32_t est, hipStream_t s)
{
	// (few workgroups, each looping: a workgroup ends with one global atomic per colour it saw, and atomics on one address serialise)
	// few, looping workgroups: every workgroup ends with one global atomic per colour, and those queue per colour (config 3: 512 workgroups 21 us,
	// 256: 13 us, 128: 11 us, 64: 15 us); more of them only where there is enough to count (a million bodies)
	hipLaunchKernelGGL(k_colour_count, dim3(std::min(std::max(stride_grid(est) / 8u, 128u), 512u)), dim3(TPB), 0, s, d, 1);
}
void launch_colour_finish(const DV& d, uint32_t first_round, int build_list, hipStream_t s) { hipLaunchKernelGGL(k_colour_finish, dim3(1), dim3(1024), 0, s, d, first_round, build_list); }
void launch_setup(const DV& d, uint32_t n_man, hipStream_t s)
{
	hipLaunchKernelGGL(k_setup_slots, dim3(std::max(64u, std::min(4096u, (n_man + TPB * SLOTS_PER_THREAD - 1) / (TPB * SLOTS_PER_THREAD)))), dim3(TPB), 0, s, d);
	hipLaunchKernelGGL(k_setup, dim3(stride_grid(n_man)), dim3(TPB), 0, s, d);
}
void launch_solve_colour(const DV& d, int colour, uint32_t est, int mode, hipStream_t s, int compact_rows)
{
	uint32_t blocks = (est + est / 8 + 64 + SOLVE_TPB - 1) / SOLVE_TPB;      // warm start: one thread per constraint, one wave per workgroup
	if (mode != 0) blocks = (est + est / 8 + 64 + SOLVE_VEL_TPB / 2 - 1) / (SOLVE_VEL_TPB / 2);
	if (blocks > 8192) blocks = 8192;
	if (mode != 0) blocks = (blocks + 7u) & ~7u;      // (XCD-contiguous chunks: k_solve_colour)
	if (mode != 0 && est >= 8192u && est <= 65536u) colour |= SOLVE_XCD_CHUNKS;      // (the colour's bodies and rows then fit the eight L2s)
	if (mode == 0) hipLaunchKernelGGL(k_solve_colour<0>, dim3(blocks), dim3(SOLVE_TPB), 0, s, d, colour);
	else if (mode == 1) {
		if (compact_rows == 2) hipLaunchKernelGGL((k_solve_colour<1, 2>), dim3(blocks), dim3(SOLVE_VEL_TPB), 0, s, d, colour);
		else if (compact_rows) hipLaunchKernelGGL((k_solve_colour<1, 1>), dim3(blocks), dim3(SOLVE_VEL_TPB), 0, s, d, colour);
		else hipLaunchKernelGGL((k_solve_colour<1, 0>), dim3(blocks), dim3(SOLVE_VEL_TPB), 0, s, d, colour);
	}
	else hipLaunchKernelGGL(k_solve_colour<2>, dim3(blocks), dim3(SOLVE_VEL_TPB), 0, s, d, colour);
}
void launch_solve_colour_veh(const DV& d, int colour, uint32_t est, int mode, hipStream_t s, int compact_rows)
{
	uint32_t blocks = (est + est / 8 + 64 + SOLVE_VEL_TPB / 2 - 1) / (SOLVE_VEL_TPB / 2);
	if (blocks > 8192) blocks = 8192;
	blocks = (blocks + 7u) & ~7u;
	if (est >= 8192u && est <= 65536u) colour |= SOLVE_XCD_CHUNKS;
	const uint32_t vb = (d.n_vehicles * 4u + SOLVE_VEL_TPB - 1) / SOLVE_VEL_TPB;
	if (mode == 1) {
		if (compact_rows == 2) hipLaunchKernelGGL((k_solve_colour_veh<1, 2>), dim3(vb + blocks), dim3(SOLVE_VEL_TPB), 0, s, d, colour, vb);
		else if (compact_rows) hipLaunchKernelGGL((k_solve_colour_veh<1, 1>), dim3(vb + blocks), dim3(SOLVE_VEL_TPB), 0, s, d, colour, vb);
		else hipLaunchKernelGGL((k_solve_colour_veh<1, 0>), dim3(vb + blocks), dim3(SOLVE_VEL_TPB), 0, s, d, colour, vb);
	}
	else hipLaunchKernelGGL((k_solve_colour_veh<2, -1>), dim3(vb + blocks), dim3(SOLVE_VEL_TPB), 0, s, d, colour, vb);
}
void launch_warm_bodies(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_warm_bodies, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_solve_tail(const DV& d, int first_colour, int mode, hipStream_t s, int compact_rows)
{
	if (mode == 1) {
		if (compact_rows == 2) hipLaunchKernelGGL(k_solve_tail_vel<2>, dim3(1), dim3(TAIL_VEL_TPB), 0, s, d, first_colour);
		else if (compact_rows) hipLaunchKernelGGL(k_solve_tail_vel<1>, dim3(1), dim3(TAIL_VEL_TPB), 0, s, d, first_colour);
		else hipLaunchKernelGGL(k_solve_tail_vel<0>, dim3(1), dim3(TAIL_VEL_TPB), 0, s, d, first_colour);
	}      // (position passes of the tail: one thread per constraint measured faster, 30 against 37 us)
	else hipLaunchKernelGGL(k_solve_tail, dim3(1), dim3(512), 0, s, d, first_colour, mode);
}
void launch_hc_build(const DV& d, int first_colour, uint32_t est, hipStream_t s)
{
	const uint32_t blocks = std::max(1u, std::min(1024u, (est + est / 8 + TPB - 1) / TPB));
	hipLaunchKernelGGL(k_hc_hook, dim3(blocks), dim3(TPB), 0, s, d, first_colour);
	hipLaunchKernelGGL(k_hc_count, dim3(blocks), dim3(TPB), 0, s, d, first_colour);
	hipLaunchKernelGGL(k_hc_alloc, dim3(blocks), dim3(TPB), 0, s, d, first_colour);
	hipLaunchKernelGGL(k_hc_scatter, dim3(blocks), dim3(TPB), 0, s, d, first_colour);
	hipLaunchKernelGGL(k_hc_sort, dim3(std::max(1u, std::min(2048u, (2u * est + HC_CLASSES * HC_WG_PAIRS) / HC_WG_PAIRS))), dim3(HC_WG_PAIRS), 0, s, d);
}
void launch_hc_probe(const DV& d, int probe_colour, uint32_t probe_est, hipStream_t s)
{
	const uint32_t pb = std::max(1u, std::min(1024u, (probe_est + probe_est / 8 + TPB - 1) / TPB));
	hipLaunchKernelGGL(k_hc_hook, dim3(pb), dim3(TPB), 0, s, d, probe_colour);
	hipLaunchKernelGGL(k_hc_count, dim3(pb), dim3(TPB), 0, s, d, probe_colour);
	hipLaunchKernelGGL(k_hc_probe, dim3(pb), dim3(TPB), 0, s, d, probe_colour);
	hipLaunchKernelGGL(k_hc_init, dim3(pb), dim3(TPB), 0, s, d, probe_colour);      // (every body the probe touched, i.e. also every body of the real build)
}
void launch_solve_hc(const DV& d, int first_colour, uint32_t est, int mode, hipStream_t s, int compact_rows)
{
	// list entries: a component of n constraints takes the next power of two (< 2 n), plus the padding of the classes
	const uint32_t blocks = std::max(1u, std::min(2048u, (2u * est + HC_CLASSES * HC_WG_PAIRS) / HC_WG_PAIRS));
	if (mode == 1) {
		if (compact_rows == 2) hipLaunchKernelGGL((k_solve_hc<1, 2>), dim3(blocks), dim3(HC_TPB), 0, s, d, first_colour);
		else if (compact_rows) hipLaunchKernelGGL((k_solve_hc<1, 1>), dim3(blocks), dim3(HC_TPB), 0, s, d, first_colour);
		else hipLaunchKernelGGL((k_solve_hc<1, 0>), dim3(blocks), dim3(HC_TPB), 0, s, d, first_colour);
	}
	else hipLaunchKernelGGL((k_solve_hc<2, -1>), dim3(blocks), dim3(HC_TPB), 0, s, d, first_colour);
}
void launch_solve_small(const DV& d, int warm_start, int iterations, int lane_pairs, hipStream_t s)
{
	if (lane_pairs) hipLaunchKernelGGL(k_solve_small, dim3(1), dim3(SMALL_TPB), 0, s, d, warm_start, iterations);
	else hipLaunchKernelGGL(k_solve_small_single, dim3(1), dim3(512), 0, s, d, warm_start, iterations);
}
void launch_integrate_pose(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_integrate_pose, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_finalize(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_finalize, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_island_mark(const DV& d, uint32_t n_con, int clear_cache, hipStream_t s) { hipLaunchKernelGGL(k_island_mark, dim3(stride_grid(n_con)), dim3(TPB), 0, s, d, clear_cache); }
void launch_island_hook(const DV& d, uint32_t n_con, hipStream_t s) { hipLaunchKernelGGL(k_island_hook, dim3(stride_grid(n_con)), dim3(TPB), 0, s, d); }
void launch_island_flag(const DV& d, uint32_t n_con, hipStream_t s) { hipLaunchKernelGGL(k_island_flag, dim3(stride_grid(n_con)), dim3(TPB), 0, s, d); }
void launch_sleep_apply(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_sleep_apply, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_buoyancy(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_buoyancy, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_cache_build(const DV& d, uint32_t n_con, StepCounters* host_mapped, EventCounters* host_events, hipStream_t s)
{
	// (the table was emptied by the first k_island_mark launch of the step)
	hipLaunchKernelGGL(k_cache_build, dim3(stride_grid(n_con)), dim3(TPB), 0, s, d, host_mapped, host_events);
}
// forget the previous step's contacts (the world has gone to sleep as a whole: the CPU statement's steps without an awake body leave no constraints behind either)
void launch_cache_wipe(const DV& d, hipStream_t s) { hipLaunchKernelGGL(k_cache_clear, dim3(64), dim3(TPB), 0, s, d); }
void launch_contact_events(const DV& d, uint32_t est, hipStream_t s) { hipLaunchKernelGGL(k_contact_events, dim3(stride_grid(est)), dim3(TPB), 0, s, d); }
void launch_ghost_refresh(const DV& d, const GhostRefresh* recs, uint32_t n, hipStream_t s) { if (n) hipLaunchKernelGGL(k_ghost_refresh, dim3(blocks_for(n)), dim3(TPB), 0, s, d, recs, n); }
void launch_apply_cmds(const DV& d, const BodyCmd* cmds, const uint32_t* run_start, uint32_t n_runs, hipStream_t s) { if (n_runs) hipLaunchKernelGGL(k_apply_cmds, dim3(blocks_for(n_runs)), dim3(TPB), 0, s, d, cmds, run_start, n_runs); }
void launch_gather_aabbs(const DV& d, const uint32_t* ids, uint32_t n, float4* out, hipStream_t s) { if (n) hipLaunchKernelGGL(k_gather_aabbs, dim3(blocks_for(n)), dim3(TPB), 0, s, d, ids, n, out); }
void launch_gather_states(const DV& d, const uint32_t* ids, uint32_t first, uint32_t n, sgp_body_state* out, hipStream_t s) { if (n) hipLaunchKernelGGL(k_gather_states, dim3(blocks_for(n)), dim3(TPB), 0, s, d, ids, first, n, out); }
void launch_gather_active_poses(const DV& d, uint32_t nb, void* out, uint32_t cap, hipStream_t s) { hipLaunchKernelGGL(k_gather_active_poses, dim3(blocks_for(nb)), dim3(TPB), 0, s, d, (float4*)out, cap); }
void launch_gather_active(const DV& d, uint32_t nb, sgp_body_state* out, uint32_t cap, hipStream_t s) { hipLaunchKernelGGL(k_gather_active, dim3(blocks_for(nb)), dim3(TPB), 0, s, d, out, cap); }
void launch_dump_constraints(const DV& d, uint32_t which, uint32_t n_con, void* out, uint32_t cap, hipStream_t s) { if (n_con) hipLaunchKernelGGL(k_dump_constraints, dim3(blocks_for(n_con)), dim3(TPB), 0, s, d, which, n_con, (ConstraintDumpRec*)out, cap); }
void launch_vehicle_pre(const DV& d, hipStream_t s)
{
	if (!d.n_vehicles) return;
	hipLaunchKernelGGL(k_vehicle_cast, dim3(d.n_vehicles), dim3(64), 0, s, d);
	hipLaunchKernelGGL(k_vehicle_controller, dim3(d.n_vehicles), dim3(64), 0, s, d);
}
void launch_vehicle_solve(const DV& d, int mode, hipStream_t s)
{
	const dim3 g((d.n_vehicles * 4u + VEH_SOLVE_TPB - 1) / VEH_SOLVE_TPB), b(VEH_SOLVE_TPB);
	if (mode == 0) hipLaunchKernelGGL(k_vehicle_solve<0>, g, b, 0, s, d);
	else if (mode == 1) hipLaunchKernelGGL(k_vehicle_solve<1>, g, b, 0, s, d);
	else hipLaunchKernelGGL(k_vehicle_solve<2>, g, b, 0, s, d);
}
void launch_raycast(const DV& d, const sgp_ray* rays, uint32_t n, sgp_hit* hits, hipStream_t s) { if (n) hipLaunchKernelGGL(k_raycast, dim3((n + 63) / 64), dim3(64), 0, s, d, rays, n, hits); }
void launch_collide_capsules(const DV& d, const sgp_capsule_query* q, uint32_t n, sgp_query_contact* out, uint32_t cap, uint32_t* count, hipStream_t s) { if (n) hipLaunchKernelGGL(k_collide_capsules, dim3(n), dim3(64), 0, s, d, q, n, out, cap, count); }      // a wave per query
void launch_spherecast(const DV& d, const sgp_ray* rays, const float* radii, uint32_t n, sgp_hit* hits, hipStream_t s) { if (n) hipLaunchKernelGGL(k_spherecast, dim3((n + 63) / 64), dim3(64), 0, s, d, rays, radii, n, hits); }
void launch_route_export(const DV& d, uint32_t nb, const TileRoute& t, uint32_t* block_counts, uint32_t* block_offsets, RouteHeader* header,
                         sgp_ghost_record* out, uint32_t cap, uint32_t* emigrant_ids, uint32_t emigrant_cap, hipStream_t s)
{
	const uint32_t blocks = blocks_for(nb);
	hipLaunchKernelGGL(k_route_count, dim3(blocks), dim3(TPB), 0, s, d, t, block_counts);
	hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(1024), 0, s, (const uint32_t*)block_counts, block_offsets, blocks, t.n_tiles, header);
	hipLaunchKernelGGL(k_route_write, dim3(blocks), dim3(TPB), 0, s, d, t, (const uint32_t*)block_offsets, (const RouteHeader*)header, out, cap, emigrant_ids, emigrant_cap);
}
void launch_tiles_hist(const DV& d, uint32_t nb, const TilePlanes& tp, int level, uint32_t* out, hipStream_t s) { hipLaunchKernelGGL(k_tiles_hist, dim3(blocks_for(nb)), dim3(TPB), 0, s, d, tp, level, out); }
void launch_pack_ghost_keys(const sgp_ghost_record* recs, uint32_t n, void* out, hipStream_t s)
{
	if (n) hipLaunchKernelGGL(k_pack_ghost_keys, dim3(blocks_for(n)), dim3(TPB), 0, s, recs, n, (uint4*)out);
}
void launch_ghost_refresh_records(const DV& d, const sgp_ghost_record* recs, const uint32_t* ids, uint32_t n, hipStream_t s)
{
	if (n) hipLaunchKernelGGL(k_ghost_refresh_records, dim3(blocks_for(n)), dim3(TPB), 0, s, d, recs, ids, n);
}
void launch_export_boundary(const DV& d, uint32_t nb, float3 lo, float3 hi, float margin, sgp_ghost_record* out, uint32_t cap, uint32_t* count, hipStream_t s)
{
	const uint32_t blocks = blocks_for(nb);
	hipLaunchKernelGGL(k_export_count, dim3(blocks), dim3(TPB), 0, s, d, lo, hi, margin);
	hipLaunchKernelGGL(k_export_boundary, dim3(blocks), dim3(TPB), 0, s, d, lo, hi, margin, out, cap, count);
}
