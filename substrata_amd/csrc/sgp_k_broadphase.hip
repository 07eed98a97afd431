// sgp_k_broadphase.hip -- K2 / K3 -- the step's first launch (bounds), the paged cell grid, scans, pairs by 4 x 4 x 4-cell tile in LDS, the large bodies.
// One of the stage files of the step kernels (stage map: sgp_kernels.h).  Kernels first, their launch wrappers at the end.
#include "sgp_dev_all.h"

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
	const bool leads = binned && leader == lane;
	uint32_t* const entry = &d.tile_slot[leads ? tile : 0u];
	if (leads) sl = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (relaxed: the slot NUMBER is all that travels through the entry)
	// a tile nobody has asked for yet: whoever turns its entry from NONE to PENDING gives it a slot.  The wave's winners take their slots with ONE atomic on the counter
	// (round 6: every new tile for itself was 12 k atomics on one address at config 5 -- scattered debris, a wave's 64 bodies in 64 tiles -- and most of the kernel's 48 us)
	bool won = false;
	if (leads && sl >= BP_TILE_PENDING) {
		const uint32_t old = atomicCAS(entry, BP_TILE_NONE, BP_TILE_PENDING);
		if (old == BP_TILE_NONE) won = true; else sl = old;      // (a slot, or PENDING: somebody else is about to publish one)
	}
	{
		const unsigned long long wm = __ballot(won);
		if (wm) {
			uint32_t base = 0;
			const int first = __ffsll((long long)wm) - 1;
			if (lane == first) base = atomicAdd(&d.ctr->n_tiles_used, (uint32_t)__popcll(wm));
			base = (uint32_t)__shfl((int)base, first, 64);
			if (won) {
				sl = base + (uint32_t)__popcll(wm & ((1ull << lane) - 1ull));
				d.tile_of_slot[sl] = tile;
				__hip_atomic_store(entry, sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
	}
	// (a lane that waits, waits for a lane of ANOTHER wave: the tiles of this wave's leaders differ, and its winners have published above)
	if (leads) for (int tries = 0; tries < (1 << 24) && sl >= BP_TILE_PENDING; ++tries) sl = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
#ifndef BP_LDS_CAP_SMALL
#define BP_LDS_CAP_SMALL 768
#endif
#ifndef BP_PAIR_CAP_SMALL
#define BP_PAIR_CAP_SMALL 1024
#endif
#define BP_LDS_CAP_LARGE 1536
#define BP_PAIR_CAP_LARGE 2048
#ifndef BP_SPLIT
#define BP_SPLIT 4
#endif

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
__global__ void __launch_bounds__(TPB) k_gather_aabbs(DV d, const uint32_t* ids, uint32_t n, float4* out)
{
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= n) return;
	out[2 * (size_t)k] = d.aabb_min[ids[k]]; out[2 * (size_t)k + 1] = d.aabb_max[ids[k]];
}
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
void launch_gather_aabbs(const DV& d, const uint32_t* ids, uint32_t n, float4* out, hipStream_t s) { if (n) hipLaunchKernelGGL(k_gather_aabbs, dim3(blocks_for(n)), dim3(TPB), 0, s, d, ids, n, out); }
