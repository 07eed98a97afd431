// sgp_k_constraints.hip -- K5 / K6 -- colouring, constraint setup with the contact-cache match, the cache rebuild, contact events.
// One of the stage files of the step kernels (stage map: sgp_kernels.h).  Kernels first, their launch wrappers at the end.
#include "sgp_dev_all.h"

// Colour inheritance through the contact cache: a persisted manifold keeps last step's colour when both of its movable
// bodies were already movable when that colour was chosen (last step's proper colouring then guarantees that no two
// inheritors sharing a movable body carry the same colour).  Only the remaining manifolds go through the rounds.
__global__ void __launch_bounds__(TPB) k_colour_inherit(DV d)
{
	const uint32_t n = min(d.ctr->n_manifolds, d.cap_manifolds);
	for (uint32_t m = blockIdx.x * TPB + threadIdx.x; m < n; m += gridDim.x * TPB) {
		const uint2 ab = d.man_ab[m];
		// the one hash probe per manifold (unless the narrow phase already made it for a contact-cache attempt): every later kernel reads man_prev.
		// The probe's 16-byte entry also holds the previous constraint's colour; a manifold the narrow phase probed carries it in man_colour (-(3 + colour))
		uint32_t mp = d.man_prev[m];
		const int mc = d.man_colour[m];
		int pc = mc <= -3 ? -3 - mc : -1;
		if (mp == MAN_PREV_LOOKUP) {
			int pnc = 0;
			const uint32_t f = cache_find(d, ((uint64_t)ab.x << 32) | ab.y, &pnc);
			mp = f == 0xFFFFFFFFu ? MAN_PREV_NONE : (f | ((uint32_t)(pnc & 0xFF) << MAN_PREV_PNP_SHIFT)); d.man_prev[m] = mp;
			if (f != 0xFFFFFFFFu) pc = (pnc >> 8) & 0xFF;
		}
		int col = -1;
		if (pc >= 0 && pc < SGP_OVERFLOW_COLOUR && (mp & MAN_PREV_SLOT_MASK) != MAN_PREV_NONE) {
			const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
			const bool ma = fa & BF_MOVABLE_CUR, mb = fb & BF_MOVABLE_CUR;
			if (!((ma && !(fa & BF_MOVABLE_PREV)) || (mb && !(fb & BF_MOVABLE_PREV))) &&
			    !(pc == 0 && ((ma && chassis_colours(d, ab.x, fa)) || (mb && chassis_colours(d, ab.y, fb))))) {      // (the body became a chassis since: colour 0 is the vehicle's)
				col = pc;
				if (ma) atomicOr((unsigned long long*)&d.colour_mask[ab.x], 1ull << pc);
				if (mb) atomicOr((unsigned long long*)&d.colour_mask[ab.y], 1ull << pc);
			}
		}
		if (col != mc) d.man_colour[m] = col;      // (-1: through the colouring rounds)
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
		const float4 n4 = d.man_n[m];
		const int npb = __float_as_int(n4.w);
		const int np = (npb & MAN_NP_SENSOR) ? 0 : (npb & 0xFF);          // sensor pairs carry no points
		const v3 nrm = V3(n4);
		const uint32_t mprev = d.man_prev[m];
		const bool reused = mprev & MAN_PREV_REUSED;                 // the manifold came from the body-pair contact cache
		const uint32_t fslot = (mprev & MAN_PREV_SLOT_MASK) == MAN_PREV_NONE ? 0xFFFFFFFFu : (mprev & MAN_PREV_SLOT_MASK);
		const uint32_t pslot = d.st.warm_start ? fslot : 0xFFFFFFFFu;
		const int pnp_all = fslot != 0xFFFFFFFFu ? (int)((mprev >> MAN_PREV_PNP_SHIFT) & 7u) : 0;      // points of the previous constraint (came with the hash probe: no gather)
		const int pnp = pslot != 0xFFFFFFFFu ? pnp_all : 0;                                           // ... whose impulses may be taken over
		// the previous constraint's points in body space (requested together with the bodies' records) and, for a manifold taken from the body-pair contact
		// cache, the record of the relative pose it was computed at
		v3 pl1[4], pl2[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			pl1[j] = pl2[j] = V3(0.0f, 0.0f, 0.0f);
			if (j < pnp_all) { pl1[j] = V3(PRV(d).loc1[j][fslot]); pl2[j] = V3(PRV(d).loc2[j][fslot]); }
		}
		float4 pr0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pr1 = pr0, pr2 = pr0;
		if (reused) { const float4* rec = PRV(d).prec + (size_t)fslot * PREC_F4; pr0 = rec[0]; pr1 = rec[1]; pr2 = rec[2]; }
		// per body: pose record, velocity record (velocities after gravity + the effective inverse mass: k_pre_solve), property record
		const float4 pa4 = d.pose[POSE_F4 * (size_t)ab.x], qa4 = d.pose[POSE_F4 * (size_t)ab.x + 1], pb4 = d.pose[POSE_F4 * (size_t)ab.y], qb4 = d.pose[POSE_F4 * (size_t)ab.y + 1];
		const float4 va4 = d.vel[VEL_F4 * (size_t)ab.x], wa4 = d.vel[VEL_F4 * (size_t)ab.x + 1], vb4 = d.vel[VEL_F4 * (size_t)ab.y], wb4 = d.vel[VEL_F4 * (size_t)ab.y + 1];
		const float4 ia4 = d.pose[POSE_F4 * (size_t)ab.x + 2], sa4 = d.pose[POSE_F4 * (size_t)ab.x + 3], ib4 = d.pose[POSE_F4 * (size_t)ab.y + 2], sb4 = d.pose[POSE_F4 * (size_t)ab.y + 3];
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
		const v3 g = V3(d.gx, d.gy, d.gz);
		CUR(d).hdr[slot] = make_uint4(ab.x, ab.y, (uint32_t)(np | (col << 8) | ((fslot != 0xFFFFFFFFu ? 1 : 0) << 16)), 0u);      // one 16-byte store: ids + np_col
		CUR(d).n_fric[slot] = F4(nrm, friction);
		// warm start (docs/CONTRACT.md): the cached impulses of the manifold summed -- linear P, angular about either centre of mass A1, A2 --, one velocity
		// change per body, written as the body's record of this colour (k_warm_bodies adds a body's records in colour order)
		v3 wP = V3(0.0f, 0.0f, 0.0f), wA1 = V3(0.0f, 0.0f, 0.0f), wA2 = V3(0.0f, 0.0f, 0.0f);
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if (i >= np) break;
			const v3 p1 = V3(d.man_p1[i][m]), p2 = V3(d.man_p2[i][m]);
			v3 local1 = m33_tmul(RA, v3_sub(p1, posA));
			v3 local2 = m33_tmul(RB, v3_sub(p2, posB));
			if (reused) { local1 = pl1[i]; local2 = pl2[i]; }      // the cached body-space points themselves: no drift from re-deriving them
			float lam_n = 0.0f, lam_t1 = 0.0f, lam_t2 = 0.0f;
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				if (j >= pnp) break;
				if (v3_len_sq(v3_sub(local1, pl1[j])) < d.st.contact_point_preserve_lambda_max_dist_sq &&
				    v3_len_sq(v3_sub(local2, pl2[j])) < d.st.contact_point_preserve_lambda_max_dist_sq) {
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
			v3 wj = v3_scale(nrm, lam_n);
			if (friction > 0.0f) { wj = v3_add(wj, v3_scale(t1, lam_t1)); wj = v3_add(wj, v3_scale(t2, lam_t2)); }
			wP = v3_add(wP, wj);
			wA1 = v3_add(wA1, v3_cross(r1, wj));
			wA2 = v3_add(wA2, v3_cross(r2, wj));
		}
		// body-pair contact cache (polytope pairs): a fresh manifold records where the bodies are relative to each other now; a reused one keeps the record of
		// the step its points were computed in (slow drift then ends the reuse)
		if (npb & MAN_NP_POLYTOPE) {
			float4* rec = CUR(d).prec + (size_t)slot * PREC_F4;
			if (reused) { rec[0] = pr0; rec[1] = pr1; rec[2] = pr2; }
			else {
				v3 dpos; quat drot;
				pair_relative_pose(posA, Q4(qa4), posB, Q4(qb4), &dpos, &drot);
				const v3 nl = m33_tmul(RB, nrm);
				rec[0] = make_float4(drot.x, drot.y, drot.z, drot.w);
				rec[1] = make_float4(dpos.x, dpos.y, dpos.z, nl.x);
				rec[2] = make_float4(nl.y, nl.z, 0.0f, 0.0f);
			}
		}
		// (body, colour) -> warm-start record: a proper colouring gives every movable body at most one constraint per colour, so this table needs no
		// clearing -- its valid entries are exactly the bits of colour_mask[body] (read by k_warm_bodies).  Body 1 loses what body 2 gains.
		if (col < SGP_OVERFLOW_COLOUR) {
			if (im1 > 0.0f) {
				float4* wr = d.warm + ((size_t)ab.x * SGP_MAX_COLOURS + col) * 2;
				wr[0] = F4(v3_neg(v3_scale(wP, im1)), 0.0f); wr[1] = F4(v3_neg(sym33_mul(I1, wA1)), 0.0f);
			}
			if (im2 > 0.0f) {
				float4* wr = d.warm + ((size_t)ab.y * SGP_MAX_COLOURS + col) * 2;
				wr[0] = F4(v3_scale(wP, im2), 0.0f); wr[1] = F4(sym33_mul(I2, wA2), 0.0f);
			}
		}
	}
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
	// nobody was awake in this step (StepCounters::any_awake): it made no constraint, the table was not emptied, and the buffer parity goes back to what it was --
	// the next step finds the cache this one found
	const bool kept = !d.ctr->any_awake;
	const uint32_t n_con = kept ? 0u : d.ctr->n_constraints;
	const uint32_t size = *d.ht_cur;
	const uint32_t mask = size - 1;
	if (kept && blockIdx.x == 0 && threadIdx.x == 0) d.sp->parity = d.sp->parity ^ 1u;
	for (uint32_t k = blockIdx.x * TPB + threadIdx.x; k < n_con; k += gridDim.x * TPB) {
		const uint4 hd = con_hdr(CUR(d), k);
		const uint2 ab = make_uint2(hd.x, hd.y);
		const uint32_t nc = hd.z;
		const uint64_t key = ((uint64_t)ab.x << 32) | ab.y;
		uint32_t h = ht_hash(key, mask);
		for (uint32_t probe = 0; probe < size; ++probe) {
			// (an entry is 16 bytes: the key's two words claimed with one 64-bit compare-and-swap, slot and np_col stored behind it -- nothing reads the
			// table before the next step)
			const unsigned long long old = atomicCAS((unsigned long long*)&d.ht[h], ~0ull, (unsigned long long)key);
			if (old == ~0ull || old == key) { ((uint2*)&d.ht[h])[1] = make_uint2(k, nc); break; }
			h = (h + 1) & mask;
		}
	}
	// The contacts of sleeping pairs stay in the cache: an entry of the cache this step found is carried over -- copied behind this step's constraints, entered in the
	// table -- when neither of its bodies was awake in this step (such a pair made no constraint of its own: no duplicate) and neither was created or reshaped since
	// the last step.  It is found like any other when the pair wakes (warm start, the body-pair cache's manifold, 'persisted') and is never solved, counted or reported.
	if (!kept) {
		const uint32_t par = d.sp->parity;
		const uint32_t prev_total = min(d.cache_total[par ^ 1u], d.cap_manifolds);
		const ConstraintArrays& P = PRV(d); const ConstraintArrays& Cn = CUR(d);
		for (uint32_t k0 = blockIdx.x * TPB; k0 < prev_total; k0 += gridDim.x * TPB) {
			const uint32_t k = k0 + threadIdx.x;
			bool carry = false; uint4 hd = make_uint4(0u, 0u, 0u, 0u);
			if (k < prev_total) {
				hd = con_hdr(P, k);
				const uint32_t fa = d.flags[hd.x], fb = d.flags[hd.y];
				carry = (fa & BF_ALIVE) && (fb & BF_ALIVE) && !((fa | fb) & (BF_AWAKE_STEP | BF_FRESH));
			}
			if (!carry) continue;
			const uint32_t slot = wave_alloc(&d.cache_total[par]);
			if (slot >= d.cap_manifolds) continue;
			Cn.hdr[slot] = hd;
			const int np = (int)(hd.z & 0xFFu);
#pragma unroll
			for (int j = 0; j < 4; ++j) if (j < np) { Cn.loc1[j][slot] = P.loc1[j][k]; Cn.loc2[j][slot] = P.loc2[j][k]; Cn.lam[j][slot] = P.lam[j][k]; }
			{ const float4* src = P.prec + (size_t)k * PREC_F4; float4* dst = Cn.prec + (size_t)slot * PREC_F4; dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; }
			const uint64_t key = ((uint64_t)hd.x << 32) | hd.y;
			uint32_t h = ht_hash(key, mask);
			for (uint32_t probe = 0; probe < size; ++probe) {
				const unsigned long long old = atomicCAS((unsigned long long*)&d.ht[h], ~0ull, (unsigned long long)key);
				if (old == ~0ull || old == key) { ((uint2*)&d.ht[h])[1] = make_uint2(slot, hd.z); break; }
				h = (h + 1) & mask;
			}
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
		const bool persisted = (d.man_prev[m] & MAN_PREV_SLOT_MASK) != MAN_PREV_NONE;
		// one atomic per wave and list (the lanes here are the loop's active lanes; wave_alloc serves those that call it together)
		uint32_t k;
		if (persisted) k = wave_alloc(&d.evc->n_contact_persisted); else k = wave_alloc(&d.evc->n_contact_added);
		if (k >= d.cap_contact_events) continue;
		sgp_contact_event e;
		e.id1 = ab.x; e.id2 = ab.y; e.userdata1 = 0; e.userdata2 = 0;
		const float4 la = d.vel[VEL_F4 * (size_t)ab.x], lb = d.vel[VEL_F4 * (size_t)ab.y];      // velocities after gravity, before the solve (k_pre_solve)
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
void launch_cache_build(const DV& d, uint32_t n_con, StepCounters* host_mapped, EventCounters* host_events, hipStream_t s)
{
	// (the table was emptied by the first k_island_mark launch of the step)
	hipLaunchKernelGGL(k_cache_build, dim3(stride_grid(n_con)), dim3(TPB), 0, s, d, host_mapped, host_events);
}
void launch_contact_events(const DV& d, uint32_t est, hipStream_t s) { hipLaunchKernelGGL(k_contact_events, dim3(stride_grid(est)), dim3(TPB), 0, s, d); }
