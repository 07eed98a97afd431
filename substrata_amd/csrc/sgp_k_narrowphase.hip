// sgp_k_narrowphase.hip -- K4 -- one thread per candidate pair (sphere / box / capsule), in-step activation (k_wake_pairs), convex hull pairs (a wave per pair).
// One of the stage files of the step kernels (stage map: sgp_kernels.h).  Kernels first, their launch wrappers at the end.
#include "sgp_dev_all.h"
#ifndef NP_CLIP_LDS
#define NP_CLIP_LDS 1      // the box - box clip polygons of k_narrowphase in LDS (0: private arrays, i.e. scratch -- kept for A/B measurements)
#endif

// The body-pair contact cache (ContactConstraintManager::GetContactsFromCache) for one pair of non-mesh bodies: true = *m is last step's
// manifold carried to the bodies' current poses -- the two bodies sit, relative to each other, where they sat when it was computed (within
// 1 mm and 2 degrees), so the collision test is skipped.  *prev: the pair's slot in last step's constraints (MAN_PREV_NONE if it had none),
// found with the one hash look-up every later kernel shares.
SGP_DEV bool reuse_cached_manifold(const DV& d, uint2 ab, uint32_t fa, uint32_t fb, sgd_manifold* m, uint32_t* prev, int* colour_candidate)
{
	int pnc = 0;
	const uint32_t ps = cache_find(d, ((uint64_t)ab.x << 32) | ab.y, &pnc);
	if (ps == 0xFFFFFFFFu) { *prev = MAN_PREV_NONE; return false; }
	*prev = ps | ((uint32_t)(pnc & 0xFF) << MAN_PREV_PNP_SHIFT);
	*colour_candidate = man_colour_candidate(pnc);      // (k_colour_inherit needs no probe of its own for this pair)
	if (!d.st.use_body_pair_contact_cache || ((fa | fb) & (BF_CACHE_INVALID | BF_SENSOR))) return false;
	const float4* rec = PRV(d).prec + (size_t)ps * PREC_F4;      // the relative pose the previous manifold was computed at: one 64-byte record
	const float4 cdr = rec[0], cdp = rec[1], cnl = rec[2];
	const v3 posA = V3(d.pose[POSE_F4 * (size_t)ab.x]), posB = V3(d.pose[POSE_F4 * (size_t)ab.y]);
	const quat qA = Q4(d.pose[POSE_F4 * (size_t)ab.x + 1]), qB = Q4(d.pose[POSE_F4 * (size_t)ab.y + 1]);
	v3 dpos; quat drot;
	pair_relative_pose(posA, qA, posB, qB, &dpos, &drot);
	if (!(v3_len_sq(v3_sub(dpos, V3(cdp))) <= d.st.body_pair_cache_max_delta_position_sq)) return false;
	const float dq = drot.x * cdr.x + drot.y * cdr.y + drot.z * cdr.z + drot.w * cdr.w;
	if (!(fabsf(dq) >= d.st.body_pair_cache_cos_max_delta_rotation_div2)) return false;
	const m33 RA = quat_to_m33(qA), RB = quat_to_m33(qB);
	m->np = pnc & 0xFF;
	m->n = m33_mul(RB, V3(cdp.w, cnl.x, cnl.y));
	for (int i = 0; i < 4; ++i) if (i < m->np) { m->p1[i] = v3_add(posA, m33_mul(RA, V3(PRV(d).loc1[i][ps]))); m->p2[i] = v3_add(posB, m33_mul(RB, V3(PRV(d).loc2[i][ps]))); }
	*prev |= MAN_PREV_REUSED;
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
	// the box - box clip's two polygons per lane (sgd_box_box<true>): 12 KB per wave; the activation round's few pairs keep the private arrays
	__shared__ float s_clip[ROUND == 0 && NP_CLIP_LDS ? TPB / 64 : 1][ROUND == 0 && NP_CLIP_LDS ? 2 * SGD_LPOLY_FLOATS : 1];
	const uint32_t n = ROUND ? min(d.ctr->n_wake_pairs, d.cap_wake_pairs) : min(d.ctr->n_pairs, d.cap_pairs);
	const uint2* const pairs = ROUND ? d.wake_pairs : d.pairs;
	const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
	for (uint32_t p0 = blockIdx.x * TPB; p0 < n; p0 += gridDim.x * TPB) {
		const uint32_t p = p0 + threadIdx.x;
		bool have = false;
		uint2 ab = make_uint2(0u, 0u); uint32_t fa = 0, fb = 0;
		sgd_manifold m;
		uint32_t prev = MAN_PREV_LOOKUP;
		int colour_candidate = -1;
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
				const bool polytopes = pair_is_polytopes(fa, fb);
				if (polytopes && reuse_cached_manifold(d, ab, fa, fb, &m, &prev, &colour_candidate)) have = true;
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
						if (hull_pair_is_big(sa, sb)) {      // (round 5: hundreds of thousands of edge pairs are not one thread's work -- to the list, k_narrowphase_hull_big after this launch)
							const uint32_t k = wave_alloc(&d.ctr->n_hull_pairs);
							if (k < d.cap_hull_pairs) d.hull_pairs[k] = ab; else atomicAdd(&d.ctr->pairs_dropped, 1u);
						} else
						have = sgd_collide_hull(&sa, &sb, d.st.speculative_contact_distance, &m) != 0;
					}
				} else {
					const sgd_shape sa = load_shape(d, ab.x, fa), sb = load_shape(d, ab.y, fb);
					if constexpr (ROUND == 0 && NP_CLIP_LDS) have = sgd_collide<true>(&sa, &sb, d.st.speculative_contact_distance, &m, &s_clip[wave][lane]) != 0;
					else have = sgd_collide(&sa, &sb, d.st.speculative_contact_distance, &m) != 0;
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
			emit_manifold_at(d, slot, ab, fa, fb, m, prev, colour_candidate);
		}
		__syncthreads();
	}
}
#ifndef NP_WAVES
#define NP_WAVES 3      // waves per SIMD of k_narrowphase: its 48 KB of clip polygons allow three workgroups per CU
#endif
__global__ void __launch_bounds__(TPB, NP_WAVES) k_narrowphase(DV d) { narrowphase_pairs<0>(d); }
template <bool HULLS> __global__ void __launch_bounds__(TPB, 4) k_narrowphase_wake(DV d) { narrowphase_pairs<1, HULLS>(d); }

// IN-STEP ACTIVATION (PhysicsSystem::JobFindCollisions keeps taking bodies from the active list while ProcessBodyPair appends the ones it wakes: a woken
// body collides in the step that woke it, and wakes what it touches in turn).  One extra round: a body the narrow phase or a wheel marked takes along
// everything that fell asleep in the same island (sleep_label / label_wake: sleeping bodies have not moved, so the contacts that made the island are the
// ones the cascade would follow), and every woken body is paired here with all that was not awake when the step began -- its pairs with awake
// bodies exist already.  The narrow-phase kernels then run once more over these pairs (the hull and mesh kernels from hull_base / mesh_base on).
// What those contacts wake in turn -- two islands that went to sleep apart and touch -- is woken too but meets its other contacts next step.
SGP_DEV bool body_woken(const DV& d, uint32_t j, uint32_t fj, uint32_t epoch)
{
	// woken itself (BF_WAKE: a contact of the first round, a wheel) or along with its island (the label's mark).  The first alone matters when the label is no longer
	// current -- the island's root left and its slot went to another body: such a body wakes alone (wake_body stamps nothing), and it still meets, in this step, what was
	// not awake when the step began (tools/fuzz_tiles.py seed 23: a sleeper woken by a new ghost lost its contact with the ground for a step)
	if ((fj & (BF_ALIVE | BF_ACTIVE | BF_ALIAS)) != BF_ALIVE || f_motion(fj) != SGP_MOTION_DYNAMIC) return false;
	if (fj & BF_WAKE) return true;
	return label_current(d, d.sleep_label[j]) && d.label_wake[SGP_LABEL_SLOT(d.sleep_label[j])] == epoch;
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
		if (hull_pair_is_big(sa, sb)) continue;      // (a hull beyond 32 vertices: k_narrowphase_hull_big writes this pair's work item)
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

// The pairs k_narrowphase_hull leaves out: a hull of more than 32 vertices against a hull or a box, a workgroup
// per pair (hull_sat_search_block).  Launched only in worlds that hold such a hull; in the in-step activation round the list holds nothing else.
__global__ void __launch_bounds__(HULL_BIG_TPB) k_narrowphase_hull_big(DV d)
{
	__shared__ HullBigLds L;
	const uint32_t n = min(d.ctr->n_hull_pairs, d.cap_hull_pairs);
	for (uint32_t p = d.ctr->hull_base + blockIdx.x; p < n; p += gridDim.x) {
		const uint2 ab = d.hull_pairs[p];
		const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
		const sgd_shape sa = load_shape(d, ab.x, fa), sb = load_shape(d, ab.y, fb);
		if (!hull_pair_is_big(sa, sb)) continue;
		HullWork wk; wk.ab = ab; wk.round_other = 0u;
		const bool flip = sa.type > sb.type;
		const sgd_shape* x = flip ? &sb : &sa; const sgd_shape* y = flip ? &sa : &sb;
		const sgd_hview hx = sgd_hull_view(x), hy = sgd_hull_view(y);
		if (!hull_sat_search_block(&hx, &hy, d.st.speculative_contact_distance, &wk.r, L)) { wk.round_other = 2u; memset(&wk.r, 0, sizeof(wk.r)); }
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
void launch_narrowphase(const DV& d, uint32_t est, hipStream_t s)
{
	hipLaunchKernelGGL(k_narrowphase, dim3(stride_grid(est)), dim3(TPB), 0, s, d);
}
void launch_wake_round(const DV& d, uint32_t nb, int has_hulls, bool has_meshes, hipStream_t s)      // has_hulls: 0 none, 1 some, 2 some of more than 32 vertices
{
	hipLaunchKernelGGL(k_wake_pairs, dim3(blocks_for(nb)), dim3(TPB), 0, s, d);
	if (has_hulls) hipLaunchKernelGGL(k_narrowphase_wake<true>, dim3(32), dim3(TPB), 0, s, d);      // (few pairs, 1.7 KB of scratch per lane: a small grid starts faster)
	else hipLaunchKernelGGL(k_narrowphase_wake<false>, dim3(32), dim3(TPB), 0, s, d);
	// (hull pairs of this round are collided by k_narrowphase_wake itself; hull - mesh pairs by the hull instances of the mesh kernels)
	if (has_hulls == 2) {      // (the big pairs k_narrowphase_wake put on the list: their search by workgroups, their manifolds)
		hipLaunchKernelGGL(k_narrowphase_hull_big, dim3(256), dim3(HULL_BIG_TPB), 0, s, d);
		hipLaunchKernelGGL(k_narrowphase_hull_manifold, dim3(64), dim3(64), 0, s, d);
	}
	if (has_meshes) launch_narrowphase_mesh_blocks(d, has_hulls != 0, 256, s);
}
void launch_narrowphase_hull(const DV& d, bool big_hulls, hipStream_t s)
{
	hipLaunchKernelGGL(k_narrowphase_hull, dim3(4096), dim3(64), 0, s, d);
	if (big_hulls) hipLaunchKernelGGL(k_narrowphase_hull_big, dim3(2048), dim3(HULL_BIG_TPB), 0, s, d);
	hipLaunchKernelGGL(k_narrowphase_hull_manifold, dim3(1024), dim3(64), 0, s, d);
}
