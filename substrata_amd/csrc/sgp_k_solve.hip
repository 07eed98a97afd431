// sgp_k_solve.hip -- K7 -- warm start, velocity and position iterations: a launch per colour, the high colours by connected component, the tail, small worlds.
// One of the stage files of the step kernels (stage map: sgp_kernels.h).  Kernels first, their launch wrappers at the end.
#include "sgp_dev_all.h"

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
template <int VS> SGP_DEV void load_pair(const DV& d, uint32_t slot, PairCtx& c, const float4* vel)
{
	const uint4 hd = con_hdr(CUR(d), slot);
	c.ab = make_uint2(hd.x, hd.y);
	const float4 nf = CUR(d).n_fric[slot];
	c.n = V3(nf); c.friction = nf.w;
	c.np = (int)hd.z & 0xFF;
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

// Warm start of ONE constraint (docs/CONTRACT.md, warm start; oracle: warm_start_constraint): the cached impulses of the manifold are summed first -- per point
// n lam_n (+ t1 lam_t1 + t2 lam_t2 with friction), over the points the linear impulse P and the angular impulses A1 = sum r1 x j, A2 = sum r2 x j -- and each
// body receives one velocity change, v -+ P / m, w -+ I^-1 A.  k_setup evaluates the same expressions on the same operands for the (body, colour) records
// k_warm_bodies adds up, so this form (small worlds, the tail, the overflow colour) and that one give the same bits.
template <int VS> SGP_DEV void warm_start_one_t(const DV& d, uint32_t slot, float4* vel)
{
	PairCtx c;
	load_pair<VS>(d, slot, c, vel);
	c.t1 = v3_normalized_perpendicular(c.n);
	c.t2 = v3_cross(c.n, c.t1);
	v3 P = V3(0.0f, 0.0f, 0.0f), A1 = V3(0.0f, 0.0f, 0.0f), A2 = V3(0.0f, 0.0f, 0.0f);
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (i < c.np) {
			const v3 r1 = V3(CUR(d).r1b[i][slot]), r2 = V3(CUR(d).r2e[i][slot]);
			const float4 l = CUR(d).lam[i][slot];
			v3 j = v3_scale(c.n, l.x);
			if (c.friction > 0.0f) { j = v3_add(j, v3_scale(c.t1, l.y)); j = v3_add(j, v3_scale(c.t2, l.z)); }
			P = v3_add(P, j);
			A1 = v3_add(A1, v3_cross(r1, j));
			A2 = v3_add(A2, v3_cross(r2, j));
		}
	}
	if (c.im1 > 0.0f) { c.A.lv = v3_add(c.A.lv, v3_neg(v3_scale(P, c.im1))); c.A.av = v3_add(c.A.av, v3_neg(sym33_mul(c.I1, A1))); }
	if (c.im2 > 0.0f) { c.B.lv = v3_add(c.B.lv, v3_scale(P, c.im2)); c.B.av = v3_add(c.B.av, sym33_mul(c.I2, A2)); }
	store_pair_vel<VS>(c, vel);
}
SGP_DEV void warm_start_one(const DV& d, uint32_t slot) { warm_start_one_t<VEL_F4>(d, slot, d.vel); }

// Warm start, one thread per BODY instead of one launch per colour.  What a constraint's warm start does to one of its bodies depends only on the constraint
// (cached impulses, axes, lever arms) and on that body's inverse mass / inertia -- not on any velocity --, so what the colour-by-colour order does to one body
// is a fixed sequence of additions: the velocity changes of its constraints in ascending colour (a body has at most one per colour).  k_setup writes each
// change as a 32-byte record at [body][colour]; this kernel adds a body's records in colour order.  A body's colours are its lowest ones (the colouring takes
// the lowest free colour), so its records are a few consecutive 128-byte lines.  Constraints of the overflow colour come last in the order and are applied
// serially by the last workgroup.
SGP_DEV void warm_body_one(const DV& d, uint32_t i)
{
	if (i >= d.sp->n_slots) return;
	uint64_t mask = d.colour_mask[i] & ~(1ull << SGP_OVERFLOW_COLOUR);
	if (!mask) return;
	float4* rec = d.vel + VEL_F4 * (size_t)i;
	const float4 v4 = rec[0], w4 = rec[1];
	const float im = v4.w;
	if (!(im > 0.0f)) return;
	const float4* wr = d.warm + (size_t)i * SGP_MAX_COLOURS * 2;
	v3 lv = V3(v4), av = V3(w4);
	// four records requested at a time (independent loads), added in colour order
	while (mask) {
		int col[4]; float4 a[4], b[4];
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			col[k] = mask ? __ffsll((long long)mask) - 1 : -1;
			if (mask) mask &= mask - 1;
			if (col[k] >= 0) { a[k] = wr[2 * col[k]]; b[k] = wr[2 * col[k] + 1]; }
		}
#pragma unroll
		for (int k = 0; k < 4; ++k) if (col[k] >= 0) { lv = v3_add(lv, V3(a[k])); av = v3_add(av, V3(b[k])); }
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

SGP_DEV void solve_position_one(const DV& d, uint32_t slot)
{
	const uint4 hd = con_hdr(CUR(d), slot);
	const uint2 ab = make_uint2(hd.x, hd.y);
	const float4 nf = CUR(d).n_fric[slot];
	const v3 nrm = V3(nf);
	const int np = (int)hd.z & 0xFF;
	// the pose records themselves (k_integrate_pose advanced them; the corrections are made in place) + the local inverse inertia
	float4* ra = d.pose + POSE_F4 * (size_t)ab.x;
	float4* rb = d.pose + POSE_F4 * (size_t)ab.y;
	const float4 pa = ra[0], pb = rb[0];
	const float im1 = pa.w, im2 = pb.w;                 // 0 unless dynamic (and a dynamic body in a constraint is awake: touched sleepers are woken by k_pre_solve)
	quat qa = Q4(ra[1]), qb = Q4(rb[1]);
	const v3 iiA = V3(d.pose[POSE_F4 * (size_t)ab.x + 2]), iiB = V3(d.pose[POSE_F4 * (size_t)ab.y + 2]);
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

// One launch = one colour of one pass.  The slot range comes from the device-side colour table, so the host never has
// to know the counts of the current step; the grid is sized from the previous step and the loop strides over the rest.
#define SOLVE_TPB 64      // one wave per workgroup: a colour of ~17k constraints then spreads over all 256 CUs instead of 67 of them
// OCC: waves per SIMD the kernel is built for.  A colour of config 3 (~27k constraints) is less than one wave per SIMD and wants the registers; a colour of config 4
// (300k+ constraints: nine waves per SIMD) runs in rounds, each a chain of dependent gathers, and wants the rounds to be few -- launch_solve_colour picks by size.
// (the two leading pointer arguments repeat d.cstarts and d.sp: leading scalar arguments are PRELOADED into registers when the wave starts (kernarg preload,
// -amdgpu-kernarg-preload-count), so the colour table and the buffer parity are requested at once, beside the fetch of the rest of the arguments instead of after it --
// one dependent scalar fetch less on the chain of each of the ~110 colour launches of a step)
template <int MODE, int ROWS = -1, int OCC = 1> __global__ void __launch_bounds__(MODE != 0 ? SOLVE_VEL_TPB : SOLVE_TPB, OCC) k_solve_colour(const uint32_t* cstarts_pre, StepParams* sp_pre, DV d_arg, int colour_arg)
{
	DV d = d_arg; d.sp = sp_pre;
	const int colour = colour_arg & 0xFF;
	const uint32_t first = cstarts_pre[colour], end = cstarts_pre[colour + 1];
	if (MODE != 0) {
		// velocity and position iterations: two neighbouring lanes per constraint
		const int side = (int)(threadIdx.x & 1u);
		// Workgroups are dealt to the eight XCDs in turn, each with an L2 of its own: workgroup b takes chunk (b % 8) * (n / 8) + b / 8 of the colour's
		// slots, so that one XCD works through a CONTIGUOUS eighth of them -- neighbouring slots are neighbouring manifolds, which share bodies'
		// cache lines, and the same XCD meets the same rows again in the next pass.  (The grid is a multiple of eight: launch_solve_colour.)
		const uint32_t bx = (colour_arg & SOLVE_XCD_CHUNKS) ? xcd_block() : blockIdx.x;      // (a colour of 200k constraints -- config 4 -- streams from HBM whatever the order, and lost 17 % with the chunks)
		for (uint32_t k = first + ((bx * SOLVE_VEL_TPB + threadIdx.x) >> 1); k < end; k += gridDim.x * (SOLVE_VEL_TPB / 2)) {
			if (MODE == 1) { if constexpr (ROWS == 2) solve_velocity_pair_norows<VEL_F4>(d, k, side, d.vel); else solve_velocity_pair_t<VEL_F4, ROWS>(d, k, side, d.vel); }
			else solve_position_pair(d, k, side);
		}
		return;
	}
	for (uint32_t k = first + blockIdx.x * SOLVE_TPB + threadIdx.x; k < end; k += gridDim.x * SOLVE_TPB) warm_start_one(d, k);
}


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
		const uint2 okab = con_ab(CUR(d), first + k); const uint64_t pr = sgp_mix64(((uint64_t)okab.x << 32) | okab.y);
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
			if (my_col == c) half_solve<VEL_F4>(h, side, d.vel, d.dbg_flags);
			__syncthreads();
		}
		if (mine) half_store(d, slot, side, h);
	} else
	for (int c = first_colour; c < SGP_OVERFLOW_COLOUR; ++c) {
		const uint32_t b = cs[c], e = cs[c + 1];
		if (b == e) continue;
		for (uint32_t k = b + pair; k < e; k += TAIL_VEL_TPB / 2) solve_velocity_pair_t<VEL_F4, ROWS>(d, k, side, d.vel);
		__syncthreads();
	}
	const uint32_t first = cs[SGP_OVERFLOW_COLOUR], count = cs[SGP_OVERFLOW_COLOUR + 1] - first;
	if (count == 0 || threadIdx.x >= 2) return;                   // lanes 0 and 1: the two sides of one constraint at a time
	uint64_t last = 0; bool have_last = false;
	for (uint32_t it = 0; it < count; ++it) {
		const uint32_t bslot = overflow_next(d, first, count, last, have_last);
		solve_velocity_pair_t<VEL_F4, ROWS>(d, bslot, side, d.vel);
	}
}
#define HC_CLASSES 9                  // component size classes: 1 << class constraints
#define HC_WG_PAIRS 256               // lane pairs (= constraints) per workgroup (8 waves: 2 per SIMD, a constraint half keeps its ~150 registers)
#define HC_TPB (2 * HC_WG_PAIRS)
#define HC_BIG 0xFFFFFFFFu
#define HC_NONE 0xFFFFFFFFu
#define NPCOL_CATCH_ALL (1 << 17)     // np_col: the constraint's component is too large for a workgroup
#define HC_BIG_LIST (4 * HC_WG_PAIRS)  // the catch-all's own list of such constraints (up to four per lane pair; more: it searches the colours for the flag)

SGP_DEV bool hc_can_move(const DV& d, uint32_t body) { return d.vel[VEL_F4 * (size_t)body].w > 0.0f; }      // effective inverse mass of the step (k_pre_solve)

// (1) a constraint between two bodies that can move joins their components (k_pre_solve made every body a component of its own);
//     the slot list is cleared to "no constraint"
__global__ void __launch_bounds__(TPB) k_hc_hook(DV d, int first_colour)
{
	const uint32_t b = d.cstarts[first_colour], e = d.cstarts[SGP_OVERFLOW_COLOUR];
	const uint32_t tid = blockIdx.x * TPB + threadIdx.x, stride = gridDim.x * TPB;
	const uint32_t lim = min(2u * (e - b) + HC_CLASSES * HC_WG_PAIRS, d.cap_hc_list);
	for (uint32_t i = tid; i < lim; i += stride) d.hc_list[i] = HC_NONE;
	for (uint32_t k = b + tid; k < e; k += stride) {
		const uint2 ab = con_ab(CUR(d), k);
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
		const uint32_t r = hc_root_of(d, con_ab(CUR(d), k));
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
		if (k < e && d.hc_rank[k] == 0u) { r = hc_root_of(d, con_ab(CUR(d), k)); if (r != HC_NONE) size = d.hc_count[r]; }
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
		const uint32_t r = hc_root_of(d, con_ab(CUR(d), k));
		const uint32_t place = r == HC_NONE ? HC_BIG : d.hc_base[r];
		uint32_t at = HC_NONE;
		if (place != HC_BIG) {
			const int cls = (int)(place >> 28);
			at = hc_class_first(d, cls) + ((place & 0x0FFFFFFFu) << cls) + d.hc_rank[k];
		}
		if (at < d.cap_hc_list) d.hc_list[at] = k;          // (the list has room for every constraint rounded up to its class: at is always inside)
		else {
			con_npc(CUR(d), k) |= NPCOL_CATCH_ALL;
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
		const uint2 ab = con_ab(CUR(d), k);
		if (hc_can_move(d, ab.x)) { d.hc_root[ab.x] = ab.x; d.hc_count[ab.x] = 0u; }
		if (hc_can_move(d, ab.y)) { d.hc_root[ab.y] = ab.y; d.hc_count[ab.y] = 0u; }
	}
}
__global__ void __launch_bounds__(TPB) k_hc_probe(DV d, int first_colour)
{
	const uint32_t b = d.cstarts[first_colour], e = d.cstarts[SGP_OVERFLOW_COLOUR];
	for (uint32_t k = b + blockIdx.x * TPB + threadIdx.x; k < e; k += gridDim.x * TPB) {
		if (d.hc_rank[k] != 0u) continue;
		const uint32_t r = hc_root_of(d, con_ab(CUR(d), k));
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
		const uint4 hd = slot != HC_NONE ? con_hdr(CUR(d), slot) : make_uint4(0u, 0u, 0u, 0u);
		const int npc = (int)hd.z;
		const uint2 ab = make_uint2(hd.x, hd.y);
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
template <int MODE, int ROWS = -1> __global__ void __launch_bounds__(HC_TPB) k_solve_hc(const StepCounters* ctr_pre, const uint4* hc_entry_pre, DV d, int first_colour)      // (leading pointers: preloaded with the wave, k_solve_colour; `d` itself stays as passed -- the catch-all below takes it by reference, and a modified copy would live in scratch)
{
	constexpr int RS = MODE == 1 ? 2 : 3;      // float4 per body: velocity half (lin + inverse mass, ang) / pose half (pos + inverse mass, rot, inertia)
	__shared__ float4 s_rec[HC_TABLE * RS];
	__shared__ uint32_t s_key[HC_TABLE];
	__shared__ unsigned long long s_present;
	__shared__ uint32_t s_ticket;
	const int side = (int)(threadIdx.x & 1u);
	const uint32_t pair = threadIdx.x >> 1;
	const uint32_t entries = ctr_pre->hc_entries;
	const int n_colours = (int)ctr_pre->n_colours;
	for (uint32_t e0 = blockIdx.x * HC_WG_PAIRS; e0 < entries; e0 += gridDim.x * HC_WG_PAIRS) {
		__syncthreads();      // (everyone is done with the previous round's table and mask)
		for (uint32_t i = threadIdx.x; i < HC_TABLE; i += HC_TPB) s_key[i] = HC_NONE;
		if (threadIdx.x == 0) s_present = 0ull;
		__syncthreads();
		const uint4 entry = hc_entry_pre[e0 + pair];
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
				const float4* g = (MODE == 1 ? d.vel + VEL_F4 * (size_t)body : d.pose + POSE_F4 * (size_t)body);
				s_rec[RS * at] = g[0]; s_rec[RS * at + 1] = g[1];
				if (MODE != 1) s_rec[RS * at + 2] = d.pose[POSE_F4 * (size_t)body + 2];
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
			float4* g = (MODE == 1 ? d.vel + VEL_F4 * (size_t)body : d.pose + POSE_F4 * (size_t)body);
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
		for (uint32_t e = pair; e < n_big; e += HC_WG_PAIRS) { const uint32_t k = d.hc_big_list[e]; mine[cnt] = k; mcol[cnt] = (int)((con_npc(CUR(d), k) >> 8) & 0xFF); ++cnt; }
		for (int c = first_colour; c < n_colours; ++c) {
#pragma unroll
			for (int j = 0; j < 4; ++j) if (j < cnt && mcol[j] == c) { if (MODE == 1) solve_velocity_pair_t<VEL_F4, ROWS>(d, mine[j], side, d.vel); else solve_position_pair(d, mine[j], side); }
			__syncthreads();
		}
	} else
	for (int c = first_colour; c < n_colours && n_big != 0u; ++c) {
		const uint32_t cb = d.cstarts[c], ce = d.cstarts[c + 1];
		for (uint32_t k = cb + pair; k < ce; k += HC_WG_PAIRS) {
			if (!(con_npc(CUR(d), k) & NPCOL_CATCH_ALL)) continue;
			if (MODE == 1) solve_velocity_pair_t<VEL_F4, ROWS>(d, k, side, d.vel); else solve_position_pair(d, k, side);
		}
		__syncthreads();
	}
	if (ocount == 0u || threadIdx.x >= 2u) return;               // lanes 0 and 1: the two sides of one constraint at a time
	uint64_t last = 0; bool have_last = false;
	for (uint32_t it = 0; it < ocount; ++it) {
		const uint32_t bslot = overflow_next(d, ofirst, ocount, last, have_last);
		if (MODE == 1) solve_velocity_pair_t<VEL_F4, ROWS>(d, bslot, side, d.vel); else solve_position_pair(d, bslot, side);
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
	for (uint32_t i = threadIdx.x; i < 2 * n; i += SMALL_TPB) sv[i] = d.vel[VEL_F4 * (size_t)(i >> 1) + (i & 1u)];
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
	for (uint32_t i = threadIdx.x; i < 2 * n; i += SMALL_TPB) d.vel[VEL_F4 * (size_t)(i >> 1) + (i & 1u)] = sv[i];
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
	const uint4 hd = con_hdr(CUR(d), slot);
	r.ab = make_uint2(hd.x, hd.y);
	r.nf = CUR(d).n_fric[slot];
	r.np_col = (int)hd.z;
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
	for (uint32_t i = threadIdx.x; i < 2 * n; i += 512) sv[i] = d.vel[VEL_F4 * (size_t)(i >> 1) + (i & 1u)];
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
							const uint2 okab = con_ab(CUR(d), first + k); const uint64_t pr = sgp_mix64(((uint64_t)okab.x << 32) | okab.y);
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
	for (uint32_t i = threadIdx.x; i < 2 * n; i += 512) d.vel[VEL_F4 * (size_t)(i >> 1) + (i & 1u)] = sv[i];
}
#ifndef SGP_EXPERIMENTS
// (the solver probe and the resident tile solver of round 3 are experiments: built only with -DSGP_EXPERIMENTS, as the unity file sgp_kernels_experiments.hip;
// a plain build never plans them, sgp_world.hip)
void launch_solve_probe(const DV&, int, int, uint32_t, hipStream_t) {}
void launch_ts_label(const DV&, uint32_t, hipStream_t) {}
void launch_colour_count_ts(const DV&, uint32_t, hipStream_t) {}
void launch_setup_ts(const DV&, uint32_t, hipStream_t) {}
void launch_ts_solve(const DV&, int, int, hipStream_t) {}
#endif
#define SOLVE_MANY_MIN 131072u      // constraints in a colour from which the launch runs in rounds (256 CUs x 4 SIMDs x 2-3 waves x 32 constraints = 65k-98k in flight)
#ifndef SOLVE_MANY_OCC
#define SOLVE_MANY_OCC 3
#endif
void launch_solve_colour(const DV& d, int colour, uint32_t est, int mode, hipStream_t s, int compact_rows)
{
	uint32_t blocks = (est + est / 8 + 64 + SOLVE_TPB - 1) / SOLVE_TPB;      // warm start: one thread per constraint, one wave per workgroup
	if (mode != 0) blocks = (est + est / 8 + 64 + SOLVE_VEL_TPB / 2 - 1) / (SOLVE_VEL_TPB / 2);
	if (blocks > 8192) blocks = 8192;
	if (mode != 0) blocks = (blocks + 7u) & ~7u;      // (XCD-contiguous chunks: k_solve_colour)
	if (mode != 0 && est >= 8192u && est <= 65536u) colour |= SOLVE_XCD_CHUNKS;      // (the colour's bodies and rows then fit the eight L2s)
	if (mode == 0) hipLaunchKernelGGL(k_solve_colour<0>, dim3(blocks), dim3(SOLVE_TPB), 0, s, (const uint32_t*)d.cstarts, d.sp, d, colour);
	else if (mode == 1) {
		const bool many = est > SOLVE_MANY_MIN;
		if (compact_rows == 2) hipLaunchKernelGGL((k_solve_colour<1, 2>), dim3(blocks), dim3(SOLVE_VEL_TPB), 0, s, (const uint32_t*)d.cstarts, d.sp, d, colour);      // (108 VGPRs: four waves per SIMD as it is, solve_velocity_pair_norows; five spill and lose)
		else if (compact_rows) { if (many) hipLaunchKernelGGL((k_solve_colour<1, 1, SOLVE_MANY_OCC>), dim3(blocks), dim3(SOLVE_VEL_TPB), 0, s, (const uint32_t*)d.cstarts, d.sp, d, colour); else hipLaunchKernelGGL((k_solve_colour<1, 1>), dim3(blocks), dim3(SOLVE_VEL_TPB), 0, s, (const uint32_t*)d.cstarts, d.sp, d, colour); }
		else hipLaunchKernelGGL((k_solve_colour<1, 0>), dim3(blocks), dim3(SOLVE_VEL_TPB), 0, s, (const uint32_t*)d.cstarts, d.sp, d, colour);
	}
	else if (est > SOLVE_MANY_MIN) hipLaunchKernelGGL((k_solve_colour<2, -1, SOLVE_MANY_OCC>), dim3(blocks), dim3(SOLVE_VEL_TPB), 0, s, (const uint32_t*)d.cstarts, d.sp, d, colour);
	else hipLaunchKernelGGL(k_solve_colour<2>, dim3(blocks), dim3(SOLVE_VEL_TPB), 0, s, (const uint32_t*)d.cstarts, d.sp, d, colour);
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
		if (compact_rows == 2) hipLaunchKernelGGL((k_solve_hc<1, 2>), dim3(blocks), dim3(HC_TPB), 0, s, (const StepCounters*)d.ctr, (const uint4*)d.hc_entry, d, first_colour);
		else if (compact_rows) hipLaunchKernelGGL((k_solve_hc<1, 1>), dim3(blocks), dim3(HC_TPB), 0, s, (const StepCounters*)d.ctr, (const uint4*)d.hc_entry, d, first_colour);
		else hipLaunchKernelGGL((k_solve_hc<1, 0>), dim3(blocks), dim3(HC_TPB), 0, s, (const StepCounters*)d.ctr, (const uint4*)d.hc_entry, d, first_colour);
	}
	else hipLaunchKernelGGL((k_solve_hc<2, -1>), dim3(blocks), dim3(HC_TPB), 0, s, (const StepCounters*)d.ctr, (const uint4*)d.hc_entry, d, first_colour);
}
void launch_solve_small(const DV& d, int warm_start, int iterations, int lane_pairs, hipStream_t s)
{
	if (lane_pairs) hipLaunchKernelGGL(k_solve_small, dim3(1), dim3(SMALL_TPB), 0, s, d, warm_start, iterations);
	else hipLaunchKernelGGL(k_solve_small_single, dim3(1), dim3(512), 0, s, d, warm_start, iterations);
}
