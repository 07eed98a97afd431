// sgp_dev_solve.h -- sequential impulses on values: lane pairs (ConHalf / half_solve), position halves, union-find helpers.
// Device-inline functions only (no kernels), shared between stage files; included through sgp_dev_all.h, whose order is the dependency order.
#pragma once

// `vel` / VS: where the velocity records live -- the global array (vel = d.vel) or a workgroup's copy in LDS; VS = float4 per record (VEL_F4 in the global array, 2 in LDS).
// The world-space inverse inertia is derived here from the pose and property records (read-only during the solve).
SGP_DEV sym33 body_world_inv_inertia(const DV& d, uint32_t body)
{
	return world_inv_inertia(quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)body + 1])), V3(d.pose[POSE_F4 * (size_t)body + 2]));
}
// the same matrix from the record k_pre_solve wrote for this step (compact rows only; bodies that cannot move have none and need none: their rows are never applied)
SGP_DEV sym33 body_world_inv_inertia_rec(const DV& d, uint32_t body)
{
	const float4 a = d.vel[VEL_F4 * (size_t)body + 2], b = d.vel[VEL_F4 * (size_t)body + 3];
	sym33 s; s.xx = a.x; s.xy = a.y; s.xz = a.z; s.yy = a.w; s.yz = b.x; s.zz = b.y;
	return s;
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
		const sym33 I = body_world_inv_inertia_rec(d, h.body);
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
		const sym33 I = body_world_inv_inertia_rec(d, h.body);
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
	const uint4 hd = con_hdr(CUR(d), slot);      // ids + np_col: one 16-byte load
	half_load_known<ROWS>(d, slot, side, (int)hd.z, side ? hd.y : hd.x, h);
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
// `vel` / VS: where the velocity records live (the global array d.vel or an LDS copy; VS float4 per record).
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

// The same for the layout without rows (ROWS = 2: worlds of a million constraints and more), written so that nothing is kept that is used once: a colour launch
// loads a constraint, iterates it ONCE and stores it, so r x axis and I (r x axis) -- 72 registers of a ConHalf -- are built where the row is applied instead of
// where the constraint is loaded.  The same expressions on the same operands in the same order as half_load_rows<2> + half_solve_core (each value is computed once
// either way): the same bits.  What it buys is registers: the launch fits four waves per SIMD, and a colour of 300k constraints is nine waves per SIMD.
template <int VS> SGP_DEV void solve_velocity_pair_norows(const DV& d, uint32_t slot, int side, float4* vel)
{
	const uint4 hd = con_hdr(CUR(d), slot);      // ids + np_col: one 16-byte load
	const int np = (int)hd.z & 0xFF;
	const uint32_t body = side ? hd.y : hd.x;
	const float4 nf = CUR(d).n_fric[slot];
	float4 r4[4]; float2 et[4]; v3 lam[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (i == 0 || i < np) { r4[i] = side ? CUR(d).r2e[i][slot] : CUR(d).r1b[i][slot]; et[i] = CUR(d).efft[i][slot]; lam[i] = V3(CUR(d).lam[i][slot]); }
		else { r4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); et[i] = make_float2(0.0f, 0.0f); lam[i] = V3(0.0f, 0.0f, 0.0f); }
	}
	if (np == 0) return;                                    // a sensor pair: kept in the contact list, nothing to solve
	const sym33 I = body_world_inv_inertia_rec(d, body);
	float4 v4 = vel[VS * (size_t)body], w4 = vel[VS * (size_t)body + 1];
	const float im = v4.w, friction = nf.w;
	v3 lv = V3(v4), av = V3(w4);
	const v3 n = V3(nf);
	const v3 t1 = v3_normalized_perpendicular(n);
	const v3 t2 = v3_cross(n, t1);
	float eff0[4], bias[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) { const float ow = lane_swap1(r4[i].w); eff0[i] = side ? r4[i].w : ow; bias[i] = side ? ow : r4[i].w; }
	if (friction > 0.0f) {
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if (i < np && !(et[i].x <= 0.0f && et[i].y <= 0.0f)) {
				const v3 r = V3(r4[i]);
				const v3 c1 = v3_cross(r, t1), c2 = v3_cross(r, t2);
				float l1 = lam[i].y + et[i].x * half_jv(lv, av, t1, c1, side);
				float l2 = lam[i].z + et[i].y * half_jv(lv, av, t2, c2, side);
				const float max_f = friction * lam[i].x;
				const float tot_sq = l1 * l1 + l2 * l2;
				if (tot_sq > max_f * max_f) { const float sc = max_f / sqrtf(tot_sq); l1 = l1 * sc; l2 = l2 * sc; }
				half_apply(lv, av, im, t1, sym33_mul(I, c1), l1 - lam[i].y, side); lam[i].y = l1;
				half_apply(lv, av, im, t2, sym33_mul(I, c2), l2 - lam[i].z, side); lam[i].z = l2;
			}
		}
	}
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		if (i < np && eff0[i] > 0.0f) {
			const v3 c0 = v3_cross(V3(r4[i]), n);
			const float jv = half_jv(lv, av, n, c0, side);
			const float lambda = eff0[i] * (jv - bias[i]);
			const float nl = max0f(lam[i].x + lambda);
			half_apply(lv, av, im, n, sym33_mul(I, c0), nl - lam[i].x, side);
			lam[i].x = nl;
		}
	}
	if (im > 0.0f) { vel[VS * (size_t)body] = F4(lv, im); vel[VS * (size_t)body + 1] = F4(av, 0.0f); }
	if (!side) {
#pragma unroll
		for (int i = 0; i < 4; ++i) { if (i < np) CUR(d).lam[i][slot] = F4(lam[i], 0.0f); }
	}
}

// The position iteration of one manifold on TWO LANES (side 0 / 1 = body 1 / body 2, neighbouring lanes, both must call it): each lane carries
// its own body's pose, computes its own contact point and its own share of the effective mass, swaps them with its neighbour, and corrects
// its own body.  Same operands, same operations as solve_position_one (the effective mass is share of body 1 + share of body 2 there too),
// hence the same bits -- at about half the instructions per lane, which is what a position launch is made of (4700 of them per manifold).
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
SGP_DEV void solve_position_pair(const DV& d, uint32_t slot, int side)
{
	const uint4 hd = con_hdr(CUR(d), slot);      // ids + np_col: one 16-byte load
	const uint32_t body = side ? hd.y : hd.x;
	PosHalf ph;
	pos_half_load(d, slot, side, (int)hd.z, ph);
	pos_half_solve(d, ph, side, d.pose + POSE_F4 * (size_t)body, V3(d.pose[POSE_F4 * (size_t)body + 2]));      // this lane's body's pose record + its local inverse inertia
}
SGP_DEV void solve_position_pair_at(const DV& d, uint32_t slot, int side, float4* rec, v3 ii)
{
	PosHalf ph;
	pos_half_load(d, slot, side, con_npc(CUR(d), slot), ph);
	pos_half_solve(d, ph, side, rec, ii);
}
// (velocity and position iterations: two neighbouring lanes per constraint; workgroups of four waves = 128 constraints.  Measured on config 3:
// one-wave workgroups dispatch ~1 us longer per launch than two-wave ones, two-wave ones another 0.15 us longer than four-wave ones; eight
// waves are as fast for the velocity launches and slower for the position launches)
#define SOLVE_VEL_TPB 256
#define SOLVE_XCD_CHUNKS 0x100      // flag in the colour argument: XCD-contiguous chunks (launch_solve_colour sets it for colours that fit the L2s)

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
