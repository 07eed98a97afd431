// sgp_k_sweep.hip -- K8 / K1 / K9 / A3 -- the body-array sweep (k_pre_solve, k_integrate_pose, k_finalize), islands and sleeping, buoyancy.
// One of the stage files of the step kernels (stage map: sgp_kernels.h).  Kernels first, their launch wrappers at the end.
#include "sgp_dev_all.h"

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
	const float4 lv4 = d.vel[VEL_F4 * (size_t)i], av4 = d.vel[VEL_F4 * (size_t)i + 1];
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
		reset_sleep(d, i, f_shape(f), d.pose[POSE_F4 * (size_t)i + 3], V3(d.pose[POSE_F4 * (size_t)i]), Q4(d.pose[POSE_F4 * (size_t)i + 1]));
	}
	if ((f & BF_ACTIVE) && f_motion(f) != SGP_MOTION_STATIC) {
		if (!(f & BF_ALIAS)) d.ctr->any_awake = 1u;      // (every awake lane stores the same word: no atomic)
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
					Iw = world_inv_inertia(quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)i + 1])), V3(d.pose[POSE_F4 * (size_t)i + 2]));
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
		d.vel[VEL_F4 * (size_t)i] = F4(lv, im);
		d.vel[VEL_F4 * (size_t)i + 1] = F4(av, 0.0f);
		if (im > 0.0f && d.sp->compact_rows != 0u) {
			// compact rows: the lanes of the velocity iterations rebuild I (r x axis) -- from this record (the expression k_setup evaluates on the same
			// pose and property records, hence the same bits), one 32-byte gather instead of 48 bytes and a rotation matrix per lane and launch
			const sym33 I = world_inv_inertia(quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)i + 1])), V3(d.pose[POSE_F4 * (size_t)i + 2]));
			d.vel[VEL_F4 * (size_t)i + 2] = make_float4(I.xx, I.xy, I.xz, I.yy);
			d.vel[VEL_F4 * (size_t)i + 3] = make_float4(I.yz, I.zz, 0.0f, 0.0f);
		}
	}
	d.hc_root[i] = i; d.hc_count[i] = 0u;      // every body a component of its own (k_hc_hook joins them along the high-colour constraints)
	// remember whether the body was movable when the previous step coloured its constraints (colour inheritance)
	uint32_t nf = f & ~(BF_MOVABLE_PREV | BF_MOVABLE_CUR | BF_CACHE_INVALID | BF_AWAKE_STEP | BF_FRESH);      // (the narrow phase of this step has seen the flag)
	if ((f & BF_ACTIVE) && f_motion(f) != SGP_MOTION_STATIC && !(f & BF_ALIAS)) nf |= BF_AWAKE_STEP;      // (what k_cache_build asks of the bodies of a cached contact)
	if (f & BF_CACHE_INVALID) nf |= BF_FRESH;
	if (f & BF_MOVABLE_CUR) nf |= BF_MOVABLE_PREV;
	if (f_movable(f)) nf |= BF_MOVABLE_CUR;
	if (nf != f0) d.flags[i] = nf;
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
	const float4 v4 = d.vel[VEL_F4 * (size_t)i], w4 = d.vel[VEL_F4 * (size_t)i + 1];
	float4 p = d.pose[POSE_F4 * (size_t)i], r4 = d.pose[POSE_F4 * (size_t)i + 1];
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
		if (cl) d.vel[VEL_F4 * (size_t)i] = F4(lv, v4.w);                       // (only a clamped velocity changes)
		if (ca) d.vel[VEL_F4 * (size_t)i + 1] = F4(av, w4.w);
	}
	const v3 np = v3_add(V3(p), v3_scale(lv, dt));
	const quat q = quat_add_rotation_step(Q4(r4), v3_scale(av, dt));
	d.pose[POSE_F4 * (size_t)i] = F4(np, p.w);
	d.pose[POSE_F4 * (size_t)i + 1] = make_float4(q.x, q.y, q.z, q.w);
	if (f_shape(f) == SGP_SHAPE_MESH) {
		// a kinematic mesh body (a scripted platform): the two alias slots behind it -- second / third contact manifold of a pair -- share its pose
		for (uint32_t k = 1; k <= 2; ++k) { d.pose[POSE_F4 * (size_t)(i + k)] = F4(np, p.w); d.pose[POSE_F4 * (size_t)(i + k) + 1] = make_float4(q.x, q.y, q.z, q.w); }
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
	const float4 sh = d.pose[POSE_F4 * (size_t)i + 3];
	const float4 p4 = d.pose[POSE_F4 * (size_t)i], r4 = d.pose[POSE_F4 * (size_t)i + 1];      // (the position iterations corrected the pose records in place)
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
// Edge k of the island graph: the contact constraints, then one link per wheel of an active vehicle that stands on a dynamic body (chassis, that
// body) -- VehicleConstraint::BuildIslands links them, so a car and the loose box under its wheel fall asleep together or not at all.
SGP_DEV uint32_t island_edges(const DV& d) { return d.ctr->n_constraints + 4u * d.n_vehicles; }
SGP_DEV bool island_edge(const DV& d, uint32_t k, uint32_t n_con, uint2& ab)
{
	if (k < n_con) { ab = con_ab(CUR(d), k); return true; }
	const uint32_t e = k - n_con, v = e >> 2, i = e & 3u;
	const float4 h0 = d.veh_head[(size_t)v * VEH_HEAD_F4];
	if (!(__float_as_uint(h0.y) & 1u) || i >= __float_as_uint(h0.z)) return false;
	const uint32_t wbits = __float_as_uint(d.veh_rows[veh_chunk_at(d, v, (int)i, VEH_CHUNK_NORMAL)].w);
	if (!(wbits >> 5)) return false;
	ab = make_uint2(__float_as_uint(h0.x), (wbits >> 5) - 1u);
	return true;
}
__global__ void __launch_bounds__(TPB) k_island_mark(DV d, int clear_cache)
{
	// The first of the marking launches also empties the contact-cache table for this step's rebuild (nothing reads the old table after the set-up;
	// the stores ride along with a launch that waits for its gathers: a launch of its own was 6 us on the step's chain)
	if (clear_cache && d.ctr->any_awake) {      // (nobody awake: the step leaves the contact cache as it found it, StepCounters::any_awake)
		const uint32_t size = cache_table_size(d);
		for (uint32_t i = blockIdx.x * TPB + threadIdx.x; i < size; i += gridDim.x * TPB) d.ht[i] = make_uint4(~0u, ~0u, 0u, 0u);
		if (blockIdx.x == 0 && threadIdx.x == 0) { *d.ht_cur = size; d.cache_total[d.sp->parity] = d.ctr->n_constraints; }      // (k_cache_build appends the carried entries)
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
			d.sleep_label[i] = SGP_LABEL(root, d.slot_gen[root] & 0x7Fu);        // the island goes to sleep as a whole and is remembered by its root: what wakes a member wakes them all (k_wake_pairs)
			// (the record of a body that is not awake reads (0, 0, 0 | effective inverse mass 0): k_pre_solve then has nothing to write for it)
			d.vel[VEL_F4 * (size_t)i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			d.vel[VEL_F4 * (size_t)i + 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			push_event(d.ev_deactivated, &d.evc->n_deactivated, d.cap_bodies, i);
		}
	} else if (f_motion(f) == SGP_MOTION_KINEMATIC && (f & BF_ACTIVE)) {
		const v3 lv = V3(d.vel[VEL_F4 * (size_t)i]), av = V3(d.vel[VEL_F4 * (size_t)i + 1]);
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
		const float4 sh = d.pose[POSE_F4 * (size_t)i + 3];
		const float4 pim = d.pose[POSE_F4 * (size_t)i];
		const v3 pos = V3(pim);
		const m33 R = quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)i + 1]));
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
			float4 lv4 = d.vel[VEL_F4 * (size_t)i], av4 = d.vel[VEL_F4 * (size_t)i + 1];
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
			const sym33 Iw = world_inv_inertia(R, V3(d.pose[POSE_F4 * (size_t)i + 2]));
			v3 ddrag = sym33_mul(Iw, drag_ang_imp);
			if (v3_len_sq(ddrag) > v3_len_sq(av)) ddrag = v3_neg(av);
			const v3 dang = v3_add(ddrag, sym33_mul(Iw, v3_cross(rc, v3_add(buoy_imp, drag_imp))));
			d.vel[VEL_F4 * (size_t)i] = F4(v3_add(lv, dlin), lv4.w);
			d.vel[VEL_F4 * (size_t)i + 1] = F4(v3_add(av, dang), av4.w);
			applied = true;
		}
		if (applied) {
			if (!(f & BF_UNDERWATER)) { push_event(d.ev_water, &d.evc->n_water, d.cap_bodies, i); f |= BF_UNDERWATER; d.flags[i] = f; }
			d.submerged[i] = sub;
		} else { if (f & BF_UNDERWATER) d.flags[i] = f & ~BF_UNDERWATER; d.submerged[i] = 0.0f; }
	} else if (f & BF_UNDERWATER) { d.flags[i] = f & ~BF_UNDERWATER; d.submerged[i] = 0.0f; }
}
void launch_pre_solve(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_pre_solve, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_integrate_pose(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_integrate_pose, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_finalize(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_finalize, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_island_mark(const DV& d, uint32_t n_con, int clear_cache, hipStream_t s) { hipLaunchKernelGGL(k_island_mark, dim3(stride_grid(n_con)), dim3(TPB), 0, s, d, clear_cache); }
void launch_island_hook(const DV& d, uint32_t n_con, hipStream_t s) { hipLaunchKernelGGL(k_island_hook, dim3(stride_grid(n_con)), dim3(TPB), 0, s, d); }
void launch_island_flag(const DV& d, uint32_t n_con, hipStream_t s) { hipLaunchKernelGGL(k_island_flag, dim3(stride_grid(n_con)), dim3(TPB), 0, s, d); }
void launch_sleep_apply(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_sleep_apply, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
void launch_buoyancy(const DV& d, uint32_t nb, hipStream_t s) { hipLaunchKernelGGL(k_buoyancy, dim3(blocks_for(nb)), dim3(TPB), 0, s, d); }
