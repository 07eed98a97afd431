// sgp_dev_narrowphase.h -- shape records, the contact-cache probe, wake-ups: what setup, colouring, queries and vehicles share with the narrow phase.
// Device-inline functions only (no kernels), shared between stage files; included through sgp_dev_all.h, whose order is the dependency order.
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// K4: narrow phase, one thread per candidate pair

SGP_DEV sgd_shape load_shape(const DV& d, uint32_t i, uint32_t f)
{
	sgd_shape s;
	s.pos = V3(d.pose[POSE_F4 * (size_t)i]);
	s.R = quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)i + 1]));
	s.type = (int)f_shape(f);
	const float4 sh = d.pose[POSE_F4 * (size_t)i + 3];
	s.p0 = sh.x; s.p1 = sh.y; s.p2 = sh.z;
	s.hull = s.type == SGP_SHAPE_HULL ? body_hull(d, sh) : (s.type == SGP_SHAPE_BOX ? &d.hulls[0] : nullptr);
	return s;
}

SGP_DEV uint32_t cache_find(const DV& d, uint64_t key, int* np_col_prev);
SGP_DEV int man_colour_candidate(int np_col_prev);
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

// a label is current while the slot it names still holds the body (generation) it was made from
SGP_DEV bool label_current(const DV& d, uint32_t lbl) { return (lbl >> 25) == (d.slot_gen[SGP_LABEL_SLOT(lbl)] & 0x7Fu); }
// a new body in slot i: the slot's next generation, and the body's own label
SGP_DEV void label_new_body(const DV& d, uint32_t i) { const uint32_t g = (d.slot_gen[i] + 1u) & 0x7Fu; d.slot_gen[i] = g; d.sleep_label[i] = SGP_LABEL(i, g); }
// A sleeping dynamic body is touched by an awake one (or stands under a wheel): k_pre_solve wakes it, and k_wake_pairs wakes, in the same step, everything
// that fell asleep in the same island (the label's mark carries this step's epoch)
SGP_DEV void wake_body(const DV& d, uint32_t id)
{
	atomicOr(&d.flags[id], BF_WAKE);
	const uint32_t lbl = d.sleep_label[id];
	if (label_current(d, lbl)) d.label_wake[SGP_LABEL_SLOT(lbl)] = *d.veh_epoch;      // (a label whose slot has since been given to another body names nobody: the body wakes alone)
	d.ctr->wake_any = 1u;
}

// the manifold goes to slot `slot` of the step's manifold list (the caller allocated it)
// bit 9 of a manifold's point-count word: a polytope pair (box / hull against box / hull), the pairs the body-pair contact cache serves -- k_setup keeps the second
// sector of the cache record for them only
#define MAN_NP_SENSOR   0x100
#define MAN_NP_POLYTOPE 0x200
SGP_DEV bool pair_is_polytopes(uint32_t fa, uint32_t fb)
{
	return (f_shape(fa) == SGP_SHAPE_BOX || f_shape(fa) == SGP_SHAPE_HULL) && (f_shape(fb) == SGP_SHAPE_BOX || f_shape(fb) == SGP_SHAPE_HULL);
}
// colour_candidate: -1, or man_colour_candidate() of the previous constraint the caller's probe found
SGP_DEV void emit_manifold_at(const DV& d, uint32_t slot, uint2 ab, uint32_t fa, uint32_t fb, const sgd_manifold& m, uint32_t prev, int colour_candidate = -1)
{
	if (slot >= d.cap_manifolds) { atomicAdd(&d.ctr->manifolds_dropped, 1u); return; }
	d.man_ab[slot] = ab;
	const bool sensor = (fa | fb) & BF_SENSOR;
	// bit 8 = sensor pair (mIsSensor, PhysicsWorld.cpp:1235): reported in the contact events, kept in the contact list
	// (so that it is 'persisted' next step) but with zero points for the solver
	d.man_n[slot] = make_float4(m.n.x, m.n.y, m.n.z, __int_as_float(m.np | (sensor ? MAN_NP_SENSOR : 0) | (pair_is_polytopes(fa, fb) ? MAN_NP_POLYTOPE : 0)));
	for (int k = 0; k < 4; ++k) if (k < m.np) { d.man_p1[k][slot] = F4(m.p1[k], 0.0f); d.man_p2[k][slot] = F4(m.p2[k], 0.0f); }
	d.man_prio[slot] = sgp_mix64(((uint64_t)ab.x << 32) | ab.y);
	d.man_prev[slot] = prev;          // (MAN_PREV_LOOKUP: k_colour_inherit -- a light kernel that hides the hash probe's latency -- resolves it)
	d.man_colour[slot] = colour_candidate;
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
