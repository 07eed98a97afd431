// sgp_k_vehicle.hip -- wheel casts, controller, and the vehicle rows inside the solver passes.
// One of the stage files of the step kernels (stage map: sgp_kernels.h).  Kernels first, their launch wrappers at the end.
#include "sgp_dev_all.h"

// rs: what the moving shape reaches around the path of its centre -- the cast sphere's radius, or (wh != nullptr: the wheel itself is cast, sgd_cast_disc) the wheel's
// CYL: the instance that can cast the wheel itself (a world without such a vehicle runs the one without: half the registers)
template <bool CYL> SGP_DEV void veh_cast_test(const DV& d, const sgd_vehicle* v, v3 o, v3 dir, float rs, float cast_len, uint32_t j, float& best, uint32_t& bid, v3& bn, v3& bp, const sgd_wheel* wh = nullptr)
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
	const float4 sh = d.pose[POSE_F4 * (size_t)j + 3];
	const float prm[3] = { sh.x, sh.y, sh.z };
	v3 n, p;
	float t;
	if (CYL && wh) {
		// the wheel itself: always over the whole travel (the search's path must not depend on what another candidate returned); a hit behind the best so far loses
		t = f_shape(f) == SGP_SHAPE_MESH ? cast_disc_mesh(d, j, o, dir, wh->cast_e, wh->cast_din, wh->disc_r, wh->cast_rho, cast_len, &n, &p)
		  : sgd_cast_disc_body((int)f_shape(f), prm, f_shape(f) == SGP_SHAPE_HULL ? body_hull(d, sh) : nullptr, V3(d.pose[POSE_F4 * (size_t)j]), quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)j + 1])), o, dir, wh->cast_e, wh->cast_din, wh->disc_r, wh->cast_rho, cast_len, &n, &p);
		if (t > best) t = -1.0f;
	} else
	t = f_shape(f) == SGP_SHAPE_MESH ? cast_sphere_mesh(d, j, o, dir, best, rs, &n, &p)
	  : sgd_cast_sphere_body((int)f_shape(f), prm, f_shape(f) == SGP_SHAPE_HULL ? body_hull(d, sh) : nullptr, V3(d.pose[POSE_F4 * (size_t)j]), quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)j + 1])), o, dir, best, rs, &n, &p);
	if (t < 0.0f || (!(CYL && wh) && n.z < v->cos_max_slope)) return;      // (VehicleCollisionTesterCastCylinder has no slope limit)
	// closest accepted hit; on equal distance the lower body id wins (the oracle visits ids in ascending order)
	if (t < best || bid == SGP_INVALID_ID || (t == best && j < bid)) { best = t; bid = j; bn = n; bp = p; }
}

SGP_DEV sgd_chassis veh_chassis_pose_vel(const DV& d, uint32_t b)
{
	sgd_chassis c;
	const float4 p = d.pose[POSE_F4 * (size_t)b];
	c.pos = V3(p); c.rot = Q4(d.pose[POSE_F4 * (size_t)b + 1]); c.v = V3(d.vel[VEL_F4 * (size_t)b]); c.w = V3(d.vel[VEL_F4 * (size_t)b + 1]);
	c.im = p.w; c.inv_inertia_local = V3(d.pose[POSE_F4 * (size_t)b + 2]);
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
template <bool CYL> __global__ void __launch_bounds__(64) k_vehicle_cast(DV d)
{
	__shared__ sgd_vehicle sv;
	__shared__ int s_dormant;
	const uint32_t k = blockIdx.x;
	sgd_vehicle* gv = &d.vehicles[k];
	if ((k & 31u) == 0u && threadIdx.x == 0) d.veh_defer_bits[k >> 5] = 0u;      // (k_vehicle_controller, the next launch, sets the bits of this step)
	if (!gv->alive) return;
	veh_stage_in(&sv, gv);
	if (threadIdx.x == 0) {
		const sgp_vehicle_input in = d.vehicle_inputs[k];
		sv.in_forward = in.forward; sv.in_right = in.right; sv.in_brake = in.brake; sv.in_handbrake = in.hand_brake;
		const uint32_t b = sv.body;
		const uint32_t fb = b < d.sp->n_slots ? d.flags[b] : 0u;
		sv.active = f_movable(fb) ? 1 : 0;
		// a vehicle whose chassis sleeps casts too (VehicleConstraint::OnStep runs every step; the constraint is active when the chassis OR a body under
		// a wheel is active, and VehicleConstraint::BuildIslands then activates the chassis): on the staged copy only -- its record stays as it is unless
		// this step wakes it
		s_dormant = (!sv.active && (fb & BF_ALIVE) && !(fb & BF_ACTIVE) && f_motion(fb) == SGP_MOTION_DYNAMIC) ? 1 : 0;
	}
	__syncthreads();
	const bool dormant = s_dormant != 0;
	if (sv.active || dormant) {
		{ const sgd_chassis c = veh_chassis_pose_vel(d, sv.body); sgd_vehicle_precast_lanes(&sv, &c, d.sp->dt, (int)threadIdx.x); }
		__syncthreads();
		const int wi = (int)(threadIdx.x >> 4); const uint32_t sub = threadIdx.x & 15u;
		float best = 0.0f; uint32_t bid = SGP_INVALID_ID; v3 bn = V3(0.0f, 0.0f, 0.0f), bp = bn;
		if (wi < sv.num_wheels) {
			const sgd_wheel* wh = &sv.wheels[wi];
			const v3 o = wh->cast_origin, dir = wh->cast_dir;
			const bool cyl = CYL && sv.tester == SGP_VEHICLE_TESTER_CYLINDER;
			const float rs = cyl ? wh->radius : sv.cast_radius;
			const sgd_wheel* const whc = cyl ? wh : nullptr;
			best = wh->cast_len;
			for (uint32_t l = sub; l < d.sp->n_large; l += 16) veh_cast_test<CYL>(d, &sv, o, dir, rs, wh->cast_len, d.large_ids[l], best, bid, bn, bp, whc);
			{
				// static large bodies under the swept sphere's bounds, dealt to the wheel's 16 lanes in the order the grid yields them
				const v3 e2 = v3_add(o, v3_scale(dir, wh->cast_len));
				const float m2 = rs + 2.0e-3f;
				uint32_t seen = 0;
				large_grid_query(d, V3(fminf(o.x, e2.x) - m2, fminf(o.y, e2.y) - m2, fminf(o.z, e2.z) - m2), V3(fmaxf(o.x, e2.x) + m2, fmaxf(o.y, e2.y) + m2, fmaxf(o.z, e2.z) + m2),
				                 [&](uint32_t i) { if ((seen++ & 15u) == sub) veh_cast_test<CYL>(d, &sv, o, dir, rs, wh->cast_len, i, best, bid, bn, bp, whc); });
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
					grid_row_runs(d, g, x0, x1, y, z, [&](uint32_t q0, uint32_t q1) { for (uint32_t q = q0 + sub; q < q1; q += 16) veh_cast_test<CYL>(d, &sv, o, dir, rs, wh->cast_len, __float_as_uint(d.sorted_max[q].w), best, bid, bn, bp, whc); });
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
		if (dormant) {
			// does a wheel touch something that is awake?  (one wave per vehicle: the vote is the workgroup's)
			const bool hit = wi < sv.num_wheels && sub == 0 && bid != SGP_INVALID_ID;
			const uint32_t fh = hit ? d.flags[bid] : 0u;
			if (__ballot(hit && (fh & BF_ACTIVE) && f_motion(fh) != SGP_MOTION_STATIC) == 0ull) {      // nothing: the vehicle sleeps on, its record untouched ...
				if (threadIdx.x == 0) gv->active = 0;      // ... but for the flag the controller launch reads (it was awake in the step before it fell asleep)
				return;
			}
			if (threadIdx.x == 0) { sv.active = 1; wake_body(d, sv.body); }      // woken like a body an active one touches: by this step's k_pre_solve, before the solve
		}
		if (wi < sv.num_wheels && sub == 0 && bid != SGP_INVALID_ID) {
			const uint32_t fo = d.flags[bid];
			v3 gvel = V3(0.0f, 0.0f, 0.0f);
			if (f_motion(fo) != SGP_MOTION_STATIC) gvel = v3_add(V3(d.vel[VEL_F4 * (size_t)bid]), v3_cross(V3(d.vel[VEL_F4 * (size_t)bid + 1]), v3_sub(bp, V3(d.pose[POSE_F4 * (size_t)bid]))));
			sgd_vehicle_set_hit(&sv, wi, bid, best, bn, bp, gvel, d.pose[POSE_F4 * (size_t)bid + 3].w);
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
		const float lvw = d.vel[VEL_F4 * (size_t)b].w, avw = d.vel[VEL_F4 * (size_t)b + 1].w;
		// lane i: the dynamic body under wheel i, as the row set-up needs it (inverse mass of the body, not of the step: a sleeper is woken by this step's k_pre_solve)
		sgd_ground g; g.dyn = 0; g.pos = V3(0.0f, 0.0f, 0.0f); g.im = 0.0f; g.I = sym33_zero();
		bool lost = false;
		const unsigned long long my_claim = ((unsigned long long)*d.veh_epoch << 32) | (unsigned long long)(0xFFFFFFFFu - blockIdx.x);
		if ((int)threadIdx.x < sv.num_wheels && sv.wheels[threadIdx.x].has_contact && sv.wheels[threadIdx.x].ground_dynamic) {
			const uint32_t gb = sv.wheels[threadIdx.x].contact_body;
			const float4 gp = d.pose[POSE_F4 * (size_t)gb];
			g.dyn = 1; g.pos = V3(gp); g.im = gp.w;
			g.I = world_inv_inertia(quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)gb + 1])), V3(d.pose[POSE_F4 * (size_t)gb + 2]));
			lost = d.veh_claim[gb] != my_claim;
		}
		if (threadIdx.x == 0 && d.veh_claim[b] != my_claim) lost = true;
		// a vehicle that shares a movable body with one of lower index waits for it in every pass (veh_block_solve)
		if (__any(lost)) { sv_deferred = 1; if (threadIdx.x == 0) { atomicOr(&d.veh_defer_bits[blockIdx.x >> 5], 1u << (blockIdx.x & 31u)); atomicAdd(&d.ctr->veh_deferred, 1u); } }
		const int spinning = sgd_vehicle_controller_lanes(&sv, &c, &g, d.sp->dt, (int)threadIdx.x);
		if (threadIdx.x == 0) {
			if (spinning) d.sleep_timer[b] = 0.0f;
			d.vel[VEL_F4 * (size_t)b] = F4(c.v, lvw); d.vel[VEL_F4 * (size_t)b + 1] = F4(c.w, avw);      // (the anti-roll impulses)
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
		const float4 p4 = d.pose[POSE_F4 * (size_t)b], q4 = d.pose[POSE_F4 * (size_t)b + 1];
		const v3 iil = V3(d.pose[POSE_F4 * (size_t)b + 2]);
		v3 pos = V3(p4); quat rot = Q4(q4);
		const float im = p4.w, baumgarte = d.st.baumgarte;
		float4 gp4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), gq4 = make_float4(0.0f, 0.0f, 0.0f, 1.0f); v3 giil = V3(0.0f, 0.0f, 0.0f);
		if (gid != SGP_INVALID_ID) { gp4 = d.pose[POSE_F4 * (size_t)gid]; gq4 = d.pose[POSE_F4 * (size_t)gid + 1]; giil = V3(d.pose[POSE_F4 * (size_t)gid + 2]); }
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
		if (L == 0) { d.pose[POSE_F4 * (size_t)b] = F4(pos, p4.w); d.pose[POSE_F4 * (size_t)b + 1] = make_float4(rot.x, rot.y, rot.z, rot.w); }
		if (gmoved) { d.pose[POSE_F4 * (size_t)gid] = F4(gpos, gp4.w); d.pose[POSE_F4 * (size_t)gid + 1] = make_float4(grot.x, grot.y, grot.z, grot.w); }      // (the lane that moved it: a later lane on the same body started from this pose)
		return;
	}
	// the rows of this lane's wheel
	float4 ra[4], ri[4];
#pragma unroll
	for (int r = 0; r < 4; ++r) { ra[r] = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_ROW + 2 * r)]; ri[r] = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_ROW + 2 * r + 1)]; }
	const float4 s0 = d.vel[VEL_F4 * (size_t)b], s1 = d.vel[VEL_F4 * (size_t)b + 1];
	VehBody c;
	c.v = V3(s0); c.im = s0.w; c.w = V3(s1);                  // (s0.w: the effective inverse mass of this step, k_pre_solve)
	const quat crot = Q4(d.pose[POSE_F4 * (size_t)b + 1]);
	c.I = c.im > 0.0f ? world_inv_inertia(quat_to_m33(crot), V3(d.pose[POSE_F4 * (size_t)b + 2])) : sym33_zero();
	float4 cp = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_CPOS)];
	// the dynamic body under the wheel: velocity record (live), inverse mass, world inverse inertia, lever arm
	VehGround g; g.id = gid; g.v = V3(0.0f, 0.0f, 0.0f); g.w = g.v; g.im = 0.0f; g.I = sym33_zero(); g.r2 = g.v;
	float4 g0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), g1 = g0;
	if (gid != SGP_INVALID_ID) {
		g0 = d.vel[VEL_F4 * (size_t)gid]; g1 = d.vel[VEL_F4 * (size_t)gid + 1];
		const float4 gp4 = d.pose[POSE_F4 * (size_t)gid];
		g.v = V3(g0); g.w = V3(g1); g.im = gp4.w;
		g.I = world_inv_inertia(quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)gid + 1])), V3(d.pose[POSE_F4 * (size_t)gid + 2]));
		g.r2 = v3_sub(V3(cp), V3(gp4));
	}
	if (MODE == 0) {
		// VehicleConstraint::WarmStartVelocityConstraint: suspension, upper stop, lateral (the longitudinal row starts every step from zero)
		const v3 neg_lat = v3_neg(V3(d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_LAT)]));
		VEH_TURNS(if (contact) { if (wbits & 2u) veh_row_apply(c, g, ri[0], neg_n); if (wbits & 4u) veh_row_apply(c, g, ri[1], neg_n); if (wbits & 16u) veh_row_apply(c, g, ri[3], neg_lat); })
		if (L == 0) { d.vel[VEL_F4 * (size_t)b] = F4(c.v, s0.w); d.vel[VEL_F4 * (size_t)b + 1] = F4(c.w, s1.w); }
		if (gid != SGP_INVALID_ID) { d.vel[VEL_F4 * (size_t)gid] = F4(g.v, g0.w); d.vel[VEL_F4 * (size_t)gid + 1] = F4(g.w, g1.w); }
		return;
	}
	const float4 cl = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_LONG)], ct = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_LAT)], cg = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_GVEL)];
	const float4 cm = d.veh_rows[veh_chunk_at(d, k, L, VEH_CHUNK_MISC)];
	const v3 cpos = V3(d.pose[POSE_F4 * (size_t)b]);
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
	if (gid != SGP_INVALID_ID) { d.vel[VEL_F4 * (size_t)gid] = F4(g.v, g0.w); d.vel[VEL_F4 * (size_t)gid + 1] = F4(g.w, g1.w); }      // (lanes on the same body hold the same values)
	if (L == 0) {
		d.vel[VEL_F4 * (size_t)b] = F4(c.v, s0.w); d.vel[VEL_F4 * (size_t)b + 1] = F4(c.w, s1.w);
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
		if (MODE == 1) solve_velocity_pair_t<VEL_F4, ROWS>(d, k, side, d.vel); else solve_position_pair(d, k, side);
	}
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
void launch_vehicle_pre(const DV& d, bool cylinder_testers, hipStream_t s)
{
	if (!d.n_vehicles) return;
	if (cylinder_testers) hipLaunchKernelGGL(k_vehicle_cast<true>, dim3(d.n_vehicles), dim3(64), 0, s, d);
	else hipLaunchKernelGGL(k_vehicle_cast<false>, dim3(d.n_vehicles), dim3(64), 0, s, d);
	hipLaunchKernelGGL(k_vehicle_controller, dim3(d.n_vehicles), dim3(64), 0, s, d);
}
void launch_vehicle_solve(const DV& d, int mode, hipStream_t s)
{
	const dim3 g((d.n_vehicles * 4u + VEH_SOLVE_TPB - 1) / VEH_SOLVE_TPB), b(VEH_SOLVE_TPB);
	if (mode == 0) hipLaunchKernelGGL(k_vehicle_solve<0>, g, b, 0, s, d);
	else if (mode == 1) hipLaunchKernelGGL(k_vehicle_solve<1>, g, b, 0, s, d);
	else hipLaunchKernelGGL(k_vehicle_solve<2>, g, b, 0, s, d);
}
