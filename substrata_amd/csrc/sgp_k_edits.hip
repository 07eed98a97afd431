// sgp_k_edits.hip -- A5 / A6 -- host edit commands, ghost refresh, read-back.
// One of the stage files of the step kernels (stage map: sgp_kernels.h).  Kernels first, their launch wrappers at the end.
#include "sgp_dev_all.h"

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
	d.pose[POSE_F4 * (size_t)i] = make_float4(c.pos[0], c.pos[1], c.pos[2], d.pose[POSE_F4 * (size_t)i].w);
	d.pose[POSE_F4 * (size_t)i + 1] = make_float4(c.rot[0], c.rot[1], c.rot[2], c.rot[3]);
	if (f_motion(f) != SGP_MOTION_STATIC) {
		d.vel[VEL_F4 * (size_t)i] = make_float4(c.linv[0], c.linv[1], c.linv[2], d.vel[VEL_F4 * (size_t)i].w);
		d.vel[VEL_F4 * (size_t)i + 1] = make_float4(c.angv[0], c.angv[1], c.angv[2], d.vel[VEL_F4 * (size_t)i + 1].w);
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
			d.pose[POSE_F4 * (size_t)i] = make_float4(c.pos[0], c.pos[1], c.pos[2], c.inv_mass);
			d.pose[POSE_F4 * (size_t)i + 1] = make_float4(c.rot[0], c.rot[1], c.rot[2], c.rot[3]);
			d.vel[VEL_F4 * (size_t)i] = make_float4(c.linv[0], c.linv[1], c.linv[2], 0.0f);        // (effective inverse mass: set by k_pre_solve once the body is awake)
			d.vel[VEL_F4 * (size_t)i + 1] = make_float4(c.angv[0], c.angv[1], c.angv[2], 0.0f);
			d.dyn[i] = make_float4(c.lin_damp, c.ang_damp, c.gravity_factor, c.inv_mass);
			d.force[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
			d.torque[i] = make_float4(0.0f, 0.0f, 0.0f, c.mass);
			d.pose[POSE_F4 * (size_t)i + 2] = make_float4(c.inv_inertia[0], c.inv_inertia[1], c.inv_inertia[2], c.restitution);
			d.pose[POSE_F4 * (size_t)i + 3] = make_float4(c.shape[0], c.shape[1], c.shape[2], c.friction);
			d.submerged[i] = 0.0f;
			d.userdata[i] = c.userdata;
			label_new_body(d, i);
			refresh_aabb(d, i, f);
			reset_sleep(d, i, f_shape(f), d.pose[POSE_F4 * (size_t)i + 3], V3(d.pose[POSE_F4 * (size_t)i]), Q4(d.pose[POSE_F4 * (size_t)i + 1]));
			continue;
		}
		if (c.ops & CMD_REMOVE) { f = 0; continue; }
		if (!(f & BF_ALIVE)) continue;
		if (c.ops & CMD_SET_CHASSIS) { f = (f & ~BF_CHASSIS) | (c.flags & BF_CHASSIS); continue; }
		if (c.ops & CMD_SET_LAYER) f = (f & ~BF_LAYER_MASK) | ((c.flags & 0x3u) << BF_LAYER_SHIFT);
		if (c.ops & CMD_MOVE_KINEMATIC) {
			// MotionProperties::MoveKinematic: velocities that reach the target in dt
			if (f_motion(f) == SGP_MOTION_KINEMATIC && c.dt > 0.0f) {
				const v3 pos = V3(d.pose[POSE_F4 * (size_t)i]);
				const quat q = Q4(d.pose[POSE_F4 * (size_t)i + 1]);
				const v3 lv = v3_scale(v3_sub(V3(c.pos[0], c.pos[1], c.pos[2]), pos), 1.0f / c.dt);
				quat t; t.x = c.rot[0]; t.y = c.rot[1]; t.z = c.rot[2]; t.w = c.rot[3];
				quat cj; cj.x = -q.x; cj.y = -q.y; cj.z = -q.z; cj.w = q.w;
				quat dq = quat_mul(t, cj);
				if (dq.w < 0.0f) { dq.x = -dq.x; dq.y = -dq.y; dq.z = -dq.z; dq.w = -dq.w; }
				const float sl = sqrtf(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z);
				v3 av = V3(0.0f, 0.0f, 0.0f);
				if (sl > 1.0e-12f) { const float angle = sgd_quat_angle(sl, dq.w); av = v3_scale(V3(dq.x / sl, dq.y / sl, dq.z / sl), angle / c.dt); }
				d.vel[VEL_F4 * (size_t)i] = F4(lv, d.vel[VEL_F4 * (size_t)i].w);
				d.vel[VEL_F4 * (size_t)i + 1] = F4(av, d.vel[VEL_F4 * (size_t)i + 1].w);
				if (!(f & BF_ALIAS)) f = activate_body(d, i, f);      // (a mesh body's alias slots follow its pose and velocities, they are never awake themselves)
			}
			continue;
		}
		bool pose = false;
		if (c.ops & CMD_SET_POS) { d.pose[POSE_F4 * (size_t)i] = make_float4(c.pos[0], c.pos[1], c.pos[2], d.pose[POSE_F4 * (size_t)i].w); pose = true; }
		if (c.ops & CMD_SET_ROT) { d.pose[POSE_F4 * (size_t)i + 1] = make_float4(c.rot[0], c.rot[1], c.rot[2], c.rot[3]); pose = true; }
		if (c.ops & CMD_SET_SHAPE) {
			d.pose[POSE_F4 * (size_t)i + 3] = make_float4(c.shape[0], c.shape[1], c.shape[2], d.pose[POSE_F4 * (size_t)i + 3].w); pose = true;
			f = ((f & ~BF_LARGE) | (c.flags & BF_LARGE)) | BF_CACHE_INVALID;      // a new scale can move the body across the broad phase's large-body radius (host: note_radius)
		}
		if ((c.ops & CMD_SET_VEL) && f_motion(f) != SGP_MOTION_STATIC) {
			d.vel[VEL_F4 * (size_t)i] = make_float4(c.linv[0], c.linv[1], c.linv[2], d.vel[VEL_F4 * (size_t)i].w);
			d.vel[VEL_F4 * (size_t)i + 1] = make_float4(c.angv[0], c.angv[1], c.angv[2], d.vel[VEL_F4 * (size_t)i + 1].w);
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
				d.torque[i] = F4(v3_add(V3(T), v3_cross(v3_sub(V3(c.pos[0], c.pos[1], c.pos[2]), V3(d.pose[POSE_F4 * (size_t)i])), Fv)), T.w);
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
	const float4 p = d.pose[POSE_F4 * (size_t)i], q = d.pose[POSE_F4 * (size_t)i + 1], lv = d.vel[VEL_F4 * (size_t)i], av = d.vel[VEL_F4 * (size_t)i + 1];
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
		const float4 p = d.pose[POSE_F4 * (size_t)i];
		out[2 * (size_t)k] = make_float4(p.x, p.y, p.z, __uint_as_float(i));
		out[2 * (size_t)k + 1] = d.pose[POSE_F4 * (size_t)i + 1];
	}
}

struct ConstraintDumpRec { uint32_t a, b; int32_t colour; int32_t np; float n[3]; float lam_n[4]; float lam_t1[4]; float lam_t2[4]; float bias[4]; };

__global__ void __launch_bounds__(TPB) k_dump_constraints(DV d, uint32_t which, uint32_t n_con, ConstraintDumpRec* out, uint32_t cap)
{
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= n_con || k >= cap) return;
	const ConstraintArrays& ca = d.ca[which & 1];
	ConstraintDumpRec r;
	const uint2 ab = con_ab(ca, k);
	const int nc = con_npc(ca, k);
	const float4 nf = ca.n_fric[k];
	r.a = ab.x; r.b = ab.y; r.colour = (nc >> 8) & 0xFF; r.np = nc & 0xFF;
	r.n[0] = nf.x; r.n[1] = nf.y; r.n[2] = nf.z;
	for (int i = 0; i < 4; ++i) {
		if (i < r.np) { const float4 l = ca.lam[i][k]; r.lam_n[i] = l.x; r.lam_t1[i] = l.y; r.lam_t2[i] = l.z; r.bias[i] = ca.r1b[i][k].w; }
		else { r.lam_n[i] = 0; r.lam_t1[i] = 0; r.lam_t2[i] = 0; r.bias[i] = 0; }
	}
	out[k] = r;
}
void launch_ghost_refresh(const DV& d, const GhostRefresh* recs, uint32_t n, hipStream_t s) { if (n) hipLaunchKernelGGL(k_ghost_refresh, dim3(blocks_for(n)), dim3(TPB), 0, s, d, recs, n); }
void launch_apply_cmds(const DV& d, const BodyCmd* cmds, const uint32_t* run_start, uint32_t n_runs, hipStream_t s) { if (n_runs) hipLaunchKernelGGL(k_apply_cmds, dim3(blocks_for(n_runs)), dim3(TPB), 0, s, d, cmds, run_start, n_runs); }
void launch_gather_states(const DV& d, const uint32_t* ids, uint32_t first, uint32_t n, sgp_body_state* out, hipStream_t s) { if (n) hipLaunchKernelGGL(k_gather_states, dim3(blocks_for(n)), dim3(TPB), 0, s, d, ids, first, n, out); }
void launch_gather_active_poses(const DV& d, uint32_t nb, void* out, uint32_t cap, hipStream_t s) { hipLaunchKernelGGL(k_gather_active_poses, dim3(blocks_for(nb)), dim3(TPB), 0, s, d, (float4*)out, cap); }
void launch_gather_active(const DV& d, uint32_t nb, sgp_body_state* out, uint32_t cap, hipStream_t s) { hipLaunchKernelGGL(k_gather_active, dim3(blocks_for(nb)), dim3(TPB), 0, s, d, out, cap); }
void launch_dump_constraints(const DV& d, uint32_t which, uint32_t n_con, void* out, uint32_t cap, hipStream_t s) { if (n_con) hipLaunchKernelGGL(k_dump_constraints, dim3(blocks_for(n_con)), dim3(TPB), 0, s, d, which, n_con, (ConstraintDumpRec*)out, cap); }
