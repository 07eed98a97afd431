// sgp_dev_common.h -- flag accessors, layer table, shape bounds / sleep points, event lists, wave- and block-level slot allocation: what every stage uses.
// Device-inline functions only (no kernels), shared between stage files; included through sgp_dev_all.h, whose order is the dependency order.
#pragma once
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
// a constraint's header: the two body ids (8 bytes), np_col, or all of it in one 16-byte load
SGP_DEV uint2 con_ab(const ConstraintArrays& c, size_t k) { return *(const uint2*)(c.hdr + k); }
SGP_DEV int& con_npc(const ConstraintArrays& c, size_t k) { return ((int*)(c.hdr + k))[2]; }
SGP_DEV uint4 con_hdr(const ConstraintArrays& c, size_t k) { return c.hdr[k]; }

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

// ---------------------------------------------------------------------------------------------------------------
// launch wrappers

static inline uint32_t blocks_for(uint32_t n) { return n ? (n + TPB - 1) / TPB : 1; }
static inline uint32_t stride_grid(uint32_t estimate) { uint32_t b = blocks_for(estimate); if (b < 64) b = 64; if (b > 4096) b = 4096; return (b + 7u) & ~7u; }      // (a multiple of eight: xcd_block)
