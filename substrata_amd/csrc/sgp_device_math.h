// sgp_device_math.h -- fp32 vector maths for the gfx950 kernels.
//
// Built with -ffp-contract=off: every expression rounds exactly as written, so results are comparable with the
// CPU oracle to rounding.  No libm transcendental on the step path (half-angle sin/cos use a fixed polynomial,
// Jolt Body::AddRotationStep role).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SGP_DEV __device__ __forceinline__

struct v3 { float x, y, z; };
struct quat { float x, y, z, w; };
struct m33 { v3 c0, c1, c2; };                 // columns
struct sym33 { float xx, xy, xz, yy, yz, zz; };

SGP_DEV v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
SGP_DEV v3 V3(float4 a) { return V3(a.x, a.y, a.z); }
SGP_DEV float4 F4(v3 a, float w) { return make_float4(a.x, a.y, a.z, w); }
SGP_DEV v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
SGP_DEV v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
SGP_DEV v3 v3_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
SGP_DEV v3 v3_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
SGP_DEV v3 v3_min(v3 a, v3 b) { return V3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
SGP_DEV v3 v3_max(v3 a, v3 b) { return V3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
SGP_DEV float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
SGP_DEV v3 v3_cross(v3 a, v3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
SGP_DEV float v3_len_sq(v3 a) { return v3_dot(a, a); }
SGP_DEV float v3_len(v3 a) { return sqrtf(v3_dot(a, a)); }
SGP_DEV float v3_get(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
SGP_DEV void v3_set(v3& a, int i, float v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else a.z = v; }
SGP_DEV v3 v3_abs(v3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
/* comparisons, not fminf / fmaxf: for operands that compare equal (+0 and -0, e.g. a friction limit of zero) those may return either one, and the
   choice differs between processors; this form returns the same bits everywhere */
SGP_DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
/* max(v, 0) with a definite sign of zero: fmaxf may return either zero for v = -0, adding +0 turns both into +0 (x + 0 is not folded
   away without fast-math precisely because of that case); one v_max + one v_add on the device, where the select form cost 2 % of the solve */
SGP_DEV float max0f(float v) { return fmaxf(v, 0.0f) + 0.0f; }

SGP_DEV m33 quat_to_m33(quat q)
{
	const float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
	const float xx = q.x * x2, yy = q.y * y2, zz = q.z * z2;
	const float xy = q.x * y2, xz = q.x * z2, yz = q.y * z2;
	const float wx = q.w * x2, wy = q.w * y2, wz = q.w * z2;
	m33 m;
	m.c0 = V3(1.0f - (yy + zz), xy + wz, xz - wy);
	m.c1 = V3(xy - wz, 1.0f - (xx + zz), yz + wx);
	m.c2 = V3(xz + wy, yz - wx, 1.0f - (xx + yy));
	return m;
}
SGP_DEV quat Q4(float4 a) { quat q; q.x = a.x; q.y = a.y; q.z = a.z; q.w = a.w; return q; }
SGP_DEV v3 m33_mul(m33 m, v3 v)
{
	return V3(m.c0.x * v.x + m.c1.x * v.y + m.c2.x * v.z,
	          m.c0.y * v.x + m.c1.y * v.y + m.c2.y * v.z,
	          m.c0.z * v.x + m.c1.z * v.y + m.c2.z * v.z);
}
SGP_DEV v3 m33_tmul(m33 m, v3 v) { return V3(v3_dot(m.c0, v), v3_dot(m.c1, v), v3_dot(m.c2, v)); }
SGP_DEV v3 m33_col(m33 m, int i) { return i == 0 ? m.c0 : (i == 1 ? m.c1 : m.c2); }

SGP_DEV sym33 world_inv_inertia(m33 R, v3 d)
{
	sym33 s;
	s.xx = R.c0.x * d.x * R.c0.x + R.c1.x * d.y * R.c1.x + R.c2.x * d.z * R.c2.x;
	s.xy = R.c0.x * d.x * R.c0.y + R.c1.x * d.y * R.c1.y + R.c2.x * d.z * R.c2.y;
	s.xz = R.c0.x * d.x * R.c0.z + R.c1.x * d.y * R.c1.z + R.c2.x * d.z * R.c2.z;
	s.yy = R.c0.y * d.x * R.c0.y + R.c1.y * d.y * R.c1.y + R.c2.y * d.z * R.c2.y;
	s.yz = R.c0.y * d.x * R.c0.z + R.c1.y * d.y * R.c1.z + R.c2.y * d.z * R.c2.z;
	s.zz = R.c0.z * d.x * R.c0.z + R.c1.z * d.y * R.c1.z + R.c2.z * d.z * R.c2.z;
	return s;
}
SGP_DEV sym33 sym33_zero() { sym33 s; s.xx = s.xy = s.xz = s.yy = s.yz = s.zz = 0.0f; return s; }
SGP_DEV v3 sym33_mul(sym33 s, v3 v)
{
	return V3(s.xx * v.x + s.xy * v.y + s.xz * v.z,
	          s.xy * v.x + s.yy * v.y + s.yz * v.z,
	          s.xz * v.x + s.yz * v.y + s.zz * v.z);
}

SGP_DEV quat quat_mul(quat a, quat b)
{
	quat r;
	r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
	r.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
	r.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
	r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
	return r;
}
SGP_DEV quat quat_normalized(quat q)
{
	const float l = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
	quat r; r.x = q.x / l; r.y = q.y / l; r.z = q.z / l; r.w = q.w / l;
	return r;
}

SGP_DEV void sgp_sincos_poly(float x, float* s, float* c)
{
	if (fabsf(x) > 1.5f) { *s = sinf(x); *c = cosf(x); return; }
	const float x2 = x * x;
	float ps = -2.50521083854417e-8f;
	ps = ps * x2 + 2.75573192239859e-6f;
	ps = ps * x2 - 1.98412698412698e-4f;
	ps = ps * x2 + 8.33333333333333e-3f;
	ps = ps * x2 - 1.66666666666667e-1f;
	ps = ps * x2 + 1.0f;
	*s = ps * x;
	float pc = 2.08767569878681e-9f;
	pc = pc * x2 - 2.75573192239859e-7f;
	pc = pc * x2 + 2.48015873015873e-5f;
	pc = pc * x2 - 1.38888888888889e-3f;
	pc = pc * x2 + 4.16666666666667e-2f;
	pc = pc * x2 - 0.5f;
	pc = pc * x2 + 1.0f;
	*c = pc;
}

SGP_DEV quat quat_add_rotation_step(quat q, v3 w)
{
	const float len = v3_len(w);
	if (len > 1.0e-6f) {
		float s, c;
		sgp_sincos_poly(0.5f * len, &s, &c);
		const float k = s / len;
		quat dq; dq.x = w.x * k; dq.y = w.y * k; dq.z = w.z * k; dq.w = c;
		return quat_normalized(quat_mul(dq, q));
	}
	return q;
}

SGP_DEV v3 v3_normalized_perpendicular(v3 n)
{
	if (fabsf(n.x) > fabsf(n.y)) {
		const float len = sqrtf(n.x * n.x + n.z * n.z);
		return V3(n.z / len, 0.0f, -n.x / len);
	} else {
		const float len = sqrtf(n.y * n.y + n.z * n.z);
		return V3(0.0f, n.z / len, -n.y / len);
	}
}

SGP_DEV uint64_t sgp_mix64(uint64_t z)
{
	z += 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
