// sgp_device_collide.h -- gfx950 narrow phase: sphere / box / capsule contact manifolds (<= 4 points).
//
// Role of Jolt v5.3.0 CollideShape + ManifoldBetweenTwoFaces + PruneContactPoints behind
// PhysicsSystem::Update (/root/reference/gui_client/PhysicsWorld.cpp:1363) for the primitives Substrata creates
// natively: box half 0.5*scale (PhysicsWorld.cpp:1249-1255), sphere r 0.5*scale.x (:1221-1227), capsule
// (PlayerPhysics.cpp:31-32,74).  Closed-form closest-feature / separating-axis tests instead of GJK/EPA.
//
// Convention (JPH::ContactManifold): n points from A to B, p1 on A, p2 on B, penetration = dot(p1 - p2, n).
// The arithmetic (expression order included) is the contract checked by tests/test_parity_*.py against the
// CPU oracle; this file is device code only and never runs on the host.
#pragma once
#include "sgp_device_math.h"

#define SGD_SHAPE_SPHERE  0
#define SGD_SHAPE_BOX     1
#define SGD_SHAPE_CAPSULE 2
#define SGD_CAPSULE_SLOP 0.02f

#define SGD_SHAPE_HULL    3   // convex hull (sgp_device_hull.h): `hull` = the shape; for a box `hull` = the +-1 cube template
struct sgd_hull_s;
struct sgd_shape { v3 pos; m33 R; int type; float p0, p1, p2; const sgd_hull_s* hull; };
struct sgd_manifold { v3 n; int np; v3 p1[8]; v3 p2[8]; };

SGP_DEV static int sgd_sphere_sphere_pts(v3 ca, float ra, v3 cb, float rb, float max_sep, sgd_manifold* m)
{
	const v3 d = v3_sub(cb, ca);
	const float dist_sq = v3_len_sq(d);
	const float lim = ra + rb + max_sep;
	if (dist_sq > lim * lim) return 0;
	const float dist = sqrtf(dist_sq);
	v3 n = V3(0.0f, 0.0f, 1.0f);
	if (dist > 1.0e-12f) n = v3_scale(d, 1.0f / dist);
	m->n = n;
	m->np = 1;
	m->p1[0] = v3_add(ca, v3_scale(n, ra));
	m->p2[0] = v3_sub(cb, v3_scale(n, rb));
	return 1;
}

/* A = sphere, B = box */
SGP_DEV static int sgd_sphere_box(const sgd_shape* s, const sgd_shape* b, float max_sep, sgd_manifold* m)
{
	const float r = s->p0;
	const v3 h = V3(b->p0, b->p1, b->p2);
	const v3 cl = m33_tmul(b->R, v3_sub(s->pos, b->pos));
	const v3 q = V3(clampf(cl.x, -h.x, h.x), clampf(cl.y, -h.y, h.y), clampf(cl.z, -h.z, h.z));
	const v3 d = v3_sub(cl, q);
	const float dist_sq = v3_len_sq(d);
	v3 nl; v3 qb = q;
	if (dist_sq > 1.0e-12f) {
		const float dist = sqrtf(dist_sq);
		if (dist - r > max_sep) return 0;
		nl = v3_scale(d, 1.0f / dist);              /* box -> sphere, box local */
	} else {
		/* centre inside the box: leave through the nearest face */
		const float dx = h.x - fabsf(cl.x), dy = h.y - fabsf(cl.y), dz = h.z - fabsf(cl.z);
		int k = 0; float dm = dx;
		if (dy < dm) { dm = dy; k = 1; }
		if (dz < dm) { dm = dz; k = 2; }
		const float sg = v3_get(cl, k) >= 0.0f ? 1.0f : -1.0f;
		nl = V3(0.0f, 0.0f, 0.0f); v3_set(nl, k, sg);
		v3_set(qb, k, sg * v3_get(h, k));
	}
	const v3 n = v3_neg(m33_mul(b->R, nl));          /* sphere -> box */
	m->n = n;
	m->np = 1;
	m->p1[0] = v3_add(s->pos, v3_scale(n, r));
	m->p2[0] = v3_add(b->pos, m33_mul(b->R, qb));
	return 1;
}

SGP_DEV static v3 sgd_closest_on_segment(v3 p0, v3 p1, v3 c)
{
	const v3 d = v3_sub(p1, p0);
	const float dd = v3_len_sq(d);
	float t = 0.0f;
	if (dd > 1.0e-12f) t = clampf(v3_dot(v3_sub(c, p0), d) / dd, 0.0f, 1.0f);
	return v3_add(p0, v3_scale(d, t));
}

/* A = sphere, B = capsule */
SGP_DEV static int sgd_sphere_capsule(const sgd_shape* s, const sgd_shape* c, float max_sep, sgd_manifold* m)
{
	const v3 ax = c->R.c2;
	const v3 p0 = v3_sub(c->pos, v3_scale(ax, c->p1));
	const v3 p1 = v3_add(c->pos, v3_scale(ax, c->p1));
	const v3 q = sgd_closest_on_segment(p0, p1, s->pos);
	return sgd_sphere_sphere_pts(s->pos, s->p0, q, c->p0, max_sep, m);
}

/* Closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9). */
SGP_DEV static void sgd_closest_seg_seg(v3 p1, v3 q1, v3 p2, v3 q2, float* s_out, float* t_out)
{
	const v3 d1 = v3_sub(q1, p1), d2 = v3_sub(q2, p2), r = v3_sub(p1, p2);
	const float a = v3_len_sq(d1), e = v3_len_sq(d2), f = v3_dot(d2, r);
	const float eps = 1.0e-12f;
	float s, t;
	if (a <= eps && e <= eps) { s = 0.0f; t = 0.0f; }
	else if (a <= eps) { s = 0.0f; t = clampf(f / e, 0.0f, 1.0f); }
	else {
		const float c = v3_dot(d1, r);
		if (e <= eps) { t = 0.0f; s = clampf(-c / a, 0.0f, 1.0f); }
		else {
			const float b = v3_dot(d1, d2);
			const float denom = a * e - b * b;
			s = denom > 1.0e-12f ? clampf((b * f - c * e) / denom, 0.0f, 1.0f) : 0.0f;
			t = (b * s + f) / e;
			if (t < 0.0f) { t = 0.0f; s = clampf(-c / a, 0.0f, 1.0f); }
			else if (t > 1.0f) { t = 1.0f; s = clampf((b - c) / a, 0.0f, 1.0f); }
		}
	}
	*s_out = s; *t_out = t;
}

/* A = capsule, B = capsule */
SGP_DEV static int sgd_capsule_capsule(const sgd_shape* a, const sgd_shape* b, float max_sep, sgd_manifold* m)
{
	const v3 axa = a->R.c2, axb = b->R.c2;
	const float ra = a->p0, ha = a->p1, rb = b->p0, hb = b->p1;
	const v3 a0 = v3_sub(a->pos, v3_scale(axa, ha)), a1 = v3_add(a->pos, v3_scale(axa, ha));
	const v3 b0 = v3_sub(b->pos, v3_scale(axb, hb)), b1 = v3_add(b->pos, v3_scale(axb, hb));
	float s, t;
	sgd_closest_seg_seg(a0, a1, b0, b1, &s, &t);
	const v3 ca = v3_add(a0, v3_scale(v3_sub(a1, a0), s));
	const v3 cb = v3_add(b0, v3_scale(v3_sub(b1, b0), t));
	if (!sgd_sphere_sphere_pts(ca, ra, cb, rb, max_sep, m)) return 0;
	/* Two supporting edges (both axes perpendicular to the normal) that run parallel: 2-point manifold. */
	const v3 n = m->n;
	if (fabsf(v3_dot(n, axa)) < SGD_CAPSULE_SLOP && fabsf(v3_dot(n, axb)) < SGD_CAPSULE_SLOP &&
	    fabsf(v3_dot(axa, axb)) > 0.999f) {
		const float u0 = v3_dot(v3_sub(b0, a->pos), axa), u1 = v3_dot(v3_sub(b1, a->pos), axa);
		const float lo = fmaxf(-ha, fminf(u0, u1)), hi = fminf(ha, fmaxf(u0, u1));
		if (hi - lo > 1.0e-4f) {
			const float us[2] = { lo, hi };
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				const v3 pa = v3_add(a->pos, v3_scale(axa, us[i]));
				const float tb = clampf(v3_dot(v3_sub(pa, b->pos), axb), -hb, hb);
				const v3 pb = v3_add(b->pos, v3_scale(axb, tb));
				m->p1[i] = v3_add(pa, v3_scale(n, ra));
				m->p2[i] = v3_sub(pb, v3_scale(n, rb));
			}
			m->np = 2;
		}
	}
	return 1;
}

/* g(t) = 0.5 d/dt dist^2(S(t), box) for S(t) = s0 + t d, box [-h,h]. Monotone non-decreasing in t. */
SGP_DEV static float sgd_seg_box_grad(v3 s0, v3 d, v3 h, float t)
{
	const v3 p = v3_add(s0, v3_scale(d, t));
	const v3 c = V3(p.x - clampf(p.x, -h.x, h.x), p.y - clampf(p.y, -h.y, h.y), p.z - clampf(p.z, -h.z, h.z));
	return v3_dot(c, d);
}

/* A = box, B = capsule */
SGP_DEV static int sgd_box_capsule(const sgd_shape* b, const sgd_shape* c, float max_sep, sgd_manifold* m)
{
	const v3 h = V3(b->p0, b->p1, b->p2);
	const float r = c->p0, hh = c->p1;
	const v3 axw = c->R.c2;
	const v3 cl = m33_tmul(b->R, v3_sub(c->pos, b->pos));
	const v3 axl = m33_tmul(b->R, axw);
	const v3 s0 = v3_sub(cl, v3_scale(axl, hh));
	const v3 d = v3_scale(axl, 2.0f * hh);
	/* minimise the convex piecewise-quadratic dist^2(t) on [0,1] */
	float tstar;
	const float g0 = sgd_seg_box_grad(s0, d, h, 0.0f), g1 = sgd_seg_box_grad(s0, d, h, 1.0f);
	if (g0 >= 0.0f) tstar = 0.0f;
	else if (g1 <= 0.0f) tstar = 1.0f;
	else {
		float lo = 0.0f, glo = g0, hi = 1.0f, ghi = g1;
		for (int i = 0; i < 3; ++i) {
			const float di = v3_get(d, i);
			if (fabsf(di) > 1.0e-12f) {
				for (int sgn = 0; sgn < 2; ++sgn) {
					const float hb = sgn ? v3_get(h, i) : -v3_get(h, i);
					const float tk = (hb - v3_get(s0, i)) / di;
					if (tk > 0.0f && tk < 1.0f) {
						const float gk = sgd_seg_box_grad(s0, d, h, tk);
						if (gk <= 0.0f) { if (tk > lo) { lo = tk; glo = gk; } }
						else { if (tk < hi) { hi = tk; ghi = gk; } }
					}
				}
			}
		}
		const float den = ghi - glo;
		tstar = den > 0.0f ? lo + (hi - lo) * (-glo / den) : lo;
	}
	const v3 S = v3_add(s0, v3_scale(d, tstar));
	const v3 q = V3(clampf(S.x, -h.x, h.x), clampf(S.y, -h.y, h.y), clampf(S.z, -h.z, h.z));
	const v3 dv = v3_sub(S, q);
	const float dist_sq = v3_len_sq(dv);
	if (dist_sq > 1.0e-12f) {
		const float dist = sqrtf(dist_sq);
		if (dist - r > max_sep) return 0;
		const v3 nl = v3_scale(dv, 1.0f / dist);     /* box -> capsule, box local */
		m->n = m33_mul(b->R, nl);
		m->np = 1;
		m->p1[0] = v3_add(b->pos, m33_mul(b->R, q));
		m->p2[0] = v3_add(b->pos, m33_mul(b->R, v3_sub(S, v3_scale(nl, r))));
		/* capsule lying along a face: supporting edge of the capsule clipped against the box face */
		if (hh > 0.0f && fabsf(v3_dot(nl, axl)) < SGD_CAPSULE_SLOP) {
			int k = 0; float nm = fabsf(nl.x);
			if (fabsf(nl.y) > nm) { nm = fabsf(nl.y); k = 1; }
			if (fabsf(nl.z) > nm) { nm = fabsf(nl.z); k = 2; }
			if (nm > 0.95f) {
				const float sg = v3_get(nl, k) >= 0.0f ? 1.0f : -1.0f;
				float t0 = 0.0f, t1 = 1.0f; int ok = 1;
				for (int j = 1; j <= 2 && ok; ++j) {
					const int a = (k + j) % 3;
					const float sa = v3_get(s0, a), da = v3_get(d, a), ha = v3_get(h, a);
					if (fabsf(da) < 1.0e-12f) { if (sa < -ha || sa > ha) ok = 0; }
					else {
						float ta = (-ha - sa) / da, tb = (ha - sa) / da;
						if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
						if (ta > t0) t0 = ta;
						if (tb < t1) t1 = tb;
						if (t0 > t1) ok = 0;
					}
				}
				if (ok && (t1 - t0) * (2.0f * hh) > 1.0e-4f) {
					const float ts[2] = { t0, t1 };
					int np = 0;
#pragma unroll
					for (int i = 0; i < 2; ++i) {
						const v3 P = v3_add(s0, v3_scale(d, ts[i]));
						const float sep = sg * v3_get(P, k) - v3_get(h, k) - r;
						if (sep <= max_sep) {
							v3 pb = P; v3_set(pb, k, sg * v3_get(h, k));
							const v3 x1 = v3_add(b->pos, m33_mul(b->R, pb)), x2 = v3_add(b->pos, m33_mul(b->R, v3_sub(P, v3_scale(nl, r))));
							if (np == 0) { m->p1[0] = x1; m->p2[0] = x2; } else { m->p1[1] = x1; m->p2[1] = x2; }      // (static slots: the manifold stays in registers)
							++np;
						}
					}
					if (np == 2) m->np = 2;
					else {
						/* restore the single closest-point contact */
						m->p1[0] = v3_add(b->pos, m33_mul(b->R, q));
						m->p2[0] = v3_add(b->pos, m33_mul(b->R, v3_sub(S, v3_scale(nl, r))));
						m->np = 1;
					}
				}
			}
		}
		return 1;
	}
	/* core segment touches / pierces the box: leave through the face that needs the least motion */
	{
		int bk = 0; float bs = 1.0f; float bd = 3.4e38f;
		for (int k = 0; k < 3; ++k) {
			const float e0 = v3_get(s0, k), e1 = e0 + v3_get(d, k);
			const float dpos = v3_get(h, k) - fminf(e0, e1);   /* push capsule towards +k */
			const float dneg = v3_get(h, k) + fmaxf(e0, e1);   /* push capsule towards -k */
			if (dpos < bd) { bd = dpos; bk = k; bs = 1.0f; }
			if (dneg < bd) { bd = dneg; bk = k; bs = -1.0f; }
		}
		v3 nl = V3(0.0f, 0.0f, 0.0f); v3_set(nl, bk, bs);
		const float e0 = v3_get(s0, bk), e1 = e0 + v3_get(d, bk);
		v3 P;
		if (fabsf(e0 - e1) < 1.0e-3f) P = v3_add(s0, v3_scale(d, 0.5f));
		else if ((bs > 0.0f) == (e0 < e1)) P = s0;
		else P = v3_add(s0, d);
		v3 pb = P; v3_set(pb, bk, bs * v3_get(h, bk));
		m->n = m33_mul(b->R, nl);
		m->np = 1;
		m->p1[0] = v3_add(b->pos, m33_mul(b->R, pb));
		m->p2[0] = v3_add(b->pos, m33_mul(b->R, v3_sub(P, v3_scale(nl, r))));
		return 1;
	}
}

/* Clip polygon (<= 8 verts, reference-box local space) against  sgn * p[axis] <= lim. */
SGP_DEV static int sgd_clip_poly(const v3* in, int n, int axis, float sgn, float lim, v3* out)
{
	int m = 0;
	for (int i = 0; i < n; ++i) {
		const v3 a = in[i], b = in[(i + 1) % n];
		const float da = sgn * v3_get(a, axis) - lim, db = sgn * v3_get(b, axis) - lim;
		if (da <= 0.0f) { if (m < 8) out[m++] = a; }
		if ((da <= 0.0f) != (db <= 0.0f)) {
			const float t = da / (da - db);
			if (m < 8) out[m++] = v3_add(a, v3_scale(v3_sub(b, a), t));
		}
	}
	return m;
}

/* Keep at most 4 of np points: deepest, farthest from it, and the extreme on either side of that chord. */
SGP_DEV static void sgd_reduce_manifold(sgd_manifold* m)
{
	const int np = m->np;
	if (np <= 4) return;
	const v3 n = m->n;
	int i0 = 0; float best = -3.4e38f;
	for (int i = 0; i < np; ++i) { const float pen = v3_dot(v3_sub(m->p1[i], m->p2[i]), n); if (pen > best) { best = pen; i0 = i; } }
	int i1 = i0; best = -1.0f;
	for (int i = 0; i < np; ++i) { const float d2 = v3_len_sq(v3_sub(m->p1[i], m->p1[i0])); if (d2 > best) { best = d2; i1 = i; } }
	const v3 e = v3_sub(m->p1[i1], m->p1[i0]);
	int i2 = -1, i3 = -1; float amax = 0.0f, amin = 0.0f;
	for (int i = 0; i < np; ++i) {
		if (i == i0 || i == i1) continue;
		const float area = v3_dot(v3_cross(e, v3_sub(m->p1[i], m->p1[i0])), n);
		if (area > amax) { amax = area; i2 = i; }
		if (area < amin) { amin = area; i3 = i; }
	}
	int idx[4]; int k = 0;
	idx[k++] = i0;
	if (i1 != i0) idx[k++] = i1;
	if (i2 >= 0) idx[k++] = i2;
	if (i3 >= 0) idx[k++] = i3;
	v3 q1[4], q2[4];
	for (int i = 0; i < k; ++i) { q1[i] = m->p1[idx[i]]; q2[i] = m->p2[idx[i]]; }
	for (int i = 0; i < k; ++i) { m->p1[i] = q1[i]; m->p2[i] = q2[i]; }
	m->np = k;
}

// ---- polygons in LDS, one column per lane (corner i, component c of lane l at [(3 i + c) * 64 + l]: no bank conflicts), indexed at run time by plain loops ----
// (first used by the triangle - box manifold of the mesh kernels, sgp_device_mesh.h, which tells why; round 6: the box - box clip of k_narrowphase)
#define SGD_LPOLY_FLOATS (8 * 3 * 64)      // one polygon column set for the 64 lanes of a wave
struct sgd_lpoly { float* b; };             // b = the wave's buffer + lane
SGP_DEV static v3 sgd_lp_get(sgd_lpoly a, int i) { return V3(a.b[(3 * i) * 64], a.b[(3 * i + 1) * 64], a.b[(3 * i + 2) * 64]); }
SGP_DEV static void sgd_lp_set(sgd_lpoly a, int i, v3 v) { a.b[(3 * i) * 64] = v.x; a.b[(3 * i + 1) * 64] = v.y; a.b[(3 * i + 2) * 64] = v.z; }
// = sgd_hull_reduce / sgd_reduce_manifold for <= 8 candidate points in LDS; writes m->n, m->np, m->p1 / p2 [0 .. 3] (static slots: the manifold stays in registers)
SGP_DEV static void sgd_lp_reduce(v3 n, sgd_lpoly P1, sgd_lpoly P2, int np, sgd_manifold* m)
{
	m->n = n;
	int pick[4] = { 0, 1, 2, 3 }; int k = np;
	if (np > 4) {
		int i0 = 0; float best = -3.4e38f;
		for (int i = 0; i < np; ++i) { const float pen = v3_dot(v3_sub(sgd_lp_get(P1, i), sgd_lp_get(P2, i)), n); if (pen > best) { best = pen; i0 = i; } }
		const v3 p0 = sgd_lp_get(P1, i0);
		int i1 = i0; best = -1.0f;
		for (int i = 0; i < np; ++i) { const float d2 = v3_len_sq(v3_sub(sgd_lp_get(P1, i), p0)); if (d2 > best) { best = d2; i1 = i; } }
		const v3 e = v3_sub(sgd_lp_get(P1, i1), p0);
		int i2 = -1, i3 = -1; float amax = 0.0f, amin = 0.0f;
		for (int i = 0; i < np; ++i) {
			if (i == i0 || i == i1) continue;
			const float area = v3_dot(v3_cross(e, v3_sub(sgd_lp_get(P1, i), p0)), n);
			if (area > amax) { amax = area; i2 = i; }
			if (area < amin) { amin = area; i3 = i; }
		}
		// the survivors in the order i0, i1 (unless it is i0), i2, i3 (those that exist)
		k = 0;
		pick[0] = i0; k = 1;
		if (i1 != i0) { pick[1] = i1; k = 2; }
		if (i2 >= 0) { if (k == 1) pick[1] = i2; else pick[2] = i2; ++k; }
		if (i3 >= 0) { if (k == 1) pick[1] = i3; else if (k == 2) pick[2] = i3; else pick[3] = i3; ++k; }
	}
#pragma unroll
	for (int j = 0; j < 4; ++j) if (j < k) { m->p1[j] = sgd_lp_get(P1, pick[j]); m->p2[j] = sgd_lp_get(P2, pick[j]); }
	m->np = k;
}

/* A = box, B = box: 15-axis SAT, then reference-face / incident-face clipping or an edge-edge point. */
// LDS = true: the clip polygons and the (up to eight) candidate points live in two LDS polygon columns of the lane (lds = the wave's 2 x SGD_LPOLY_FLOATS + lane)
// instead of arrays indexed at run time, i.e. scratch: at 1 M bodies k_narrowphase moved 8 GB of scratch per launch for 0.9 GB of pairs, bodies and manifolds
// (profiles/NOTES_r06.md 7).  Same expressions, same order of corners and planes: the same bits.
SGP_DEV static int sgd_clip_poly_lds(sgd_lpoly in, int n, int axis, float sgn, float lim, sgd_lpoly out)
{
	int m = 0;
	for (int i = 0; i < n; ++i) {
		const v3 a = sgd_lp_get(in, i), b = sgd_lp_get(in, i + 1 < n ? i + 1 : 0);
		const float da = sgn * v3_get(a, axis) - lim, db = sgn * v3_get(b, axis) - lim;
		if (da <= 0.0f) { if (m < 8) sgd_lp_set(out, m++, a); }
		if ((da <= 0.0f) != (db <= 0.0f)) {
			const float t = da / (da - db);
			if (m < 8) sgd_lp_set(out, m++, v3_add(a, v3_scale(v3_sub(b, a), t)));
		}
	}
	return m;
}
template <bool LDS = false> SGP_DEV static int sgd_box_box(const sgd_shape* A, const sgd_shape* B, float max_sep, sgd_manifold* m, float* lds = nullptr)
{
	const v3 hA = V3(A->p0, A->p1, A->p2), hB = V3(B->p0, B->p1, B->p2);
	const v3 T = v3_sub(B->pos, A->pos);
	float R[3][3], AR[3][3], tA[3], tB[3];
	for (int i = 0; i < 3; ++i) {
		const v3 ai = m33_col(A->R, i);
		tA[i] = v3_dot(T, ai);
		for (int j = 0; j < 3; ++j) { R[i][j] = v3_dot(ai, m33_col(B->R, j)); AR[i][j] = fabsf(R[i][j]) + 1.0e-6f; }
	}
	for (int j = 0; j < 3; ++j) tB[j] = v3_dot(T, m33_col(B->R, j));
	float sA = -3.4e38f, sB = -3.4e38f; int kA = 0, kB = 0;
	for (int i = 0; i < 3; ++i) {
		const float s = fabsf(tA[i]) - (v3_get(hA, i) + (hB.x * AR[i][0] + hB.y * AR[i][1] + hB.z * AR[i][2]));
		if (s > max_sep) return 0;
		if (s > sA) { sA = s; kA = i; }
	}
	for (int j = 0; j < 3; ++j) {
		const float s = fabsf(tB[j]) - (v3_get(hB, j) + (hA.x * AR[0][j] + hA.y * AR[1][j] + hA.z * AR[2][j]));
		if (s > max_sep) return 0;
		if (s > sB) { sB = s; kB = j; }
	}
	float sE = -3.4e38f; int eI = -1, eJ = -1;
	for (int i = 0; i < 3; ++i) {
		const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
		for (int j = 0; j < 3; ++j) {
			const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
			const float l2 = 1.0f - R[i][j] * R[i][j];
			if (l2 < 1.0e-6f) continue;
			const float tl = tA[i2] * R[i1][j] - tA[i1] * R[i2][j];
			const float ra = v3_get(hA, i1) * AR[i2][j] + v3_get(hA, i2) * AR[i1][j];
			const float rb = v3_get(hB, j1) * AR[i][j2] + v3_get(hB, j2) * AR[i][j1];
			const float s = (fabsf(tl) - (ra + rb)) / sqrtf(l2);
			if (s > max_sep) return 0;
			if (s > sE) { sE = s; eI = i; eJ = j; }
		}
	}
	const float sF = fmaxf(sA, sB);
	if (eI >= 0 && sE > sF + 1.0e-3f) {
		/* edge - edge */
		const v3 ua = m33_col(A->R, eI), ub = m33_col(B->R, eJ);
		v3 n = v3_cross(ua, ub);
		n = v3_scale(n, 1.0f / v3_len(n));
		if (v3_dot(n, T) < 0.0f) n = v3_neg(n);
		v3 pa = A->pos, pb = B->pos;
		for (int k = 0; k < 3; ++k) {
			if (k != eI) { const v3 ak = m33_col(A->R, k); const float sg = v3_dot(n, ak) >= 0.0f ? 1.0f : -1.0f; pa = v3_add(pa, v3_scale(ak, sg * v3_get(hA, k))); }
			if (k != eJ) { const v3 bk = m33_col(B->R, k); const float sg = v3_dot(n, bk) >= 0.0f ? -1.0f : 1.0f; pb = v3_add(pb, v3_scale(bk, sg * v3_get(hB, k))); }
		}
		const v3 dd = v3_sub(pb, pa);
		const float uaub = v3_dot(ua, ub), q1 = v3_dot(ua, dd), q2 = -v3_dot(ub, dd);
		const float den = 1.0f - uaub * uaub;
		float alpha = 0.0f, beta = 0.0f;
		if (den > 1.0e-4f) { alpha = (q1 + uaub * q2) / den; beta = (uaub * q1 + q2) / den; }
		alpha = clampf(alpha, -v3_get(hA, eI), v3_get(hA, eI));
		beta = clampf(beta, -v3_get(hB, eJ), v3_get(hB, eJ));
		m->n = n; m->np = 1;
		m->p1[0] = v3_add(pa, v3_scale(ua, alpha));
		m->p2[0] = v3_add(pb, v3_scale(ub, beta));
		return 1;
	}
	/* face contact: reference box X owns the axis, incident box Y is clipped against X's face */
	const int refA = !(sB > sA + 1.0e-4f);
	const sgd_shape* X = refA ? A : B; const sgd_shape* Y = refA ? B : A;
	const v3 hX = refA ? hA : hB, hY = refA ? hB : hA;
	const int k = refA ? kA : kB;
	const float tk = refA ? tA[k] : -tB[k];          /* (cY - cX) . x_k */
	const float sg = tk >= 0.0f ? 1.0f : -1.0f;
	const v3 nref = v3_scale(m33_col(X->R, k), sg);  /* X -> Y */
	int j = 0; float dj = 0.0f; float bestd = -1.0f;
	for (int jj = 0; jj < 3; ++jj) { const float dd = v3_dot(nref, m33_col(Y->R, jj)); if (fabsf(dd) > bestd) { bestd = fabsf(dd); j = jj; dj = dd; } }
	const float sj = dj > 0.0f ? -1.0f : 1.0f;
	const int u = (j + 1) % 3, v = (j + 2) % 3;
	const v3 yu = v3_scale(m33_col(Y->R, u), v3_get(hY, u)), yv = v3_scale(m33_col(Y->R, v), v3_get(hY, v));
	const v3 fc = v3_add(Y->pos, v3_scale(m33_col(Y->R, j), sj * v3_get(hY, j)));
	const v3 w0 = v3_add(v3_add(fc, yu), yv), w1 = v3_add(v3_sub(fc, yu), yv);
	const v3 w2 = v3_sub(v3_sub(fc, yu), yv), w3 = v3_sub(v3_add(fc, yu), yv);
	const int a1 = (k + 1) % 3, a2 = (k + 2) % 3;
	if constexpr (LDS) {
		sgd_lpoly P, Q; P.b = lds; Q.b = lds + SGD_LPOLY_FLOATS;
		sgd_lp_set(P, 0, m33_tmul(X->R, v3_sub(w0, X->pos))); sgd_lp_set(P, 1, m33_tmul(X->R, v3_sub(w1, X->pos)));
		sgd_lp_set(P, 2, m33_tmul(X->R, v3_sub(w2, X->pos))); sgd_lp_set(P, 3, m33_tmul(X->R, v3_sub(w3, X->pos)));
		int np = 4;
		np = sgd_clip_poly_lds(P, np, a1, 1.0f, v3_get(hX, a1), Q);
		np = sgd_clip_poly_lds(Q, np, a1, -1.0f, v3_get(hX, a1), P);
		np = sgd_clip_poly_lds(P, np, a2, 1.0f, v3_get(hX, a2), Q);
		np = sgd_clip_poly_lds(Q, np, a2, -1.0f, v3_get(hX, a2), P);
		// the candidates over the polygon they come from: slot cnt <= i is written after corner i was read (P: points on body 1, Q: on body 2)
		int cnt = 0;
		for (int i = 0; i < np; ++i) {
			const v3 pi = sgd_lp_get(P, i);
			const float sep = sg * v3_get(pi, k) - v3_get(hX, k);
			if (sep <= max_sep) {
				v3 pr = pi; v3_set(pr, k, sg * v3_get(hX, k));
				const v3 wi = v3_add(X->pos, m33_mul(X->R, pi));   /* on Y */
				const v3 wr = v3_add(X->pos, m33_mul(X->R, pr));   /* on X */
				sgd_lp_set(P, cnt, refA ? wr : wi); sgd_lp_set(Q, cnt, refA ? wi : wr);
				++cnt;
			}
		}
		if (cnt == 0) return 0;
		sgd_lp_reduce(refA ? nref : v3_neg(nref), P, Q, cnt, m);
		return 1;
	} else {
	// (arrays indexed at run time: scratch.  The same clip with its polygons in registers -- eight slots and a count, corners appended by select chains, no
	// run-time index -- was measured here and is SLOWER: k_narrowphase runs four waves per SIMD on 128 registers, the four polygons spill as much as the
	// arrays held, and the select chains are instructions the scratch accesses were not: config 3 108 -> 124 us.  k_narrowphase itself takes the LDS branch above.)
	v3 poly[8], tmp[8];
	poly[0] = m33_tmul(X->R, v3_sub(w0, X->pos)); poly[1] = m33_tmul(X->R, v3_sub(w1, X->pos));
	poly[2] = m33_tmul(X->R, v3_sub(w2, X->pos)); poly[3] = m33_tmul(X->R, v3_sub(w3, X->pos));
	int np = 4;
	np = sgd_clip_poly(poly, np, a1, 1.0f, v3_get(hX, a1), tmp);
	np = sgd_clip_poly(tmp, np, a1, -1.0f, v3_get(hX, a1), poly);
	np = sgd_clip_poly(poly, np, a2, 1.0f, v3_get(hX, a2), tmp);
	np = sgd_clip_poly(tmp, np, a2, -1.0f, v3_get(hX, a2), poly);
	m->n = refA ? nref : v3_neg(nref);
	int cnt = 0;
	for (int i = 0; i < np; ++i) {
		const float sep = sg * v3_get(poly[i], k) - v3_get(hX, k);
		if (sep <= max_sep) {
			v3 pr = poly[i]; v3_set(pr, k, sg * v3_get(hX, k));
			const v3 wi = v3_add(X->pos, m33_mul(X->R, poly[i]));   /* on Y */
			const v3 wr = v3_add(X->pos, m33_mul(X->R, pr));        /* on X */
			if (refA) { m->p1[cnt] = wr; m->p2[cnt] = wi; } else { m->p1[cnt] = wi; m->p2[cnt] = wr; }
			++cnt;
		}
	}
	if (cnt == 0) return 0;
	m->np = cnt;
	sgd_reduce_manifold(m);
	return 1;
	}
}

SGP_DEV static void sgd_flip_manifold(sgd_manifold* m)
{
	m->n = v3_neg(m->n);
#pragma unroll
	for (int i = 0; i < 4; ++i) if (i < m->np) { const v3 t = m->p1[i]; m->p1[i] = m->p2[i]; m->p2[i] = t; }      // (a finished manifold: at most four points; static slots)
}

/* Dispatch on the (type_a, type_b) pair; canonical order sphere < box < capsule. */
#include "sgp_device_hull.h"

SGP_DEV static sgd_hview sgd_hull_view(const sgd_shape* s)
{
	sgd_hview v;
	v.pos = s->pos; v.R = s->R; v.h = s->hull;
	v.scale = s->type == SGD_SHAPE_BOX ? V3(s->p0, s->p1, s->p2) : V3(1.0f, 1.0f, 1.0f);
	return v;
}

// pairs with a convex hull (canonical order sphere < box < capsule < hull); run by k_narrowphase_hull only
SGP_DEV static int sgd_collide_hull(const sgd_shape* a, const sgd_shape* b, float max_sep, sgd_manifold* m)
{
	int hit, flip = 0;
	const sgd_shape* x = a; const sgd_shape* y = b;
	if (a->type > b->type) { x = b; y = a; flip = 1; }
	const sgd_hview hy = sgd_hull_view(y);
	if (x->type == SGD_SHAPE_SPHERE) { hit = sgd_hull_sphere(&hy, x->pos, x->p0, max_sep, m); flip = !flip; }       // computed hull -> sphere
	else if (x->type == SGD_SHAPE_CAPSULE) {
		const v3 ax = v3_scale(m33_col(x->R, 2), x->p1);
		hit = sgd_hull_capsule(&hy, v3_sub(x->pos, ax), v3_add(x->pos, ax), x->p0, max_sep, m); flip = !flip;
	} else { const sgd_hview hx = sgd_hull_view(x); hit = sgd_hull_hull(&hx, &hy, max_sep, m); }
	if (hit && flip) sgd_flip_manifold(m);
	return hit;
}

template <bool LDS = false> SGP_DEV static int sgd_collide(const sgd_shape* a, const sgd_shape* b, float max_sep, sgd_manifold* m, float* lds = nullptr)
{
	int hit, flip = 0;
	const sgd_shape* x = a; const sgd_shape* y = b;
	if (a->type > b->type) { x = b; y = a; flip = 1; }
	if (x->type == SGD_SHAPE_SPHERE) {
		if (y->type == SGD_SHAPE_SPHERE) hit = sgd_sphere_sphere_pts(x->pos, x->p0, y->pos, y->p0, max_sep, m);
		else if (y->type == SGD_SHAPE_BOX) hit = sgd_sphere_box(x, y, max_sep, m);
		else hit = sgd_sphere_capsule(x, y, max_sep, m);
	} else if (x->type == SGD_SHAPE_BOX) {
		if (y->type == SGD_SHAPE_BOX) hit = sgd_box_box<LDS>(x, y, max_sep, m, lds);
		else hit = sgd_box_capsule(x, y, max_sep, m);
	} else hit = sgd_capsule_capsule(x, y, max_sep, m);
	if (hit && flip) sgd_flip_manifold(m);
	return hit;
}

