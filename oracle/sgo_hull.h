/*
 * sgo_hull.h -- ORACLE (test infrastructure only): convex hull shapes.
 *
 * What it restates.  Substrata gives dynamic mesh objects and vehicle bodies a JPH::ConvexHullShape
 * (/root/reference/gui_client/PhysicsWorld.cpp:735-1166 createJoltShapeFor...Mesh with is_dynamic; CarPhysics.cpp:66-78;
 * BikePhysics.cpp:76-100), optionally wrapped in OffsetCenterOfMassShape / ScaledShape.  Jolt (v5.3.0, not in the tree) collides
 * convex shapes with GJK/EPA, builds the manifold from the two supporting faces and prunes it to 4 points; like the rest of the
 * oracle this file restates that from upstream knowledge with closed-form tests instead (parity unpinned; pinned by
 * tests/test_oracle_hull.py and the brute-force support-function test in tests/test_collide_independent.py):
 *   hull - hull / hull - box : separating axis test over face normals and edge pairs, reference-face / incident-face clipping,
 *                             or the closest points of the two edges for an edge-edge axis;
 *   hull - sphere            : closest point on the hull surface (deepest face when the centre is inside);
 *   hull - capsule           : the sphere test at the point of the capsule axis closest to the hull (fixed-count ternary search),
 *                             two points when the axis lies along a face.
 * A hull lives in its body frame: origin = centre of mass, axes = principal axes of inertia (the builder returns the transform
 * from the frame of the input points, which is what OffsetCenterOfMassShape / MassProperties::DecomposePrincipalMomentsOfInertia
 * hide inside Jolt).  A box can be viewed as a hull (the +-1 cube template scaled by its half extents).
 */
#ifndef SGO_HULL_H
#define SGO_HULL_H

#include "sgo_math.h"
#include <stdlib.h>
/* (included from sgo_collide.h after sgo_manifold, SGO_CAPSULE_SLOP and sgo_closest_on_segment are defined) */

#define SGO_HULL_MAX_VERTS 256         /* JPH::ConvexHullShape::cMaxPointsInHull (round 5; rounds 1-4 kept 32) */
#define SGO_HULL_MAX_FACES 512         /* <= 2 V - 4 triangles, fewer once coplanar ones are merged; faces of more than 16 corners are split */
#define SGO_HULL_MAX_EDGES 768         /* <= 3 V - 6 (+ the diagonals of split faces) */
#define SGO_HULL_MAX_FACE_IDX 1792     /* sum of the face loops = 2 E */
#define SGO_HULL_MAX_FACE_VERTS 16
#define SGO_HULL_SMALL_VERTS 32        /* up to here the builder and the separating-axis search are those of rounds 1-4, bit for bit; a pair with a larger hull: the Gauss-map test selects the edge pairs worth an axis */
#define SGO_HULL_CLIP_CAP 24

struct sgo_hull_s {
	int nv, nf, ne, is_box_template;
	v3 verts[SGO_HULL_MAX_VERTS];
	v3 normals[SGO_HULL_MAX_FACES]; float plane_d[SGO_HULL_MAX_FACES];       /* inside: n.x <= d */
	unsigned short face_start[SGO_HULL_MAX_FACES + 1]; unsigned char face_idx[SGO_HULL_MAX_FACE_IDX];   /* CCW seen from outside */
	unsigned char edge_a[SGO_HULL_MAX_EDGES], edge_b[SGO_HULL_MAX_EDGES];
	unsigned short edge_f0[SGO_HULL_MAX_EDGES], edge_f1[SGO_HULL_MAX_EDGES];   /* the two faces an edge lies between: f0 has it as a -> b, f1 as b -> a (Gauss-map test of edge pairs) */
	v3 aabb_min, aabb_max;
	float bound_radius, volume;
	v3 unit_inertia;                   /* principal moments for density 1 */
};
typedef struct sgo_hull_s sgo_hull;

/* a hull (or the cube template scaled to a box) placed in the world */
typedef struct { v3 pos; m33 R; v3 scale; const sgo_hull* h; } sgo_hview;

static inline v3 sgo_hv_local(const sgo_hview* v, int i) { const v3 p = v->h->verts[i]; return V3(p.x * v->scale.x, p.y * v->scale.y, p.z * v->scale.z); }
static inline v3 sgo_hv_world(const sgo_hview* v, int i) { return v3_add(v->pos, m33_mul(v->R, sgo_hv_local(v, i))); }
static inline v3 sgo_hv_normal(const sgo_hview* v, int f) { return m33_mul(v->R, v->h->normals[f]); }
static inline float sgo_hv_plane_d(const sgo_hview* v, int f)
{
	if (v->h->is_box_template) { const v3 n = v->h->normals[f]; return fabsf(n.x) * v->scale.x + fabsf(n.y) * v->scale.y + fabsf(n.z) * v->scale.z; }
	return v->h->plane_d[f];
}
/* min / max over the vertices of w . x (world direction w) */
static inline float sgo_hv_proj_min(const sgo_hview* v, v3 w)
{
	const v3 l = m33_tmul(v->R, w);
	float best = 3.4e38f;
	for (int i = 0; i < v->h->nv; ++i) { const float d = v3_dot(l, sgo_hv_local(v, i)); if (d < best) best = d; }
	return v3_dot(w, v->pos) + best;
}
static inline float sgo_hv_proj_max(const sgo_hview* v, v3 w)
{
	const v3 l = m33_tmul(v->R, w);
	float best = -3.4e38f;
	for (int i = 0; i < v->h->nv; ++i) { const float d = v3_dot(l, sgo_hv_local(v, i)); if (d > best) best = d; }
	return v3_dot(w, v->pos) + best;
}

/* Clip a world-space polygon against the half space (p - a) . side <= 0. */
static inline int sgo_hull_clip(const v3* in, int n, v3 a, v3 side, v3* out)
{
	int m = 0;
	for (int i = 0; i < n; ++i) {
		const v3 p = in[i], q = in[(i + 1) % n];
		const float dp = v3_dot(v3_sub(p, a), side), dq = v3_dot(v3_sub(q, a), side);
		if (dp <= 0.0f) { if (m < SGO_HULL_CLIP_CAP) out[m++] = p; }
		if ((dp <= 0.0f) != (dq <= 0.0f)) {
			const float t = dp / (dp - dq);
			if (m < SGO_HULL_CLIP_CAP) out[m++] = v3_add(p, v3_scale(v3_sub(q, p), t));
		}
	}
	return m;
}

/* Keep at most 4 of np (<= SGO_HULL_CLIP_CAP) contact points: deepest, farthest from it, the extremes either side of that chord
   (same rule as sgo_reduce_manifold).  Writes the survivors to the manifold. */
static inline void sgo_hull_reduce(v3 n, const v3* p1, const v3* p2, int np, sgo_manifold* m)
{
	m->n = n;
	if (np <= 4) { for (int i = 0; i < np; ++i) { m->p1[i] = p1[i]; m->p2[i] = p2[i]; } m->np = np; return; }
	int i0 = 0; float best = -3.4e38f;
	for (int i = 0; i < np; ++i) { const float pen = v3_dot(v3_sub(p1[i], p2[i]), n); if (pen > best) { best = pen; i0 = i; } }
	int i1 = i0; best = -1.0f;
	for (int i = 0; i < np; ++i) { const float d2 = v3_len_sq(v3_sub(p1[i], p1[i0])); if (d2 > best) { best = d2; i1 = i; } }
	const v3 e = v3_sub(p1[i1], p1[i0]);
	int i2 = -1, i3 = -1; float amax = 0.0f, amin = 0.0f;
	for (int i = 0; i < np; ++i) {
		if (i == i0 || i == i1) continue;
		const float area = v3_dot(v3_cross(e, v3_sub(p1[i], p1[i0])), n);
		if (area > amax) { amax = area; i2 = i; }
		if (area < amin) { amin = area; i3 = i; }
	}
	int k = 0;
	m->p1[k] = p1[i0]; m->p2[k] = p2[i0]; ++k;
	if (i1 != i0) { m->p1[k] = p1[i1]; m->p2[k] = p2[i1]; ++k; }
	if (i2 >= 0) { m->p1[k] = p1[i2]; m->p2[k] = p2[i2]; ++k; }
	if (i3 >= 0) { m->p1[k] = p1[i3]; m->p2[k] = p2[i3]; ++k; }
	m->np = k;
}

/* closest points of two segments (a0,a1), (b0,b1) */
static inline void sgo_seg_seg_closest(v3 a0, v3 a1, v3 b0, v3 b1, v3* pa, v3* pb)
{
	const v3 d1 = v3_sub(a1, a0), d2 = v3_sub(b1, b0), r = v3_sub(a0, b0);
	const float a = v3_dot(d1, d1), e = v3_dot(d2, d2), f = v3_dot(d2, r);
	float s = 0.0f, t = 0.0f;
	if (a > 1.0e-12f && e > 1.0e-12f) {
		const float c = v3_dot(d1, r), b = v3_dot(d1, d2);
		const float den = a * e - b * b;
		if (den > 1.0e-12f) s = clampf((b * f - c * e) / den, 0.0f, 1.0f);
		t = (b * s + f) / e;
		if (t < 0.0f) { t = 0.0f; s = clampf(-c / a, 0.0f, 1.0f); }
		else if (t > 1.0f) { t = 1.0f; s = clampf((b - c) / a, 0.0f, 1.0f); }
	} else if (a > 1.0e-12f) { s = clampf(-v3_dot(d1, r) / a, 0.0f, 1.0f); }
	else if (e > 1.0e-12f) { t = clampf(f / e, 0.0f, 1.0f); }
	*pa = v3_add(a0, v3_scale(d1, s));
	*pb = v3_add(b0, v3_scale(d2, t));
}
/* ... and only the parameter of the closest point along the first segment (same arithmetic) */
static inline float sgo_seg_seg_param(v3 a0, v3 a1, v3 b0, v3 b1)
{
	const v3 d1 = v3_sub(a1, a0), d2 = v3_sub(b1, b0), r = v3_sub(a0, b0);
	const float a = v3_dot(d1, d1), e = v3_dot(d2, d2), f = v3_dot(d2, r);
	float s = 0.0f, t = 0.0f;
	if (a > 1.0e-12f && e > 1.0e-12f) {
		const float c = v3_dot(d1, r), b = v3_dot(d1, d2);
		const float den = a * e - b * b;
		if (den > 1.0e-12f) s = clampf((b * f - c * e) / den, 0.0f, 1.0f);
		t = (b * s + f) / e;
		if (t < 0.0f) { t = 0.0f; s = clampf(-c / a, 0.0f, 1.0f); }
		else if (t > 1.0f) { t = 1.0f; s = clampf((b - c) / a, 0.0f, 1.0f); }
	} else if (a > 1.0e-12f) { s = clampf(-v3_dot(d1, r) / a, 0.0f, 1.0f); }
	(void)t;
	return s;
}

/* Result of the separating-axis search: best face axis of A, of B, best (supporting) edge pair. */
typedef struct { float sA, sB, sE; int fA, fB, eA, eB; v3 nE; } sgo_hull_sat;

/* separation of B in front of face f of X (X, Y in either role) */
static inline float sgo_hull_axis_face(const sgo_hview* X, const sgo_hview* Y, int f)
{
	const v3 n = sgo_hv_normal(X, f);
	return sgo_hv_proj_min(Y, n) - (v3_dot(n, X->pos) + sgo_hv_plane_d(X, f));
}

/* edge pair (i of A, j of B): returns 0 when the edges are (nearly) parallel, else 1 with the axis (oriented A -> B), the
   separation along it and whether this pair really supports the two hulls along it (parallel edges give the same axis: only the
   supporting pair is the contact) */
static inline int sgo_hull_axis_edge(const sgo_hview* A, const sgo_hview* B, int i, int j, v3 T, v3* ax_out, float* s_out, int* supporting)
{
	const v3 da = m33_mul(A->R, v3_sub(sgo_hv_local(A, A->h->edge_b[i]), sgo_hv_local(A, A->h->edge_a[i])));
	const v3 db = m33_mul(B->R, v3_sub(sgo_hv_local(B, B->h->edge_b[j]), sgo_hv_local(B, B->h->edge_a[j])));
	v3 ax = v3_cross(da, db);
	const float l2 = v3_len_sq(ax);
	if (l2 < 1.0e-6f * v3_len_sq(da) * v3_len_sq(db)) return 0;
	ax = v3_scale(ax, 1.0f / sqrtf(l2));
	if (v3_dot(ax, T) < 0.0f) ax = v3_neg(ax);
	const float s = sgo_hv_proj_min(B, ax) - sgo_hv_proj_max(A, ax);
	const v3 a0 = sgo_hv_world(A, A->h->edge_a[i]), b0 = sgo_hv_world(B, B->h->edge_a[j]);
	const float s_edge = v3_dot(ax, b0) - v3_dot(ax, a0);        /* (both ends of an edge project alike: ax is perpendicular to it) */
	*ax_out = ax; *s_out = s; *supporting = !(s_edge - s > 1.0e-4f);
	return 1;
}

/* The same for an edge pair the Gauss-map test has picked (a, bb: world normals of the faces either side of A's edge): the two edges ARE what supports the
   hulls along +-(da x db), so the separation is that of the edges themselves -- no walk over the vertices -- and the axis points the way A's two faces do. */
static inline int sgo_hull_axis_edge_picked(const sgo_hview* A, const sgo_hview* B, int i, int j, v3 a, v3 bb, v3* ax_out, float* s_out)
{
	const v3 da = m33_mul(A->R, v3_sub(sgo_hv_local(A, A->h->edge_b[i]), sgo_hv_local(A, A->h->edge_a[i])));
	const v3 db = m33_mul(B->R, v3_sub(sgo_hv_local(B, B->h->edge_b[j]), sgo_hv_local(B, B->h->edge_a[j])));
	v3 ax = v3_cross(da, db);
	const float l2 = v3_len_sq(ax);
	if (l2 < 1.0e-6f * v3_len_sq(da) * v3_len_sq(db)) return 0;
	ax = v3_scale(ax, 1.0f / sqrtf(l2));
	if (v3_dot(ax, v3_add(a, bb)) < 0.0f) ax = v3_neg(ax);
	const v3 a0 = sgo_hv_world(A, A->h->edge_a[i]), b0 = sgo_hv_world(B, B->h->edge_a[j]);
	*ax_out = ax; *s_out = v3_dot(ax, b0) - v3_dot(ax, a0);
	return 1;
}

/* Sequential search (first maximum wins).  Returns 0 when some axis separates the hulls by more than max_sep. */
static inline int sgo_hull_sat_search(const sgo_hview* A, const sgo_hview* B, float max_sep, sgo_hull_sat* r)
{
	r->sA = -3.4e38f; r->sB = -3.4e38f; r->sE = -3.4e38f; r->fA = 0; r->fB = 0; r->eA = -1; r->eB = -1; r->nE = V3(0, 0, 0);
	for (int f = 0; f < A->h->nf; ++f) {
		const float s = sgo_hull_axis_face(A, B, f);
		if (s > max_sep) return 0;
		if (s > r->sA) { r->sA = s; r->fA = f; }
	}
	for (int f = 0; f < B->h->nf; ++f) {
		const float s = sgo_hull_axis_face(B, A, f);
		if (s > max_sep) return 0;
		if (s > r->sB) { r->sB = s; r->fB = f; }
	}
	const v3 T = v3_sub(B->pos, A->pos);
	if (A->h->nv == 3 && A->h->nf == 2 && B->h->nv > SGO_HULL_SMALL_VERTS && !getenv("SGO_HULL_TRIANGLE_FULL_SEARCH")) {      /* (the variable: tests/test_oracle_hull.py compares with the full search) */
		/* A mesh triangle against a hull beyond 32 vertices (round 5): 3 x up to 768 edge pairs, each a walk over every vertex in the full search.  The Gauss-map
		   test for a triangle: edge k supports the triangle along the directions of the half circle about it through its outward in-plane normal m_k (from the
		   triangle's normal to its opposite); edge j of the hull supports the hull along minus the arc between its two faces' normals.  The two meet -- the pair
		   is a face of the Minkowski difference -- when the arc crosses the plane perpendicular to edge k on m_k's side.  In the hull's frame (its normals as
		   stored); a picked pair's axis and separation from its two edges.  Hull edge outermost: its normals are fetched once for the three triangle edges. */
		const v3 nT = sgo_hv_normal(A, 0);
		v3 da[3], dal[3], ml[3];
		for (int k = 0; k < 3; ++k) {
			const int ia = A->h->edge_a[k], ib = A->h->edge_b[k], io = 3 - ia - ib;
			da[k] = m33_mul(A->R, v3_sub(sgo_hv_local(A, ib), sgo_hv_local(A, ia)));
			v3 m = v3_cross(da[k], nT);
			if (v3_dot(m, v3_sub(sgo_hv_world(A, ia), sgo_hv_world(A, io))) < 0.0f) m = v3_neg(m);
			dal[k] = m33_tmul(B->R, da[k]); ml[k] = m33_tmul(B->R, m);
		}
		for (int j = 0; j < B->h->ne; ++j) {
			if (B->h->edge_f0[j] == 0xFFFF) {      /* (an edge without its two faces: its three pairs in full) */
				for (int k = 0; k < 3; ++k) {
					v3 ax; float s; int sup;
					if (!sgo_hull_axis_edge(A, B, k, j, T, &ax, &s, &sup)) continue;
					if (s > max_sep) return 0;
					if (s > r->sE && sup) { r->sE = s; r->eA = k; r->eB = j; r->nE = ax; }
				}
				continue;
			}
			const v3 c = v3_neg(B->h->normals[B->h->edge_f0[j]]), dd = v3_neg(B->h->normals[B->h->edge_f1[j]]);
			for (int k = 0; k < 3; ++k) {
				const float cd = v3_dot(c, dal[k]), ddd = v3_dot(dd, dal[k]);
				if (!(cd * ddd < 0.0f)) continue;
				const v3 x = v3_add(v3_scale(c, fabsf(ddd)), v3_scale(dd, fabsf(cd)));      /* (where the arc crosses the plane: the Minkowski face's normal, hull frame) */
				if (!(v3_dot(x, ml[k]) > 0.0f)) continue;
				const v3 db = m33_mul(B->R, v3_sub(sgo_hv_local(B, B->h->edge_b[j]), sgo_hv_local(B, B->h->edge_a[j])));
				v3 ax = v3_cross(da[k], db);
				const float l2 = v3_len_sq(ax);
				if (l2 < 1.0e-6f * v3_len_sq(da[k]) * v3_len_sq(db)) continue;
				ax = v3_scale(ax, 1.0f / sqrtf(l2));
				if (v3_dot(m33_tmul(B->R, ax), x) < 0.0f) ax = v3_neg(ax);
				const v3 a0 = sgo_hv_world(A, A->h->edge_a[k]), b0 = sgo_hv_world(B, B->h->edge_a[j]);
				const float s = v3_dot(ax, b0) - v3_dot(ax, a0);
				if (s > max_sep) return 0;
				if (s > r->sE) { r->sE = s; r->eA = k; r->eB = j; r->nE = ax; }
			}
		}
		return 1;
	}
	if ((A->h->nv > SGO_HULL_SMALL_VERTS || B->h->nv > SGO_HULL_SMALL_VERTS) && A->h->nv > 3 && B->h->nv > 3) {      /* (not against a mesh triangle's thin hull: its two faces span no arc) */
		/* A hull beyond 32 vertices is involved (round 5; up to 768 x 768 edge pairs): only the pairs whose cross product can be a face of the Minkowski difference are
		   evaluated -- the arcs between the normals of the faces either side of edge i of A and of (minus) those either side of edge j of B cross on the unit
		   sphere (the Gauss-map test; 4 dot products per pair instead of a projection of every vertex of both hulls), and a picked pair's separation is that of
		   its two edges (sgo_hull_axis_edge_picked).  The minimum-penetration axis is a face
		   normal of A, of B, or such a pair, so the answer is that of the full search wherever the full search is decided by more than rounding. */
		v3* na = (v3*)malloc(sizeof(v3) * (size_t)(A->h->nf + B->h->nf + A->h->ne + B->h->ne));
		v3* nb = na + A->h->nf; v3* ea = nb + B->h->nf; v3* eb = ea + A->h->ne;
		for (int f = 0; f < A->h->nf; ++f) na[f] = sgo_hv_normal(A, f);
		for (int f = 0; f < B->h->nf; ++f) nb[f] = v3_neg(sgo_hv_normal(B, f));
		for (int i = 0; i < A->h->ne; ++i) ea[i] = A->h->edge_f0[i] == 0xFFFF ? V3(0, 0, 0) : v3_cross(na[A->h->edge_f1[i]], na[A->h->edge_f0[i]]);
		for (int j = 0; j < B->h->ne; ++j) eb[j] = B->h->edge_f0[j] == 0xFFFF ? V3(0, 0, 0) : v3_cross(nb[B->h->edge_f1[j]], nb[B->h->edge_f0[j]]);
		int separated = 0;
		for (int i = 0; i < A->h->ne && !separated; ++i) {
			const int open_a = A->h->edge_f0[i] == 0xFFFF;      /* (an edge without its two faces, sgo_hull_build.h: its pairs in full) */
			const v3 a = open_a ? V3(0, 0, 0) : na[A->h->edge_f0[i]], bb = open_a ? V3(0, 0, 0) : na[A->h->edge_f1[i]], bxa = ea[i];
			for (int j = 0; j < B->h->ne; ++j) {
				if (open_a || B->h->edge_f0[j] == 0xFFFF) {
					v3 ax; float s; int sup;
					if (!sgo_hull_axis_edge(A, B, i, j, T, &ax, &s, &sup)) continue;
					if (s > max_sep) { separated = 1; break; }
					if (s > r->sE && sup) { r->sE = s; r->eA = i; r->eB = j; r->nE = ax; }
					continue;
				}
				const v3 c = nb[B->h->edge_f0[j]], dd = nb[B->h->edge_f1[j]], dxc = eb[j];
				const float cba = v3_dot(c, bxa), dba = v3_dot(dd, bxa), adc = v3_dot(a, dxc), bdc = v3_dot(bb, dxc);
				if (!(cba * dba < 0.0f && adc * bdc < 0.0f && cba * bdc > 0.0f)) continue;
				v3 ax; float s;
				if (!sgo_hull_axis_edge_picked(A, B, i, j, a, bb, &ax, &s)) continue;
				if (s > max_sep) { separated = 1; break; }
				if (s > r->sE) { r->sE = s; r->eA = i; r->eB = j; r->nE = ax; }
			}
		}
		free(na);
		return !separated;
	}
	for (int i = 0; i < A->h->ne; ++i) {
		for (int j = 0; j < B->h->ne; ++j) {
			v3 ax; float s; int sup;
			if (!sgo_hull_axis_edge(A, B, i, j, T, &ax, &s, &sup)) continue;
			if (s > max_sep) return 0;
			if (s > r->sE && sup) { r->sE = s; r->eA = i; r->eB = j; r->nE = ax; }
		}
	}
	return 1;
}

/* Face contact: reference hull X owns the axis (its face fX), the most anti-parallel face of Y is clipped against X's face.  refA: X is the
   pair's first hull (the manifold's normal runs from the first to the second). */
static inline int sgo_hull_face_contact(const sgo_hview* X, const sgo_hview* Y, int fX, int refA, float max_sep, sgo_manifold* m)
{
	const v3 nref = sgo_hv_normal(X, fX);
	int fY = 0; float bestd = 3.4e38f;
	for (int f = 0; f < Y->h->nf; ++f) { const float d = v3_dot(nref, sgo_hv_normal(Y, f)); if (d < bestd) { bestd = d; fY = f; } }
	v3 poly[SGO_HULL_CLIP_CAP], tmp[SGO_HULL_CLIP_CAP];
	int np = 0;
	/* (a face of more than SGO_HULL_MAX_FACE_VERTS corners takes part with every step-th of them: the polygon inscribed in it) */
	const int y0 = Y->h->face_start[fY], y1 = Y->h->face_start[fY + 1], ystep = (y1 - y0 + SGO_HULL_MAX_FACE_VERTS - 1) / SGO_HULL_MAX_FACE_VERTS;
	for (int k = y0; k < y1; k += ystep) poly[np++] = sgo_hv_world(Y, Y->h->face_idx[k]);
	const int x0 = X->h->face_start[fX], x1 = X->h->face_start[fX + 1], xstep = (x1 - x0 + SGO_HULL_MAX_FACE_VERTS - 1) / SGO_HULL_MAX_FACE_VERTS;
	for (int k = x0; k < x1 && np > 0; k += xstep) {
		const v3 a = sgo_hv_world(X, X->h->face_idx[k]);
		const v3 b = sgo_hv_world(X, X->h->face_idx[k + xstep < x1 ? k + xstep : x0]);
		const v3 side = v3_cross(v3_sub(b, a), nref);
		np = sgo_hull_clip(poly, np, a, side, tmp);
		for (int i = 0; i < np; ++i) poly[i] = tmp[i];
	}
	const float off = v3_dot(nref, X->pos) + sgo_hv_plane_d(X, fX);
	v3 q1[SGO_HULL_CLIP_CAP], q2[SGO_HULL_CLIP_CAP];
	int cnt = 0;
	for (int i = 0; i < np; ++i) {
		const float sep = v3_dot(nref, poly[i]) - off;
		if (sep <= max_sep) {
			const v3 pr = v3_sub(poly[i], v3_scale(nref, sep));      /* on X's face */
			if (refA) { q1[cnt] = pr; q2[cnt] = poly[i]; } else { q1[cnt] = poly[i]; q2[cnt] = pr; }
			++cnt;
		}
	}
	if (cnt == 0) {
		/* nothing of the incident face lies over the reference face (the closest features are an edge / a vertex of X): fall back
		   to the support vertex of Y along the axis */
		int bi = 0; float bp = 3.4e38f;
		for (int i = 0; i < Y->h->nv; ++i) { const float pr = v3_dot(nref, sgo_hv_world(Y, i)); if (pr < bp) { bp = pr; bi = i; } }
		const float sep = bp - off;
		if (sep > max_sep) return 0;
		const v3 py = sgo_hv_world(Y, bi), px = v3_sub(py, v3_scale(nref, sep));
		if (refA) { q1[0] = px; q2[0] = py; } else { q1[0] = py; q2[0] = px; }
		cnt = 1;
	}
	sgo_hull_reduce(refA ? nref : v3_neg(nref), q1, q2, cnt, m);
	return 1;
}

/* Manifold from the result of the search.  Normal from A to B. */
static inline int sgo_hull_manifold(const sgo_hview* A, const sgo_hview* B, float max_sep, const sgo_hull_sat* r, sgo_manifold* m)
{
	const float sA = r->sA, sB = r->sB, sE = r->sE; const int fA = r->fA, fB = r->fB, eA = r->eA, eB = r->eB; const v3 nE = r->nE;
	const float sF = fmaxf(sA, sB);
	if (eA >= 0 && sE > sF + 1.0e-3f) {
		v3 pa, pb;
		sgo_seg_seg_closest(sgo_hv_world(A, A->h->edge_a[eA]), sgo_hv_world(A, A->h->edge_b[eA]),
		                    sgo_hv_world(B, B->h->edge_a[eB]), sgo_hv_world(B, B->h->edge_b[eB]), &pa, &pb);
		m->n = nE; m->np = 1; m->p1[0] = pa; m->p2[0] = pb;
		return 1;
	}
	/* face contact: reference hull X owns the axis, the most anti-parallel face of Y is clipped against X's face */
	const int refA = !(sB > sA + 1.0e-4f);
	return refA ? sgo_hull_face_contact(A, B, fA, 1, max_sep, m) : sgo_hull_face_contact(B, A, fB, 0, max_sep, m);
}

/* A, B = hull views (either may be the scaled cube template): SAT + clipping.  Normal from A to B. */
static inline int sgo_hull_hull(const sgo_hview* A, const sgo_hview* B, float max_sep, sgo_manifold* m)
{
	sgo_hull_sat r;
	if (!sgo_hull_sat_search(A, B, max_sep, &r)) return 0;
	return sgo_hull_manifold(A, B, max_sep, &r, m);
}

/* Closest point on the hull surface to the hull-local point l (scale 1 hulls only).  Returns the signed distance (negative
   inside), the closest point q and the outward direction n at q (unit). */
static inline float sgo_hull_closest(const sgo_hull* h, v3 l, v3* q_out, v3* n_out)
{
	float smax = -3.4e38f; int fmax = 0;
	for (int f = 0; f < h->nf; ++f) { const float s = v3_dot(h->normals[f], l) - h->plane_d[f]; if (s > smax) { smax = s; fmax = f; } }
	/* (a mesh triangle's thin hull has no side planes: a point exactly in its plane is 'inside' only above the triangle itself -- the loop below decides) */
	const int thin = h->nf == 2 && h->nv == 3;
	if (smax <= 0.0f && !thin) {
		*n_out = h->normals[fmax];
		*q_out = v3_sub(l, v3_scale(h->normals[fmax], smax));
		return smax;
	}
	float best = 3.4e38f; v3 bq = l; int on_face = -1;
	for (int f = 0; f < h->nf; ++f) {
		const v3 n = h->normals[f];
		const float s = v3_dot(n, l) - h->plane_d[f];
		if (thin ? s < 0.0f : s <= 0.0f) continue;
		const v3 p = v3_sub(l, v3_scale(n, s));
		const int k0 = h->face_start[f], k1 = h->face_start[f + 1];
		int inside = 1;
		for (int k = k0; k < k1; ++k) {
			const v3 a = h->verts[h->face_idx[k]], b = h->verts[h->face_idx[k + 1 < k1 ? k + 1 : k0]];
			if (v3_dot(v3_cross(v3_sub(b, a), n), v3_sub(p, a)) > 0.0f) { inside = 0; break; }
		}
		if (inside) { const float d2 = s * s; if (d2 < best) { best = d2; bq = p; on_face = f; } continue; }
		for (int k = k0; k < k1; ++k) {
			const v3 a = h->verts[h->face_idx[k]], b = h->verts[h->face_idx[k + 1 < k1 ? k + 1 : k0]];
			const v3 c = sgo_closest_on_segment(a, b, l);
			const float d2 = v3_len_sq(v3_sub(l, c));
			if (d2 < best) { best = d2; bq = c; on_face = -1; }
		}
	}
	const float dist = sqrtf(best);
	*q_out = bq;
	/* over a face the direction IS the face normal: l - q would cancel to nothing once the distance drops below the rounding of l
	   (the distance itself, taken from the plane equation, stays accurate) -- and a zero direction poisons the whole solve */
	if (on_face >= 0) *n_out = h->normals[on_face];
	else {
		const v3 v = v3_sub(l, bq);
		const float len = v3_len(v);
		*n_out = len > 1.0e-12f ? v3_scale(v, 1.0f / len) : h->normals[fmax];
	}
	return dist;
}

/* hull H (scale 1) vs sphere: normal from the hull to the sphere */
static inline int sgo_hull_sphere(const sgo_hview* H, v3 c, float r, float max_sep, sgo_manifold* m)
{
	const v3 l = m33_tmul(H->R, v3_sub(c, H->pos));
	v3 q, n;
	const float d = sgo_hull_closest(H->h, l, &q, &n);
	if (d - r > max_sep) return 0;
	const v3 nw = m33_mul(H->R, n);
	m->n = nw; m->np = 1;
	m->p1[0] = v3_add(H->pos, m33_mul(H->R, q));
	m->p2[0] = v3_sub(c, v3_scale(nw, r));
	return 1;
}

/* hull H (scale 1) vs capsule (end points e0, e1 of the axis, radius r): normal from the hull to the capsule */
static inline int sgo_hull_capsule(const sgo_hview* H, v3 e0, v3 e1, float r, float max_sep, sgo_manifold* m)
{
	const v3 s0 = m33_tmul(H->R, v3_sub(e0, H->pos)), s1 = m33_tmul(H->R, v3_sub(e1, H->pos));
	const v3 d = v3_sub(s1, s0);
	v3 q, n;
	float ts;
	if (H->h->nv == 3 && H->h->nf == 2) {
		/* a mesh triangle (thin hull).  The distance to a convex set is C1 outside the set, so along the axis it is least at an end of the axis, at its
		   closest approach to one of the three edges, or where it pierces the triangle: at most six evaluations, no search (a capsule on a mesh is
		   the player on the world: this is the character controller's inner loop) */
		float cand[6]; int ncand = 0;
		cand[ncand++] = 0.0f; cand[ncand++] = 1.0f;
		for (int k = 0; k < 3; ++k) cand[ncand++] = sgo_seg_seg_param(s0, s1, H->h->verts[H->h->edge_a[k]], H->h->verts[H->h->edge_b[k]]);
		const float h0 = v3_dot(H->h->normals[0], s0) - H->h->plane_d[0], h1 = v3_dot(H->h->normals[0], s1) - H->h->plane_d[0];
		if ((h0 > 0.0f) != (h1 > 0.0f)) {
			/* (only a crossing INSIDE the triangle counts: a thin hull has no side planes, and a point of its plane beside the triangle would
			   pass for "inside" in the closest-point function) */
			const float tp = h0 / (h0 - h1);
			const v3 P = v3_add(s0, v3_scale(d, tp));
			int inside = 1;
			for (int k = H->h->face_start[0]; k < H->h->face_start[1]; ++k) {
				const v3 a = H->h->verts[H->h->face_idx[k]], b = H->h->verts[H->h->face_idx[k + 1 < H->h->face_start[1] ? k + 1 : H->h->face_start[0]]];
				if (v3_dot(v3_cross(v3_sub(b, a), H->h->normals[0]), v3_sub(P, a)) > 0.0f) { inside = 0; break; }
			}
			if (inside) cand[ncand++] = tp;
		}
		float best = 3.4e38f; ts = 0.0f;
		for (int k = 0; k < ncand; ++k) {
			const float fk = sgo_hull_closest(H->h, v3_add(s0, v3_scale(d, cand[k])), &q, &n);
			if (fk < best) { best = fk; ts = cand[k]; }
		}
	} else {
		/* the distance to a convex set is convex along the segment: fixed-count ternary search */
		float lo = 0.0f, hi = 1.0f;
		for (int it = 0; it < 40; ++it) {
			const float t1 = lo + (hi - lo) * (1.0f / 3.0f), t2 = hi - (hi - lo) * (1.0f / 3.0f);
			const float f1 = sgo_hull_closest(H->h, v3_add(s0, v3_scale(d, t1)), &q, &n);
			const float f2 = sgo_hull_closest(H->h, v3_add(s0, v3_scale(d, t2)), &q, &n);
			if (f1 <= f2) hi = t2; else lo = t1;
		}
		ts = 0.5f * (lo + hi);
	}
	const v3 S = v3_add(s0, v3_scale(d, ts));
	const float dist = sgo_hull_closest(H->h, S, &q, &n);
	if (dist - r > max_sep) return 0;
	m->n = m33_mul(H->R, n); m->np = 1;
	m->p1[0] = v3_add(H->pos, m33_mul(H->R, q));
	m->p2[0] = v3_add(H->pos, m33_mul(H->R, v3_sub(S, v3_scale(n, r))));
	/* axis (nearly) parallel to the supporting face: both ends of the overlap between the axis and that face */
	const float dl = v3_len(d);
	if (dl > 1.0e-6f && dist > 0.0f && fabsf(v3_dot(n, d)) < SGO_CAPSULE_SLOP * dl) {
		int fb = 0; float bd = -3.4e38f;
		for (int f = 0; f < H->h->nf; ++f) { const float dd = v3_dot(H->h->normals[f], n); if (dd > bd) { bd = dd; fb = f; } }
		if (bd > 0.95f) {
			const v3 nf = H->h->normals[fb];
			float t0 = 0.0f, t1 = 1.0f; int ok = 1;
			const int k0 = H->h->face_start[fb], k1 = H->h->face_start[fb + 1];
			for (int k = k0; k < k1 && ok; ++k) {
				const v3 a = H->h->verts[H->h->face_idx[k]], b = H->h->verts[H->h->face_idx[k + 1 < k1 ? k + 1 : k0]];
				const v3 side = v3_cross(v3_sub(b, a), nf);
				const float g0 = v3_dot(side, v3_sub(s0, a)), gd = v3_dot(side, d);
				if (fabsf(gd) < 1.0e-12f) { if (g0 > 0.0f) ok = 0; }
				else { const float tk = -g0 / gd; if (gd > 0.0f) { if (tk < t1) t1 = tk; } else { if (tk > t0) t0 = tk; } if (t0 > t1) ok = 0; }
			}
			if (ok && (t1 - t0) * dl > 1.0e-4f) {
				const float tt[2] = { t0, t1 };
				v3 a1[2], a2[2]; int np = 0;
				for (int i = 0; i < 2; ++i) {
					const v3 P = v3_add(s0, v3_scale(d, tt[i]));
					const float sep = v3_dot(nf, P) - H->h->plane_d[fb] - r;
					if (sep <= max_sep) {
						a1[np] = v3_sub(P, v3_scale(nf, v3_dot(nf, P) - H->h->plane_d[fb]));
						a2[np] = v3_sub(P, v3_scale(n, r));
						++np;
					}
				}
				if (np == 2) {
					m->np = 2;
					for (int i = 0; i < 2; ++i) { m->p1[i] = v3_add(H->pos, m33_mul(H->R, a1[i])); m->p2[i] = v3_add(H->pos, m33_mul(H->R, a2[i])); }
				}
			}
		}
	}
	return 1;
}

/* ray against a hull (scale 1) grown by `grow` along every face normal, hull-local: clip against every face plane.  Returns t
   or -1; normal of the entry face (-dl when the origin is inside).  grow > 0 serves the sphere cast of the wheel tester: the
   planes-only offset is the Minkowski sum with a ball except near edges and corners, where it is slightly larger. */
static inline float sgo_ray_hull(const sgo_hull* h, v3 ol, v3 dl, float max_t, float grow, v3* n_out)
{
	float t0 = 0.0f, t1 = max_t; int fin = -1;
	for (int f = 0; f < h->nf; ++f) {
		const v3 n = h->normals[f];
		const float den = v3_dot(n, dl), num = (h->plane_d[f] + grow) - v3_dot(n, ol);
		if (fabsf(den) < 1.0e-12f) { if (num < 0.0f) return -1.0f; continue; }
		const float t = num / den;
		if (den < 0.0f) { if (t > t0) { t0 = t; fin = f; } } else { if (t < t1) t1 = t; }
		if (t0 > t1) return -1.0f;
	}
	if (fin < 0) { *n_out = v3_neg(dl); return 0.0f; }
	*n_out = h->normals[fin];
	return t0;
}

#endif
