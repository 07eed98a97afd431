/*
 * sgo_mesh.h -- ORACLE (test infrastructure only): static triangle-mesh shapes.
 *
 * What it restates.  Substrata gives every static mesh object a JPH::MeshShape and the terrain a JPH::HeightFieldShape
 * (/root/reference/gui_client/PhysicsWorld.cpp:735-1166 createJoltShapeFor...Mesh with is_dynamic = false,
 * createJoltHeightFieldShape :1020-1120; TerrainSystem.cpp:1300).  Jolt (v5.3.0, not in the tree) walks the mesh's tree with the
 * other shape's bounds and collides the convex shape with every triangle it reaches (back faces ignored), each hit becoming a
 * manifold; manifolds of one body pair with similar normals are merged and pruned to 4 points.  Restated from upstream
 * knowledge (parity unpinned; pinned by tests/test_oracle_mesh.py):
 *   - a triangle is collided as a thin convex hull (3 vertices, front and back face, 3 edges) with the hull routines of
 *     sgo_hull.h (SAT + clipping for boxes / hulls, closest point for spheres / capsules); contacts whose normal points to the
 *     triangle's back side are dropped;
 *   - the triangles reached by a body are taken in index order; their manifolds are grouped by normal (within ~18 degrees of the
 *     group's first normal), at most 3 groups per body pair, each group pruned to 4 points -- a body in a corner of the mesh
 *     keeps one constraint per wall.  The groups are carried by the mesh body and its two alias body slots, so that the
 *     (body a, body b) constraint key needs no sub-shape part.
 *   - active edges (round 4; MeshShapeSettings::mActiveEdgeCosThresholdAngle = cos 5 deg, PhysicsWorld.cpp:1028-1060 leaves the default;
 *     CollideShapeSettings::mActiveEdgeMode = CollideOnlyWithActive with the bodies' relative velocity as movement hint, as
 *     PhysicsSystem::ProcessBodyPair sets them): an edge shared by two triangles is INACTIVE when it is concave or its triangles' normals are
 *     within 5 degrees of each other (sgo_mesh_active_edges, restating MeshShape::sFindActiveEdges + ActiveEdges::IsEdgeActive); a contact whose
 *     normal is not the triangle's and whose deepest point lies on an inactive edge / vertex takes the triangle's normal instead
 *     (sgo_active_edge_fix, restating ActiveEdges::FixNormal) -- a body sliding over the seams of a flat triangulated floor meets no bumps.
 *     UNVERIFIED: upstream.
 */
#ifndef SGO_MESH_H
#define SGO_MESH_H

#include "sgo_hull.h"
/* (sgo_ray_sphere / sgo_ray_capsule_z come from sgo_vehicle.h, included before this header) */

#define SGO_MESH_MAX_GROUPS 3
#define SGO_MESH_GROUP_COS 0.95f
#define SGO_ACTIVE_EDGE_COS 0.99619469809f      /* cos(5 degrees): MeshShapeSettings::mActiveEdgeCosThresholdAngle */

/* ---- which edges of which triangles are active (mesh build time, doubles) ---------------------------------------------------------- */
typedef struct { uint32_t lo, hi, tri, k; } sgo_edge_rec;
static int sgo_edge_rec_cmp(const void* x, const void* y)
{
	const sgo_edge_rec* a = (const sgo_edge_rec*)x; const sgo_edge_rec* b = (const sgo_edge_rec*)y;
	if (a->lo != b->lo) return a->lo < b->lo ? -1 : 1;
	if (a->hi != b->hi) return a->hi < b->hi ? -1 : 1;
	if (a->tri != b->tri) return a->tri < b->tri ? -1 : 1;
	return a->k < b->k ? -1 : (a->k > b->k ? 1 : 0);
}
static int sgo_tri_normal_d(const float* verts, const uint32_t* t, double n[3])
{
	const float* a = verts + 3 * t[0]; const float* b = verts + 3 * t[1]; const float* c = verts + 3 * t[2];
	const double e1[3] = { (double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2] }, e2[3] = { (double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2] };
	n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
	const double l = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
	if (!(l > 1.0e-30)) return 0;
	n[0] /= l; n[1] /= l; n[2] /= l;
	return 1;
}
/* flags[t] bit k set = edge k of triangle t (k = 0: v0-v1, 1: v1-v2, 2: v2-v0) is active.  An edge is keyed by its two vertex INDICES (as Jolt's
   indexed triangle list): used by one triangle or by more than two -> active; by two -> ActiveEdges::IsEdgeActive of their normals. */
static void sgo_mesh_active_edges(const float* verts, const uint32_t* idx, uint32_t nt, double cos_threshold, unsigned char* flags)
{
	sgo_edge_rec* e = (sgo_edge_rec*)malloc(sizeof(sgo_edge_rec) * 3 * (size_t)nt);
	for (uint32_t t = 0; t < nt; ++t) for (uint32_t k = 0; k < 3; ++k) {
		const uint32_t a = idx[3 * t + k], b = idx[3 * t + (k + 1) % 3];
		sgo_edge_rec r; r.lo = a < b ? a : b; r.hi = a < b ? b : a; r.tri = t; r.k = k;
		e[3 * (size_t)t + k] = r;
		flags[t] = 7;
	}
	qsort(e, 3 * (size_t)nt, sizeof(sgo_edge_rec), sgo_edge_rec_cmp);
	for (size_t i = 0; i < 3 * (size_t)nt; ) {
		size_t j = i + 1;
		while (j < 3 * (size_t)nt && e[j].lo == e[i].lo && e[j].hi == e[i].hi) ++j;
		if (j - i == 2 && e[i].lo != e[i].hi) {
			double n1[3], n2[3];
			if (sgo_tri_normal_d(verts, idx + 3 * e[i].tri, n1) && sgo_tri_normal_d(verts, idx + 3 * e[i + 1].tri, n2)) {
				const uint32_t va = idx[3 * e[i].tri + e[i].k], vb = idx[3 * e[i].tri + (e[i].k + 1) % 3];       /* the edge in the first triangle's winding */
				const double d[3] = { (double)verts[3 * vb] - verts[3 * va], (double)verts[3 * vb + 1] - verts[3 * va + 1], (double)verts[3 * vb + 2] - verts[3 * va + 2] };
				const double cosn = n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
				const double cx = n1[1] * n2[2] - n1[2] * n2[1], cy = n1[2] * n2[0] - n1[0] * n2[2], cz = n1[0] * n2[1] - n1[1] * n2[0];
				int active;
				if (cosn < -0.999848) active = 1;                                   /* back to back */
				else if (cx * d[0] + cy * d[1] + cz * d[2] < 0.0) active = 0;       /* concave */
				else active = cosn < cos_threshold;                                 /* convex: active beyond the threshold angle */
				if (!active) { flags[e[i].tri] &= (unsigned char)~(1u << e[i].k); flags[e[i + 1].tri] &= (unsigned char)~(1u << e[i + 1].k); }
			}
		}
		i = j;
	}
	free(e);
}

/* ---- ActiveEdges::FixNormal: does the triangle's normal nt replace the contact normal n (both unit, triangle -> body)? ----------------------
   a, b, c: the triangle (world), edges: its active-edge bits, p: the contact point on the triangle that decides (the deepest one), movement: velocity
   of the body relative to the mesh.  Jolt's axes point the other way (convex -> triangle): its test m . n_J < m . t_J reads m . n > m . nt here. */
static inline int sgo_active_edge_fix(v3 a, v3 b, v3 c, v3 nt, unsigned edges, v3 p, v3 n, v3 movement)
{
	if (edges == 7u) return 0;
	if (v3_dot(movement, n) > v3_dot(movement, nt)) return 0;        /* the computed normal opposes the motion less than the triangle's: keep it */
	if (edges == 0u) return 1;
	if (v3_dot(nt, n) > 0.999848f) return 0;                          /* within a degree of the triangle's normal anyway */
	/* barycentric coordinates of p (weights of a, b, c) */
	const v3 v0 = v3_sub(b, a), v1 = v3_sub(c, a), v2 = v3_sub(p, a);
	const float d00 = v3_dot(v0, v0), d01 = v3_dot(v0, v1), d11 = v3_dot(v1, v1), d20 = v3_dot(v2, v0), d21 = v3_dot(v2, v1);
	const float den = d00 * d11 - d01 * d01;
	if (!(fabsf(den) > 1.0e-20f)) return 0;
	const float bv = (d11 * d20 - d01 * d21) / den, bw = (d00 * d21 - d01 * d20) / den, bu = (1.0f - bv) - bw;
	const float eps = 1.0e-4f, one = 1.0f - 1.0e-4f;
	unsigned colliding;
	if (bu > one) colliding = 5u;            /* vertex a: edge 0 or 2 */
	else if (bv > one) colliding = 3u;       /* vertex b: edge 0 or 1 */
	else if (bw > one) colliding = 6u;       /* vertex c: edge 1 or 2 */
	else if (bu < eps) colliding = 2u;       /* edge b - c */
	else if (bv < eps) colliding = 4u;       /* edge c - a */
	else if (bw < eps) colliding = 1u;       /* edge a - b */
	else return 0;                           /* interior */
	return (edges & colliding) ? 0 : 1;
}

/* the thin hull of one triangle; vertices relative to the centroid (mesh frame) */
static inline void sgo_tri_hull(v3 a, v3 b, v3 c, sgo_hull* h, v3* centroid_out, v3* normal_out)
{
	const v3 cen = v3_scale(v3_add(v3_add(a, b), c), 1.0f / 3.0f);
	v3 n = v3_cross(v3_sub(b, a), v3_sub(c, a));
	const float l = v3_len(n);
	n = l > 1.0e-20f ? v3_scale(n, 1.0f / l) : V3(0.0f, 0.0f, 1.0f);
	h->nv = 3; h->nf = 2; h->ne = 3; h->is_box_template = 0;
	h->verts[0] = v3_sub(a, cen); h->verts[1] = v3_sub(b, cen); h->verts[2] = v3_sub(c, cen);
	h->normals[0] = n; h->plane_d[0] = v3_dot(n, h->verts[0]);
	h->normals[1] = v3_neg(n); h->plane_d[1] = -h->plane_d[0];
	h->face_start[0] = 0; h->face_start[1] = 3; h->face_start[2] = 6;
	h->face_idx[0] = 0; h->face_idx[1] = 1; h->face_idx[2] = 2;          /* counter-clockwise seen from +n */
	h->face_idx[3] = 0; h->face_idx[4] = 2; h->face_idx[5] = 1;
	h->edge_a[0] = 0; h->edge_b[0] = 1; h->edge_a[1] = 1; h->edge_b[1] = 2; h->edge_a[2] = 0; h->edge_b[2] = 2;
	*centroid_out = cen; *normal_out = n;
}

typedef struct { v3 n; int np; v3 p_mesh[SGO_HULL_CLIP_CAP]; v3 p_body[SGO_HULL_CLIP_CAP]; } sgo_mesh_group;
typedef struct { int ng; sgo_mesh_group g[SGO_MESH_MAX_GROUPS]; } sgo_mesh_contacts;

/* m: manifold of one triangle, normal from the triangle to the body, p1 on the triangle, p2 on the body */
static inline void sgo_mesh_add(sgo_mesh_contacts* mc, const sgo_manifold* m)
{
	int gi = -1;
	for (int k = 0; k < mc->ng; ++k) if (v3_dot(mc->g[k].n, m->n) >= SGO_MESH_GROUP_COS) { gi = k; break; }
	if (gi < 0) {
		if (mc->ng == SGO_MESH_MAX_GROUPS) return;
		gi = mc->ng++;
		mc->g[gi].n = m->n; mc->g[gi].np = 0;
	}
	sgo_mesh_group* g = &mc->g[gi];
	for (int i = 0; i < m->np; ++i) {
		if (g->np == SGO_HULL_CLIP_CAP) break;
		/* the same point reached through two triangles that share it (an edge or a vertex of the mesh) counts once */
		int dup = 0;
		for (int j = 0; j < g->np; ++j) if (v3_len_sq(v3_sub(g->p_body[j], m->p2[i])) < 1.0e-8f) { dup = 1; break; }
		if (dup) continue;
		g->p_mesh[g->np] = m->p1[i]; g->p_body[g->np] = m->p2[i]; g->np++;
	}
}

/* X against one triangle (world-space view T of its thin hull, world normal nt).  Normal of the result: triangle -> X.
   edges: the triangle's active-edge bits (7: no fixing, e.g. a shape query), movement: X's velocity relative to the mesh. */
static inline int sgo_collide_tri(const sgo_shape* X, const sgo_hview* T, v3 nt, float max_sep, sgo_manifold* m, unsigned edges, v3 movement)
{
	int hit;
	if (X->type == SGO_SHAPE_SPHERE) hit = sgo_hull_sphere(T, X->pos, X->p[0], max_sep, m);
	else if (X->type == SGO_SHAPE_CAPSULE) {
		const v3 ax = v3_scale(m33_col(X->R, 2), X->p[1]);
		hit = sgo_hull_capsule(T, v3_sub(X->pos, ax), v3_add(X->pos, ax), X->p[0], max_sep, m);
	} else {
		sgo_hview hx;
		hx.pos = X->pos; hx.R = X->R; hx.h = X->hull;
		hx.scale = X->type == SGO_SHAPE_BOX ? V3(X->p[0], X->p[1], X->p[2]) : V3(1.0f, 1.0f, 1.0f);
		hit = sgo_hull_hull(T, &hx, max_sep, m);
	}
	if (!hit) return 0;
	if (v3_dot(m->n, nt) < 0.0f) return 0;                   /* reached from the back side */
	if (edges != 7u && m->np > 0) {
		/* the point that decides: the deepest one (Jolt has one point at this stage, the deepest) */
		int bi = 0; float bd = -3.4e38f;
		for (int i = 0; i < m->np; ++i) { const float dd = v3_dot(v3_sub(m->p1[i], m->p2[i]), m->n); if (dd > bd) { bd = dd; bi = i; } }
		if (sgo_active_edge_fix(sgo_hv_world(T, 0), sgo_hv_world(T, 1), sgo_hv_world(T, 2), nt, edges, m->p1[bi], m->n, movement)) {
			if (X->type == SGO_SHAPE_SPHERE || X->type == SGO_SHAPE_CAPSULE) m->n = nt;      /* the points stay, the direction changes */
			else {
				/* a polytope: the contact as the triangle's FACE makes it (reference face = the triangle's front, clipped incident face of X) */
				sgo_hview hx;
				hx.pos = X->pos; hx.R = X->R; hx.h = X->hull;
				hx.scale = X->type == SGO_SHAPE_BOX ? V3(X->p[0], X->p[1], X->p[2]) : V3(1.0f, 1.0f, 1.0f);
				if (!sgo_hull_face_contact(T, &hx, 0, 1, max_sep, m)) return 0;
			}
		}
	}
	return 1;
}

/* the groups as manifolds (normal mesh -> body, p1 on the mesh, p2 on the body), each pruned to <= 4 points */
static inline int sgo_mesh_finish(const sgo_mesh_contacts* mc, sgo_manifold* out)
{
	for (int k = 0; k < mc->ng; ++k) sgo_hull_reduce(mc->g[k].n, mc->g[k].p_mesh, mc->g[k].p_body, mc->g[k].np, &out[k]);
	return mc->ng;
}

/* ray against one triangle (Moeller-Trumbore, front face only): t or -1; uv_out (may be NULL) = barycentric coordinates of the hit,
   point = (1 - u - v) a + u b + v c */
static inline float sgo_ray_tri_uv(v3 o, v3 d, v3 a, v3 b, v3 c, float max_t, float* uv_out)
{
	const v3 e1 = v3_sub(b, a), e2 = v3_sub(c, a);
	const v3 pv = v3_cross(d, e2);
	const float det = v3_dot(e1, pv);
	if (det < 1.0e-12f) return -1.0f;                          /* parallel or hitting the back face */
	const v3 tv = v3_sub(o, a);
	const float u = v3_dot(tv, pv);
	if (u < 0.0f || u > det) return -1.0f;
	const v3 qv = v3_cross(tv, e1);
	const float vv = v3_dot(d, qv);
	if (vv < 0.0f || u + vv > det) return -1.0f;
	const float t = v3_dot(e2, qv) / det;
	if (t < 0.0f || t > max_t) return -1.0f;
	if (uv_out) { uv_out[0] = u / det; uv_out[1] = vv / det; }
	return t;
}
static inline float sgo_ray_tri(v3 o, v3 d, v3 a, v3 b, v3 c, float max_t) { return sgo_ray_tri_uv(o, d, a, b, c, max_t, (float*)0); }

/* ray against a capsule with end points a, b and radius r (any orientation): t or -1, normal at the hit */
static inline float sgo_ray_capsule_seg(v3 o, v3 d, v3 a, v3 b, float r, float max_t, v3* n_out)
{
	const v3 ab = v3_sub(b, a);
	const float len = v3_len(ab);
	if (len < 1.0e-12f) { const float t = sgo_ray_sphere(v3_sub(o, a), d, r, max_t, n_out); return t; }
	const v3 ez = v3_scale(ab, 1.0f / len);
	const v3 ex = v3_normalized_perpendicular(ez);
	const v3 ey = v3_cross(ez, ex);
	const v3 mid = v3_scale(v3_add(a, b), 0.5f);
	const v3 ro = v3_sub(o, mid);
	const v3 ol = V3(v3_dot(ro, ex), v3_dot(ro, ey), v3_dot(ro, ez)), dl = V3(v3_dot(d, ex), v3_dot(d, ey), v3_dot(d, ez));
	v3 nl;
	const float t = sgo_ray_capsule_z(ol, dl, r, 0.5f * len, max_t, &nl);
	if (t < 0.0f) return -1.0f;
	*n_out = v3_add(v3_add(v3_scale(ex, nl.x), v3_scale(ey, nl.y)), v3_scale(ez, nl.z));
	return t;
}

/* A sphere of radius rs moving from o along d against the FRONT of triangle (a, b, c): the face plane moved out by rs, plus the
   three edges as capsules (they cover the vertices too).  Returns the travel distance or -1; n_out = normal at the touch point. */
static inline float sgo_cast_sphere_tri(v3 o, v3 d, v3 a, v3 b, v3 c, float max_t, float rs, v3* n_out)
{
	v3 nt = v3_cross(v3_sub(b, a), v3_sub(c, a));
	const float l = v3_len(nt);
	if (l < 1.0e-20f) return -1.0f;
	nt = v3_scale(nt, 1.0f / l);
	float best = -1.0f; v3 bn = nt; float lim = max_t;
	const v3 off = v3_scale(nt, rs);
	if (rs > 0.0f && v3_dot(d, nt) < 0.0f) {
		/* the sphere STARTS in touch with the face's interior (centre less than rs in front of the plane, its foot point inside the triangle)
		   and moves into it: that is a hit at distance 0 (JPH::CastShape reports fraction 0 for an initial overlap); the offset-plane test
		   below only sees a centre that is still in front of the offset plane */
		const float h = v3_dot(v3_sub(o, a), nt);
		if (h >= 0.0f && h < rs) {
			const v3 q = v3_sub(o, v3_scale(nt, h));
			const float e0 = v3_dot(v3_cross(v3_sub(b, a), v3_sub(q, a)), nt), e1 = v3_dot(v3_cross(v3_sub(c, b), v3_sub(q, b)), nt), e2 = v3_dot(v3_cross(v3_sub(a, c), v3_sub(q, c)), nt);
			if (e0 >= 0.0f && e1 >= 0.0f && e2 >= 0.0f) { *n_out = nt; return 0.0f; }
		}
	}
	const float tf = sgo_ray_tri(o, d, v3_add(a, off), v3_add(b, off), v3_add(c, off), lim);
	if (tf >= 0.0f) { best = tf; bn = nt; lim = tf; }
	if (rs > 0.0f) {
		const v3 ea[3] = { a, b, c }, eb[3] = { b, c, a };
		for (int k = 0; k < 3; ++k) {
			v3 nn = nt;
			const float tk = sgo_ray_capsule_seg(o, d, ea[k], eb[k], rs, lim, &nn);
			if (tk >= 0.0f && v3_dot(nn, nt) >= 0.0f && (best < 0.0f || tk < best)) { best = tk; bn = nn; lim = tk; }
		}
	}
	if (best < 0.0f) return -1.0f;
	*n_out = bn;
	return best;
}

#endif
