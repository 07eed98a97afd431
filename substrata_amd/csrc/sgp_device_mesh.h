// sgp_device_mesh.h -- gfx950 static triangle meshes: per-triangle collision (a triangle is a thin 3-vertex hull), grouping of the
// triangle manifolds of one body pair by normal (<= 3 groups, <= 4 points each), ray - triangle (device code only).
//
// Role of JPH::MeshShape / HeightFieldShape in CollideShape / CastRay for Substrata's static meshes and terrain
// (/root/reference/gui_client/PhysicsWorld.cpp:735-1166 with is_dynamic = false, :1020-1120; TerrainSystem.cpp:1300).
// Included after sgp_device_collide.h.
#pragma once
#include "sgp_device_vehicle.h"     // sgd_ray_sphere / sgd_ray_capsule_z, sgd_hull through sgp_device_collide.h

#define SGD_MESH_MAX_GROUPS 3
#define SGD_MESH_GROUP_COS 0.95f

// the thin hull of one triangle; vertices relative to the centroid (mesh frame).  The record has the members of sgd_hull the collision
// functions read, with room for exactly one triangle: 100 bytes a lane can keep near, where the full record is 2.2 KB of scratch memory
struct sgd_tri_hull_t {
	int nv, nf, ne, is_box_template;
	v3 verts[3]; v3 normals[2]; float plane_d[2];
	unsigned char face_start[3], face_idx[6], edge_a[3], edge_b[3];
};
template <> struct sgd_is_thin<sgd_tri_hull_t> { static constexpr bool value = true; };
typedef sgd_hview_t<sgd_tri_hull_t> sgd_tri_view;
SGP_DEV static void sgd_tri_hull(v3 a, v3 b, v3 c, sgd_tri_hull_t* h, v3* centroid_out, v3* normal_out)
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
	h->face_idx[0] = 0; h->face_idx[1] = 1; h->face_idx[2] = 2;          // counter-clockwise seen from +n
	h->face_idx[3] = 0; h->face_idx[4] = 2; h->face_idx[5] = 1;
	h->edge_a[0] = 0; h->edge_b[0] = 1; h->edge_a[1] = 1; h->edge_b[1] = 2; h->edge_a[2] = 0; h->edge_b[2] = 2;
	*centroid_out = cen; *normal_out = n;
}

struct sgd_mesh_group { v3 n; int np; v3 p_mesh[SGD_HULL_CLIP_CAP]; v3 p_body[SGD_HULL_CLIP_CAP]; };
struct sgd_mesh_contacts { int ng; sgd_mesh_group g[SGD_MESH_MAX_GROUPS]; };

// m: manifold of one triangle, normal from the triangle to the body, p1 on the triangle, p2 on the body
SGP_DEV static void sgd_mesh_add(sgd_mesh_contacts* mc, const sgd_manifold* m)
{
	int gi = -1;
	for (int k = 0; k < mc->ng; ++k) if (v3_dot(mc->g[k].n, m->n) >= SGD_MESH_GROUP_COS) { gi = k; break; }
	if (gi < 0) {
		if (mc->ng == SGD_MESH_MAX_GROUPS) return;
		gi = mc->ng++;
		mc->g[gi].n = m->n; mc->g[gi].np = 0;
	}
	sgd_mesh_group* g = &mc->g[gi];
	for (int i = 0; i < m->np; ++i) {
		if (g->np == SGD_HULL_CLIP_CAP) break;
		// the same point reached through two triangles that share it (an edge or a vertex of the mesh) counts once
		int dup = 0;
		for (int j = 0; j < g->np; ++j) if (v3_len_sq(v3_sub(g->p_body[j], m->p2[i])) < 1.0e-8f) { dup = 1; break; }
		if (dup) continue;
		g->p_mesh[g->np] = m->p1[i]; g->p_body[g->np] = m->p2[i]; g->np++;
	}
}

// X against one triangle (world-space view T of its thin hull, world normal nt).  Normal of the result: triangle -> X.
SGP_DEV static int sgd_collide_tri(const sgd_shape* X, const sgd_tri_view* T, v3 nt, float max_sep, sgd_manifold* m)
{
	int hit;
	if (X->type == SGD_SHAPE_SPHERE) hit = sgd_hull_sphere(T, X->pos, X->p0, max_sep, m);
	else if (X->type == SGD_SHAPE_CAPSULE) {
		const v3 ax = v3_scale(m33_col(X->R, 2), X->p1);
		hit = sgd_hull_capsule(T, v3_sub(X->pos, ax), v3_add(X->pos, ax), X->p0, max_sep, m);
	} else {
		sgd_hview hx;
		hx.pos = X->pos; hx.R = X->R; hx.h = X->hull;
		hx.scale = X->type == SGD_SHAPE_BOX ? V3(X->p0, X->p1, X->p2) : V3(1.0f, 1.0f, 1.0f);
		hit = sgd_hull_hull(T, &hx, max_sep, m);
	}
	if (!hit) return 0;
	if (v3_dot(m->n, nt) < 0.0f) return 0;                   // reached from the back side
	return 1;
}

// the groups as manifolds (normal mesh -> body, p1 on the mesh, p2 on the body), each pruned to <= 4 points
SGP_DEV static int sgd_mesh_finish(const sgd_mesh_contacts* mc, sgd_manifold* out)
{
	for (int k = 0; k < mc->ng; ++k) sgd_hull_reduce(mc->g[k].n, mc->g[k].p_mesh, mc->g[k].p_body, mc->g[k].np, &out[k]);
	return mc->ng;
}

/* ray against one triangle (Moeller-Trumbore, front face only): t or -1; uv_out (may be NULL) = barycentric coordinates of the hit,
   point = (1 - u - v) a + u b + v c */
SGP_DEV static float sgd_ray_tri_uv(v3 o, v3 d, v3 a, v3 b, v3 c, float max_t, float* uv_out)
{
	const v3 e1 = v3_sub(b, a), e2 = v3_sub(c, a);
	const v3 pv = v3_cross(d, e2);
	const float det = v3_dot(e1, pv);
	if (det < 1.0e-12f) return -1.0f;                          // parallel or hitting the back face
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
SGP_DEV static float sgd_ray_tri(v3 o, v3 d, v3 a, v3 b, v3 c, float max_t) { return sgd_ray_tri_uv(o, d, a, b, c, max_t, (float*)0); }

// ray against a capsule with end points a, b and radius r (any orientation): t or -1, normal at the hit
SGP_DEV static float sgd_ray_capsule_seg(v3 o, v3 d, v3 a, v3 b, float r, float max_t, v3* n_out)
{
	const v3 ab = v3_sub(b, a);
	const float len = v3_len(ab);
	if (len < 1.0e-12f) { const float t = sgd_ray_sphere(v3_sub(o, a), d, r, max_t, n_out); return t; }
	const v3 ez = v3_scale(ab, 1.0f / len);
	const v3 ex = v3_normalized_perpendicular(ez);
	const v3 ey = v3_cross(ez, ex);
	const v3 mid = v3_scale(v3_add(a, b), 0.5f);
	const v3 ro = v3_sub(o, mid);
	const v3 ol = V3(v3_dot(ro, ex), v3_dot(ro, ey), v3_dot(ro, ez)), dl = V3(v3_dot(d, ex), v3_dot(d, ey), v3_dot(d, ez));
	v3 nl;
	const float t = sgd_ray_capsule_z(ol, dl, r, 0.5f * len, max_t, &nl);
	if (t < 0.0f) return -1.0f;
	*n_out = v3_add(v3_add(v3_scale(ex, nl.x), v3_scale(ey, nl.y)), v3_scale(ez, nl.z));
	return t;
}

/* A sphere of radius rs moving from o along d against the FRONT of triangle (a, b, c): the face plane moved out by rs, plus the
   three edges as capsules (they cover the vertices too).  Returns the travel distance or -1; n_out = normal at the touch point. */
SGP_DEV static float sgd_cast_sphere_tri(v3 o, v3 d, v3 a, v3 b, v3 c, float max_t, float rs, v3* n_out)
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
	const float tf = sgd_ray_tri(o, d, v3_add(a, off), v3_add(b, off), v3_add(c, off), lim);
	if (tf >= 0.0f) { best = tf; bn = nt; lim = tf; }
	if (rs > 0.0f) {
		const v3 ea[3] = { a, b, c }, eb[3] = { b, c, a };
		for (int k = 0; k < 3; ++k) {
			v3 nn = nt;
			const float tk = sgd_ray_capsule_seg(o, d, ea[k], eb[k], rs, lim, &nn);
			if (tk >= 0.0f && v3_dot(nn, nt) >= 0.0f && (best < 0.0f || tk < best)) { best = tk; bn = nn; lim = tk; }
		}
	}
	if (best < 0.0f) return -1.0f;
	*n_out = bn;
	return best;
}

