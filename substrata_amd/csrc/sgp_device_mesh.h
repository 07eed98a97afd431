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
// uint4.w of a mesh triangle: index in the caller's order | active-edge flags << 29 (bit k: edge vertex k -> vertex k + 1 collides with its own normal)
#define MESH_TRI_INDEX(w) ((w) & 0x1FFFFFFFu)
#define MESH_TRI_EDGES(w) ((w) >> 29)

// ActiveEdges::FixNormal (round 4): does the triangle's normal nt replace the contact normal n (both unit, triangle -> body)?  a, b, c: the triangle
// (world), edges: its active-edge bits, p: the contact point on the triangle that decides (the deepest one), movement: velocity of the body relative to
// the mesh.  Jolt's axes point the other way (convex -> triangle): its test m . n_J < m . t_J reads m . n > m . nt here.
SGP_DEV static int sgd_active_edge_fix(v3 a, v3 b, v3 c, v3 nt, unsigned edges, v3 p, v3 n, v3 movement)
{
	if (edges == 7u) return 0;
	if (v3_dot(movement, n) > v3_dot(movement, nt)) return 0;        // the computed normal opposes the motion less than the triangle's: keep it
	if (edges == 0u) return 1;
	if (v3_dot(nt, n) > 0.999848f) return 0;                          // within a degree of the triangle's normal anyway
	// barycentric coordinates of p (weights of a, b, c)
	const v3 v0 = v3_sub(b, a), v1 = v3_sub(c, a), v2 = v3_sub(p, a);
	const float d00 = v3_dot(v0, v0), d01 = v3_dot(v0, v1), d11 = v3_dot(v1, v1), d20 = v3_dot(v2, v0), d21 = v3_dot(v2, v1);
	const float den = d00 * d11 - d01 * d01;
	if (!(fabsf(den) > 1.0e-20f)) return 0;
	const float bv = (d11 * d20 - d01 * d21) / den, bw = (d00 * d21 - d01 * d20) / den, bu = (1.0f - bv) - bw;
	const float eps = 1.0e-4f, one = 1.0f - 1.0e-4f;
	unsigned colliding;
	if (bu > one) colliding = 5u;            // vertex a: edge 0 or 2
	else if (bv > one) colliding = 3u;       // vertex b: edge 0 or 1
	else if (bw > one) colliding = 6u;       // vertex c: edge 1 or 2
	else if (bu < eps) colliding = 2u;       // edge b - c
	else if (bv < eps) colliding = 4u;       // edge c - a
	else if (bw < eps) colliding = 1u;       // edge a - b
	else return 0;                           // interior
	return (edges & colliding) ? 0 : 1;
}

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
#pragma unroll
	for (int i = 0; i < 4; ++i) {      // (a triangle's manifold: at most four points)
		if (i >= m->np) break;
		if (g->np == SGD_HULL_CLIP_CAP) break;
		// the same point reached through two triangles that share it (an edge or a vertex of the mesh) counts once
		int dup = 0;
		for (int j = 0; j < g->np; ++j) if (v3_len_sq(v3_sub(g->p_body[j], m->p2[i])) < 1.0e-8f) { dup = 1; break; }
		if (dup) continue;
		g->p_mesh[g->np] = m->p1[i]; g->p_body[g->np] = m->p2[i]; g->np++;
	}
}

// does the active-edge rule replace the normal of manifold m (X against triangle T, world normal nt) by the triangle's?
SGP_DEV static bool sgd_tri_needs_face_normal(const sgd_tri_view* T, v3 nt, unsigned edges, v3 movement, const sgd_manifold* m)
{
	if (edges == 7u || m->np <= 0) return false;
	// the point that decides: the deepest one (Jolt has one point at this stage, the deepest)
	// (a triangle's manifold has at most four points by now -- sgd_hull_reduce --: loops over the four slots with the count as a predicate keep it in registers)
	int bi = 0; float bd = -3.4e38f;
#pragma unroll
	for (int i = 0; i < 4; ++i) if (i < m->np) { const float dd = v3_dot(v3_sub(m->p1[i], m->p2[i]), m->n); if (dd > bd) { bd = dd; bi = i; } }
	const v3 pb = bi == 0 ? m->p1[0] : (bi == 1 ? m->p1[1] : (bi == 2 ? m->p1[2] : m->p1[3]));
	return sgd_active_edge_fix(sgd_hv_world(T, 0), sgd_hv_world(T, 1), sgd_hv_world(T, 2), nt, edges, pb, m->n, movement) != 0;
}
// ---- a mesh triangle against a BOX: the separating-axis search of sgd_hull_sat_search(T, box) in closed form ----------------------------
// The general search walks the cube template like any hull: eight corners per projection, tables of direction pairs indexed at run time (720 B of
// scratch memory per lane), twelve edges.  For the +-1 cube scaled by the half extents every one of those sums has a closed form with THE SAME BITS:
//   min over the corners of  l . (+-sx, +-sy, +-sz)  =  -((|l.x sx| + |l.y sy|) + |l.z sz|)
// (a product with a negated factor is the negated product, a sum of negated terms the negated sum, and rounding is monotonic -- so the all-negative corner
// is the minimum of the eight rounded sums, whichever corner the loop met first), and an edge pair's axis depends only on the DIRECTION of the cube edge:
// the twelve edges are 3 directions x 2 senses, the sense only decides which of +-axis survives the "points from the triangle to the box" flip -- the same
// axis either way unless axis . T is exactly zero, which is kept apart below.  What remains per edge pair is the supporting test and "first maximum wins"
// in the template's edge order, which is read from the template (sgd_box_code_of), not assumed.
struct sgd_box_code { uint64_t bits; };      // per cube edge j five bits: direction k 0..2 | sense << 2 | (corner a negative along the two other axes (k + 1) % 3, (k + 2) % 3) << 3
SGP_DEV static sgd_box_code sgd_box_code_of(const sgd_hull* t)
{
	sgd_box_code c; c.bits = 0ull;
#pragma unroll
	for (int j = 0; j < 12; ++j) {
		const v3 va = t->verts[t->edge_a[j]], vb = t->verts[t->edge_b[j]];
		const v3 e = v3_sub(vb, va);
		const uint32_t cb = e.x != 0.0f ? (e.x < 0.0f ? 1u : 0u) : (e.y != 0.0f ? (e.y < 0.0f ? 3u : 2u) : (e.z < 0.0f ? 5u : 4u));      // (the classes of sgd_hull_sat_search)
		const uint32_t k = cb >> 1;
		// (along k itself the edge starts at -1 when it runs the positive way and at +1 otherwise: the sense says it)
		const float u = k == 0u ? va.y : (k == 1u ? va.z : va.x), v = k == 0u ? va.z : (k == 1u ? va.x : va.y);
		const uint64_t five = k | ((cb & 1u) << 2) | ((u < 0.0f ? 1u : 0u) << 3) | ((v < 0.0f ? 1u : 0u) << 4);
		c.bits |= five << (5 * j);
	}
	return c;
}
// sgd_hv_proj_min / max of the scaled cube along the world direction w
SGP_DEV static float sgd_box_proj_min(const sgd_hview* B, v3 w)
{
	const v3 l = m33_tmul(B->R, w);
	return v3_dot(w, B->pos) + -((fabsf(l.x * B->scale.x) + fabsf(l.y * B->scale.y)) + fabsf(l.z * B->scale.z));
}
// ... and of the triangle (its three corners relative to the centroid, mesh frame: the thin hull's vertices, held in registers)
struct sgd_tri_corners { v3 v0, v1, v2; };
SGP_DEV static float sgd_tri_proj_min(const sgd_tri_view* T, const sgd_tri_corners& c, v3 w)
{
	const v3 l = m33_tmul(T->R, w);
	float best = 3.4e38f;
	{ const float d = v3_dot(l, c.v0); if (d < best) best = d; }
	{ const float d = v3_dot(l, c.v1); if (d < best) best = d; }
	{ const float d = v3_dot(l, c.v2); if (d < best) best = d; }
	return v3_dot(w, T->pos) + best;
}
SGP_DEV static float sgd_tri_proj_max(const sgd_tri_view* T, const sgd_tri_corners& c, v3 w)
{
	const v3 l = m33_tmul(T->R, w);
	float best = -3.4e38f;
	{ const float d = v3_dot(l, c.v0); if (d > best) best = d; }
	{ const float d = v3_dot(l, c.v1); if (d > best) best = d; }
	{ const float d = v3_dot(l, c.v2); if (d > best) best = d; }
	return v3_dot(w, T->pos) + best;
}
// = sgd_hull_sat_search(T, B, max_sep, r) for B = the cube template scaled (B->h->is_box_template)
SGP_DEV static int sgd_tri_box_sat(const sgd_tri_view* T, const sgd_hview* B, const sgd_box_code& code, float max_sep, sgd_hull_sat* r)
{
	r->sA = -3.4e38f; r->sB = -3.4e38f; r->sE = -3.4e38f; r->fA = 0; r->fB = 0; r->eA = -1; r->eB = -1; r->nE = V3(0, 0, 0);
	sgd_tri_corners tc; tc.v0 = T->h->verts[0]; tc.v1 = T->h->verts[1]; tc.v2 = T->h->verts[2];
	// the triangle's two faces (n, -n)
#pragma unroll
	for (int f = 0; f < 2; ++f) {
		const v3 n = m33_mul(T->R, T->h->normals[f]);
		const float s = sgd_box_proj_min(B, n) - (v3_dot(n, T->pos) + T->h->plane_d[f]);
		if (s > max_sep) return 0;
		if (s > r->sA) { r->sA = s; r->fA = f; }
	}
	// the cube's six faces, in the template's order
	for (int f = 0; f < 6; ++f) {
		const v3 nl = B->h->normals[f];
		const v3 n = m33_mul(B->R, nl);
		const float pd = fabsf(nl.x) * B->scale.x + fabsf(nl.y) * B->scale.y + fabsf(nl.z) * B->scale.z;
		const float s = sgd_tri_proj_min(T, tc, n) - (v3_dot(n, B->pos) + pd);
		if (s > max_sep) return 0;
		if (s > r->sB) { r->sB = s; r->fB = f; }
	}
	const v3 Tv = v3_sub(B->pos, T->pos);
	// the triangle's corners in the world (sgd_hv_world with scale 1) -- edge i starts at corner (0, 1, 0)[i] and ends at (1, 2, 2)[i] (sgd_tri_hull)
	const v3 w0 = v3_add(T->pos, m33_mul(T->R, tc.v0)), w1 = v3_add(T->pos, m33_mul(T->R, tc.v1));
#pragma unroll 1
	for (int i = 0; i < 3; ++i) {
		const v3 la = i == 1 ? tc.v1 : tc.v0, lb = i == 0 ? tc.v1 : tc.v2;
		const v3 da = m33_mul(T->R, v3_sub(lb, la));
		const v3 a0 = i == 1 ? w1 : w0;
		const float da2 = v3_len_sq(da);
		// per direction k of the cube: parallel?, the axis and the separation for an edge running the positive way (sense 0) and the negative way (sense 1)
		bool par[3]; v3 axp[3], axn[3]; float sp[3], sn[3];
#pragma unroll
		for (int k = 0; k < 3; ++k) {
			// the edge vector of a cube edge of direction k as sgd_hv_local(b) - sgd_hv_local(a) gives it: (+s) - (-s) along k for the positive sense, (-s) - (+s)
			// for the negative one, s - s = +0 elsewhere; each sense goes through the general expressions on its own (the results are each other's negation
			// wherever they are not zero, and the flip below makes them the same vector -- but a zero component keeps the sign its own sums give it)
			const float sk = k == 0 ? B->scale.x : (k == 1 ? B->scale.y : B->scale.z);
			const float ep = 1.0f * sk - -1.0f * sk, en = -1.0f * sk - 1.0f * sk;
			const v3 dbp = m33_mul(B->R, V3(k == 0 ? ep : 0.0f, k == 1 ? ep : 0.0f, k == 2 ? ep : 0.0f));
			const v3 dbn = m33_mul(B->R, V3(k == 0 ? en : 0.0f, k == 1 ? en : 0.0f, k == 2 ? en : 0.0f));
			v3 ap = v3_cross(da, dbp), an = v3_cross(da, dbn);
			const float l2 = v3_len_sq(ap);
			par[k] = l2 < 1.0e-6f * da2 * v3_len_sq(dbp);
			ap = v3_scale(ap, 1.0f / sqrtf(l2)); an = v3_scale(an, 1.0f / sqrtf(v3_len_sq(an)));
			const float t0 = v3_dot(ap, Tv);
			axp[k] = t0 < 0.0f ? v3_neg(ap) : ap;
			axn[k] = v3_dot(an, Tv) < 0.0f ? v3_neg(an) : an;
			sp[k] = 0.0f; sn[k] = 0.0f;
			if (!par[k]) {
				sp[k] = sgd_box_proj_min(B, axp[k]) - sgd_tri_proj_max(T, tc, axp[k]);
				// (t0 != 0: both senses end at the same vector, zero signs apart, and the separation along it is the same number; t0 = +-0: neither is turned, the
				// two axes are opposite and each has its own separation)
				sn[k] = (t0 < 0.0f || t0 > 0.0f) ? sp[k] : sgd_box_proj_min(B, axn[k]) - sgd_tri_proj_max(T, tc, axn[k]);
			}
		}
#pragma unroll 1
		for (int j = 0; j < 12; ++j) {
			const uint32_t five = (uint32_t)(code.bits >> (5 * j)) & 31u;
			const uint32_t k = five & 3u; const bool neg = (five & 4u) != 0u;
			const bool pk = k == 0u ? par[0] : (k == 1u ? par[1] : par[2]);
			if (pk) continue;
			const v3 ax = neg ? (k == 0u ? axn[0] : (k == 1u ? axn[1] : axn[2])) : (k == 0u ? axp[0] : (k == 1u ? axp[1] : axp[2]));
			const float s = neg ? (k == 0u ? sn[0] : (k == 1u ? sn[1] : sn[2])) : (k == 0u ? sp[0] : (k == 1u ? sp[1] : sp[2]));
			if (s > max_sep) return 0;
			// the cube corner the edge starts at (sgd_hv_world: the template's +-1 times the half extents)
			const float ck = neg ? 1.0f : -1.0f, cu = (five & 8u) ? -1.0f : 1.0f, cv = (five & 16u) ? -1.0f : 1.0f;
			const v3 cs = k == 0u ? V3(ck, cu, cv) : (k == 1u ? V3(cv, ck, cu) : V3(cu, cv, ck));
			const v3 lc = V3(cs.x * B->scale.x, cs.y * B->scale.y, cs.z * B->scale.z);
			const v3 b0 = v3_add(B->pos, m33_mul(B->R, lc));
			const float s_edge = v3_dot(ax, b0) - v3_dot(ax, a0);
			const int sup = !(s_edge - s > 1.0e-4f);
			if (s > r->sE && sup) { r->sE = s; r->eA = i; r->eB = j; r->nE = ax; }
		}
	}
	return 1;
}

// ---- ... and its manifold, sgd_hull_manifold(T, box), with every polygon in LDS -------------------------------------------------------------
// The general routine keeps its clip polygons and the candidate points in arrays indexed at run time: scratch memory, a round trip per access, and one lane
// of a wave-per-SIMD kernel has nothing to hide it behind -- a box against ONE triangle took ~100 us, most of it waiting.  Two things were tried in their
// place.  Polygons in REGISTERS (eight slots and a count, loops over the slots with the count as a predicate, corners appended by select chains): no scratch,
// but 100 KB of straight-line code per kernel that every wave runs once, cold -- the test then waits for its own instructions (66 us).  Polygons in LDS, one
// column per lane (corner i, component c of lane l at [(3 i + c) * 64 + l]: no bank conflicts), indexed at run time by plain loops: a few hundred
// instructions.  A triangle and a quad never make a polygon of more than seven corners (sgd_hull_clip), so eight slots do.
// The arithmetic -- every expression, the order of the corners, the order of the clip planes, "first minimum wins" -- is the general routine's: same bits.
// (SGD_LPOLY_FLOATS, sgd_lpoly, sgd_lp_get / sgd_lp_set / sgd_lp_reduce: sgp_device_collide.h -- the box - box clip of k_narrowphase uses them too)
// = sgd_hull_clip<8>: the half space (p - a) . side <= 0
SGP_DEV static int sgd_lp_clip(sgd_lpoly in, int n, v3 a, v3 side, sgd_lpoly out)
{
	int m = 0;
	for (int i = 0; i < n; ++i) {
		const v3 p = sgd_lp_get(in, i), q = sgd_lp_get(in, i + 1 < n ? i + 1 : 0);
		const float dp = v3_dot(v3_sub(p, a), side), dq = v3_dot(v3_sub(q, a), side);
		if (dp <= 0.0f) { if (m < 8) sgd_lp_set(out, m++, p); }
		if ((dp <= 0.0f) != (dq <= 0.0f)) {
			const float t = dp / (dp - dq);
			if (m < 8) sgd_lp_set(out, m++, v3_add(p, v3_scale(v3_sub(q, p), t)));
		}
	}
	return m;
}
// the triangle of a thin hull view, in registers: corners relative to the centroid (mesh frame), unit normal, plane offset of the front face
struct sgd_tri_regs { v3 pos; m33 R; v3 v0, v1, v2; v3 n; float d0; };
SGP_DEV static v3 sgd_trr_corner(const sgd_tri_regs& t, int j) { return v3_add(t.pos, m33_mul(t.R, j == 0 ? t.v0 : (j == 1 ? t.v1 : t.v2))); }      // (sgd_hv_world with scale 1)
// = sgd_hull_manifold(T, B, max_sep, r, m) for B = the cube template scaled.  Normal from the triangle to the box.  lds: 3 x SGD_LPOLY_FLOATS of the wave + lane.
SGP_DEV static int sgd_tri_box_manifold(const sgd_tri_view* T, const sgd_hview* B, float max_sep, const sgd_hull_sat* r, sgd_manifold* m, float* lds)
{
	sgd_tri_regs tr;
	tr.pos = T->pos; tr.R = T->R; tr.v0 = T->h->verts[0]; tr.v1 = T->h->verts[1]; tr.v2 = T->h->verts[2]; tr.n = T->h->normals[0]; tr.d0 = T->h->plane_d[0];
	const float sA = r->sA, sB = r->sB, sE = r->sE; const int fA = r->fA, fB = r->fB, eA = r->eA, eB = r->eB;
	const float sF = fmaxf(sA, sB);
	if (eA >= 0 && sE > sF + 1.0e-3f) {
		v3 pa, pb;      // (edge i of the triangle: corners (0, 1, 0)[i] -> (1, 2, 2)[i])
		sgd_seg_seg_closest(sgd_trr_corner(tr, eA == 1 ? 1 : 0), sgd_trr_corner(tr, eA == 0 ? 1 : 2),
		                    sgd_hv_world(B, B->h->edge_a[eB]), sgd_hv_world(B, B->h->edge_b[eB]), &pa, &pb);
		m->n = r->nE; m->np = 1; m->p1[0] = pa; m->p2[0] = pb;
		return 1;
	}
	const int refA = !(sB > sA + 1.0e-4f);
	sgd_lpoly P, Q, S; P.b = lds; Q.b = lds + SGD_LPOLY_FLOATS; S.b = lds + 2 * SGD_LPOLY_FLOATS;
	int np = 0;
	v3 nref; float off;
	if (refA) {
		// reference face = face fA of the triangle (0: front, corners 0 1 2; 1: back, corners 0 2 1), incident face = the cube's most anti-parallel one
		nref = m33_mul(tr.R, fA == 0 ? tr.n : v3_neg(tr.n));
		int fY = 0; float bestd = 3.4e38f;
		for (int f = 0; f < B->h->nf; ++f) { const float dd = v3_dot(nref, sgd_hv_normal(B, f)); if (dd < bestd) { bestd = dd; fY = f; } }
		for (int k = B->h->face_start[fY]; k < B->h->face_start[fY + 1]; ++k) { if (np < 8) sgd_lp_set(P, np++, sgd_hv_world(B, B->h->face_idx[k])); }
#pragma unroll 1
		for (int k = 0; k < 3 && np > 0; ++k) {
			const int ia = fA == 0 ? k : (k == 0 ? 0 : 3 - k), ib = fA == 0 ? (k == 2 ? 0 : k + 1) : (k == 0 ? 2 : (k == 1 ? 1 : 0));
			const v3 a = sgd_trr_corner(tr, ia), b = sgd_trr_corner(tr, ib);
			np = sgd_lp_clip(P, np, a, v3_cross(v3_sub(b, a), nref), Q);
			const sgd_lpoly t = P; P = Q; Q = t;
		}
		off = v3_dot(nref, tr.pos) + (fA == 0 ? tr.d0 : -tr.d0);
	} else {
		// reference face = face fB of the cube, incident face = the side of the triangle that looks at it
		nref = sgd_hv_normal(B, fB);
		int fY = 0; float bestd = 3.4e38f;
		{ const float dd = v3_dot(nref, m33_mul(tr.R, tr.n)); if (dd < bestd) { bestd = dd; fY = 0; } }
		{ const float dd = v3_dot(nref, m33_mul(tr.R, v3_neg(tr.n))); if (dd < bestd) { bestd = dd; fY = 1; } }
		np = 3;
		sgd_lp_set(P, 0, sgd_trr_corner(tr, 0)); sgd_lp_set(P, 1, sgd_trr_corner(tr, fY == 0 ? 1 : 2)); sgd_lp_set(P, 2, sgd_trr_corner(tr, fY == 0 ? 2 : 1));
		const int x0 = B->h->face_start[fB], x1 = B->h->face_start[fB + 1];
#pragma unroll 1
		for (int k = x0; k < x1 && np > 0; ++k) {
			const v3 a = sgd_hv_world(B, B->h->face_idx[k]);
			const v3 b = sgd_hv_world(B, B->h->face_idx[k + 1 < x1 ? k + 1 : x0]);
			np = sgd_lp_clip(P, np, a, v3_cross(v3_sub(b, a), nref), Q);
			const sgd_lpoly t = P; P = Q; Q = t;
		}
		off = v3_dot(nref, B->pos) + sgd_hv_plane_d(B, fB);
	}
	// the corners of the clipped incident face within reach of the reference plane, each with its foot point on that plane (Q, S: the two free columns)
	int cnt = 0;
	for (int i = 0; i < np; ++i) {
		const v3 pi = sgd_lp_get(P, i);
		const float sep = v3_dot(nref, pi) - off;
		if (sep <= max_sep) {
			const v3 pr = v3_sub(pi, v3_scale(nref, sep));
			if (refA) { sgd_lp_set(Q, cnt, pr); sgd_lp_set(S, cnt, pi); } else { sgd_lp_set(Q, cnt, pi); sgd_lp_set(S, cnt, pr); }
			++cnt;
		}
	}
	if (cnt == 0) {
		// nothing of the incident face lies over the reference face: the support vertex of the incident hull along the axis
		v3 py; float bp = 3.4e38f;
		if (refA) {
			int bi = 0;
			for (int i = 0; i < B->h->nv; ++i) { const float pr = v3_dot(nref, sgd_hv_world(B, i)); if (pr < bp) { bp = pr; bi = i; } }
			py = sgd_hv_world(B, bi);
		} else {
			int bi = 0;
#pragma unroll 1
			for (int i = 0; i < 3; ++i) { const float pr = v3_dot(nref, sgd_trr_corner(tr, i)); if (pr < bp) { bp = pr; bi = i; } }
			py = sgd_trr_corner(tr, bi);
		}
		const float sep = bp - off;
		if (sep > max_sep) return 0;
		const v3 px = v3_sub(py, v3_scale(nref, sep));
		if (refA) { sgd_lp_set(Q, 0, px); sgd_lp_set(S, 0, py); } else { sgd_lp_set(Q, 0, py); sgd_lp_set(S, 0, px); }
		cnt = 1;
	}
	sgd_lp_reduce(refA ? nref : v3_neg(nref), Q, S, cnt, m);
	return 1;
}

// X against one triangle (world-space view T of its thin hull, world normal nt).  Normal of the result: triangle -> X.
// edges: the triangle's active-edge bits (7: no fixing, e.g. a shape query), movement: X's velocity relative to the mesh.
// KINDS: the shapes X can be (bit SGD_SHAPE_*): an instance for spheres, boxes and capsules carries nothing of the general hull search.
#define SGD_KINDS_ALL 15
#define SGD_KINDS_PRIMITIVES 7      // sphere | box | capsule
// code, lpoly: what a box needs (the cube template's edge code, sgd_box_code_of; three polygon columns of the wave in LDS + lane, sgd_tri_box_manifold)
template <int KINDS = SGD_KINDS_ALL> SGP_DEV static int sgd_collide_tri(const sgd_shape* X, const sgd_tri_view* T, v3 nt, float max_sep, sgd_manifold* m, unsigned edges, v3 movement, const sgd_box_code* code = nullptr, float* lpoly = nullptr)
{
	if ((KINDS & 5) && (X->type == SGD_SHAPE_SPHERE || X->type == SGD_SHAPE_CAPSULE)) {
		int hit;
		if ((KINDS & 1) && (!(KINDS & 4) || X->type == SGD_SHAPE_SPHERE)) hit = sgd_hull_sphere(T, X->pos, X->p0, max_sep, m);
		else {
			const v3 ax = v3_scale(m33_col(X->R, 2), X->p1);
			hit = sgd_hull_capsule(T, v3_sub(X->pos, ax), v3_add(X->pos, ax), X->p0, max_sep, m);
		}
		if (!hit) return 0;
		if (v3_dot(m->n, nt) < 0.0f) return 0;                   // reached from the back side
		if (sgd_tri_needs_face_normal(T, nt, edges, movement, m)) m->n = nt;      // (active edges: the points stay, the direction changes)
		return 1;
	}
	if (!(KINDS & 10)) return 0;
	sgd_hview hx;
	hx.pos = X->pos; hx.R = X->R; hx.h = X->hull;
	hx.scale = X->type == SGD_SHAPE_BOX ? V3(X->p0, X->p1, X->p2) : V3(1.0f, 1.0f, 1.0f);
	sgd_hull_sat r;
	if constexpr ((KINDS & 2) && !(KINDS & 8)) { if (!sgd_tri_box_sat(T, &hx, *code, max_sep, &r)) return 0; }      // a box, and the caller brought the template's edge code
	else if constexpr (!(KINDS & 2)) { if (!sgd_hull_sat_search<false>(T, &hx, max_sep, &r)) return 0; }      // a convex hull, never the cube template
	else if (X->type == SGD_SHAPE_BOX && code) { if (!sgd_tri_box_sat(T, &hx, *code, max_sep, &r)) return 0; }
	else if (!sgd_hull_sat_search(T, &hx, max_sep, &r)) return 0;
	// ONE call site for the manifold (its clip polygons are a kilobyte of scratch per inlined copy): the second turn of the loop is the active-edge
	// rule's -- the contact as the triangle's FACE makes it (reference face = the triangle's front, clipped incident face of X)
	for (int turn = 0; turn < 2; ++turn) {
		if constexpr ((KINDS & 2) && !(KINDS & 8)) { if (!sgd_tri_box_manifold(T, &hx, max_sep, &r, m, lpoly)) return 0; }      // (a box: polygons in LDS)
		else { if (!sgd_hull_manifold(T, &hx, max_sep, &r, m)) return 0; }
		if (turn == 1) break;
		if (v3_dot(m->n, nt) < 0.0f) return 0;                   // reached from the back side
		if (!sgd_tri_needs_face_normal(T, nt, edges, movement, m)) break;
		r.eA = -1; r.eB = -1; r.fA = 0; r.sA = 0.0f; r.sB = -3.4e38f; r.sE = -3.4e38f;
	}
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

