// sgp_k_queries.hip -- A7 -- rays, the character's capsule queries, sphere casts.
// One of the stage files of the step kernels (stage map: sgp_kernels.h).  Kernels first, their launch wrappers at the end.
#include "sgp_dev_all.h"

// ---------------------------------------------------------------------------------------------------------------
// ray queries (traceRay, PhysicsWorld.cpp:1668-1725), one thread per ray, brute force over bodies with an AABB slab test

struct RaySub { uint32_t tri, mat; float u, v; };      // which triangle of a mesh a ray hit, its user data, barycentrics

SGP_DEV float ray_body(const DV& d, uint32_t type, float4 sh, v3 pos, quat q, v3 o, v3 dir, float max_t, v3* n_out, RaySub* sub)
{
	const m33 R = quat_to_m33(q);
	const v3 ol = m33_tmul(R, v3_sub(o, pos)), dl = m33_tmul(R, dir);
	sub->tri = SGP_INVALID_ID; sub->mat = 0; sub->u = 0.0f; sub->v = 0.0f;
	if (type == SGP_SHAPE_MESH) {
		// closest front-facing triangle; on equal distance the lower triangle index (caller's order) wins
		const MeshHeader mh = d.meshes[(uint32_t)sh.x];
		float best = max_t; uint32_t best_idx = 0xFFFFFFFFu; v3 bn = V3(0.0f, 0.0f, 0.0f);
		const v3 inv = V3(fabsf(dl.x) > 1.0e-12f ? 1.0f / dl.x : 3.0e38f, fabsf(dl.y) > 1.0e-12f ? 1.0f / dl.y : 3.0e38f, fabsf(dl.z) > 1.0e-12f ? 1.0f / dl.z : 3.0e38f);
		uint32_t stack[48]; int sp = 0;
		stack[sp++] = 0;
		while (sp > 0) {
			const MeshNode nd = d.mesh_nodes[mh.node_off + stack[--sp]];
			// slab test against the node box grown a little (never rejects a triangle the exact test would accept)
			const float g = 1.0e-4f * (1.0f + fabsf(nd.mxx) + fabsf(nd.mxy) + fabsf(nd.mxz) + fabsf(nd.mnx) + fabsf(nd.mny) + fabsf(nd.mnz));
			float t0 = 0.0f, t1 = best; bool miss = false;
			const float lo3[3] = { nd.mnx - g, nd.mny - g, nd.mnz - g }, hi3[3] = { nd.mxx + g, nd.mxy + g, nd.mxz + g };
			const float o3[3] = { ol.x, ol.y, ol.z }, d3[3] = { dl.x, dl.y, dl.z }, i3[3] = { inv.x, inv.y, inv.z };
			for (int a = 0; a < 3 && !miss; ++a) {
				if (fabsf(d3[a]) <= 1.0e-12f) { if (o3[a] < lo3[a] || o3[a] > hi3[a]) miss = true; }
				else { float ta = (lo3[a] - o3[a]) * i3[a], tb = (hi3[a] - o3[a]) * i3[a]; if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; } t0 = fmaxf(t0, ta - g); t1 = fminf(t1, tb + g); if (t0 > t1) miss = true; }
			}
			if (miss) continue;
			if (nd.count == 0) { if (sp + 2 <= 48) { stack[sp++] = nd.left; stack[sp++] = nd.right; } continue; }
			for (uint32_t k = 0; k < nd.count; ++k) {
				const uint4 tri = d.mesh_tris[mh.tri_off + nd.left + k];
				const v3 pa = V3(d.mesh_verts[mh.vert_off + tri.x]), pb = V3(d.mesh_verts[mh.vert_off + tri.y]), pc = V3(d.mesh_verts[mh.vert_off + tri.z]);
				float uv[2];
				const float tt = sgd_ray_tri_uv(ol, dl, pa, pb, pc, best, uv);
				if (tt >= 0.0f && (tt < best || best_idx == 0xFFFFFFFFu || (tt == best && MESH_TRI_INDEX(tri.w) < best_idx))) {
					best = tt; best_idx = MESH_TRI_INDEX(tri.w);
					const v3 nn = v3_cross(v3_sub(pb, pa), v3_sub(pc, pa)); bn = v3_scale(nn, 1.0f / v3_len(nn));
					sub->tri = MESH_TRI_INDEX(tri.w); sub->mat = d.mesh_tri_mat[mh.tri_off + nd.left + k]; sub->u = uv[0]; sub->v = uv[1];
				}
			}
		}
		if (best_idx == 0xFFFFFFFFu) return -1.0f;
		*n_out = m33_mul(R, bn);
		return best;
	}
	if (type == SGP_SHAPE_HULL) {
		v3 nl;
		const float t = sgd_ray_hull(body_hull(d, sh), ol, dl, max_t, 0.0f, &nl);
		if (t < 0.0f) return -1.0f;
		*n_out = m33_mul(R, nl);
		return t;
	}
	if (type == SGP_SHAPE_SPHERE) {
		const float r = sh.x;
		const float B = v3_dot(ol, dl), C = v3_len_sq(ol) - r * r;
		if (C <= 0.0f) { *n_out = v3_neg(dir); return 0.0f; }
		const float disc = B * B - C;
		if (disc < 0.0f) return -1.0f;
		const float t = -B - sqrtf(disc);
		if (t < 0.0f || t > max_t) return -1.0f;
		*n_out = m33_mul(R, v3_scale(v3_add(ol, v3_scale(dl, t)), 1.0f / r));
		return t;
	}
	if (type == SGP_SHAPE_BOX) {
		const v3 h = V3(sh.x, sh.y, sh.z);
		float t0 = 0.0f, t1 = max_t; int ax = -1; float sg = 0.0f;
		for (int k = 0; k < 3; ++k) {
			const float ok = v3_get(ol, k), dk = v3_get(dl, k), hk = v3_get(h, k);
			if (fabsf(dk) < 1.0e-12f) { if (ok < -hk || ok > hk) return -1.0f; continue; }
			float ta = (-hk - ok) / dk, tb = (hk - ok) / dk; float s = -1.0f;
			if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; s = 1.0f; }
			if (ta > t0) { t0 = ta; ax = k; sg = s; }
			if (tb < t1) t1 = tb;
			if (t0 > t1) return -1.0f;
		}
		if (ax < 0) { *n_out = v3_neg(dir); return 0.0f; }
		v3 nl = V3(0.0f, 0.0f, 0.0f); v3_set(nl, ax, sg);
		*n_out = m33_mul(R, nl);
		return t0;
	}
	{
		const float r = sh.x, hh = sh.y;
		{      // starting inside comes first (else an interior cap-sphere entry can win, depending on max_t; see sgd_ray_capsule_z)
			const v3 qq = sgd_closest_on_segment(V3(0.0f, 0.0f, -hh), V3(0.0f, 0.0f, hh), ol);
			if (v3_len_sq(v3_sub(ol, qq)) <= r * r) { *n_out = v3_neg(dir); return 0.0f; }
		}
		float best = -1.0f; v3 bn = V3(0.0f, 0.0f, 0.0f);
		const float a = dl.x * dl.x + dl.y * dl.y;
		const float bq = ol.x * dl.x + ol.y * dl.y, c = ol.x * ol.x + ol.y * ol.y - r * r;
		if (a > 1.0e-12f) {
			const float disc = bq * bq - a * c;
			if (disc >= 0.0f) {
				const float t = (-bq - sqrtf(disc)) / a;
				const float z = ol.z + dl.z * t;
				if (t >= 0.0f && t <= max_t && fabsf(z) <= hh) { best = t; bn = V3((ol.x + dl.x * t) / r, (ol.y + dl.y * t) / r, 0.0f); }
			}
		}
		for (int sgn = -1; sgn <= 1; sgn += 2) {
			const v3 oc = V3(ol.x, ol.y, ol.z - (float)sgn * hh);
			const float B = v3_dot(oc, dl), C = v3_len_sq(oc) - r * r;
			const float disc = B * B - C;
			if (disc < 0.0f) continue;
			const float t = -B - sqrtf(disc);
			if (t < 0.0f || t > max_t) continue;
			if (best < 0.0f || t < best) { best = t; bn = v3_scale(v3_add(oc, v3_scale(dl, t)), 1.0f / r); }
		}
		if (best < 0.0f) return -1.0f;
		*n_out = m33_mul(R, bn);
		return best;
	}
}

struct RayBest { float t; uint32_t id; v3 n; RaySub sub; };

// body i with its records already at hand (the same tests in the same order as ray_test_body below)
SGP_DEV void ray_test_loaded(const DV& d, const sgp_ray& ry, v3 o, v3 dir, uint32_t i, uint32_t f, float4 amin, float4 amax, float4 prop1, float4 pose0, float4 pose1, RayBest& best)
{
	if (i == ry.ignore_id) return;
	if (!(f & BF_ALIVE) || (f & BF_ALIAS)) return;
	const uint32_t layer = f_layer(f);
	if (ry.collidable_only && !(layer == SGP_LAYER_NON_MOVING || layer == SGP_LAYER_MOVING)) return;
	if (!ray_aabb(o, dir, amin, amax, best.t)) return;
	v3 nn; RaySub sub;
	const float t = ray_body(d, f_shape(f), prop1, V3(pose0), Q4(pose1), o, dir, best.t, &nn, &sub);
	// closest hit; ties go to the lower body id so the result does not depend on the traversal order
	if (t >= 0.0f && t <= best.t && (t < best.t || best.id == SGP_INVALID_ID || i < best.id)) { best.t = t; best.id = i; best.n = nn; best.sub = sub; }
}
SGP_DEV void ray_test_body(const DV& d, const sgp_ray& ry, v3 o, v3 dir, uint32_t i, RayBest& best)
{
	if (i == ry.ignore_id) return;
	const uint32_t f = d.flags[i];
	if (!(f & BF_ALIVE) || (f & BF_ALIAS)) return;
	const uint32_t layer = f_layer(f);
	if (ry.collidable_only && !(layer == SGP_LAYER_NON_MOVING || layer == SGP_LAYER_MOVING)) return;
	if (!ray_aabb(o, dir, d.aabb_min[i], d.aabb_max[i], best.t)) return;
	v3 nn; RaySub sub;
	const float t = ray_body(d, f_shape(f), d.pose[POSE_F4 * (size_t)i + 3], V3(d.pose[POSE_F4 * (size_t)i]), Q4(d.pose[POSE_F4 * (size_t)i + 1]), o, dir, best.t, &nn, &sub);
	// closest hit; ties go to the lower body id so the result does not depend on the traversal order
	if (t >= 0.0f && t <= best.t && (t < best.t || best.id == SGP_INVALID_ID || i < best.id)) { best.t = t; best.id = i; best.n = nn; best.sub = sub; }
}
// ... with all of the body's records fetched at once (the resident server: a lone wave waits for every dependent fetch in full -- flags, then bounds, then
// pose and shape is three round trips to memory where this is one; the batched kernel, bound by throughput, keeps the tests between the fetches)
SGP_DEV void ray_test_body_eager(const DV& d, const sgp_ray& ry, v3 o, v3 dir, uint32_t i, RayBest& best)
{
	const uint32_t f = d.flags[i];
	const float4 amin = d.aabb_min[i], amax = d.aabb_max[i], prop1 = d.pose[POSE_F4 * (size_t)i + 3], pose0 = d.pose[POSE_F4 * (size_t)i], pose1 = d.pose[POSE_F4 * (size_t)i + 1];
	ray_test_loaded(d, ry, o, dir, i, f, amin, amax, prop1, pose0, pose1, best);
}

// traceRay (PhysicsWorld.cpp:1668-1725), batched: one thread per ray.  Large bodies (ground quad ...) are tested directly;
// small bodies through a 3D-DDA walk of the broad-phase cell grid (bodies are binned by centre and reach at most one cell
// beyond it, so every visited cell also looks at its 26 neighbours), stopping once the cell entry distance passes the best hit.
// one ray against the world: large bodies, the static large bodies' grid, then a DDA walk of the cell grid
SGP_DEV sgp_hit raycast_one(const DV& d, const sgp_ray& ry)
{
	const v3 o = V3(ry.origin[0], ry.origin[1], ry.origin[2]), dir = V3(ry.dir[0], ry.dir[1], ry.dir[2]);
	RayBest best; best.t = ry.max_t; best.id = SGP_INVALID_ID; best.n = V3(0.0f, 0.0f, 0.0f);
	best.sub.tri = SGP_INVALID_ID; best.sub.mat = 0; best.sub.u = best.sub.v = 0.0f;
	for (uint32_t l = 0; l < d.sp->n_large; ++l) ray_test_body(d, ry, o, dir, d.large_ids[l], best);
	large_grid_ray(d, o, dir, &best.t, [&](uint32_t i) { ray_test_body(d, ry, o, dir, i, best); });
	const BpGrid g = *d.grid;
	if (g.n_cells > 0 && g.min_x <= g.max_x) {
		// clip the ray to the grid box inflated by one cell (bodies reach one cell beyond their centre cell)
		const float c = g.cell;
		const v3 lo = V3(g.ox - c, g.oy - c, g.oz - c);
		const v3 hi = V3(g.ox + ((float)g.nx + 1.0f) * c, g.oy + ((float)g.ny + 1.0f) * c, g.oz + ((float)g.nz + 1.0f) * c);
		float t0 = 0.0f, t1 = best.t; bool miss = false;
		const float oo[3] = { o.x, o.y, o.z }, dd[3] = { dir.x, dir.y, dir.z };
		const float bl[3] = { lo.x, lo.y, lo.z }, bh[3] = { hi.x, hi.y, hi.z };
		for (int a = 0; a < 3 && !miss; ++a) {
			if (fabsf(dd[a]) < 1.0e-12f) { if (oo[a] < bl[a] || oo[a] > bh[a]) miss = true; }
			else {
				float ta = (bl[a] - oo[a]) / dd[a], tb = (bh[a] - oo[a]) / dd[a];
				if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
				t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
				if (t0 > t1) miss = true;
			}
		}
		if (!miss) {
			// DDA over cells (cell coordinates may run one cell outside the grid on each side)
			const v3 p0 = v3_add(o, v3_scale(dir, t0));
			int cx = (int)floorf((p0.x - g.ox) * g.inv_cell), cy = (int)floorf((p0.y - g.oy) * g.inv_cell), cz = (int)floorf((p0.z - g.oz) * g.inv_cell);
			cx = min(max(cx, -1), g.nx); cy = min(max(cy, -1), g.ny); cz = min(max(cz, -1), g.nz);
			const int sx = dir.x > 0.0f ? 1 : -1, sy = dir.y > 0.0f ? 1 : -1, sz = dir.z > 0.0f ? 1 : -1;
			const float inf = 3.0e38f;
			const float tdx = fabsf(dir.x) > 1.0e-12f ? c / fabsf(dir.x) : inf, tdy = fabsf(dir.y) > 1.0e-12f ? c / fabsf(dir.y) : inf, tdz = fabsf(dir.z) > 1.0e-12f ? c / fabsf(dir.z) : inf;
			float tmx = fabsf(dir.x) > 1.0e-12f ? ((g.ox + (float)(cx + (sx > 0 ? 1 : 0)) * c) - o.x) / dir.x : inf;
			float tmy = fabsf(dir.y) > 1.0e-12f ? ((g.oy + (float)(cy + (sy > 0 ? 1 : 0)) * c) - o.y) / dir.y : inf;
			float tmz = fabsf(dir.z) > 1.0e-12f ? ((g.oz + (float)(cz + (sz > 0 ? 1 : 0)) * c) - o.z) / dir.z : inf;
			float t_enter = t0;
			for (int iter = 0; iter < 100000; ++iter) {
				if (t_enter - 2.0f * c > best.t) break;           // nothing nearer can come from cells this far along the ray
				for (int dz = -1; dz <= 1; ++dz) for (int dy = -1; dy <= 1; ++dy) {
					const int y = cy + dy, z = cz + dz;
					if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
					const int xa = max(cx - 1, 0), xb = min(cx + 1, g.nx - 1);
					if (xa > xb) continue;
					grid_row_runs(d, g, xa, xb, y, z, [&](uint32_t q0, uint32_t q1) { for (uint32_t q = q0; q < q1; ++q) ray_test_body(d, ry, o, dir, __float_as_uint(d.sorted_max[q].w), best); });
				}
				// next cell
				if (tmx <= tmy && tmx <= tmz) { t_enter = tmx; tmx += tdx; cx += sx; if (cx < -1 || cx > g.nx) break; }
				else if (tmy <= tmz) { t_enter = tmy; tmy += tdy; cy += sy; if (cy < -1 || cy > g.ny) break; }
				else { t_enter = tmz; tmz += tdz; cz += sz; if (cz < -1 || cz > g.nz) break; }
				if (t_enter > t1) break;
			}
		}
	}
	sgp_hit h;
	h.id = best.id; h.t = best.id == SGP_INVALID_ID ? 0.0f : best.t;
	h.normal[0] = best.n.x; h.normal[1] = best.n.y; h.normal[2] = best.n.z;
	h.triangle = best.sub.tri; h.material = best.sub.mat; h.bary[0] = best.sub.u; h.bary[1] = best.sub.v; h.sub_shape = 0;
	h.userdata = 0;
	return h;
}
__global__ void __launch_bounds__(64) k_raycast(DV d, const sgp_ray* rays, uint32_t n, sgp_hit* hits)
{
	const uint32_t k = blockIdx.x * 64 + threadIdx.x;
	if (k >= n) return;
	hits[k] = raycast_one(d, rays[k]);
}

// The same ray by all 64 lanes of a wave together (the resident server's form of raycast_one): the candidates are dealt to the lanes -- the large bodies
// by index, the bodies of the static large bodies' grid in the order it yields them, the nine neighbour rows of a visited cell one row per lane -- and
// every lane keeps its own closest hit.  The answer is the lexicographic minimum of (t, body id) over the lanes: ray_test_body breaks ties by body id
// precisely so that the result does not depend on the order of the tests, so this is raycast_one's answer bit for bit.  `tmin` (the wave's closest t so
// far, refreshed after every cell) bounds the walk for all lanes alike, which keeps the control flow -- and the dealing -- uniform.
SGP_DEV float wave_min_f(float x) { for (int off = 32; off >= 1; off >>= 1) x = fminf(x, __shfl_xor(x, off, 64)); return x; }
SGP_DEV void ray_share_bound(RayBest& best, float& tmin)
{
	tmin = wave_min_f(best.t);
	if (best.t > tmin) { best.t = tmin; best.id = SGP_INVALID_ID; }      // (somebody is closer: this lane's hit cannot win; it keeps pruning with the wave's bound)
}
// What the resident server keeps of the world between rays (LDS): the server is told to leave before anything touches the world (RayMailbox::stop_gen), so the
// grid headers and the large bodies' records it read when it started are those of every ray it answers.
#define RAY_CACHE_LARGE 16
struct RayServerCache {
	BpGrid g; LargeGrid lg; uint32_t n_large;
	uint32_t id[RAY_CACHE_LARGE], f[RAY_CACHE_LARGE]; float4 amin[RAY_CACHE_LARGE], amax[RAY_CACHE_LARGE], prop1[RAY_CACHE_LARGE], pose0[RAY_CACHE_LARGE], pose1[RAY_CACHE_LARGE];
};
SGP_DEV void ray_cache_fill(const DV& d, RayServerCache& C)
{
	const int lane = (int)(threadIdx.x & 63u);
	if (lane == 0) { C.g = *d.grid; C.lg = *d.lgrid; C.n_large = d.sp->n_large; }
	__syncthreads();
	if (lane < RAY_CACHE_LARGE && (uint32_t)lane < C.n_large) {
		const uint32_t i = d.large_ids[lane];
		C.id[lane] = i; C.f[lane] = d.flags[i]; C.amin[lane] = d.aabb_min[i]; C.amax[lane] = d.aabb_max[i];
		C.prop1[lane] = d.pose[POSE_F4 * (size_t)i + 3]; C.pose0[lane] = d.pose[POSE_F4 * (size_t)i]; C.pose1[lane] = d.pose[POSE_F4 * (size_t)i + 1];
	}
	__syncthreads();
}
#define RAY_WAVE_AHEAD 7      // cells of the walk looked at together: 7 x 9 neighbour rows = 63 lanes
SGP_DEV sgp_hit raycast_wave(const DV& d, const sgp_ray& ry, const RayServerCache& C)
{
	const int lane = (int)(threadIdx.x & 63u);
	const v3 o = V3(ry.origin[0], ry.origin[1], ry.origin[2]), dir = V3(ry.dir[0], ry.dir[1], ry.dir[2]);
	RayBest best; best.t = ry.max_t; best.id = SGP_INVALID_ID; best.n = V3(0.0f, 0.0f, 0.0f);
	best.sub.tri = SGP_INVALID_ID; best.sub.mat = 0; best.sub.u = best.sub.v = 0.0f;
	float tmin = ry.max_t;
	if (C.n_large <= RAY_CACHE_LARGE) { if ((uint32_t)lane < C.n_large) ray_test_loaded(d, ry, o, dir, C.id[lane], C.f[lane], C.amin[lane], C.amax[lane], C.prop1[lane], C.pose0[lane], C.pose1[lane], best); }
	else for (uint32_t l = (uint32_t)lane; l < C.n_large; l += 64u) ray_test_body(d, ry, o, dir, d.large_ids[l], best);
	ray_share_bound(best, tmin);
	if (C.lg.n_items) {
		uint32_t dealt = 0;
		const float bound = tmin;      // (fixed for the walk: every lane walks the same cells and counts the same candidates)
		large_grid_ray(d, o, dir, &bound, [&](uint32_t i) { if ((int)(dealt++ & 63u) == lane) ray_test_body_eager(d, ry, o, dir, i, best); });
		ray_share_bound(best, tmin);
	}
	const BpGrid g = C.g;
	if (g.n_cells > 0 && g.min_x <= g.max_x) {
		const float c = g.cell;
		const v3 lo = V3(g.ox - c, g.oy - c, g.oz - c);
		const v3 hi = V3(g.ox + ((float)g.nx + 1.0f) * c, g.oy + ((float)g.ny + 1.0f) * c, g.oz + ((float)g.nz + 1.0f) * c);
		float t0 = 0.0f, t1 = tmin; bool miss = false;
		const float oo[3] = { o.x, o.y, o.z }, dd[3] = { dir.x, dir.y, dir.z };
		const float bl[3] = { lo.x, lo.y, lo.z }, bh[3] = { hi.x, hi.y, hi.z };
		for (int a = 0; a < 3 && !miss; ++a) {
			if (fabsf(dd[a]) < 1.0e-12f) { if (oo[a] < bl[a] || oo[a] > bh[a]) miss = true; }
			else {
				float ta = (bl[a] - oo[a]) / dd[a], tb = (bh[a] - oo[a]) / dd[a];
				if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
				t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
				if (t0 > t1) miss = true;
			}
		}
		if (!miss) {
			const v3 p0 = v3_add(o, v3_scale(dir, t0));
			int cx = (int)floorf((p0.x - g.ox) * g.inv_cell), cy = (int)floorf((p0.y - g.oy) * g.inv_cell), cz = (int)floorf((p0.z - g.oz) * g.inv_cell);
			cx = min(max(cx, -1), g.nx); cy = min(max(cy, -1), g.ny); cz = min(max(cz, -1), g.nz);
			const int sx = dir.x > 0.0f ? 1 : -1, sy = dir.y > 0.0f ? 1 : -1, sz = dir.z > 0.0f ? 1 : -1;
			const float inf = 3.0e38f;
			const float tdx = fabsf(dir.x) > 1.0e-12f ? c / fabsf(dir.x) : inf, tdy = fabsf(dir.y) > 1.0e-12f ? c / fabsf(dir.y) : inf, tdz = fabsf(dir.z) > 1.0e-12f ? c / fabsf(dir.z) : inf;
			float tmx = fabsf(dir.x) > 1.0e-12f ? ((g.ox + (float)(cx + (sx > 0 ? 1 : 0)) * c) - o.x) / dir.x : inf;
			float tmy = fabsf(dir.y) > 1.0e-12f ? ((g.oy + (float)(cy + (sy > 0 ? 1 : 0)) * c) - o.y) / dir.y : inf;
			float tmz = fabsf(dir.z) > 1.0e-12f ? ((g.oz + (float)(cz + (sz > 0 ? 1 : 0)) * c) - o.z) / dir.z : inf;
			float t_enter = t0;
			// The walk, RAY_WAVE_AHEAD cells at a time: the cells of the walk follow from arithmetic alone, so the wave works out the next seven, lane l takes
			// neighbour row l % 9 of cell l / 9 of them, and the fetches of seven cells (page table -> cell table -> body records) are in flight together where the
			// cell-by-cell walk paid them one after the other (a 20 m ray: 23 -> see profiles/NOTES_r05.md section 2).  A cell beyond the one where raycast_one stops
			// can only hold bodies farther than the hit that stopped it (that is what stops it), so looking at it changes nothing.
			const int my_step = lane / 9, row = lane % 9, dy = row % 3 - 1, dz = row / 3 - 1;
			bool done = false;
			for (int iter = 0; iter < 100000 && !done; ++iter) {
				if (t_enter - 2.0f * c > tmin) break;
				int mx = 0, my = 0, mz = 0; bool mine = false;
#pragma unroll
				for (int k = 0; k < RAY_WAVE_AHEAD; ++k) {
					if (!done) {
						if (k == my_step && !(t_enter - 2.0f * c > tmin)) { mx = cx; my = cy; mz = cz; mine = true; }
						if (tmx <= tmy && tmx <= tmz) { t_enter = tmx; tmx += tdx; cx += sx; if (cx < -1 || cx > g.nx) done = true; }
						else if (tmy <= tmz) { t_enter = tmy; tmy += tdy; cy += sy; if (cy < -1 || cy > g.ny) done = true; }
						else { t_enter = tmz; tmz += tdz; cz += sz; if (cz < -1 || cz > g.nz) done = true; }
						if (t_enter > t1) done = true;
					}
				}
				if (mine && lane < 9 * RAY_WAVE_AHEAD) {
					const int y = my + dy, z = mz + dz;
					const int xa = max(mx - 1, 0), xb = min(mx + 1, g.nx - 1);
					if (y >= 0 && y < g.ny && z >= 0 && z < g.nz && xa <= xb)
						grid_row_runs(d, g, xa, xb, y, z, [&](uint32_t q0, uint32_t q1) { for (uint32_t q = q0; q < q1; ++q) ray_test_body_eager(d, ry, o, dir, __float_as_uint(d.sorted_max[q].w), best); });
				}
				ray_share_bound(best, tmin);
			}
		}
	}
	// the wave's answer: lowest (t, id) among the lanes that hold a hit
	float wt = best.id != SGP_INVALID_ID ? best.t : 3.0e38f; uint32_t wid = best.id;
	for (int off = 32; off >= 1; off >>= 1) {
		const float ot = __shfl_xor(wt, off, 64); const uint32_t oid = (uint32_t)__shfl_xor((int)wid, off, 64);
		if (oid != SGP_INVALID_ID && (wid == SGP_INVALID_ID || ot < wt || (ot == wt && oid < wid))) { wt = ot; wid = oid; }
	}
	sgp_hit h;
	h.id = wid; h.t = 0.0f; h.normal[0] = h.normal[1] = h.normal[2] = 0.0f; h.triangle = SGP_INVALID_ID; h.material = 0; h.bary[0] = h.bary[1] = 0.0f; h.sub_shape = 0; h.userdata = 0;
	if (wid != SGP_INVALID_ID) {
		const unsigned long long owners = __ballot(best.id == wid && best.t == wt);
		const int src = __ffsll((long long)owners) - 1;
		h.t = wt;
		h.normal[0] = __shfl(best.n.x, src, 64); h.normal[1] = __shfl(best.n.y, src, 64); h.normal[2] = __shfl(best.n.z, src, 64);
		h.triangle = (uint32_t)__shfl((int)best.sub.tri, src, 64); h.material = (uint32_t)__shfl((int)best.sub.mat, src, 64);
		h.bary[0] = __shfl(best.sub.u, src, 64); h.bary[1] = __shfl(best.sub.v, src, 64);
	}
	return h;
}

// The resident ray server (RayMailbox, sgp_kernels.h): one wave.  Its 64 lanes trace the ray together (raycast_wave); the mailbox lines are read and written by lanes 0..15, one word each, as single 64-byte transactions over the host link.
__global__ void __launch_bounds__(64) k_ray_server(DV d, RayMailbox* mb, uint32_t first_seq, uint32_t generation, uint64_t idle_ticks, uint64_t max_ticks)
{
	const int lane = (int)threadIdx.x;
	__shared__ RayServerCache cache;
	ray_cache_fill(d, cache);
	uint32_t* req_line = (uint32_t*)mb;
	uint32_t* res_line = req_line + 16;
	uint32_t seen = first_seq;
	const uint64_t t_start = wall_clock64();
	uint64_t t_last = t_start;
	for (uint32_t poll = 0;; ++poll) {
		const uint32_t wv = lane < 16 ? __hip_atomic_load(&req_line[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
		const uint32_t req = (uint32_t)__shfl((int)wv, 0, 64), stop_gen = (uint32_t)__shfl((int)wv, 1, 64);
		// told to leave (the host writes that BEFORE anything it does to the world and before any later request; the line is read as a whole, so a request
		// seen here without the order to leave was made while this server's view of the world was current)
		if ((int32_t)(stop_gen - generation) >= 0) break;
		if (req != seen) {
			sgp_ray ry;
			uint32_t* dst = (uint32_t*)&ry;
#pragma unroll
			for (int i = 0; i < (int)(sizeof(sgp_ray) / 4); ++i) dst[i] = (uint32_t)__shfl((int)wv, 2 + i, 64);
			const sgp_hit h = raycast_wave(d, ry, cache);
			const uint32_t* hs = (const uint32_t*)&h;
			uint32_t out = 0u;
			if (lane == 0 || lane == 15) out = req; else if (lane == 1) out = generation - 1u;      // (exited_gen: not this one)
#pragma unroll
			for (int i = 0; i < (int)(sizeof(sgp_hit) / 4); ++i) if (lane == 2 + i) out = hs[i];
			if (lane < 16) __hip_atomic_store(&res_line[lane], out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			seen = req;
			t_last = wall_clock64();
			continue;
		}
		if ((poll & 15u) == 15u) { const uint64_t now = wall_clock64(); if (now - t_last > idle_ticks || now - t_start > max_ticks) break; }
	}
	// a request that arrived while this wave was deciding to leave is answered by the next server: the host sees exited_gen == this generation with its request open
	if (lane == 1) __hip_atomic_store(&res_line[1], generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // exited_gen
}

// ---------------------------------------------------------------------------------------------------------------
// Shape queries of the character controller (JPH::CharacterVirtual: CollideShape with a maximum separation, swept test).

// the points of one manifold (normal: body -> capsule) as contacts of query k with body j
SGP_DEV void capsule_emit(const DV& d, uint32_t k, uint32_t j, uint32_t f, int g, const sgd_manifold& m, sgp_query_contact* out, uint32_t cap, uint32_t* count)
{
	for (int i = 0; i < m.np; ++i) {
		const uint32_t slot = atomicAdd(count, 1u);
		if (slot >= cap) continue;
		sgp_query_contact c;
		c.query = k; c.body = j; c.sub_shape = (uint32_t)(4 * g + i);      // point index for the host's sort; the host then stores the compound child index here
		c.point[0] = m.p1[i].x; c.point[1] = m.p1[i].y; c.point[2] = m.p1[i].z;
		c.normal[0] = m.n.x; c.normal[1] = m.n.y; c.normal[2] = m.n.z;
		c.distance = v3_dot(v3_sub(m.p2[i], m.p1[i]), m.n);
		v3 pv = V3(0.0f, 0.0f, 0.0f);
		if (f_motion(f) != SGP_MOTION_STATIC) pv = v3_add(V3(d.vel[VEL_F4 * (size_t)j]), v3_cross(V3(d.vel[VEL_F4 * (size_t)j + 1]), v3_sub(m.p1[i], V3(d.pose[POSE_F4 * (size_t)j]))));
		c.point_velocity[0] = pv.x; c.point_velocity[1] = pv.y; c.point_velocity[2] = pv.z;
		c.motion_type = f_motion(f); c.is_sensor = (f & BF_SENSOR) ? 1u : 0u; c.inv_mass = d.pose[POSE_F4 * (size_t)j].w; c.userdata = 0;
		out[slot] = c;
	}
}

// One candidate body of a query, by one lane: the filters, then the collision test -- except for mesh bodies, which go on the wave's list (their
// triangles are the whole wave's work).
#define QUERY_MESH_LIST 32
SGP_DEV void capsule_query_body(const DV& d, const sgp_capsule_query& q, uint32_t k, const sgd_shape& sc, v3 lo, v3 hi, uint32_t j, sgp_query_contact* out, uint32_t cap, uint32_t* count, uint32_t* mesh_list, uint32_t* n_mesh)
{
	if (j == q.ignore_id) return;
	const uint32_t f = d.flags[j];
	if (!(f & BF_ALIVE) || (f & BF_ALIAS)) return;
	const uint32_t layer = f_layer(f);
	if (q.collidable_only && !(layer == SGP_LAYER_NON_MOVING || layer == SGP_LAYER_MOVING)) return;
	const float4 mn = d.aabb_min[j], mx = d.aabb_max[j];
	if (mx.x < lo.x || mn.x > hi.x || mx.y < lo.y || mn.y > hi.y || mx.z < lo.z || mn.z > hi.z) return;
	const sgd_shape sb = load_shape(d, j, f);
	sgd_manifold mm[SGD_MESH_MAX_GROUPS]; int ng; bool dropped = false;
	if (sb.type == SGP_SHAPE_MESH) {
		const uint32_t at = atomicAdd(n_mesh, 1u);
		if (at < QUERY_MESH_LIST) { mesh_list[at] = j; return; }
		ng = collide_with_mesh(d, j, sc, lo, hi, q.max_separation, mm, &dropped);      // (more meshes around one capsule than the list holds: this lane walks the rest)
	}
	else ng = (sb.type == SGP_SHAPE_HULL ? sgd_collide_hull(&sb, &sc, q.max_separation, &mm[0]) : sgd_collide(&sb, &sc, q.max_separation, &mm[0])) ? 1 : 0;   // normal: body -> capsule
	for (int g = 0; g < ng; ++g) capsule_emit(d, k, j, f, g, mm[g], out, cap, count);
}

// ONE WAVE PER QUERY CAPSULE (the character controller asks for one or a few per update, and waits for the answer): the candidate bodies -- the
// large ones, and those of the broad-phase cells its bounds reach -- dealt to the 64 lanes; the mesh bodies among them (a player stands on one and
// next to others all the time) are then taken one after the other by the whole wave, 64 candidate triangles per round (mesh_pair_groups).
__global__ void __launch_bounds__(64) k_collide_capsules(DV d, const sgp_capsule_query* qs, uint32_t n, sgp_query_contact* out, uint32_t cap, uint32_t* count)
{
	__shared__ MeshPairLds<64> L;
	__shared__ uint32_t mesh_list[QUERY_MESH_LIST];
	__shared__ uint32_t n_mesh;
	const uint32_t k = blockIdx.x;
	if (k >= n) return;
	const uint32_t lane = threadIdx.x;
	const sgp_capsule_query q = qs[k];
	sgd_shape sc;
	sc.pos = V3(q.pos[0], q.pos[1], q.pos[2]);
	quat qq; qq.x = q.rot[0]; qq.y = q.rot[1]; qq.z = q.rot[2]; qq.w = q.rot[3];
	sc.R = quat_to_m33(qq); sc.type = SGP_SHAPE_CAPSULE; sc.p0 = q.radius; sc.p1 = q.half_height; sc.p2 = 0.0f; sc.hull = nullptr;
	const v3 ax = v3_scale(sc.R.c2, q.half_height);
	const float e = q.radius + q.max_separation;
	const v3 ext = V3(fabsf(ax.x) + e, fabsf(ax.y) + e, fabsf(ax.z) + e);
	const v3 lo = v3_sub(sc.pos, ext), hi = v3_add(sc.pos, ext);
	if (lane == 0) n_mesh = 0;
	__syncthreads();
	for (uint32_t l = lane; l < d.sp->n_large; l += 64) capsule_query_body(d, q, k, sc, lo, hi, d.large_ids[l], out, cap, count, mesh_list, &n_mesh);
	{
		uint32_t seen = 0;      // static large bodies around the capsule, dealt to the lanes in the order the grid yields them
		large_grid_query(d, lo, hi, [&](uint32_t i) { if ((seen++ & 63u) == lane) capsule_query_body(d, q, k, sc, lo, hi, i, out, cap, count, mesh_list, &n_mesh); });
	}
	const BpGrid g = *d.grid;
	if (g.n_cells > 0 && g.min_x <= g.max_x) {
		const int x0 = max((int)floorf((lo.x - g.ox) * g.inv_cell) - 1, 0), x1 = min((int)floorf((hi.x - g.ox) * g.inv_cell) + 1, g.nx - 1);
		const int y0 = max((int)floorf((lo.y - g.oy) * g.inv_cell) - 1, 0), y1 = min((int)floorf((hi.y - g.oy) * g.inv_cell) + 1, g.ny - 1);
		const int z0 = max((int)floorf((lo.z - g.oz) * g.inv_cell) - 1, 0), z1 = min((int)floorf((hi.z - g.oz) * g.inv_cell) + 1, g.nz - 1);
		if (x0 <= x1) for (int z = z0; z <= z1; ++z) for (int y = y0; y <= y1; ++y) {
			grid_row_runs(d, g, x0, x1, y, z, [&](uint32_t c0, uint32_t c1) { for (uint32_t c = c0 + lane; c < c1; c += 64) capsule_query_body(d, q, k, sc, lo, hi, __float_as_uint(d.sorted_max[c].w), out, cap, count, mesh_list, &n_mesh); });
		}
	}
	__syncthreads();
	const uint32_t nm = min(n_mesh, (uint32_t)QUERY_MESH_LIST);
	const v3 es = V3(q.max_separation, q.max_separation, q.max_separation);
	for (uint32_t mi = 0; mi < nm; ++mi) {
		const uint32_t mid = mesh_list[mi];
		bool valid = true, dropped = false;
		sgd_shape X = sc;
		mesh_pair_groups<64, 4>(d, L, valid, X, mid, v3_sub(lo, es), v3_add(hi, es), q.max_separation, 0, (int)lane, 0u, dropped, V3(q.movement[0], q.movement[1], q.movement[2]), q.active_edges != 0u);      // (CharacterVirtual::GetContactsAtPosition: CollideOnlyWithActive + its direction of travel; 0: every edge with its own normal)
		if ((int)lane < L.mc.ng) {
			const sgd_mesh_group& grp = L.mc.g[lane];
			sgd_manifold mm;
			sgd_hull_reduce(grp.n, grp.p_mesh, grp.p_body, grp.np, &mm);
			capsule_emit(d, k, mid, d.flags[mid], (int)lane, mm, out, cap, count);
		}
		__syncthreads();
	}
}

SGP_DEV void spherecast_body(const DV& d, const sgp_ray& ry, float rs, v3 o, v3 dir, uint32_t j, RayBest& best)
{
	if (j == ry.ignore_id) return;
	const uint32_t f = d.flags[j];
	if (!(f & BF_ALIVE) || (f & (BF_SENSOR | BF_ALIAS))) return;
	const uint32_t layer = f_layer(f);
	if (ry.collidable_only && !(layer == SGP_LAYER_NON_MOVING || layer == SGP_LAYER_MOVING)) return;
	const float4 mn = d.aabb_min[j], mx = d.aabb_max[j];
	const float e = rs + 1.0e-3f;
	if (!ray_aabb(o, dir, make_float4(mn.x - e, mn.y - e, mn.z - e, 0.0f), make_float4(mx.x + e, mx.y + e, mx.z + e, 0.0f), ry.max_t)) return;      // full length: see veh_cast_test
	const float4 sh = d.pose[POSE_F4 * (size_t)j + 3];
	const float prm[3] = { sh.x, sh.y, sh.z };
	v3 n, p;
	const float t = f_shape(f) == SGP_SHAPE_MESH ? cast_sphere_mesh(d, j, o, dir, best.t, rs, &n, &p)
	              : sgd_cast_sphere_body((int)f_shape(f), prm, f_shape(f) == SGP_SHAPE_HULL ? body_hull(d, sh) : nullptr, V3(d.pose[POSE_F4 * (size_t)j]), quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)j + 1])), o, dir, best.t, rs, &n, &p);
	if (t >= 0.0f && t <= best.t && (t < best.t || best.id == SGP_INVALID_ID || j < best.id)) { best.t = t; best.id = j; best.n = n; }
}

// one thread per cast; cells under the swept sphere's bounds (casts are short: a character's step)
__global__ void __launch_bounds__(64) k_spherecast(DV d, const sgp_ray* rays, const float* radii, uint32_t n, sgp_hit* hits)
{
	const uint32_t k = blockIdx.x * 64 + threadIdx.x;
	if (k >= n) return;
	const sgp_ray ry = rays[k];
	const float rs = radii[k];
	const v3 o = V3(ry.origin[0], ry.origin[1], ry.origin[2]), dir = V3(ry.dir[0], ry.dir[1], ry.dir[2]);
	RayBest best; best.t = ry.max_t; best.id = SGP_INVALID_ID; best.n = V3(0.0f, 0.0f, 0.0f);
	for (uint32_t l = 0; l < d.sp->n_large; ++l) spherecast_body(d, ry, rs, o, dir, d.large_ids[l], best);
	{
		// the static large bodies under the swept sphere's bounds (casts are short)
		const v3 e = v3_add(o, v3_scale(dir, ry.max_t));
		const float m = rs + 2.0e-3f;
		large_grid_query(d, V3(fminf(o.x, e.x) - m, fminf(o.y, e.y) - m, fminf(o.z, e.z) - m), V3(fmaxf(o.x, e.x) + m, fmaxf(o.y, e.y) + m, fmaxf(o.z, e.z) + m),
		                 [&](uint32_t i) { spherecast_body(d, ry, rs, o, dir, i, best); });
	}
	const BpGrid g = *d.grid;
	if (g.n_cells > 0 && g.min_x <= g.max_x) {
		const v3 e = v3_add(o, v3_scale(dir, ry.max_t));
		const float m = rs + 1.0e-3f;
		const int x0 = max((int)floorf((fminf(o.x, e.x) - m - g.ox) * g.inv_cell) - 1, 0), x1 = min((int)floorf((fmaxf(o.x, e.x) + m - g.ox) * g.inv_cell) + 1, g.nx - 1);
		const int y0 = max((int)floorf((fminf(o.y, e.y) - m - g.oy) * g.inv_cell) - 1, 0), y1 = min((int)floorf((fmaxf(o.y, e.y) + m - g.oy) * g.inv_cell) + 1, g.ny - 1);
		const int z0 = max((int)floorf((fminf(o.z, e.z) - m - g.oz) * g.inv_cell) - 1, 0), z1 = min((int)floorf((fmaxf(o.z, e.z) + m - g.oz) * g.inv_cell) + 1, g.nz - 1);
		if (x0 <= x1) for (int z = z0; z <= z1; ++z) for (int y = y0; y <= y1; ++y) {
			grid_row_runs(d, g, x0, x1, y, z, [&](uint32_t c0, uint32_t c1) { for (uint32_t c = c0; c < c1; ++c) spherecast_body(d, ry, rs, o, dir, __float_as_uint(d.sorted_max[c].w), best); });
		}
	}
	sgp_hit h;
	h.id = best.id; h.t = best.id == SGP_INVALID_ID ? 0.0f : best.t;
	h.normal[0] = best.n.x; h.normal[1] = best.n.y; h.normal[2] = best.n.z;
	h.triangle = SGP_INVALID_ID; h.material = 0; h.bary[0] = h.bary[1] = 0.0f; h.sub_shape = 0;
	h.userdata = 0;
	hits[k] = h;
}
void launch_ray_server(const DV& d, RayMailbox* mb, uint32_t first_seq, uint32_t generation, uint64_t idle_ticks, uint64_t max_ticks, hipStream_t s) { hipLaunchKernelGGL(k_ray_server, dim3(1), dim3(64), 0, s, d, mb, first_seq, generation, idle_ticks, max_ticks); }
void launch_raycast(const DV& d, const sgp_ray* rays, uint32_t n, sgp_hit* hits, hipStream_t s) { if (n) hipLaunchKernelGGL(k_raycast, dim3((n + 63) / 64), dim3(64), 0, s, d, rays, n, hits); }
void launch_collide_capsules(const DV& d, const sgp_capsule_query* q, uint32_t n, sgp_query_contact* out, uint32_t cap, uint32_t* count, hipStream_t s) { if (n) hipLaunchKernelGGL(k_collide_capsules, dim3(n), dim3(64), 0, s, d, q, n, out, cap, count); }      // a wave per query
void launch_spherecast(const DV& d, const sgp_ray* rays, const float* radii, uint32_t n, sgp_hit* hits, hipStream_t s) { if (n) hipLaunchKernelGGL(k_spherecast, dim3((n + 63) / 64), dim3(64), 0, s, d, rays, radii, n, hits); }
