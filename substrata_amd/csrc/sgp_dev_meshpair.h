// sgp_dev_meshpair.h -- a (body, static mesh) pair by lanes: tree walk, candidate sort, triangle tests, manifold groups -- shared by the narrow phase and the character's capsule query.
// Device-inline functions only (no kernels), shared between stage files; included through sgp_dev_all.h, whose order is the dependency order.
#pragma once

// ---- static triangle meshes ----------------------------------------------------------------------------------------
#define MESH_CAND_CAP 192

// Triangles of mesh body M (header mh, pose pos / R) whose tree leaves overlap the mesh-local box [llo, lhi]: indices (caller's order)
// into cand[], ascending.  Returns the count (capped; *overflow set).
SGP_DEV int mesh_candidates(const DV& d, const MeshHeader& mh, v3 llo, v3 lhi, uint32_t* cand, bool* overflow)
{
	int n = 0; *overflow = false;
	uint32_t stack[48]; int sp = 0;
	stack[sp++] = 0;
	while (sp > 0) {
		const MeshNode nd = d.mesh_nodes[mh.node_off + stack[--sp]];
		if (nd.mxx < llo.x || nd.mnx > lhi.x || nd.mxy < llo.y || nd.mny > lhi.y || nd.mxz < llo.z || nd.mnz > lhi.z) continue;
		if (nd.count == 0) { if (sp + 2 <= 48) { stack[sp++] = nd.left; stack[sp++] = nd.right; } else *overflow = true; continue; }
		for (uint32_t k = 0; k < nd.count; ++k) {
			if (n == MESH_CAND_CAP) { *overflow = true; break; }
			cand[n++] = nd.left + k;                     // position in the tree-ordered triangle array
		}
	}
	// order by the triangle's index in the caller's order (what the sequential reference walks): insertion sort on (orig, pos)
	for (int i = 1; i < n; ++i) {
		const uint32_t pos = cand[i]; const uint32_t key = MESH_TRI_INDEX(d.mesh_tris[mh.tri_off + pos].w);
		int j = i - 1;
		while (j >= 0 && MESH_TRI_INDEX(d.mesh_tris[mh.tri_off + cand[j]].w) > key) { cand[j + 1] = cand[j]; --j; }
		cand[j + 1] = pos;
	}
	return n;
}

// X against mesh body M: every triangle whose world bounds come within max_sep of [lo, hi], in index order, manifolds grouped by normal.
// Returns the number of groups (manifolds mesh -> X).  Sequential (one thread).
SGP_DEV int collide_with_mesh(const DV& d, uint32_t mbody, const sgd_shape& X, v3 lo, v3 hi, float max_sep, sgd_manifold* out, bool* dropped)      // (X: a capsule)
{
	const float4 msh = d.pose[POSE_F4 * (size_t)mbody + 3];
	const MeshHeader mh = d.meshes[(uint32_t)msh.x];
	const v3 mpos = V3(d.pose[POSE_F4 * (size_t)mbody]); const m33 R = quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)mbody + 1]));
	const v3 e = V3(max_sep, max_sep, max_sep);
	const v3 qlo = v3_sub(lo, e), qhi = v3_add(hi, e);
	// the query box in the mesh frame (bounds of its 8 corners), a little generous
	v3 llo = V3(3.4e38f, 3.4e38f, 3.4e38f), lhi = V3(-3.4e38f, -3.4e38f, -3.4e38f);
	for (int k = 0; k < 8; ++k) {
		const v3 c = V3((k & 1) ? qhi.x : qlo.x, (k & 2) ? qhi.y : qlo.y, (k & 4) ? qhi.z : qlo.z);
		const v3 l = m33_tmul(R, v3_sub(c, mpos));
		llo = V3(fminf(llo.x, l.x), fminf(llo.y, l.y), fminf(llo.z, l.z)); lhi = V3(fmaxf(lhi.x, l.x), fmaxf(lhi.y, l.y), fmaxf(lhi.z, l.z));
	}
	const float pad = 1.0e-4f * (1.0f + fabsf(llo.x) + fabsf(llo.y) + fabsf(llo.z) + fabsf(lhi.x) + fabsf(lhi.y) + fabsf(lhi.z));
	llo = v3_sub(llo, V3(pad, pad, pad)); lhi = v3_add(lhi, V3(pad, pad, pad));
	uint32_t cand[MESH_CAND_CAP];
	const int nc = mesh_candidates(d, mh, llo, lhi, cand, dropped);
	sgd_mesh_contacts mc; mc.ng = 0;
	for (int k = 0; k < nc; ++k) {
		const uint4 tri = d.mesh_tris[mh.tri_off + cand[k]];
		const v3 a = V3(d.mesh_verts[mh.vert_off + tri.x]), b = V3(d.mesh_verts[mh.vert_off + tri.y]), c = V3(d.mesh_verts[mh.vert_off + tri.z]);
		const v3 wa = v3_add(mpos, m33_mul(R, a)), wb = v3_add(mpos, m33_mul(R, b)), wc = v3_add(mpos, m33_mul(R, c));
		const v3 tmin = V3(fminf(fminf(wa.x, wb.x), wc.x), fminf(fminf(wa.y, wb.y), wc.y), fminf(fminf(wa.z, wb.z), wc.z));
		const v3 tmax = V3(fmaxf(fmaxf(wa.x, wb.x), wc.x), fmaxf(fmaxf(wa.y, wb.y), wc.y), fmaxf(fmaxf(wa.z, wb.z), wc.z));
		if (tmax.x < qlo.x || tmin.x > qhi.x || tmax.y < qlo.y || tmin.y > qhi.y || tmax.z < qlo.z || tmin.z > qhi.z) continue;
		sgd_tri_hull_t th; v3 cen, n;
		sgd_tri_hull(a, b, c, &th, &cen, &n);
		sgd_tri_view T; T.pos = v3_add(mpos, m33_mul(R, cen)); T.R = R; T.scale = V3(1.0f, 1.0f, 1.0f); T.h = &th;
		sgd_manifold m;
		if (sgd_collide_tri<4>(&X, &T, m33_mul(R, n), max_sep, &m, 7u, V3(0.0f, 0.0f, 0.0f))) sgd_mesh_add(&mc, &m);      // (a shape query: no active-edge fixing; X is the character's capsule)
	}
	return sgd_mesh_finish(&mc, out);
}

// Pairs with a static mesh: EIGHT LANES PER PAIR, eight pairs per wave.  The sequential statement walks a pair's candidate triangles in
// index order, tests each against the body and merges the triangle's manifold into <= 3 groups by normal -- the tests are independent and
// are the cost (a thin-hull SAT with clipping for a box or hull), the merge depends on the order and is cheap.  So: lane 0 of the group
// walks the mesh's tree and drops the candidates into LDS, the eight lanes order them by triangle index (rank sort: keys are unique), then
// round after round each lane tests one of the next eight candidates and the hits are merged one lane at a time, in candidate order, into
// the group table in LDS -- the sequence of sgd_mesh_add calls of the sequential walk.  The <= 3 groups are reduced and emitted by three lanes.
// (Eight, not 64: a body on a terrain or floor mesh touches a handful of triangles, and a wave per pair would idle 56 lanes.  A chassis across 150
// triangles of a detailed mesh is another matter -- a box - triangle test is ~40 us of one lane's instructions, 24 rounds of them ~2 ms --: a pair
// with more than MESH_BIG_MIN candidates is passed on to a second launch of the same kernel with all 64 lanes on one pair.)
#define MESH_BIG_MIN 32      // pairs with more candidates than this go to the wave-per-pair launch ...
#define MESH_BIG_CAP 1024    // ... which holds this many (a body across more triangles than that loses the rest: counted in manifolds_dropped)
// (the tables of a pair in LDS; G = 8: a pair that fills more than MESH_BIG_MIN entries is passed on, so 64 entries do, and no copy of the polytope)
struct MeshNoHull {};
// (HULL: room for a copy of the body's polytope -- a wave per pair, and the eight-lane instance for convex hulls: a hull against a triangle walks the hull's
// corners twice per axis, ~130 axes per triangle)
template <int G, bool HULL = (G == 64)> struct MeshPairLds {
	static constexpr int CAP = G == 64 ? MESH_BIG_CAP : 64;
	uint32_t found[CAP]; uint32_t key[CAP]; uint32_t cand[CAP]; uint32_t n_front[2], n_found, redo;
	sgd_mesh_contacts mc;
	typename std::conditional<HULL, sgd_hull, MeshNoHull>::type hull;      // the body's polytope (cube template / convex hull) next to the lanes that walk its vertices once per axis
};
#define MESH_LDS_T(G, KINDS) MeshPairLds<G, (G) == 64>      // (round 5: a hull record holds up to 256 vertices, 19 KB: only the wave-per-pair launch, one pair per workgroup, stages it)

// every triangle whose leaf box overlaps [llo, lhi] (mesh frame): positions in the tree-ordered triangle array and the triangles' indices in the
// caller's order, as found (unsorted)
SGP_DEV int mesh_candidates_found(const DV& d, const MeshHeader& mh, v3 llo, v3 lhi, uint32_t* found, uint32_t* key, bool* overflow, int stop_after = MESH_CAND_CAP, int cap = MESH_CAND_CAP)
{      // (stop_after: the caller only wants to know that there are more than this many)
	int n = 0; *overflow = false;
	uint32_t stack[48]; int sp = 0;
	stack[sp++] = 0;
	while (sp > 0) {
		const MeshNode nd = d.mesh_nodes[mh.node_off + stack[--sp]];
		if (nd.mxx < llo.x || nd.mnx > lhi.x || nd.mxy < llo.y || nd.mny > lhi.y || nd.mxz < llo.z || nd.mnz > lhi.z) continue;
		if (nd.count == 0) { if (sp + 2 <= 48) { stack[sp++] = nd.left; stack[sp++] = nd.right; } else *overflow = true; continue; }
		for (uint32_t k = 0; k < nd.count; ++k) {
			if (n == cap) { *overflow = true; break; }
			found[n] = nd.left + k; key[n] = MESH_TRI_INDEX(d.mesh_tris[mh.tri_off + nd.left + k].w); ++n;
		}
		if (n > stop_after) return n;
	}
	return n;
}

// sgd_mesh_add by the lanes of a pair together (whole wave: every lane calls this; `mine`: this lane's manifold m is the one its pair merges in this turn -- at
// most one lane per pair).  The sequential function compares every point of the manifold with every point its group already holds, one lane at work while the
// others wait: ~4 us per manifold, and a wave-per-pair round merges dozens of them.  Here the manifold is handed to all lanes of the pair, which look for its
// group (the same reads, the same answer) and share the comparisons; lane 0 of the pair appends.  The same decisions in the same order: the same groups.
template <int G> SGP_DEV void mesh_add_coop(sgd_mesh_contacts& mc, bool mine, const sgd_manifold& m, int grp, int sub)
{
	const unsigned long long gmask = G == 64 ? ~0ull : ((1ull << (G & 63)) - 1ull);
	const unsigned long long who = (__ballot(mine) >> (G == 64 ? 0 : grp * G)) & gmask;
	const bool any = who != 0ull;
	const int src = (any ? __ffsll((long long)who) - 1 : 0) + (G == 64 ? 0 : grp * G);
	const v3 n = V3(__shfl(m.n.x, src, 64), __shfl(m.n.y, src, 64), __shfl(m.n.z, src, 64));
	const int np = any ? __shfl(m.np, src, 64) : 0;
	int gi = -1; bool open_new = false;
	if (any) {
		const int ng = mc.ng;
		for (int k = 0; k < ng; ++k) if (v3_dot(mc.g[k].n, n) >= SGD_MESH_GROUP_COS) { gi = k; break; }
		if (gi < 0 && ng < SGD_MESH_MAX_GROUPS) { gi = ng; open_new = true; }
	}
	__syncthreads();
	if (open_new && sub == 0) { mc.g[gi].n = n; mc.g[gi].np = 0; mc.ng = gi + 1; }
	__syncthreads();
#pragma unroll
	for (int i = 0; i < 4; ++i) {      // (a triangle's manifold: at most four points)
		const v3 p1 = V3(__shfl(m.p1[i].x, src, 64), __shfl(m.p1[i].y, src, 64), __shfl(m.p1[i].z, src, 64));
		const v3 p2 = V3(__shfl(m.p2[i].x, src, 64), __shfl(m.p2[i].y, src, 64), __shfl(m.p2[i].z, src, 64));
		const bool act = any && gi >= 0 && i < np;
		bool dup = false; int gnp = 0;
		if (act) {
			gnp = mc.g[gi].np;
			// the same point reached through two triangles that share it (an edge or a vertex of the mesh) counts once
			for (int j = sub; j < gnp; j += G) if (v3_len_sq(v3_sub(mc.g[gi].p_body[j], p2)) < 1.0e-8f) dup = true;
		}
		const bool pair_dup = ((__ballot(dup) >> (G == 64 ? 0 : grp * G)) & gmask) != 0ull;
		if (act && sub == 0 && gnp < SGD_HULL_CLIP_CAP && !pair_dup) { mc.g[gi].p_mesh[gnp] = p1; mc.g[gi].p_body[gnp] = p2; mc.g[gi].np = gnp + 1; }
		__syncthreads();
	}
}

// The groups of one (body X, mesh body mid) pair by the lanes of its group (whole workgroup: every lane calls this; lanes of a group pass the same
// pair): what lies between "here is the pair" and "here are its <= 3 groups in L.mc".  valid: false for a group without a pair, and false on return
// when the pair was handed to the wave-per-pair launch (pair = its index in mesh_pairs; G = 8 only).  [qlo, qhi]: X's bounds grown by max_sep.
// KINDS: what X can be (bits of SGD_SHAPE_*, sgd_collide_tri): an instance for the primitives carries nothing of the general hull search, one for hulls nothing of the box's.
template <int MESH_GROUP, int KINDS = SGD_KINDS_ALL> SGP_DEV void mesh_pair_groups(const DV& d, MESH_LDS_T(MESH_GROUP, KINDS)& L, bool& valid, sgd_shape& X, uint32_t mid, v3 qlo, v3 qhi, float max_sep, int grp, int sub, uint32_t pair, bool& dropped, v3 movement, bool active_edges = true, float* lpoly = nullptr)
{
	MeshHeader mh; v3 mpos = V3(0.0f, 0.0f, 0.0f); m33 R = quat_to_m33(Q4(make_float4(0.0f, 0.0f, 0.0f, 1.0f)));
	int nc = 0;
	if (valid) {
		if constexpr (MESH_GROUP == 64) if (X.hull) {
			// a triangle test walks the polytope's vertices twice per candidate axis: out of LDS, not out of a pointer into global memory (worth the copy
			// where dozens of tests follow: 2.50 -> 2.13 ms on the car-sized boxes, but 0.93 -> 1.09 ms on the small bodies of tools/experiments/mesh_terrain_bench.py when
			// every eight-lane pair did it; the eight-lane instance for convex hulls does -- ~130 axes per triangle, each a walk over the hull's corners: 0.80 -> 0.78 ms for 3.5k hulls)
			// (the live part of every array, not the record's 19 KB: a box's cube template is 8 corners)
			const sgd_hull* hs = X.hull; sgd_hull* hd = &L.hull;
			if (sub == 0) { hd->nv = hs->nv; hd->nf = hs->nf; hd->ne = hs->ne; hd->is_box_template = hs->is_box_template; hd->aabb_min = hs->aabb_min; hd->aabb_max = hs->aabb_max; hd->bound_radius = hs->bound_radius; hd->volume = hs->volume; hd->unit_inertia = hs->unit_inertia; }
			const int nv_ = hs->nv, nf_ = hs->nf, ne_ = hs->ne, ni_ = hs->face_start[hs->nf];
			for (int i = sub; i < nv_; i += MESH_GROUP) hd->verts[i] = hs->verts[i];
			for (int i = sub; i < nf_; i += MESH_GROUP) { hd->normals[i] = hs->normals[i]; hd->plane_d[i] = hs->plane_d[i]; }
			for (int i = sub; i <= nf_; i += MESH_GROUP) hd->face_start[i] = hs->face_start[i];
			for (int i = sub; i < ni_; i += MESH_GROUP) hd->face_idx[i] = hs->face_idx[i];
			for (int i = sub; i < ne_; i += MESH_GROUP) { hd->edge_a[i] = hs->edge_a[i]; hd->edge_b[i] = hs->edge_b[i]; hd->edge_f0[i] = hs->edge_f0[i]; hd->edge_f1[i] = hs->edge_f1[i]; }
			X.hull = (const sgd_hull*)(const void*)&L.hull;
		}
		mh = d.meshes[(uint32_t)d.pose[POSE_F4 * (size_t)mid + 3].x];
		mpos = V3(d.pose[POSE_F4 * (size_t)mid]); R = quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)mid + 1]));
	}
	// the query box in the mesh frame: bounds of the box's 8 corners, a little generous (every lane of the group: the same operands, the same box)
	v3 llo = V3(3.4e38f, 3.4e38f, 3.4e38f), lhi = V3(-3.4e38f, -3.4e38f, -3.4e38f);
	if (valid) {
		for (int k = 0; k < 8; ++k) {
			const v3 c = V3((k & 1) ? qhi.x : qlo.x, (k & 2) ? qhi.y : qlo.y, (k & 4) ? qhi.z : qlo.z);
			const v3 l = m33_tmul(R, v3_sub(c, mpos));
			llo = V3(fminf(llo.x, l.x), fminf(llo.y, l.y), fminf(llo.z, l.z)); lhi = V3(fmaxf(lhi.x, l.x), fmaxf(lhi.y, l.y), fmaxf(lhi.z, l.z));
		}
		const float pad = 1.0e-4f * (1.0f + fabsf(llo.x) + fabsf(llo.y) + fabsf(llo.z) + fabsf(lhi.x) + fabsf(lhi.y) + fabsf(lhi.z));
		llo = v3_sub(llo, V3(pad, pad, pad)); lhi = v3_add(lhi, V3(pad, pad, pad));
	}
	// The candidates: the tree level by level, the lanes of the group taking the nodes of a level MESH_GROUP at a time -- a level is one fetch deep whatever
	// it holds, where the depth-first walk of one lane is a chain of every node it visits (a small body: ~70 dependent fetches, 45 us; a car-sized box on a
	// fine mesh: hundreds); the two frontiers live in the arrays the sort uses afterwards.  The SET found is that of the depth-first walk unless a table
	// overflows -- then the answer depends on the order of the walk, and lane 0 repeats it depth-first (eight lanes per pair: it then gives up once it holds
	// more than MESH_BIG_MIN -- the pair is passed on to the wave-per-pair launch; a table of 64 that overflowed says as much).
	if (sub == 0) { L.n_front[0] = valid ? 1u : 0u; L.n_front[1] = 0u; L.n_found = 0u; L.redo = 0u; L.key[0] = 0u; L.mc.ng = 0; }
	__syncthreads();
	{
		for (int level = 0; level < 64; ++level) {
			uint32_t* cur = (level & 1) ? L.cand : L.key; uint32_t* nxt = (level & 1) ? L.key : L.cand;
			// (a table overflowed: what the frontiers hold no longer matters; eight lanes that hold more than they will keep: the pair is passed on as soon as that is known)
			const uint32_t ncur = (L.redo || (MESH_GROUP != 64 && L.n_found > (uint32_t)MESH_BIG_MIN)) ? 0u : L.n_front[level & 1];
			if (!__any(ncur != 0u)) break;
			for (uint32_t i = (uint32_t)sub; i < ncur; i += MESH_GROUP) {
				const MeshNode nd = d.mesh_nodes[mh.node_off + cur[i]];
				if (nd.mxx < llo.x || nd.mnx > lhi.x || nd.mxy < llo.y || nd.mny > lhi.y || nd.mxz < llo.z || nd.mnz > lhi.z) continue;
				if (nd.count == 0) {
					const uint32_t at = atomicAdd(&L.n_front[(level & 1) ^ 1], 2u);
					if (at + 2u <= (uint32_t)MESH_LDS_T(MESH_GROUP, KINDS)::CAP) { nxt[at] = nd.left; nxt[at + 1] = nd.right; } else L.redo = 1u;
				} else {
					const uint32_t at = atomicAdd(&L.n_found, nd.count);
					if (at + nd.count <= (uint32_t)MESH_LDS_T(MESH_GROUP, KINDS)::CAP) { for (uint32_t k = 0; k < nd.count; ++k) L.found[at + k] = nd.left + k; } else L.redo = 1u;
				}
			}
			__syncthreads();
			if (sub == 0) L.n_front[level & 1] = 0u;
			__syncthreads();
		}
	}
	if (valid && L.redo && !(MESH_GROUP != 64 && L.n_found > (uint32_t)MESH_BIG_MIN)) {
		if (sub == 0) L.n_found = (uint32_t)mesh_candidates_found(d, mh, llo, lhi, L.found, L.key, &dropped, MESH_GROUP == 64 ? MESH_BIG_CAP : MESH_BIG_MIN, MESH_LDS_T(MESH_GROUP, KINDS)::CAP);
	}
	__syncthreads();
	nc = valid ? (int)min(L.n_found, (uint32_t)MESH_LDS_T(MESH_GROUP, KINDS)::CAP) : 0;
	if (MESH_GROUP != 64 && (nc > MESH_BIG_MIN || (nc > 0 && X.hull && X.hull->nv > SGD_HULL_SMALL_VERTS))) {
		// too many triangles for eight lanes: the wave-per-pair launch takes the pair (and finds its candidates again).  So does a hull beyond 32 vertices whatever
		// the count (round 5): one triangle against it is a walk over up to 768 edges, and a lane per triangle with the hull's record in LDS ends after one
		// such walk where eight lanes take turns (24 hulls of 256 vertices on a terrain: narrow phase 2.8 ms -> see docs/KERNELS.md)
		if (sub == 0) { const uint32_t kb = atomicAdd(&d.ctr->n_mesh_big, 1u); if (kb < d.cap_mesh_pairs) d.mesh_big[kb] = pair; else atomicAdd(&d.ctr->pairs_dropped, 1u); }      // (four lists feed this one: bounded like them, the excess is counted)
		valid = false; nc = 0; dropped = false;
	}
	for (int i = sub; i < nc; i += MESH_GROUP) L.key[i] = MESH_TRI_INDEX(d.mesh_tris[mh.tri_off + L.found[i]].w);
	__syncthreads();
	// candidates in the order of the caller's triangle indices: the rank of a key is the number of smaller keys
	for (int i = sub; i < nc; i += MESH_GROUP) {
		const uint32_t ki = L.key[i];
		int rank = 0;
		for (int j = 0; j < nc; ++j) rank += L.key[j] < ki ? 1 : 0;
		L.cand[rank] = L.found[i];
	}
	__syncthreads();
	// A round lasts as long as its slowest lane, and a lane whose triangle's own bounds miss the body's is done at once while its neighbour clips polygons: the
	// candidates of a tree leaf are mostly such misses (a 0.4 m box on a terrain: 8 - 32 candidates, 2 - 6 of them near it -- four rounds of which three held one
	// real test among their 64 lanes).  So the cheap bounds test runs first, over all candidates, and the survivors are packed (order kept: the merge below
	// depends on it, the set of hits does not) -- the rounds then hold real tests only.  L.key is free after the sort and takes the packed list.
	{
		int pre = (nc + MESH_GROUP - 1) / MESH_GROUP;
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) pre = max(pre, __shfl_xor(pre, off, 64));
		int kept = 0;
		for (int rd = 0; rd < pre; ++rd) {
			const int k = rd * MESH_GROUP + sub;
			bool keep = false; uint32_t cand_k = 0u;
			if (valid && k < nc) {
				cand_k = L.cand[k];
				const uint4 tri = d.mesh_tris[mh.tri_off + cand_k];
				const v3 a = V3(d.mesh_verts[mh.vert_off + tri.x]), b = V3(d.mesh_verts[mh.vert_off + tri.y]), c = V3(d.mesh_verts[mh.vert_off + tri.z]);
				const v3 wa = v3_add(mpos, m33_mul(R, a)), wb = v3_add(mpos, m33_mul(R, b)), wc = v3_add(mpos, m33_mul(R, c));
				const v3 tmin = V3(fminf(fminf(wa.x, wb.x), wc.x), fminf(fminf(wa.y, wb.y), wc.y), fminf(fminf(wa.z, wb.z), wc.z));
				const v3 tmax = V3(fmaxf(fmaxf(wa.x, wb.x), wc.x), fmaxf(fmaxf(wa.y, wb.y), wc.y), fmaxf(fmaxf(wa.z, wb.z), wc.z));
				keep = !(tmax.x < qlo.x || tmin.x > qhi.x || tmax.y < qlo.y || tmin.y > qhi.y || tmax.z < qlo.z || tmin.z > qhi.z);
			}
			const unsigned long long all = __ballot(keep);
			const unsigned long long mine_m = MESH_GROUP == 64 ? all : ((all >> (grp * MESH_GROUP)) & ((1ull << (MESH_GROUP & 63)) - 1ull));
			if (keep) L.key[kept + __popcll(mine_m & ((1ull << sub) - 1ull))] = cand_k;
			kept += __popcll(mine_m);
		}
		__syncthreads();
		nc = valid ? kept : 0;
	}
	int rounds = (nc + MESH_GROUP - 1) / MESH_GROUP;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) rounds = max(rounds, __shfl_xor(rounds, off, 64));
	// a box: the order and corners of the cube template's edges, read once (the closed-form separating-axis search of sgd_tri_box_sat)
	sgd_box_code box_code; box_code.bits = 0ull;
	if ((KINDS & 2) && valid && X.type == SGD_SHAPE_BOX) box_code = sgd_box_code_of(&d.hulls[0]);
	for (int rd = 0; rd < rounds; ++rd) {
		const int k = rd * MESH_GROUP + sub;
		bool hit = false; sgd_manifold m; m.np = 0;
		if (valid && k < nc) {
			const uint4 tri = d.mesh_tris[mh.tri_off + L.key[k]];      // (the packed list: every entry passed the bounds test)
			const v3 a = V3(d.mesh_verts[mh.vert_off + tri.x]), b = V3(d.mesh_verts[mh.vert_off + tri.y]), c = V3(d.mesh_verts[mh.vert_off + tri.z]);
			{
				sgd_tri_hull_t th; v3 cen, nrm;
				sgd_tri_hull(a, b, c, &th, &cen, &nrm);
				sgd_tri_view T; T.pos = v3_add(mpos, m33_mul(R, cen)); T.R = R; T.scale = V3(1.0f, 1.0f, 1.0f); T.h = &th;
				hit = sgd_collide_tri<KINDS>(&X, &T, m33_mul(R, nrm), max_sep, &m, active_edges ? MESH_TRI_EDGES(tri.w) : 7u, movement, (KINDS & 2) ? &box_code : nullptr, lpoly) != 0;
			}
		}
		// the hits of this round into the pair's groups, in candidate order: in turn r every group of the wave merges its r-th hit (as many turns as the group
		// with the most hits has hits -- one turn per lane POSITION with a hit somewhere in the wave was up to eight turns for two or three hits per group)
		const unsigned long long hits = __ballot(hit);
		const unsigned long long mine_h = MESH_GROUP == 64 ? hits : ((hits >> (grp * MESH_GROUP)) & ((1ull << (MESH_GROUP & 63)) - 1ull));
		const int my_rank = __popcll(mine_h & ((1ull << sub) - 1ull));
		int n_turns = __popcll(mine_h);
#pragma unroll
		for (int off = 32; off >= 1; off >>= 1) n_turns = max(n_turns, __shfl_xor(n_turns, off, 64));
		// (eight lanes per pair: the lanes share a merge's comparisons; a wave per pair: dozens of turns per round, and a turn of the shared form -- thirty
		// lane-to-lane moves of the manifold and six barriers -- measured longer than one lane's walk through the group: 0.35 -> 0.40 ms on the car-sized boxes)
		for (int t = 0; t < n_turns; ++t) {
			if (MESH_GROUP == 8) mesh_add_coop<MESH_GROUP>(L.mc, hit && my_rank == t, m, grp, sub);
			else { if (hit && my_rank == t) sgd_mesh_add(&L.mc, &m); __syncthreads(); }
		}
	}
}

// The separating-axis search of one hull pair spread over the 64 lanes of a wave: lane l takes axes l, l + 64, ... of the
// flattened list [faces of A | faces of B | edge pairs]; every axis is evaluated by the same device function the sequential
// search uses, and the reduction takes (largest separation, lowest axis index) -- exactly the sequential "first maximum wins".
SGP_DEV int hull_sat_search_wave(const sgd_hview* A, const sgd_hview* B, float max_sep, sgd_hull_sat* r)
{
	const int lane = (int)(threadIdx.x & 63u);
	const int nfA = A->h->nf, nfB = B->h->nf, neA = A->h->ne, neB = B->h->ne;
	const int total = nfA + nfB + neA * neB;
	const v3 T = v3_sub(B->pos, A->pos);
	float sA = -3.4e38f, sB = -3.4e38f, sE = -3.4e38f; int iA = 0x7FFFFFFF, iB = 0x7FFFFFFF, iE = 0x7FFFFFFF;
	bool separated = false;
	// the face axes first: most candidate pairs that do not touch are told apart by one of them, and the edge pairs (nine tenths of the axes) are then never looked at
	for (int t = lane; t < nfA + nfB; t += 64) {
		if (t < nfA) {
			const float s = sgd_hull_axis_face(A, B, t);
			if (s > max_sep) separated = true;
			if (s > sA) { sA = s; iA = t; }
		} else {
			const int f = t - nfA;
			const float s = sgd_hull_axis_face(B, A, f);
			if (s > max_sep) separated = true;
			if (s > sB) { sB = s; iB = f; }
		}
	}
	if (__any(separated)) return 0;
	// (a pair with a hull beyond 32 vertices never comes here: hull_sat_search_block)
	for (int e = lane; e < total - nfA - nfB; e += 64) {
		v3 ax; float s; int sup;
		if (sgd_hull_axis_edge(A, B, e / neB, e % neB, T, &ax, &s, &sup)) {
			if (s > max_sep) separated = true;
			if (s > sE && sup) { sE = s; iE = e; }
		}
	}
	if (__any(separated)) return 0;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		float os = __shfl_xor(sA, off); int oi = __shfl_xor(iA, off);
		if (os > sA || (os == sA && oi < iA)) { sA = os; iA = oi; }
		os = __shfl_xor(sB, off); oi = __shfl_xor(iB, off);
		if (os > sB || (os == sB && oi < iB)) { sB = os; iB = oi; }
		os = __shfl_xor(sE, off); oi = __shfl_xor(iE, off);
		if (os > sE || (os == sE && oi < iE)) { sE = os; iE = oi; }
	}
	r->sA = sA; r->fA = iA == 0x7FFFFFFF ? 0 : iA; r->sB = sB; r->fB = iB == 0x7FFFFFFF ? 0 : iB;
	r->sE = sE; r->eA = -1; r->eB = -1; r->nE = V3(0.0f, 0.0f, 0.0f);
	if (iE != 0x7FFFFFFF) {
		r->eA = iE / neB; r->eB = iE % neB;
		float s; int sup;
		sgd_hull_axis_edge(A, B, r->eA, r->eB, T, &r->nE, &s, &sup);      // the axis of the winning pair (same arithmetic as above)
	}
	return 1;
}

// A pair with a hull beyond 32 vertices (up to 768 x 768 edge pairs; the Gauss-map test picks the ones worth an axis, sgd_hull_sat_search) by a WORKGROUP of 256 threads (round 5).  The wave search
// above walks such a pair with two dependent gathers out of the 17 KB hull records per edge pair -- 9 000 iterations a lane for two 256-vertex hulls, each waiting
// on L2 -- and evaluates a picked pair where it finds it, 63 lanes idle.  Here the world normals of both hulls' faces and A's edges (as the two faces each lies
// between) are staged in LDS once; a thread keeps one edge of B in registers and walks A's edges against it out of LDS (the Gauss-map test: 4 dot products);
// the pairs it picks go to a list in LDS and are evaluated afterwards, a thread each.  Same axes, same arithmetic per axis, (largest separation, lowest pair
// index) as the order of the reduction: the result of sgd_hull_sat_search bit for bit.
#define HULL_BIG_TPB 256
#define HULL_BIG_CAND_CAP 2048
struct HullBigLds {
	v3 nA[SGD_HULL_MAX_FACES], nB[SGD_HULL_MAX_FACES];      // world normals of A's faces, MINUS those of B's
	uint32_t eA[SGD_HULL_MAX_EDGES], eB[SGD_HULL_MAX_EDGES]; // edge i: face f0 | face f1 << 16 (0xFFFF: an open edge, sgp_hull_build.h)
	uint32_t cand[HULL_BIG_CAND_CAP]; uint32_t n_cand;
	float red_s[3][HULL_BIG_TPB / 64]; int red_i[3][HULL_BIG_TPB / 64]; int separated;
};
SGP_DEV bool hull_pair_is_big(const sgd_shape& sa, const sgd_shape& sb)
{
	const bool pa = sa.type == SGP_SHAPE_BOX || sa.type == SGP_SHAPE_HULL, pb = sb.type == SGP_SHAPE_BOX || sb.type == SGP_SHAPE_HULL;
	if (!pa || !pb) return false;
	return (sa.type == SGP_SHAPE_HULL && sa.hull->nv > SGD_HULL_SMALL_VERTS) || (sb.type == SGP_SHAPE_HULL && sb.hull->nv > SGD_HULL_SMALL_VERTS);
}
SGP_DEV int hull_sat_search_block(const sgd_hview* A, const sgd_hview* B, float max_sep, sgd_hull_sat* r, HullBigLds& L)
{
	const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int nfA = A->h->nf, nfB = B->h->nf, neA = A->h->ne, neB = B->h->ne;
	const v3 T = v3_sub(B->pos, A->pos);
	__syncthreads();      // (the previous pair's last readers)
	for (int f = tid; f < nfA; f += HULL_BIG_TPB) L.nA[f] = sgd_hv_normal(A, f);
	for (int f = tid; f < nfB; f += HULL_BIG_TPB) L.nB[f] = v3_neg(sgd_hv_normal(B, f));
	for (int i = tid; i < neA; i += HULL_BIG_TPB) L.eA[i] = (uint32_t)A->h->edge_f0[i] | ((uint32_t)A->h->edge_f1[i] << 16);
	for (int j = tid; j < neB; j += HULL_BIG_TPB) L.eB[j] = (uint32_t)B->h->edge_f0[j] | ((uint32_t)B->h->edge_f1[j] << 16);
	if (tid == 0) { L.n_cand = 0u; L.separated = 0; }
	float sA = -3.4e38f, sB = -3.4e38f, sE = -3.4e38f; int iA = 0x7FFFFFFF, iB = 0x7FFFFFFF, iE = 0x7FFFFFFF;
	bool separated = false;
	for (int t = tid; t < nfA + nfB; t += HULL_BIG_TPB) {
		if (t < nfA) {
			const float s = sgd_hull_axis_face(A, B, t);
			if (s > max_sep) separated = true;
			if (s > sA) { sA = s; iA = t; }
		} else {
			const int f = t - nfA;
			const float s = sgd_hull_axis_face(B, A, f);
			if (s > max_sep) separated = true;
			if (s > sB) { sB = s; iB = f; }
		}
	}
	if (__syncthreads_or(separated ? 1 : 0)) return 0;
	// the edge pairs: B's edge j stays with its thread, A's edges stream by out of LDS
	for (int j = tid; j < neB; j += HULL_BIG_TPB) {
		const uint32_t pkb = L.eB[j];
		if ((pkb & 0xFFFFu) == 0xFFFFu) continue;      // (an open edge: below)
		const v3 c = L.nB[pkb & 0xFFFFu], dd = L.nB[pkb >> 16];
		const v3 dxc = v3_cross(dd, c);
		for (int i = 0; i < neA; ++i) {
			const uint32_t pk = L.eA[i];
			if ((pk & 0xFFFFu) == 0xFFFFu) continue;
			const v3 a = L.nA[pk & 0xFFFFu], bb = L.nA[pk >> 16];
			const v3 bxa = v3_cross(bb, a);
			const float cba = v3_dot(c, bxa), dba = v3_dot(dd, bxa), adc = v3_dot(a, dxc), bdc = v3_dot(bb, dxc);
			if (!(cba * dba < 0.0f && adc * bdc < 0.0f && cba * bdc > 0.0f)) continue;
			const uint32_t k = atomicAdd(&L.n_cand, 1u);
			if (k < HULL_BIG_CAND_CAP) L.cand[k] = (uint32_t)(i * neB + j);
			else {      // (the list is full: this one where it stands)
				v3 ax; float s;
				if (sgd_hull_axis_edge_picked(A, B, i, j, a, bb, &ax, &s)) {
					if (s > max_sep) separated = true;
					if (s > sE || (s == sE && i * neB + j < iE)) { sE = s; iE = i * neB + j; }
				}
			}
		}
	}
	// the pairs of an edge without its two faces (the builder could not tell which faces it lies between: the Gauss-map test does not apply): in full, the
	// other hull's edges dealt to the threads
	for (int j = 0; j < neB; ++j) {
		if ((L.eB[j] & 0xFFFFu) != 0xFFFFu) continue;
		for (int i = tid; i < neA; i += HULL_BIG_TPB) {
			v3 ax; float s; int sup;
			if (sgd_hull_axis_edge(A, B, i, j, T, &ax, &s, &sup)) {
				if (s > max_sep) separated = true;
				if (sup && (s > sE || (s == sE && i * neB + j < iE))) { sE = s; iE = i * neB + j; }
			}
		}
	}
	for (int i = 0; i < neA; ++i) {
		if ((L.eA[i] & 0xFFFFu) != 0xFFFFu) continue;
		for (int j = tid; j < neB; j += HULL_BIG_TPB) {
			if ((L.eB[j] & 0xFFFFu) == 0xFFFFu) continue;      // (done above)
			v3 ax; float s; int sup;
			if (sgd_hull_axis_edge(A, B, i, j, T, &ax, &s, &sup)) {
				if (s > max_sep) separated = true;
				if (sup && (s > sE || (s == sE && i * neB + j < iE))) { sE = s; iE = i * neB + j; }
			}
		}
	}
	__syncthreads();
	const int nc = (int)min(L.n_cand, (uint32_t)HULL_BIG_CAND_CAP);
	for (int k = tid; k < nc; k += HULL_BIG_TPB) {
		const int key = (int)L.cand[k], i = key / neB, j = key % neB;
		const uint32_t pk = L.eA[i];
		v3 ax; float s;
		if (sgd_hull_axis_edge_picked(A, B, i, j, L.nA[pk & 0xFFFFu], L.nA[pk >> 16], &ax, &s)) {
			if (s > max_sep) separated = true;
			if (s > sE || (s == sE && key < iE)) { sE = s; iE = key; }      // (the list is in no order: the lowest index among equals, as the walk in order keeps it)
		}
	}
	if (__syncthreads_or(separated ? 1 : 0)) return 0;
#pragma unroll
	for (int off = 32; off >= 1; off >>= 1) {
		float os = __shfl_xor(sA, off); int oi = __shfl_xor(iA, off);
		if (os > sA || (os == sA && oi < iA)) { sA = os; iA = oi; }
		os = __shfl_xor(sB, off); oi = __shfl_xor(iB, off);
		if (os > sB || (os == sB && oi < iB)) { sB = os; iB = oi; }
		os = __shfl_xor(sE, off); oi = __shfl_xor(iE, off);
		if (os > sE || (os == sE && oi < iE)) { sE = os; iE = oi; }
	}
	if (lane == 0) { L.red_s[0][wave] = sA; L.red_i[0][wave] = iA; L.red_s[1][wave] = sB; L.red_i[1][wave] = iB; L.red_s[2][wave] = sE; L.red_i[2][wave] = iE; }
	__syncthreads();
	for (int w = 0; w < HULL_BIG_TPB / 64; ++w) {
		float os = L.red_s[0][w]; int oi = L.red_i[0][w];
		if (os > sA || (os == sA && oi < iA)) { sA = os; iA = oi; }
		os = L.red_s[1][w]; oi = L.red_i[1][w];
		if (os > sB || (os == sB && oi < iB)) { sB = os; iB = oi; }
		os = L.red_s[2][w]; oi = L.red_i[2][w];
		if (os > sE || (os == sE && oi < iE)) { sE = os; iE = oi; }
	}
	r->sA = sA; r->fA = iA == 0x7FFFFFFF ? 0 : iA; r->sB = sB; r->fB = iB == 0x7FFFFFFF ? 0 : iB;
	r->sE = sE; r->eA = -1; r->eB = -1; r->nE = V3(0.0f, 0.0f, 0.0f);
	if (iE != 0x7FFFFFFF) {
		r->eA = iE / neB; r->eB = iE % neB;
		float s; int sup;
		const uint32_t pk = L.eA[r->eA];
		if ((pk & 0xFFFFu) != 0xFFFFu && (L.eB[r->eB] & 0xFFFFu) != 0xFFFFu) sgd_hull_axis_edge_picked(A, B, r->eA, r->eB, L.nA[pk & 0xFFFFu], L.nA[pk >> 16], &r->nE, &s);
		else sgd_hull_axis_edge(A, B, r->eA, r->eB, T, &r->nE, &s, &sup);
	}
	return 1;
}

// pairs with a convex hull (hull - hull / box / sphere / capsule), in two launches:
//   k_narrowphase_hull       one WAVE per pair: polytope pairs search their separating axes in parallel (faces of both hulls and all edge pairs
//                            across the 64 lanes); a pair that survives, and every hull - sphere / capsule pair, becomes a work item;
//   k_narrowphase_hull_manifold  one THREAD per work item: the manifold (reference / incident face, clipping, reduction to <= 4 points), the
//                            sphere / capsule cases.  These are sequential by nature; with a thread per item 64 of them share a wave instead of
//                            each idling 63 lanes of its own.
struct HullWork { uint2 ab; sgd_hull_sat r; uint32_t round_other; };
