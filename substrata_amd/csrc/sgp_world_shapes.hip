// sgp_world_shapes.hip -- shape tables and vehicles of the C ABI: static triangle meshes (tree built here), convex hulls (sgp_hull_build.h), wheeled vehicles.
#include "sgp_world_internal.h"

// ---------------------------------------------------------------------------------------------------------------
// static triangle meshes (MeshShapeSettings::Create, PhysicsWorld.cpp:735-1166 with is_dynamic = false)


template <typename T> static int grow_pool(sgp_world* w, T*& dev, size_t& cap, size_t need, size_t used_before)
{
	if (need <= cap) return SGP_OK;
	size_t nc = std::max<size_t>(need + need / 2, 4096);
	T* nd = nullptr;
	HIP_TRY(hipMalloc((void**)&nd, sizeof(T) * nc));
	if (dev && used_before) HIP_TRY(hipMemcpyAsync(nd, dev, sizeof(T) * used_before, hipMemcpyDeviceToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	if (dev) { hipFree(dev); w->device_bytes -= sizeof(T) * cap; }
	dev = nd; cap = nc; w->device_bytes += sizeof(T) * nc;
	return SGP_OK;
}

// median-split tree over the triangles [first, first + count) of `order`; returns the node index
static uint32_t build_mesh_node(std::vector<MeshNode>& nodes, size_t node_base, std::vector<uint32_t>& order, const std::vector<float>& cen, const std::vector<float>& tmin, const std::vector<float>& tmax, uint32_t first, uint32_t count)
{
	const uint32_t me = (uint32_t)(nodes.size() - node_base);
	nodes.push_back(MeshNode{});
	float mn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, mx[3] = { -3.4e38f, -3.4e38f, -3.4e38f }, cmn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, cmx[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
	for (uint32_t k = first; k < first + count; ++k) for (int a = 0; a < 3; ++a) {
		const uint32_t t = order[k];
		mn[a] = std::min(mn[a], tmin[3 * t + a]); mx[a] = std::max(mx[a], tmax[3 * t + a]);
		cmn[a] = std::min(cmn[a], cen[3 * t + a]); cmx[a] = std::max(cmx[a], cen[3 * t + a]);
	}
	MeshNode nd{};
	nd.mnx = mn[0]; nd.mny = mn[1]; nd.mnz = mn[2]; nd.mxx = mx[0]; nd.mxy = mx[1]; nd.mxz = mx[2];
	int axis = 0; if (cmx[1] - cmn[1] > cmx[axis] - cmn[axis]) axis = 1; if (cmx[2] - cmn[2] > cmx[axis] - cmn[axis]) axis = 2;
	if (count <= 4 || !(cmx[axis] - cmn[axis] > 0.0f)) { nd.left = first; nd.right = 0; nd.count = count; nodes[node_base + me] = nd; return me; }
	const uint32_t mid = first + count / 2;
	std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](uint32_t x, uint32_t y) { return cen[3 * x + axis] < cen[3 * y + axis] || (cen[3 * x + axis] == cen[3 * y + axis] && x < y); });
	nd.count = 0;
	nd.left = build_mesh_node(nodes, node_base, order, cen, tmin, tmax, first, mid - first);
	nd.right = build_mesh_node(nodes, node_base, order, cen, tmin, tmax, mid, first + count - mid);
	nodes[node_base + me] = nd;
	return me;
}

// first-fit from the ranges destroyed shapes gave back, else the end of the pool
static uint32_t take_range(std::vector<std::pair<uint32_t, uint32_t>>& free_ranges, uint32_t len, size_t pool_end)
{
	for (size_t k = 0; k < free_ranges.size(); ++k) if (free_ranges[k].second >= len) {
		const uint32_t off = free_ranges[k].first;
		if (free_ranges[k].second == len) free_ranges.erase(free_ranges.begin() + (long)k); else { free_ranges[k].first += len; free_ranges[k].second -= len; }
		return off;
	}
	return (uint32_t)pool_end;
}
static void give_range(std::vector<std::pair<uint32_t, uint32_t>>& free_ranges, uint32_t off, uint32_t len)
{
	if (!len) return;
	free_ranges.push_back(std::make_pair(off, len));
	std::sort(free_ranges.begin(), free_ranges.end());
	for (size_t k = 0; k + 1 < free_ranges.size();) {       // merge neighbours
		if (free_ranges[k].first + free_ranges[k].second == free_ranges[k + 1].first) { free_ranges[k].second += free_ranges[k + 1].second; free_ranges.erase(free_ranges.begin() + (long)k + 1); } else ++k;
	}
}
template <typename T> static int grow_table(sgp_world* w, T*& dev, size_t& cap, size_t need)
{
	if (need <= cap) return SGP_OK;
	const size_t nc = std::max(need, 2 * cap);
	T* nd = nullptr;
	HIP_TRY(hipMalloc((void**)&nd, sizeof(T) * nc));
	HIP_TRY(hipMemsetAsync(nd, 0, sizeof(T) * nc, w->stream));
	HIP_TRY(hipMemcpyAsync(nd, dev, sizeof(T) * cap, hipMemcpyDeviceToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	hipFree(dev); w->device_bytes += sizeof(T) * (nc - cap);
	dev = nd; cap = nc;
	return SGP_OK;
}

// Which edges of which triangles are ACTIVE (MeshShape::sFindActiveEdges + ActiveEdges::IsEdgeActive with the 5 degree default the reference leaves in
// place, PhysicsWorld.cpp:1028-1060): flags[t] bit k set = edge k (v[k] - v[k + 1]) of triangle t collides with its own normal.  An edge is keyed by its two
// vertex indices: used by one triangle or by more than two -> active; by two -> inactive when concave or when their normals are within the threshold.
// Doubles: the flags must come out the same wherever this runs (tests/test_mesh_parity_gpu.py compares them with the sequential CPU statement's).
#define SGP_ACTIVE_EDGE_COS 0.99619469809f      // cos(5 degrees) as a float (MeshShapeSettings::mActiveEdgeCosThresholdAngle), widened to double for the test below
static void mesh_active_edges(const float* verts, const uint32_t* idx, uint32_t nt, std::vector<uint8_t>& flags)
{
	struct Rec { uint32_t lo, hi, tri, k; };
	std::vector<Rec> e(3 * (size_t)nt);
	flags.assign(nt, 7);
	for (uint32_t t = 0; t < nt; ++t) for (uint32_t k = 0; k < 3; ++k) { const uint32_t a = idx[3 * t + k], b = idx[3 * t + (k + 1) % 3]; e[3 * (size_t)t + k] = Rec{ std::min(a, b), std::max(a, b), t, k }; }
	std::sort(e.begin(), e.end(), [](const Rec& x, const Rec& y) { if (x.lo != y.lo) return x.lo < y.lo; if (x.hi != y.hi) return x.hi < y.hi; if (x.tri != y.tri) return x.tri < y.tri; return x.k < y.k; });
	auto normal = [&](uint32_t t, double n[3]) {
		const float* a = verts + 3 * idx[3 * t]; const float* b = verts + 3 * idx[3 * t + 1]; const float* c = verts + 3 * idx[3 * t + 2];
		const double e1[3] = { (double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2] }, e2[3] = { (double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2] };
		n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
		const double l = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
		if (!(l > 1.0e-30)) return false;
		n[0] /= l; n[1] /= l; n[2] /= l;
		return true;
	};
	const double cos_threshold = (double)SGP_ACTIVE_EDGE_COS;
	for (size_t i = 0; i < e.size(); ) {
		size_t j = i + 1;
		while (j < e.size() && e[j].lo == e[i].lo && e[j].hi == e[i].hi) ++j;
		if (j - i == 2 && e[i].lo != e[i].hi) {
			double n1[3], n2[3];
			if (normal(e[i].tri, n1) && normal(e[i + 1].tri, n2)) {
				const uint32_t va = idx[3 * e[i].tri + e[i].k], vb = idx[3 * e[i].tri + (e[i].k + 1) % 3];       // the edge in the first triangle's winding
				const double d[3] = { (double)verts[3 * vb] - verts[3 * va], (double)verts[3 * vb + 1] - verts[3 * va + 1], (double)verts[3 * vb + 2] - verts[3 * va + 2] };
				const double cosn = n1[0] * n2[0] + n1[1] * n2[1] + n1[2] * n2[2];
				const double cx = n1[1] * n2[2] - n1[2] * n2[1], cy = n1[2] * n2[0] - n1[0] * n2[2], cz = n1[0] * n2[1] - n1[1] * n2[0];
				bool active;
				if (cosn < -0.999848) active = true;                                    // back to back
				else if (cx * d[0] + cy * d[1] + cz * d[2] < 0.0) active = false;       // concave
				else active = cosn < cos_threshold;                                     // convex: active beyond the threshold angle
				if (!active) { flags[e[i].tri] &= (uint8_t)~(1u << e[i].k); flags[e[i + 1].tri] &= (uint8_t)~(1u << e[i + 1].k); }
			}
		}
		i = j;
	}
}

SGP_API int sgp_mesh_create_with_materials(sgp_world* w, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t nt, const uint32_t* tri_mats, sgp_mesh_info* info);
SGP_API int sgp_mesh_create(sgp_world* w, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t nt, sgp_mesh_info* info)
{
	return sgp_mesh_create_with_materials(w, verts, nv, idx, nt, nullptr, info);
}
SGP_API int sgp_mesh_create_with_materials(sgp_world* w, const float* verts, uint32_t nv, const uint32_t* idx, uint32_t nt, const uint32_t* tri_mats, sgp_mesh_info* info)
{
	if (w) ray_server_stop(w);      // (a resident ray server must not keep this call's stream work waiting: ADVICE r05)
	if (!w || !verts || !idx || !info || nv < 3 || nt < 1) return fail(SGP_ERR_INVALID, "sgp_mesh_create: bad arguments");
	for (uint32_t k = 0; k < 3 * nt; ++k) if (idx[k] >= nv) return fail(SGP_ERR_INVALID, "sgp_mesh_create: vertex index out of range");
	for (uint32_t k = 0; k < 3 * nv; ++k) if (!std::isfinite(verts[k])) return fail(SGP_ERR_INVALID, "sgp_mesh_create: non-finite vertex");
	hipSetDevice(w->device);
	MeshHeader mh{};
	mh.nv = nv; mh.nt = nt;
	mh.vert_off = take_range(w->free_vert_ranges, nv, w->mesh_verts.size());
	mh.tri_off = take_range(w->free_tri_ranges, nt, w->mesh_tris.size());
	if (w->mesh_verts.size() < (size_t)mh.vert_off + nv) w->mesh_verts.resize((size_t)mh.vert_off + nv);
	if (w->mesh_tris.size() < (size_t)mh.tri_off + nt) { w->mesh_tris.resize((size_t)mh.tri_off + nt); w->mesh_tri_mat.resize((size_t)mh.tri_off + nt); }
	float mn[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, mx[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
	for (uint32_t k = 0; k < nv; ++k) {
		w->mesh_verts[mh.vert_off + k] = make_float4(verts[3 * k], verts[3 * k + 1], verts[3 * k + 2], 0.0f);
		for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], verts[3 * k + a]); mx[a] = std::max(mx[a], verts[3 * k + a]); }
	}
	mh.mnx = mn[0]; mh.mny = mn[1]; mh.mnz = mn[2]; mh.mxx = mx[0]; mh.mxy = mx[1]; mh.mxz = mx[2];
	std::vector<float> cen(3 * (size_t)nt), tmin(3 * (size_t)nt), tmax(3 * (size_t)nt);
	std::vector<uint32_t> order(nt);
	for (uint32_t t = 0; t < nt; ++t) {
		order[t] = t;
		for (int a = 0; a < 3; ++a) {
			const float p0 = verts[3 * idx[3 * t] + a], p1 = verts[3 * idx[3 * t + 1] + a], p2 = verts[3 * idx[3 * t + 2] + a];
			cen[3 * t + a] = (p0 + p1 + p2) * (1.0f / 3.0f); tmin[3 * t + a] = std::min(p0, std::min(p1, p2)); tmax[3 * t + a] = std::max(p0, std::max(p1, p2));
		}
	}
	{      // the tree is built aside (its size is not known beforehand), then placed in a freed range or at the end of the node pool
		std::vector<MeshNode> nodes;
		build_mesh_node(nodes, 0, order, cen, tmin, tmax, 0, nt);
		mh.n_nodes = (uint32_t)nodes.size();
		mh.node_off = take_range(w->free_node_ranges, mh.n_nodes, w->mesh_nodes.size());
		if (w->mesh_nodes.size() < (size_t)mh.node_off + mh.n_nodes) w->mesh_nodes.resize((size_t)mh.node_off + mh.n_nodes);
		std::copy(nodes.begin(), nodes.end(), w->mesh_nodes.begin() + mh.node_off);
	}
	if (nt >= (1u << 29)) return fail(SGP_ERR_CAPACITY, "sgp_mesh_create: more than 2^29 triangles");
	std::vector<uint8_t> edge_flags;
	mesh_active_edges(verts, idx, nt, edge_flags);
	// (uint4.w of a triangle: its index in the caller's order, and in the top three bits its active-edge flags: MESH_TRI_INDEX / MESH_TRI_EDGES)
	for (uint32_t k = 0; k < nt; ++k) { const uint32_t t = order[k]; w->mesh_tris[mh.tri_off + k] = make_uint4(idx[3 * t], idx[3 * t + 1], idx[3 * t + 2], t | ((uint32_t)edge_flags[t] << 29)); w->mesh_tri_mat[mh.tri_off + k] = tri_mats ? tri_mats[t] : 0u; }
	// upload (pools may move: captured graphs carry the old pointers)
	{ int r = grow_pool(w, w->d_mesh_verts, w->cap_mesh_verts, w->mesh_verts.size(), w->cap_mesh_verts); if (r != SGP_OK) return r; }
	{ int r = grow_pool(w, w->d_mesh_tris, w->cap_mesh_tris, w->mesh_tris.size(), w->cap_mesh_tris); if (r != SGP_OK) return r; }
	{ int r = grow_pool(w, w->d_mesh_tri_mat, w->cap_mesh_tri_mat, w->mesh_tri_mat.size(), w->cap_mesh_tri_mat); if (r != SGP_OK) return r; }
	{ int r = grow_pool(w, w->d_mesh_nodes, w->cap_mesh_nodes, w->mesh_nodes.size(), w->cap_mesh_nodes); if (r != SGP_OK) return r; }
	HIP_TRY(hipMemcpyAsync(w->d_mesh_verts + mh.vert_off, w->mesh_verts.data() + mh.vert_off, sizeof(float4) * nv, hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipMemcpyAsync(w->d_mesh_tris + mh.tri_off, w->mesh_tris.data() + mh.tri_off, sizeof(uint4) * nt, hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipMemcpyAsync(w->d_mesh_tri_mat + mh.tri_off, w->mesh_tri_mat.data() + mh.tri_off, sizeof(uint32_t) * nt, hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipMemcpyAsync(w->d_mesh_nodes + mh.node_off, w->mesh_nodes.data() + mh.node_off, sizeof(MeshNode) * mh.n_nodes, hipMemcpyHostToDevice, w->stream));
	uint32_t id;
	if (!w->free_mesh_ids.empty()) { id = w->free_mesh_ids.back(); w->free_mesh_ids.pop_back(); w->meshes[id] = mh; w->mesh_refs[id] = 0; }
	else { id = (uint32_t)w->meshes.size(); w->meshes.push_back(mh); w->mesh_refs.push_back(0); }
	{ int r = grow_table(w, w->d_meshes, w->cap_mesh_table, w->meshes.size()); if (r != SGP_OK) return r; }
	w->dv.meshes = w->d_meshes;
	HIP_TRY(hipMemcpyAsync(&w->d_meshes[id], &w->meshes[id], sizeof(MeshHeader), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->dv.mesh_verts = w->d_mesh_verts; w->dv.mesh_tris = w->d_mesh_tris; w->dv.mesh_tri_mat = w->d_mesh_tri_mat; w->dv.mesh_nodes = w->d_mesh_nodes; w->dv.n_meshes = (uint32_t)w->meshes.size();
	invalidate_graphs(w);
	memset(info, 0, sizeof(*info));
	info->mesh_id = id; info->num_vertices = nv; info->num_triangles = nt; info->num_nodes = mh.n_nodes;
	memcpy(info->aabb_min, mn, sizeof(mn)); memcpy(info->aabb_max, mx, sizeof(mx));
	return SGP_OK;
}

// JPH::Ref<JPH::Shape> going out of scope: the mesh's table slot and pool ranges become reusable.  Refused while a body still uses it.
// the active-edge bits of a mesh's triangles in the caller's triangle order (tests: compared with the sequential CPU statement's)
SGP_API int sgp_mesh_edge_flags(sgp_world* w, uint32_t mesh_id, uint8_t* out, uint32_t cap)
{
	if (!w || !out || mesh_id < 1 || mesh_id >= w->meshes.size() || w->meshes[mesh_id].nt == 0) return fail(SGP_ERR_BAD_ID, "sgp_mesh_edge_flags: no such mesh");
	const MeshHeader& mh = w->meshes[mesh_id];
	for (uint32_t k = 0; k < mh.nt; ++k) { const uint32_t wv = w->mesh_tris[mh.tri_off + k].w; const uint32_t t = wv & 0x1FFFFFFFu; if (t < cap) out[t] = (uint8_t)(wv >> 29); }
	return SGP_OK;
}

SGP_API int sgp_mesh_destroy(sgp_world* w, uint32_t id)
{
	if (w) ray_server_stop(w);      // (a resident ray server must not keep this call's stream work waiting: ADVICE r05)
	if (!w || id < 1 || id >= w->meshes.size() || w->meshes[id].nt == 0) return fail(SGP_ERR_BAD_ID, "sgp_mesh_destroy: no such mesh");
	if (w->mesh_refs[id] != 0) return fail(SGP_ERR_REJECTED, "sgp_mesh_destroy: a body still uses the mesh");
	hipSetDevice(w->device);
	const MeshHeader mh = w->meshes[id];
	give_range(w->free_vert_ranges, mh.vert_off, mh.nv); give_range(w->free_tri_ranges, mh.tri_off, mh.nt); give_range(w->free_node_ranges, mh.node_off, mh.n_nodes);
	w->meshes[id] = MeshHeader{};
	HIP_TRY(hipMemcpyAsync(&w->d_meshes[id], &w->meshes[id], sizeof(MeshHeader), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->free_mesh_ids.push_back(id);
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// convex hull shapes (ConvexHullShapeSettings::Create, CarPhysics.cpp:66-78)


SGP_API int sgp_hull_create_com(sgp_world* w, const float* pts, uint32_t n, const float* com_offset, sgp_hull_info* info)
{
	if (w) ray_server_stop(w);      // (a resident ray server must not keep this call's stream work waiting: ADVICE r05)
	if (!w || !pts || !info || n < 4 || n > 100000) return fail(SGP_ERR_INVALID, "sgp_hull_create: bad arguments");
	hipSetDevice(w->device);
	sgd_hull h;
	float com[3], rot[4];
	if (sgd_hull_build(pts, (int)n, com_offset, &h, com, rot) != 0) return fail(SGP_ERR_REJECTED, "sgp_hull_create: degenerate point cloud or too many faces");
	uint32_t id;
	if (!w->free_hull_ids.empty()) { id = w->free_hull_ids.back(); w->free_hull_ids.pop_back(); w->hulls[id] = h; w->hull_refs[id] = 0; }
	else { id = (uint32_t)w->hulls.size(); w->hulls.push_back(h); w->hull_refs.push_back(0); }
	{ int r = grow_table(w, w->d_hulls, w->cap_hull_table, w->hulls.size()); if (r != SGP_OK) return r; }
	w->dv.hulls = w->d_hulls;
	HIP_TRY(hipMemcpyAsync(&w->d_hulls[id], &w->hulls[id], sizeof(sgd_hull), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->dv.n_hulls = (uint32_t)w->hulls.size();
	if (h.nv > SGD_HULL_SMALL_VERTS) ++w->n_big_hulls;
	invalidate_graphs(w);                        // DV travels by value in the captured launches
	memset(info, 0, sizeof(*info));
	info->hull_id = id; info->num_vertices = (uint32_t)h.nv; info->num_faces = (uint32_t)h.nf; info->num_edges = (uint32_t)h.ne;
	memcpy(info->com, com, sizeof(com)); memcpy(info->rot, rot, sizeof(rot));
	info->volume = h.volume;
	info->unit_inertia[0] = h.unit_inertia.x; info->unit_inertia[1] = h.unit_inertia.y; info->unit_inertia[2] = h.unit_inertia.z;
	info->aabb_min[0] = h.aabb_min.x; info->aabb_min[1] = h.aabb_min.y; info->aabb_min[2] = h.aabb_min.z;
	info->aabb_max[0] = h.aabb_max.x; info->aabb_max[1] = h.aabb_max.y; info->aabb_max[2] = h.aabb_max.z;
	return SGP_OK;
}

SGP_API int sgp_hull_create(sgp_world* w, const float* pts, uint32_t n, sgp_hull_info* info) { return sgp_hull_create_com(w, pts, n, nullptr, info); }
SGP_API int sgp_hull_destroy(sgp_world* w, uint32_t id)
{
	if (w) ray_server_stop(w);      // (a resident ray server must not keep this call's stream work waiting: ADVICE r05)
	if (!w || id < 1 || id >= w->hulls.size() || w->hulls[id].nv == 0) return fail(SGP_ERR_BAD_ID, "sgp_hull_destroy: no such hull");
	if (w->hull_refs[id] != 0) return fail(SGP_ERR_REJECTED, "sgp_hull_destroy: a body still uses the hull");
	hipSetDevice(w->device);
	if (w->hulls[id].nv > SGD_HULL_SMALL_VERTS) --w->n_big_hulls;
	memset(&w->hulls[id], 0, sizeof(sgd_hull));
	HIP_TRY(hipMemcpyAsync(&w->d_hulls[id], &w->hulls[id], sizeof(sgd_hull), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->free_hull_ids.push_back(id);
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// wheeled vehicles (VehicleConstraint + WheeledVehicleController, CarPhysics.cpp:94-231)

void invalidate_graphs(sgp_world* w)
{
	for (auto& kv : w->graphs) hipGraphExecDestroy(kv.second);
	w->graphs.clear();
	for (int k = 0; k < 2; ++k) { w->last_plan_key[k].clear(); w->plan_repeats[k] = 0; }
}

SGP_API void sgp_default_vehicle_desc(sgp_vehicle_desc* d)
{
	memset(d, 0, sizeof(*d));
	d->body = SGP_INVALID_ID;
	d->num_wheels = 4;
	for (int i = 0; i < 4; ++i) {
		sgp_wheel_desc* w = &d->wheels[i];
		const bool front = i < 2, left = (i % 2) == 0;
		w->position[0] = left ? -0.8f : 0.8f; w->position[1] = front ? 1.3f : -1.3f; w->position[2] = 0.15f;
		w->suspension_dir[2] = -1.0f; w->steering_axis[2] = 1.0f; w->wheel_up[2] = 1.0f; w->wheel_forward[1] = 1.0f;
		w->suspension_min_length = 0.2f; w->suspension_max_length = 0.5f; w->suspension_preload = 0.0f;     // Scripting.cpp:326-330
		w->spring_frequency = 2.0f; w->spring_damping = 0.5f;                                              // :335-339
		w->radius = 0.42f; w->width = 0.16f;                                                               // :320-324
		w->inertia = 0.9f; w->angular_damping = 0.2f;                                                      // JPH::WheelSettingsWV defaults
		w->max_steer_angle = front ? 0.78525f : 0.0f;                                                      // :342, CarPhysics.cpp:127,153
		w->max_brake_torque = 1500.0f; w->max_handbrake_torque = front ? 0.0f : 4000.0f;                   // :347-348, CarPhysics.cpp:129,155
		const float lf[3][2] = { { 0.0f, 0.0f }, { 0.06f, 1.2f }, { 0.2f, 1.0f } };
		const float tf[3][2] = { { 0.0f, 0.0f }, { 3.0f, 1.2f }, { 20.0f, 1.0f } };
		memcpy(w->longitudinal_friction, lf, sizeof(lf)); memcpy(w->lateral_friction, tf, sizeof(tf));
	}
	d->up[2] = 1.0f; d->forward[1] = 1.0f;
	d->cast_radius = 0.08f;                                                                              // 0.5 * front_wheel_width, CarPhysics.cpp:62
	d->max_slope_angle = 80.0f * 3.14159265358979323846f / 180.0f;
	d->engine_max_torque = 500.0f; d->engine_min_rpm = 1000.0f; d->engine_max_rpm = 6000.0f; d->engine_inertia = 0.5f; d->engine_angular_damping = 0.2f;
	const float ec[3][2] = { { 0.0f, 0.8f }, { 0.66f, 1.0f }, { 1.0f, 0.8f } };
	memcpy(d->engine_torque_curve, ec, sizeof(ec));
	d->num_gears = 5; d->num_reverse_gears = 1;
	const float gr[5] = { 2.66f, 1.78f, 1.3f, 1.0f, 0.74f };
	memcpy(d->gear_ratios, gr, sizeof(gr)); d->reverse_gear_ratios[0] = -2.9f;
	d->switch_time = 0.5f; d->clutch_release_time = 0.3f; d->switch_latency = 0.5f; d->shift_up_rpm = 4000.0f; d->shift_down_rpm = 2000.0f; d->clutch_strength = 10.0f;
	d->num_differentials = 1;                                                                            // front wheel drive, CarPhysics.cpp:191-194
	d->differentials[0].left_wheel = 0; d->differentials[0].right_wheel = 1;
	d->differentials[0].differential_ratio = 3.42f; d->differentials[0].left_right_split = 0.5f; d->differentials[0].limited_slip_ratio = 1.4f; d->differentials[0].engine_torque_ratio = 1.0f;
	d->differentials[1] = d->differentials[0]; d->differentials[1].left_wheel = 2; d->differentials[1].right_wheel = 3;
	d->differential_limited_slip_ratio = 1.4f;
	d->num_anti_roll_bars = 2;                                                                           // CarPhysics.cpp:217-221
	d->anti_roll_bars[0].left_wheel = 0; d->anti_roll_bars[0].right_wheel = 1; d->anti_roll_bars[0].stiffness = 1000.0f;
	d->anti_roll_bars[1].left_wheel = 2; d->anti_roll_bars[1].right_wheel = 3; d->anti_roll_bars[1].stiffness = 1000.0f;
	d->controller_type = SGP_VEHICLE_CONTROLLER_WHEELED;
	d->max_lean_angle = 45.0f * 3.14159265358979323846f / 180.0f; d->lean_spring_constant = 5000.0f; d->lean_spring_damping = 1000.0f;   // JPH::MotorcycleControllerSettings defaults
	d->lean_spring_integration_coefficient = 0.0f; d->lean_spring_integration_decay = 4.0f; d->lean_smoothing_factor = 0.8f; d->lean_steering_limit = 1;
}

static bool vehicle_desc_valid(const sgp_vehicle_desc* d)
{
	if (d->num_wheels < 1 || d->num_wheels > SGP_MAX_WHEELS) return false;
	if (d->num_gears < 1 || d->num_gears > SGP_MAX_GEARS || d->num_reverse_gears < 1 || d->num_reverse_gears > SGP_MAX_GEARS) return false;
	if (d->num_differentials > 2 || d->num_anti_roll_bars > 2) return false;
	for (uint32_t k = 0; k < d->num_differentials; ++k) {
		if (d->differentials[k].left_wheel >= (int)d->num_wheels || d->differentials[k].right_wheel >= (int)d->num_wheels) return false;
		if (!(d->differentials[k].limited_slip_ratio > 1.0f)) return false;
	}
	for (uint32_t k = 0; k < d->num_anti_roll_bars; ++k) {
		const sgp_anti_roll_bar_desc* r = &d->anti_roll_bars[k];
		if (r->left_wheel < 0 || r->right_wheel < 0 || r->left_wheel >= (int)d->num_wheels || r->right_wheel >= (int)d->num_wheels) return false;
	}
	for (uint32_t i = 0; i < d->num_wheels; ++i) {
		const sgp_wheel_desc* w = &d->wheels[i];
		if (!(w->radius > 0.0f) || !(w->inertia > 0.0f) || !(w->suspension_max_length >= w->suspension_min_length) || !(w->suspension_min_length >= 0.0f)) return false;
	}
	if (!(d->engine_inertia > 0.0f) || !(d->engine_max_rpm > 0.0f) || !(d->clutch_release_time > 0.0f) || !(d->differential_limited_slip_ratio > 1.0f)) return false;
	if (d->controller_type != SGP_VEHICLE_CONTROLLER_WHEELED && d->controller_type != SGP_VEHICLE_CONTROLLER_MOTORCYCLE) return false;
	if (d->controller_type == SGP_VEHICLE_CONTROLLER_MOTORCYCLE && !(d->max_lean_angle > 0.0f && d->max_lean_angle < 1.5f)) return false;
	return true;
}

// cos(max slope) by the same fixed polynomial the kernels use for their trigonometry (|x| <= 1.5)
static float host_cos_poly(float x)
{
	if (fabsf(x) > 1.5f) return cosf(x);
	const float x2 = x * x;
	float pc = 2.08767569878681e-9f;
	pc = pc * x2 - 2.75573192239859e-7f;
	pc = pc * x2 + 2.48015873015873e-5f;
	pc = pc * x2 - 1.38888888888889e-3f;
	pc = pc * x2 + 4.16666666666667e-2f;
	pc = pc * x2 - 0.5f;
	pc = pc * x2 + 1.0f;
	return pc;
}

static float host_sin_poly(float x)
{
	if (fabsf(x) > 1.5f) return sinf(x);
	const float x2 = x * x;
	float ps = -2.50521083854417e-8f;
	ps = ps * x2 + 2.75573192239859e-6f;
	ps = ps * x2 - 1.98412698412698e-4f;
	ps = ps * x2 + 8.33333333333333e-3f;
	ps = ps * x2 - 1.66666666666667e-1f;
	ps = ps * x2 + 1.0f;
	return ps * x;
}

static v3 hv3(const float* p) { v3 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; return r; }

static void vehicle_record_from_desc(sgd_vehicle* v, const sgp_vehicle_desc* d)
{
	memset(v, 0, sizeof(*v));
	v->body = d->body; v->alive = 1; v->num_wheels = (int)d->num_wheels;
	for (int i = 0; i < v->num_wheels; ++i) {
		sgd_wheel* w = &v->wheels[i]; const sgp_wheel_desc* s = &d->wheels[i];
		w->position = hv3(s->position); w->suspension_dir = hv3(s->suspension_dir); w->steering_axis = hv3(s->steering_axis);
		w->wheel_up = hv3(s->wheel_up); w->wheel_forward = hv3(s->wheel_forward);
		w->sus_min = s->suspension_min_length; w->sus_max = s->suspension_max_length; w->sus_preload = s->suspension_preload;
		w->spring_freq = s->spring_frequency; w->spring_damp = s->spring_damping;
		w->radius = s->radius; w->width = s->width; w->inertia = s->inertia; w->ang_damping = s->angular_damping;
		w->max_steer = s->max_steer_angle; w->max_brake_torque = s->max_brake_torque; w->max_handbrake_torque = s->max_handbrake_torque;
		memcpy(w->long_fric, s->longitudinal_friction, sizeof(w->long_fric)); memcpy(w->lat_fric, s->lateral_friction, sizeof(w->lat_fric));
		w->suspension_length = w->sus_max; w->contact_body = SGP_INVALID_ID;
	}
	v->up = hv3(d->up); v->forward = hv3(d->forward);
	v->cast_radius = d->cast_radius;
	v->tester = d->collision_tester == SGP_VEHICLE_TESTER_CYLINDER ? SGP_VEHICLE_TESTER_CYLINDER : SGP_VEHICLE_TESTER_SPHERE;
	v->cos_max_slope = host_cos_poly(d->max_slope_angle);
	v->engine_max_torque = d->engine_max_torque; v->engine_min_rpm = d->engine_min_rpm; v->engine_max_rpm = d->engine_max_rpm;
	v->engine_inertia = d->engine_inertia; v->engine_ang_damping = d->engine_angular_damping;
	memcpy(v->engine_curve, d->engine_torque_curve, sizeof(v->engine_curve));
	v->engine_rpm = d->engine_min_rpm;
	v->num_gears = (int)d->num_gears; v->num_reverse_gears = (int)d->num_reverse_gears;
	memcpy(v->gear_ratios, d->gear_ratios, sizeof(v->gear_ratios)); memcpy(v->reverse_gear_ratios, d->reverse_gear_ratios, sizeof(v->reverse_gear_ratios));
	v->switch_time = d->switch_time; v->clutch_release_time = d->clutch_release_time; v->switch_latency = d->switch_latency;
	v->shift_up_rpm = d->shift_up_rpm; v->shift_down_rpm = d->shift_down_rpm; v->clutch_strength = d->clutch_strength;
	v->current_gear = 0; v->clutch_friction = 1.0f;
	v->num_differentials = (int)d->num_differentials;
	for (int k = 0; k < v->num_differentials; ++k) {
		const sgp_differential_desc* s = &d->differentials[k];
		v->differentials[k].left = s->left_wheel; v->differentials[k].right = s->right_wheel; v->differentials[k].ratio = s->differential_ratio;
		v->differentials[k].left_right_split = s->left_right_split; v->differentials[k].limited_slip_ratio = s->limited_slip_ratio;
		v->differentials[k].engine_torque_ratio = s->engine_torque_ratio;
	}
	v->differential_limited_slip_ratio = d->differential_limited_slip_ratio;
	v->num_anti_roll_bars = (int)d->num_anti_roll_bars;
	for (int k = 0; k < v->num_anti_roll_bars; ++k) {
		v->anti_roll_bars[k].left = d->anti_roll_bars[k].left_wheel; v->anti_roll_bars[k].right = d->anti_roll_bars[k].right_wheel;
		v->anti_roll_bars[k].stiffness = d->anti_roll_bars[k].stiffness;
	}
	v->is_motorcycle = d->controller_type == SGP_VEHICLE_CONTROLLER_MOTORCYCLE;
	v->lean_enabled = v->is_motorcycle; v->lean_steering_limit = d->lean_steering_limit != 0;
	v->max_lean_angle = d->max_lean_angle;
	v->tan_max_lean = host_sin_poly(d->max_lean_angle) / host_cos_poly(d->max_lean_angle);
	v->lean_spring_constant = d->lean_spring_constant; v->lean_spring_damping = d->lean_spring_damping;
	v->lean_integration_coefficient = d->lean_spring_integration_coefficient; v->lean_integration_decay = d->lean_spring_integration_decay;
	v->lean_smoothing = d->lean_smoothing_factor;
	v->target_lean.x = 0.0f; v->target_lean.y = 0.0f; v->target_lean.z = 1.0f;
}

static inline bool vehicle_live(const sgp_world* w, uint32_t id) { return w && id < w->n_vehicles && w->veh_alive[id]; }

SGP_API int sgp_vehicle_create(sgp_world* w, const sgp_vehicle_desc* d, uint32_t* id_out)
{
	if (w) ray_server_stop(w);      // (a resident ray server must not keep this call's stream work waiting: ADVICE r05)
	if (!w || !d || !id_out) return fail(SGP_ERR_INVALID, "sgp_vehicle_create: NULL");
	if (!live(w, d->body) || (w->hb[d->body].flags & BF_MOTION_MASK) != SGP_MOTION_DYNAMIC) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_create: the chassis must be a live dynamic body");
	if (!vehicle_desc_valid(d)) return fail(SGP_ERR_INVALID, "sgp_vehicle_create: bad vehicle description");
	hipSetDevice(w->device);
	uint32_t id = w->n_vehicles;
	for (uint32_t k = 0; k < w->n_vehicles; ++k) if (!w->veh_alive[k]) { id = k; break; }     // lowest free slot
	if (id == w->n_vehicles) {
		if (w->n_vehicles == w->cap_vehicles) {
			// grow the device arrays (the step's captured graphs carry the old pointers)
			const uint32_t nc = w->cap_vehicles ? w->cap_vehicles * 2 : 64;
			sgd_vehicle* nv = nullptr; sgp_vehicle_input* ni = nullptr;
			HIP_TRY(hipMalloc((void**)&nv, sizeof(sgd_vehicle) * nc));
			HIP_TRY(hipMalloc((void**)&ni, sizeof(sgp_vehicle_input) * nc));
			HIP_TRY(hipMemsetAsync(nv, 0, sizeof(sgd_vehicle) * nc, w->stream));
			HIP_TRY(hipMemsetAsync(ni, 0, sizeof(sgp_vehicle_input) * nc, w->stream));
			if (w->n_vehicles) HIP_TRY(hipMemcpyAsync(nv, w->d_vehicles, sizeof(sgd_vehicle) * w->n_vehicles, hipMemcpyDeviceToDevice, w->stream));
			HIP_TRY(hipStreamSynchronize(w->stream));
			// (the row export of the solver passes is rebuilt by every step's controller kernel: nothing to carry over)
			const size_t row_bytes = sizeof(float4) * 16u * 4u * nc, head_bytes = sizeof(float4) * 5u * nc + sizeof(uint32_t) * (nc / 32u + 4u);      // (+ one bit per slot behind the heads: DV::veh_defer_bits)
			float4* nr = nullptr; float4* nh = nullptr;
			HIP_TRY(hipMalloc((void**)&nr, row_bytes));
			HIP_TRY(hipMalloc((void**)&nh, head_bytes));
			HIP_TRY(hipMemsetAsync(nr, 0, row_bytes, w->stream));
			HIP_TRY(hipMemsetAsync(nh, 0, head_bytes, w->stream));
			HIP_TRY(hipStreamSynchronize(w->stream));
			const size_t per_vehicle = sizeof(sgd_vehicle) + sizeof(sgp_vehicle_input) + sizeof(float4) * (16u * 4u + 5u);
			if (w->d_vehicles) { hipFree(w->d_vehicles); hipFree(w->d_veh_inputs); hipFree(w->d_veh_rows); hipFree(w->d_veh_head); w->device_bytes -= per_vehicle * w->cap_vehicles; }
			w->d_vehicles = nv; w->d_veh_inputs = ni; w->d_veh_rows = nr; w->d_veh_head = nh; w->cap_vehicles = nc;
			w->device_bytes += per_vehicle * nc;
			w->veh_inputs_dirty = true;
		}
		w->n_vehicles++;
		w->veh_alive.push_back(0); w->veh_body.push_back(SGP_INVALID_ID); w->veh_inputs.push_back(sgp_vehicle_input{ 0.0f, 0.0f, 0.0f, 0.0f });
	}
	sgd_vehicle rec;
	vehicle_record_from_desc(&rec, d);
	rec.gravity_len = sqrtf(w->dv.gx * w->dv.gx + w->dv.gy * w->dv.gy + w->dv.gz * w->dv.gz);
	HIP_TRY(hipMemcpyAsync(&w->d_vehicles[id], &rec, sizeof(rec), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));                  // `rec` lives on this stack frame
	if (rec.tester == SGP_VEHICLE_TESTER_CYLINDER) w->veh_cylinder_seen = true;
	w->veh_alive[id] = 1; w->veh_body[id] = d->body; w->veh_inputs[id] = sgp_vehicle_input{ 0.0f, 0.0f, 0.0f, 0.0f }; w->veh_inputs_dirty = true;
	w->dv.vehicles = w->d_vehicles; w->dv.vehicle_inputs = w->d_veh_inputs; w->dv.n_vehicles = w->n_vehicles;
	w->dv.veh_rows = w->d_veh_rows; w->dv.veh_head = w->d_veh_head; w->dv.veh_cap = w->cap_vehicles;
	w->dv.veh_defer_bits = (uint32_t*)(w->d_veh_head + 5u * (size_t)w->cap_vehicles);
	{ BodyCmd c = blank_cmd(d->body, CMD_SET_CHASSIS); c.flags = BF_CHASSIS; w->cmds.push_back(c); w->hb[d->body].flags |= BF_CHASSIS; }
	invalidate_graphs(w);
	w->dirty_since_step = true;
	*id_out = id;
	return SGP_OK;
}

SGP_API int sgp_vehicle_destroy(sgp_world* w, uint32_t id)
{
	if (w) ray_server_stop(w);      // (a resident ray server must not keep this call's stream work waiting: ADVICE r05)
	if (!vehicle_live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_destroy: id not live");
	hipSetDevice(w->device);
	const int zero = 0;
	HIP_TRY(hipMemcpyAsync((char*)&w->d_vehicles[id] + offsetof(sgd_vehicle, alive), &zero, sizeof(int), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	w->veh_alive[id] = 0;
	// the chassis gets colour 0 back once no live vehicle sits on it
	const uint32_t body = w->veh_body[id];
	bool other = false;
	for (uint32_t k = 0; k < w->n_vehicles; ++k) if (w->veh_alive[k] && w->veh_body[k] == body) other = true;
	if (!other && live(w, body)) { BodyCmd c = blank_cmd(body, CMD_SET_CHASSIS); c.flags = 0; w->cmds.push_back(c); w->hb[body].flags &= ~BF_CHASSIS; }
	w->dirty_since_step = true;
	return SGP_OK;
}

SGP_API int sgp_vehicle_set_inputs(sgp_world* w, uint32_t first, uint32_t n, const sgp_vehicle_input* in)
{
	if (!w || (!in && n)) return fail(SGP_ERR_INVALID, "sgp_vehicle_set_inputs: NULL");
	for (uint32_t k = 0; k < n; ++k) if (!vehicle_live(w, first + k)) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_set_inputs: id not live");
	auto cl = [](float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); };
	for (uint32_t k = 0; k < n; ++k) {
		sgp_vehicle_input c = { cl(in[k].forward, -1.0f, 1.0f), cl(in[k].right, -1.0f, 1.0f), cl(in[k].brake, 0.0f, 1.0f), cl(in[k].hand_brake, 0.0f, 1.0f) };
		w->veh_inputs[first + k] = c;
		// "On user input, assure that the car is active" (CarPhysics.cpp:362-363)
		if ((c.forward != 0.0f || c.right != 0.0f || c.brake != 0.0f || c.hand_brake != 0.0f) && live(w, w->veh_body[first + k])) w->cmds.push_back(blank_cmd(w->veh_body[first + k], CMD_ACTIVATE));
	}
	w->veh_inputs_dirty = true;
	return SGP_OK;
}
SGP_API int sgp_vehicle_set_input(sgp_world* w, uint32_t id, const sgp_vehicle_input* in) { return sgp_vehicle_set_inputs(w, id, 1, in); }

static void hvec_out(float* o, v3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

SGP_API int sgp_vehicle_get_states(sgp_world* w, uint32_t first, uint32_t n, sgp_vehicle_state* out)
{
	if (w) ray_server_stop(w);      // (a resident ray server must not keep this call's stream work waiting: ADVICE r05)
	if (!w || (!out && n)) return fail(SGP_ERR_INVALID, "sgp_vehicle_get_states: NULL");
	for (uint32_t k = 0; k < n; ++k) if (!vehicle_live(w, first + k)) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_get_states: id not live");
	if (!n) return SGP_OK;
	hipSetDevice(w->device);
	std::vector<sgd_vehicle> recs(n);
	HIP_TRY(hipMemcpyAsync(recs.data(), &w->d_vehicles[first], sizeof(sgd_vehicle) * n, hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	for (uint32_t k = 0; k < n; ++k) {
		const sgd_vehicle* v = &recs[k];
		sgp_vehicle_state* s = &out[k];
		memset(s, 0, sizeof(*s));
		for (int i = 0; i < v->num_wheels; ++i) {
			const sgd_wheel* wh = &v->wheels[i]; sgp_wheel_state* ws = &s->wheels[i];
			ws->suspension_length = wh->suspension_length; ws->steer_angle = wh->steer_angle; ws->rotation_angle = wh->angle; ws->angular_velocity = wh->angular_velocity;
			ws->has_contact = wh->has_contact; ws->contact_body = wh->has_contact ? wh->contact_body : SGP_INVALID_ID;
			if (wh->has_contact) {
				hvec_out(ws->contact_position, wh->contact_pos); hvec_out(ws->contact_normal, wh->contact_normal);
				hvec_out(ws->contact_longitudinal, wh->contact_long); hvec_out(ws->contact_lateral, wh->contact_lat); hvec_out(ws->contact_point_velocity, wh->contact_point_vel);
			}
			ws->suspension_lambda = wh->suspension.lambda + wh->max_up.lambda; ws->longitudinal_lambda = wh->longitudinal.lambda; ws->lateral_lambda = wh->lateral.lambda;
			ws->longitudinal_slip = wh->long_slip; ws->lateral_slip = wh->lat_slip;
		}
		s->engine_rpm = v->engine_rpm; s->current_gear = v->current_gear; s->clutch_friction = v->clutch_friction; s->active = w->last_step_idle ? 0 : v->active;      // (a skipped step ran no vehicle kernel: the record still says what the last real step saw)
	}
	return SGP_OK;
}
SGP_API int sgp_vehicle_get_state(sgp_world* w, uint32_t id, sgp_vehicle_state* out) { return sgp_vehicle_get_states(w, id, 1, out); }

SGP_API int sgp_vehicle_enable_lean_controller(sgp_world* w, uint32_t id, int enabled)
{
	if (w) ray_server_stop(w);      // (a resident ray server must not keep this call's stream work waiting: ADVICE r05)
	if (!vehicle_live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_enable_lean_controller: id not live");
	hipSetDevice(w->device);
	sgd_vehicle rec;
	HIP_TRY(hipMemcpyAsync(&rec, &w->d_vehicles[id], sizeof(rec), hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	const int on = (rec.is_motorcycle && enabled) ? 1 : 0;
	if (on != rec.lean_enabled) {
		HIP_TRY(hipMemcpyAsync((char*)&w->d_vehicles[id] + offsetof(sgd_vehicle, lean_enabled), &on, sizeof(int), hipMemcpyHostToDevice, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	return SGP_OK;
}

SGP_API int sgp_vehicle_reset_drivetrain(sgp_world* w, uint32_t id, float rpm, float wheel_w)
{
	if (w) ray_server_stop(w);      // (a resident ray server must not keep this call's stream work waiting: ADVICE r05)
	if (!vehicle_live(w, id)) return fail(SGP_ERR_BAD_ID, "sgp_vehicle_reset_drivetrain: id not live");
	hipSetDevice(w->device);
	sgd_vehicle rec;
	HIP_TRY(hipMemcpyAsync(&rec, &w->d_vehicles[id], sizeof(rec), hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	rec.engine_rpm = rpm;
	for (int i = 0; i < rec.num_wheels; ++i) rec.wheels[i].angular_velocity = wheel_w;
	HIP_TRY(hipMemcpyAsync(&w->d_vehicles[id], &rec, sizeof(rec), hipMemcpyHostToDevice, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	return SGP_OK;
}

