// sgp_k_mesh.hip -- K4 -- (body, static mesh) pairs: eight lanes per pair, a wave per pair for the big ones.
// One of the stage files of the step kernels (stage map: sgp_kernels.h).  Kernels first, their launch wrappers at the end.
#include "sgp_dev_all.h"

// KINDS: the shapes of the other body this instance serves (bits of SGD_SHAPE_*).  Two instances per group size: the primitives (spheres, boxes with the
// closed-form separating-axis search, capsules -- no general hull code, a fraction of the registers and scratch) and the convex hulls; the second one is
// launched only in worlds that have hulls.  G = 8 takes the lists of its kinds; G = 64 walks the one list of big pairs and skips the other instance's.
template <int MESH_GROUP, int KINDS> __global__ void __launch_bounds__(64) k_narrowphase_mesh(DV d)
{
	constexpr int MESH_PAIRS_PER_WAVE = 64 / MESH_GROUP;
	__shared__ MESH_LDS_T(MESH_GROUP, KINDS) lds[MESH_PAIRS_PER_WAVE];
	__shared__ float s_lpoly[((KINDS & 2) && !(KINDS & 8)) ? 3 * SGD_LPOLY_FLOATS : 1];      // a box's clip polygons, a column per lane (sgd_tri_box_manifold)
	const int grp = (int)(threadIdx.x / MESH_GROUP), sub = (int)(threadIdx.x % MESH_GROUP);
	MESH_LDS_T(MESH_GROUP, KINDS)& L = lds[grp];
	const float max_sep = d.st.speculative_contact_distance;
	// G = 8: a wave's eight pairs hold the same kind of body (one list per kind); G = 64: the list of the big pairs.  The lists of this instance's kinds are
	// ONE sequence of work items (an item = the next eight pairs of a list) dealt to the workgroups: with the lists taken one after the other by
	// "workgroup b takes pairs 8 b .. of every list" the first few hundred workgroups walked through a chain of spheres, THEN one of boxes, THEN one of capsules
	// while the others had nothing to do -- the launch lasted the sum of the three chains instead of the longest.
	uint32_t it_first[5], seg_base[4], seg_end[4];
	it_first[0] = 0u;
#pragma unroll
	for (uint32_t seg = 0; seg < 4u; ++seg) {
		uint32_t items = 0u; seg_base[seg] = 0u; seg_end[seg] = 0u;
		if (MESH_GROUP == 64 ? seg == 0u : ((KINDS >> seg) & 1) != 0) {
			const uint32_t seg0 = MESH_GROUP == 64 ? 0u : seg * d.cap_mesh_pairs;
			seg_end[seg] = seg0 + (MESH_GROUP == 64 ? min(d.ctr->n_mesh_big, d.cap_mesh_pairs) : min(d.ctr->n_mesh_pairs[seg], d.cap_mesh_pairs));
			seg_base[seg] = seg0 + (MESH_GROUP == 64 ? d.ctr->mesh_big_base : d.ctr->mesh_base[seg]);      // (0, or where the in-step activation round's pairs begin)
			if (seg_end[seg] > seg_base[seg]) items = (seg_end[seg] - seg_base[seg] + (uint32_t)MESH_PAIRS_PER_WAVE - 1u) / (uint32_t)MESH_PAIRS_PER_WAVE;
		}
		it_first[seg + 1] = it_first[seg] + items;
	}
	{
	for (uint32_t it = blockIdx.x; it < it_first[4]; it += gridDim.x) {
		const uint32_t seg = it >= it_first[3] ? 3u : (it >= it_first[2] ? 2u : (it >= it_first[1] ? 1u : 0u));
		const uint32_t s_first = seg == 3u ? it_first[3] : (seg == 2u ? it_first[2] : (seg == 1u ? it_first[1] : it_first[0]));
		const uint32_t n = seg == 3u ? seg_end[3] : (seg == 2u ? seg_end[2] : (seg == 1u ? seg_end[1] : seg_end[0]));
		const uint32_t p0 = (seg == 3u ? seg_base[3] : (seg == 2u ? seg_base[2] : (seg == 1u ? seg_base[1] : seg_base[0]))) + (it - s_first) * (uint32_t)MESH_PAIRS_PER_WAVE;
		const uint32_t p = p0 + (uint32_t)grp;
		bool valid = p < n;
		uint32_t mid = 0, xid = 0, fx = 0, pair = 0;
		if (valid) {
			pair = MESH_GROUP == 64 ? d.mesh_big[p] : p;
			const uint2 ab = d.mesh_pairs[pair];
			const uint32_t fa = d.flags[ab.x], fb = d.flags[ab.y];
			const bool mesh_a = f_shape(fa) == SGP_SHAPE_MESH, mesh_b = f_shape(fb) == SGP_SHAPE_MESH;
			if (mesh_a && mesh_b) valid = false;
			mid = mesh_a ? ab.x : ab.y; xid = mesh_a ? ab.y : ab.x; fx = mesh_a ? fb : fa;
			if (!((KINDS >> f_shape(fx)) & 1)) valid = false;      // (the other instance's pair: only the list of big pairs mixes the kinds)
		}
		sgd_shape X; v3 qlo = V3(0.0f, 0.0f, 0.0f), qhi = qlo; bool dropped = false;
		if (valid) {
			X = load_shape(d, xid, fx);
			const v3 e = V3(max_sep, max_sep, max_sep);
			qlo = v3_sub(V3(d.aabb_min[xid]), e); qhi = v3_add(V3(d.aabb_max[xid]), e);
		}
		// the movement hint of the active-edge rule (PhysicsSystem::ProcessBodyPair: mActiveEdgeMovementDirection = v1 - v2, after ApplyGravity): the forces
		// of this step are not applied yet at this point (k_pre_solve follows the narrow phase), so gravity is added here -- the same expression as the CPU statement's
		v3 movement = V3(0.0f, 0.0f, 0.0f);
		if (valid) {
			const v3 vx = v3_add(V3(d.vel[VEL_F4 * (size_t)xid]), v3_scale(v3_scale(V3(d.gx, d.gy, d.gz), d.dyn[xid].z), d.sp->dt));
			movement = v3_sub(vx, f_motion(d.flags[mid]) == SGP_MOTION_STATIC ? V3(0.0f, 0.0f, 0.0f) : V3(d.vel[VEL_F4 * (size_t)mid]));
		}
		mesh_pair_groups<MESH_GROUP, KINDS>(d, L, valid, X, mid, qlo, qhi, max_sep, grp, sub, pair, dropped, movement, true, s_lpoly + threadIdx.x);
		// the groups as manifolds (mesh -> body), each pruned to <= 4 points; the constraint runs lower id -> higher id, with the mesh's g-th slot
		const int ng = valid ? L.mc.ng : 0;
		if (sub < ng) {
			const sgd_mesh_group& g = L.mc.g[sub];
			sgd_manifold mm;
			sgd_hull_reduce(g.n, g.p_mesh, g.p_body, g.np, &mm);
			const uint32_t alias = mid + (uint32_t)sub;
			uint2 key;
			if (alias < xid) key = make_uint2(alias, xid); else { key = make_uint2(xid, alias); sgd_flip_manifold(&mm); }
			emit_manifold(d, key, d.flags[key.x], d.flags[key.y], mm);
		}
		if (valid && sub == 0 && dropped) atomicAdd(&d.ctr->manifolds_dropped, 1u);
		__syncthreads();          // (the tables are reused by the next eight pairs)
	}
	}
}
void launch_narrowphase_mesh_blocks(const DV& d, bool has_hulls, uint32_t blocks, hipStream_t s)
{
	// eight lanes per pair: the primitives, then (worlds with hulls) the hulls; both pass the pairs with many candidate triangles on to ...
	hipLaunchKernelGGL((k_narrowphase_mesh<8, SGD_KINDS_PRIMITIVES>), dim3(blocks), dim3(64), 0, s, d);
	if (has_hulls) hipLaunchKernelGGL((k_narrowphase_mesh<8, 8>), dim3(blocks), dim3(64), 0, s, d);
	// ... a wave per pair
	hipLaunchKernelGGL((k_narrowphase_mesh<64, SGD_KINDS_PRIMITIVES>), dim3(blocks), dim3(64), 0, s, d);
	if (has_hulls) hipLaunchKernelGGL((k_narrowphase_mesh<64, 8>), dim3(blocks), dim3(64), 0, s, d);
}
void launch_narrowphase_mesh(const DV& d, bool has_hulls, hipStream_t s) { launch_narrowphase_mesh_blocks(d, has_hulls, 2048, s); }      // (the in-step activation round: 256, launch_wake_round)
