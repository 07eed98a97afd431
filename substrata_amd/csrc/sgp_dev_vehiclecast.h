// sgp_dev_vehiclecast.h -- swept sphere against a mesh.
// Device-inline functions only (no kernels), shared between stage files; included through sgp_dev_all.h, whose order is the dependency order.
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// Wheeled vehicles (sgp_device_vehicle.h): one thread per vehicle.  Vehicles never share a chassis and apply no impulse to
// the body under a wheel, so each phase is race free without colouring; it runs as its own launch before the contact colours
// of the same pass (PhysicsSystem solves non-contact constraints first).

// swept sphere against mesh body j: closest front-side touch; on equal distance the lower triangle index (caller's order) wins
SGP_DEV float cast_sphere_mesh(const DV& d, uint32_t j, v3 o, v3 dir, float max_t, float rs, v3* n_out, v3* p_out)
{
	const MeshHeader mh = d.meshes[(uint32_t)d.pose[POSE_F4 * (size_t)j + 3].x];
	const v3 mpos = V3(d.pose[POSE_F4 * (size_t)j]); const m33 R = quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)j + 1]));
	const v3 ol = m33_tmul(R, v3_sub(o, mpos)), dl = m33_tmul(R, dir);
	float best = max_t; uint32_t best_idx = 0xFFFFFFFFu; v3 bn = V3(0.0f, 0.0f, 0.0f);
	uint32_t stack[48]; int sp = 0;
	stack[sp++] = 0;
	while (sp > 0) {
		const MeshNode nd = d.mesh_nodes[mh.node_off + stack[--sp]];
		// slab test of the centre's path against the node box grown by the sphere radius (+ a little)
		const float g = rs + 1.0e-4f * (1.0f + fabsf(nd.mxx) + fabsf(nd.mxy) + fabsf(nd.mxz) + fabsf(nd.mnx) + fabsf(nd.mny) + fabsf(nd.mnz));
		float t0 = 0.0f, t1 = best; bool miss = false;
		const float lo3[3] = { nd.mnx - g, nd.mny - g, nd.mnz - g }, hi3[3] = { nd.mxx + g, nd.mxy + g, nd.mxz + g };
		const float o3[3] = { ol.x, ol.y, ol.z }, d3[3] = { dl.x, dl.y, dl.z };
		for (int a = 0; a < 3 && !miss; ++a) {
			if (fabsf(d3[a]) <= 1.0e-12f) { if (o3[a] < lo3[a] || o3[a] > hi3[a]) miss = true; }
			else { float ta = (lo3[a] - o3[a]) / d3[a], tb = (hi3[a] - o3[a]) / d3[a]; if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; } t0 = fmaxf(t0, ta - 1.0e-4f); t1 = fminf(t1, tb + 1.0e-4f); if (t0 > t1) miss = true; }
		}
		if (miss) continue;
		if (nd.count == 0) { if (sp + 2 <= 48) { stack[sp++] = nd.left; stack[sp++] = nd.right; } continue; }
		for (uint32_t k = 0; k < nd.count; ++k) {
			const uint4 tri = d.mesh_tris[mh.tri_off + nd.left + k];
			const v3 pa = V3(d.mesh_verts[mh.vert_off + tri.x]), pb = V3(d.mesh_verts[mh.vert_off + tri.y]), pc = V3(d.mesh_verts[mh.vert_off + tri.z]);
			v3 nn;
			const float tt = sgd_cast_sphere_tri(ol, dl, pa, pb, pc, best, rs, &nn);
			if (tt >= 0.0f && (tt < best || best_idx == 0xFFFFFFFFu || (tt == best && MESH_TRI_INDEX(tri.w) < best_idx))) { best = tt; best_idx = MESH_TRI_INDEX(tri.w); bn = nn; }
		}
	}
	if (best_idx == 0xFFFFFFFFu) return -1.0f;
	const v3 n = m33_mul(R, bn);
	*n_out = n;
	*p_out = v3_sub(v3_add(o, v3_scale(dir, best)), v3_scale(n, rs));
	return best;
}

// the wheel itself (sgd_cast_disc) against mesh body j: the search per triangle whose bounds the swept wheel can reach; closest touch, on equal distance the lower
// triangle index wins (the tree only skips what the wheel cannot reach: the answer is that of a walk over every triangle)
SGP_DEV float cast_disc_mesh(const DV& d, uint32_t j, v3 o, v3 dir, v3 e, v3 din, float disc_r, float rho, float max_t, v3* n_out, v3* p_out)
{
	const MeshHeader mh = d.meshes[(uint32_t)d.pose[POSE_F4 * (size_t)j + 3].x];
	const v3 mpos = V3(d.pose[POSE_F4 * (size_t)j]); const m33 R = quat_to_m33(Q4(d.pose[POSE_F4 * (size_t)j + 1]));
	const v3 ol = m33_tmul(R, v3_sub(o, mpos)), dl = m33_tmul(R, dir), el = m33_tmul(R, e), dinl = m33_tmul(R, din);
	const v3 end = v3_add(ol, v3_scale(dl, max_t));
	const float m = disc_r + rho + 2.0e-3f;
	const v3 lo = v3_sub(v3_min(ol, end), V3(m, m, m)), hi = v3_add(v3_max(ol, end), V3(m, m, m));
	float best = 0.0f; uint32_t best_idx = 0xFFFFFFFFu; v3 bn = V3(0.0f, 0.0f, 0.0f), bp = bn;
	uint32_t stack[48]; int sp = 0;
	stack[sp++] = 0;
	while (sp > 0) {
		const MeshNode nd = d.mesh_nodes[mh.node_off + stack[--sp]];
		// the node's box against the box the swept wheel can reach (never stricter than the per-triangle test below)
		if (nd.mxx < lo.x || nd.mnx > hi.x || nd.mxy < lo.y || nd.mny > hi.y || nd.mxz < lo.z || nd.mnz > hi.z) continue;
		if (nd.count == 0) { if (sp + 2 <= 48) { stack[sp++] = nd.left; stack[sp++] = nd.right; } continue; }
		for (uint32_t k = 0; k < nd.count; ++k) {
			const uint4 tri = d.mesh_tris[mh.tri_off + nd.left + k];
			const v3 pa = V3(d.mesh_verts[mh.vert_off + tri.x]), pb = V3(d.mesh_verts[mh.vert_off + tri.y]), pc = V3(d.mesh_verts[mh.vert_off + tri.z]);
			const v3 tlo = v3_min(v3_min(pa, pb), pc), thi = v3_max(v3_max(pa, pb), pc);
			if (thi.x < lo.x || tlo.x > hi.x || thi.y < lo.y || tlo.y > hi.y || thi.z < lo.z || tlo.z > hi.z) continue;
			v3 nn, pp;
			const float tt = sgd_cast_disc([&](v3 start, v3* n, v3* q) {
				const float t = sgd_cast_sphere_tri(start, dl, pa, pb, pc, max_t, rho, n);
				if (t >= 0.0f) *q = v3_sub(v3_add(start, v3_scale(dl, t)), v3_scale(*n, rho));
				return t; }, ol, el, dinl, disc_r, &nn, &pp);
			if (tt >= 0.0f && (best_idx == 0xFFFFFFFFu || tt < best || (tt == best && MESH_TRI_INDEX(tri.w) < best_idx))) { best = tt; best_idx = MESH_TRI_INDEX(tri.w); bn = nn; bp = pp; }
		}
	}
	if (best_idx == 0xFFFFFFFFu) return -1.0f;
	*n_out = m33_mul(R, bn);
	*p_out = v3_add(mpos, m33_mul(R, bp));
	return best;
}

