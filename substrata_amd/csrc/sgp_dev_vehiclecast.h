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
