// sgp_dev_queries.h -- ray against a box, best-hit record.
// Device-inline functions only (no kernels), shared between stage files; included through sgp_dev_all.h, whose order is the dependency order.
#pragma once

SGP_DEV bool ray_aabb(v3 o, v3 dir, float4 mn, float4 mx, float tmax)
{
	float t0 = 0.0f, t1 = tmax;
	const float oo[3] = { o.x, o.y, o.z }, dd[3] = { dir.x, dir.y, dir.z };
	const float lo[3] = { mn.x, mn.y, mn.z }, hi[3] = { mx.x, mx.y, mx.z };
#pragma unroll
	for (int a = 0; a < 3; ++a) {
		if (fabsf(dd[a]) < 1.0e-12f) { if (oo[a] < lo[a] - 1.0e-4f || oo[a] > hi[a] + 1.0e-4f) return false; }
		else {
			float ta = (lo[a] - 1.0e-4f - oo[a]) / dd[a], tb = (hi[a] + 1.0e-4f - oo[a]) / dd[a];
			if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
			t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
			if (t0 > t1) return false;
		}
	}
	return true;
}
