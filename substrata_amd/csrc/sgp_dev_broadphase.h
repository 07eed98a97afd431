// sgp_dev_broadphase.h -- walkers of the broad-phase structures other stages use: cell runs of the paged grid, the static large bodies' grid (boxes and rays), the pair filter.
// Device-inline functions only (no kernels), shared between stage files; included through sgp_dev_all.h, whose order is the dependency order.
#pragma once

// fn(q0, q1): the cell-sorted records [q0, q1) of cells xa .. xb of row (y, z), tile by tile (the cells of a tile's row are neighbours in the arrays)
template <class F> SGP_DEV void grid_row_runs(const DV& d, const BpGrid& g, int xa, int xb, int y, int z, F fn)
{
	const uint32_t trow = ((uint32_t)(z >> 2) * (uint32_t)g.tny + (uint32_t)(y >> 2)) * (uint32_t)g.tnx;
	const uint32_t lrow = (uint32_t)((((z & 3) << 2) | (y & 3)) << 2);
	for (int x = xa; x <= xb; ) {
		const int xe = min(xb, x | 3);
		const uint32_t slot = d.tile_slot[trow + (uint32_t)(x >> 2)];
		if (slot < BP_TILE_PENDING) { const uint32_t c0 = slot * 64u + lrow + (uint32_t)(x & 3); fn(d.cell_start[c0], d.cell_start[c0 + (uint32_t)(xe - x) + 1u]); }
		x = xe + 1;
	}
}

SGP_DEV bool pair_passes(const DV& d, uint32_t fi, float4 mni, float4 mxi, uint32_t j)
{
	const uint32_t fj = d.flags[j];
	if (!(fj & BF_ALIVE)) return false;
	if (f_motion(fi) != SGP_MOTION_DYNAMIC && f_motion(fj) != SGP_MOTION_DYNAMIC) return false;
	if (!layers_collide(f_layer(fi), f_layer(fj))) return false;
	const float4 mnj = d.aabb_min[j], mxj = d.aabb_max[j];
	const float s = d.st.speculative_contact_distance;
	if (mni.x - s > mxj.x || mnj.x - s > mxi.x) return false;
	if (mni.y - s > mxj.y || mnj.y - s > mxi.y) return false;
	if (mni.z - s > mxj.z || mnj.z - s > mxi.z) return false;
	return true;
}

SGP_DEV uint32_t wave_alloc(uint32_t* counter);

// ---- the static large bodies' grid (LargeGrid) ---------------------------------------------------------------------------------
// cell coordinate of x, clamped into the grid (the host sorts the bodies into cells with this very expression on the bounds it read back)
SGP_DEV int lg_cell(float x, float o, float inv, int n) { return min(max((int)floorf((x - o) * inv), 0), n - 1); }
// fn(body) for every body of the grid whose cells the box [lo, hi] touches -- each body ONCE: a body sits in several cells, and it is reported from
// the one that holds the lower corner of (box intersected with the body's bounds), a cell both ranges contain whenever the two overlap.
template <class F> SGP_DEV void large_grid_query(const DV& d, v3 lo, v3 hi, F fn)
{
	const LargeGrid g = *d.lgrid;
	if (!g.n_items) return;
	const int x0 = lg_cell(lo.x, g.ox, g.inv_cell, g.nx), x1 = lg_cell(hi.x, g.ox, g.inv_cell, g.nx);
	const int y0 = lg_cell(lo.y, g.oy, g.inv_cell, g.ny), y1 = lg_cell(hi.y, g.oy, g.inv_cell, g.ny);
	const int z0 = lg_cell(lo.z, g.oz, g.inv_cell, g.nz), z1 = lg_cell(hi.z, g.oz, g.inv_cell, g.nz);
	for (int z = z0; z <= z1; ++z) for (int y = y0; y <= y1; ++y) for (int x = x0; x <= x1; ++x) {
		const uint32_t c = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx + (uint32_t)x;
		const uint32_t q0 = d.lg_start[c], q1 = d.lg_start[c + 1];
		for (uint32_t q = q0; q < q1; ++q) {
			const uint32_t b = d.lg_items[q];
			const float4 mn = d.aabb_min[b];
			if (lg_cell(fmaxf(lo.x, mn.x), g.ox, g.inv_cell, g.nx) != x || lg_cell(fmaxf(lo.y, mn.y), g.oy, g.inv_cell, g.ny) != y || lg_cell(fmaxf(lo.z, mn.z), g.oz, g.inv_cell, g.nz) != z) continue;
			fn(b);
		}
	}
}
// fn(body) for the bodies of the cells a ray (origin o, unit direction dir) crosses up to *max_t (which fn may shorten); a body may come more than once
template <class F> SGP_DEV void large_grid_ray(const DV& d, v3 o, v3 dir, const float* max_t, F fn)
{
	const LargeGrid g = *d.lgrid;
	if (!g.n_items) return;
	const float c = g.cell;
	const float oo[3] = { o.x, o.y, o.z }, dd[3] = { dir.x, dir.y, dir.z };
	const float bl[3] = { g.ox, g.oy, g.oz }, bh[3] = { g.ox + (float)g.nx * c, g.oy + (float)g.ny * c, g.oz + (float)g.nz * c };
	float t0 = 0.0f, t1 = *max_t;
	for (int a = 0; a < 3; ++a) {
		if (fabsf(dd[a]) < 1.0e-12f) { if (oo[a] < bl[a] - c || oo[a] > bh[a] + c) return; }
		else {
			float ta = (bl[a] - c - oo[a]) / dd[a], tb = (bh[a] + c - oo[a]) / dd[a];      // (one cell of slack: bounds outside the grid box sit in its border cells)
			if (ta > tb) { const float tmp = ta; ta = tb; tb = tmp; }
			t0 = fmaxf(t0, ta); t1 = fminf(t1, tb);
			if (t0 > t1) return;
		}
	}
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
		if (t_enter - c > *max_t) break;
		// (the cell and, against rounding at cell faces, nothing else: a body is in every cell its bounds touch; cells one step outside the box are its border cells)
		const int x = min(max(cx, 0), g.nx - 1), y = min(max(cy, 0), g.ny - 1), z = min(max(cz, 0), g.nz - 1);
		const uint32_t cell = ((uint32_t)z * (uint32_t)g.ny + (uint32_t)y) * (uint32_t)g.nx + (uint32_t)x;
		const uint32_t q0 = d.lg_start[cell], q1 = d.lg_start[cell + 1];
		for (uint32_t q = q0; q < q1; ++q) fn(d.lg_items[q]);
		if (tmx <= tmy && tmx <= tmz) { t_enter = tmx; tmx += tdx; cx += sx; if (cx < -1 || cx > g.nx) break; }
		else if (tmy <= tmz) { t_enter = tmy; tmy += tdy; cy += sy; if (cy < -1 || cy > g.ny) break; }
		else { t_enter = tmz; tmz += tdz; cz += sz; if (cz < -1 || cz > g.nz) break; }
		if (t_enter > t1) break;
	}
}

// large bodies (ground quad, PhysicsWorld.cpp:1123) against every body
SGP_DEV uint32_t block_alloc(uint32_t* counter, bool want);
