// sgp_dev_constraints.h -- constraint rows, effective masses and the pair-key hash table of the contact cache.
// Device-inline functions only (no kernels), shared between stage files; included through sgp_dev_all.h, whose order is the dependency order.
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// K6: deterministic round-based greedy colouring.  priority = mix64(pair key); per round each uncoloured manifold
// claims its movable bodies with atomicMin; the manifold that holds both claims takes the lowest colour free on
// both bodies.  The result depends only on the SET of manifolds (spec: DESIGN.md "Colouring").

SGP_DEV uint32_t cache_find(const DV& d, uint64_t key, int* np_col_prev);

// ---------------------------------------------------------------------------------------------------------------
// K5: contact constraint setup (Jolt ContactConstraintManager::TemplatedAddContactConstraint): contact-cache match
// for warm starting, restitution / speculative bias, effective masses.  Constraints are written colour-sorted.

SGP_DEV float axis_eff_mass(float im1, const sym33& I1, v3 r1, float im2, const sym33& I2, v3 r2, v3 axis)
{
	float inv = 0.0f;
	if (im1 > 0.0f) { const v3 c = v3_cross(r1, axis); inv = im1 + v3_dot(sym33_mul(I1, c), c); }
	if (im2 > 0.0f) { const v3 c = v3_cross(r2, axis); inv = inv + (im2 + v3_dot(sym33_mul(I2, c), c)); }
	return inv > 0.0f ? 1.0f / inv : 0.0f;
}

// rows of one (point, axis) for the velocity iterations: the two lever-arm cross products and their inverse-inertia images
SGP_DEV float4* axis_rows(const DV& d, uint32_t slot, int point, int axis) { return d.rows + (size_t)((point * 3 + axis) * 4) * d.cap_manifolds + slot; }

SGP_DEV uint32_t ht_hash(uint64_t key, uint32_t mask) { return (uint32_t)(sgp_mix64(key) >> 20) & mask; }

// -> the pair's slot in the previous step's constraints (0xFFFFFFFF: it had none) and that constraint's np_col: ONE 16-byte entry per probe
SGP_DEV uint32_t cache_find(const DV& d, uint64_t key, int* np_col_prev)
{
	const uint32_t size = *d.ht_cur;          // (the part of the table the last rebuild used: k_island_mark empties it)
	const uint32_t mask = size - 1;
	uint32_t h = ht_hash(key, mask);
	for (uint32_t probe = 0; probe < size; ++probe) {
		const uint4 e = d.ht[h];
		const uint64_t k = ((uint64_t)e.y << 32) | e.x;
		if (k == key) { *np_col_prev = (int)e.w; return e.z; }
		if (k == ~0ull) return 0xFFFFFFFFu;
		h = (h + 1) & mask;
	}
	return 0xFFFFFFFFu;
}
// man_colour of a manifold between the narrow phase and k_colour_inherit: -1, or -(3 + c) when the narrow phase's own probe of the contact cache found the
// pair's previous constraint in colour c (k_colour_inherit then needs neither a probe nor the previous constraint's header)
SGP_DEV int man_colour_candidate(int np_col_prev) { return -(3 + ((np_col_prev >> 8) & 0xFF)); }

// ---------------------------------------------------------------------------------------------------------------
// contact cache (pair key -> constraint slot) for the next step's warm start; contact events

// The table is allocated for the world's manifold capacity, but a step uses (and clears) only the power of two that holds four times its
// constraints: 8 MB instead of 33 MB per step at config 3, and a table that stays in the caches between the rebuild and the next step's probes.
SGP_DEV uint32_t cache_table_size(const DV& d)
{
	// (room for the step's constraints and for every entry of the previous cache -- all of them may be carried over, k_cache_build: half full at the very worst, a
	// quarter when few pairs sleep; called by k_island_mark, where the parity is this step's)
	const uint32_t want = 2u * max(d.ctr->n_constraints + min(d.cache_total[d.sp->parity ^ 1u], d.cap_manifolds), 512u);
	uint32_t size = 1024u;
	while (size < want && size < d.ht_size) size <<= 1;
	return min(size, d.ht_size);
}
