// sgp_dev_sweep.h -- island union-find helpers, vehicle row chunk addressing, contact-cache table size.
// Device-inline functions only (no kernels), shared between stage files; included through sgp_dev_all.h, whose order is the dependency order.
#pragma once

// Union-find over the sleepy bodies, linked by a random priority (a bijective hash of the body id) instead of by id: the
// body ids of a lattice-like pile are spatially ordered, and "smaller id wins" then builds parent chains as long as a row of
// the pile; with random priorities the expected depth is logarithmic.  Which member ends up as the root of a component is
// irrelevant (only the per-component awake flag is read), so this does not change any result.
SGP_DEV uint32_t uf_prio(uint32_t x) { uint32_t h = x * 0x9E3779B1u; h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13; return h; }
SGP_DEV uint32_t uf_find(const uint32_t* parent, uint32_t x)
{
	uint32_t p = parent[x];
	while (p != x) { x = p; p = parent[x]; }
	return x;
}
// Marking rounds before the union-find.  A sleepy body that touches a movable body which failed the sleep test, or a sleepy body
// already marked, is certainly awake: k_island_mark propagates that along the constraints for a few rounds (plain stores of 1; a
// round also sees marks made earlier in the same launch, so a mark usually travels several hops per round).  Every mark is a true
// "stays awake", so the exact union-find below only has to process what is still unmarked -- in a jittering pile, where awake
// bodies are spread everywhere, that is almost nothing, instead of one giant component of a hundred thousand sleepy bodies; an island
// that really is about to sleep, or one whose only awake member is many hops away, still goes through the union-find, which
// yields the same set of sleepers as before.
// (the vehicles' row export, defined with the vehicle kernels below)
#define VEH_HEAD_F4 5
#define VEH_CHUNK_NORMAL 0
SGP_DEV size_t veh_chunk_at(const DV& d, uint32_t k, int i, int c) { return (size_t)c * (4u * (size_t)d.veh_cap) + 4u * (size_t)k + (size_t)i; }

SGP_DEV uint32_t cache_table_size(const DV& d);

// Workgroup-wide allocation from ONE counter with one atomic: returns this thread's index if `want` (every thread of the workgroup must call it).
// Same-address atomics serialise at ~12 ns each in L2, so a per-wave atomic (1.5k of them for 100k bodies) is already ~20 us.
SGP_DEV uint32_t block_alloc(uint32_t* counter, bool want)
{
	__shared__ uint32_t s_cnt[TPB / 64];
	__shared__ uint32_t s_b;
	const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
	const unsigned long long m = __ballot(want);
	if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t tot = 0; for (int k = 0; k < TPB / 64; ++k) tot += s_cnt[k]; s_b = tot ? atomicAdd(counter, tot) : 0u; }
	__syncthreads();
	uint32_t idx = s_b + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
	for (int k = 0; k < wave; ++k) idx += s_cnt[k];
	__syncthreads();
	return idx;
}
