// xcd_cluster_bench.hip -- what does a phase barrier cost among the workgroups of ONE XCD (round 6, VERDICT r05 task 6)?
//
// profiles/r03_barrier_xcd.md: a grid barrier across the eight XCDs carries an L2 write-back + invalidate and loses to a kernel boundary.  Workgroups that
// share an XCD share its L2: data exchanged between them needs no write-back, only loads that do not stop at the per-CU vector cache.  On gfx942 / gfx950
// a WORKGROUP-scope atomic load / store carries sc0 = 1 (the memory model must cover workgroups split over two CUs), i.e. it is served by the L2, and a
// workgroup-scope atomic RMW executes in the L2 of the issuing XCD -- so a cluster of workgroups pinned to one XCD (workgroup b sits on XCD b % 8: take
// every eighth) can synchronise through workgroup-scope atomics alone.  This program measures it and CHECKS the data really crosses workgroups.
//   variant 0: 32 workgroups on XCD 0, workgroup-scope atomics for the counter and the exchanged data
//   variant 1: the same 32 workgroups, agent-scope atomics (sc1) + agent fences (what a cross-XCD barrier needs)
//   variant 2: 32 workgroups on 8 XCDs (blocks 0..31), agent scope (the cross-XCD cost at the same size)
//   variant 3: as 0 with 16 workgroups;  variant 4: as 0 with 64 workgroups (two per CU)
// Build: hipcc --offload-arch=gfx950 -O3 tools/experiments/xcd_cluster_bench.hip -o tools/experiments/xcd_cluster_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define TPBX 256

template <int SCOPE> __device__ inline uint32_t ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE> __device__ inline void st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }

// phases: every thread reads what the same lane of ANOTHER workgroup wrote in the previous phase, checks it, writes its own entry for this phase
template <int SCOPE, bool PINNED> __global__ void __launch_bounds__(TPBX) k_cluster(uint32_t* data, uint32_t* counter, uint32_t* errors, int n_wg, int phases, int chain)
{
	int w;
	if (PINNED) { if (blockIdx.x & 7u) return; w = (int)(blockIdx.x >> 3); } else w = (int)blockIdx.x;
	if (w >= n_wg) return;
	const int t = (int)threadIdx.x;
	uint32_t err = 0;
	st<SCOPE>(&data[w * TPBX + t], 0x1000u + (uint32_t)(w * TPBX + t));      // phase 0 -> buffer 0
	for (int p = 1; p <= phases; ++p) {
		// barrier: everybody's stores of phase p - 1 are out, then arrive, then wait for all
		if (SCOPE == __HIP_MEMORY_SCOPE_AGENT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__syncthreads();
		if (t == 0) {
			__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, SCOPE);
			const uint32_t want = (uint32_t)(n_wg * p);
			long spins = 0;
			while (ld<SCOPE>(counter) < want) { __builtin_amdgcn_s_sleep(1); if (++spins > 400000L) { atomicAdd(errors + 1, 1u); break; } }
		}
		__syncthreads();
		if (ld<SCOPE>(errors + 1)) break;      // (somebody gave up waiting: everybody leaves)
		if (SCOPE == __HIP_MEMORY_SCOPE_AGENT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		const int src = (w + 1 + (p % (n_wg - 1))) % n_wg;
		uint32_t v = ld<SCOPE>(&data[((p - 1) & 1) * n_wg * TPBX + src * TPBX + t]);      // buffer by phase parity: a fast workgroup writes phase p while a slow one still reads p - 1
		const uint32_t expect = (uint32_t)(p - 1) * 0x10000u + 0x1000u + (uint32_t)(src * TPBX + t);
		if (v != expect) ++err;
		for (int c = 0; c < chain; ++c) v = v * 1664525u + 1013904223u;      // (stand-in for a constraint's arithmetic)
		if (v == 0x12345678u) ++err;
		st<SCOPE>(&data[(p & 1) * n_wg * TPBX + w * TPBX + t], (uint32_t)p * 0x10000u + 0x1000u + (uint32_t)(w * TPBX + t));
	}
	if (err) atomicAdd(errors, err);
}

int main()
{
	uint32_t *data, *counter, *errors;
	CHECK(hipMalloc(&data, 4 * 2 * 64 * TPBX)); CHECK(hipMalloc(&counter, 256)); CHECK(hipMalloc(&errors, 8));
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	const int phases = 500;
	struct V { const char* name; int scope_agent, pinned, n_wg; } vs[] = {
		{"32 WGs on one XCD, workgroup-scope atomics", 0, 1, 32}, {"32 WGs on one XCD, agent scope + fences", 1, 1, 32}, {"32 WGs over 8 XCDs, agent scope + fences", 1, 0, 32},
		{"16 WGs on one XCD, workgroup scope", 0, 1, 16}, {"64 WGs on one XCD, workgroup scope", 0, 1, 64} };
	for (int chain = 0; chain <= 256; chain += 256) for (const V& v : vs) {
		float best = 1e30f; uint32_t herr[2] = { 0, 0 };
		for (int rep = 0; rep < 3; ++rep) {
			CHECK(hipMemset(counter, 0, 256)); CHECK(hipMemset(errors, 0, 8)); CHECK(hipMemset(data, 0, 4 * 2 * 64 * TPBX));
			const int grid = v.pinned ? v.n_wg * 8 : v.n_wg;
			CHECK(hipEventRecord(e0, 0));
			if (!v.scope_agent && v.pinned) hipLaunchKernelGGL((k_cluster<__HIP_MEMORY_SCOPE_WORKGROUP, true>), dim3(grid), dim3(TPBX), 0, 0, data, counter, errors, v.n_wg, phases, chain);
			else if (v.scope_agent && v.pinned) hipLaunchKernelGGL((k_cluster<__HIP_MEMORY_SCOPE_AGENT, true>), dim3(grid), dim3(TPBX), 0, 0, data, counter, errors, v.n_wg, phases, chain);
			else hipLaunchKernelGGL((k_cluster<__HIP_MEMORY_SCOPE_AGENT, false>), dim3(grid), dim3(TPBX), 0, 0, data, counter, errors, v.n_wg, phases, chain);
			CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
			CHECK(hipMemcpy(herr, errors, 8, hipMemcpyDeviceToHost));
		}
		printf("chain %3d  %-46s  %.3f us per phase   wrong values %u  timeouts %u\n", chain, v.name, best * 1000.0f / phases, herr[0], herr[1]); fflush(stdout);
	}
	return 0;
}
