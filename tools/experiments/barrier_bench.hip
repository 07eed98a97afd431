// Micro-benchmark: what does a hand-rolled grid-wide barrier cost on MI355X (8 XCDs, non-coherent L2s) compared with a kernel boundary?
// Each round every workgroup writes one line, crosses the barrier, then reads the line its neighbour (another XCD) wrote in that round and
// checks it -- i.e. the barrier carries the agent-scope release / acquire a colour-by-colour constraint solve would need.
// Build: hipcc --offload-arch=gfx950 -O3 -o barrier_bench barrier_bench.hip ;  run: ./barrier_bench [workgroups] [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ bool grid_barrier(unsigned* counter, unsigned target)
{
	__shared__ int ok;
	__syncthreads();
	if (threadIdx.x == 0) {
		__threadfence();                                    // release: make this workgroup's stores visible device-wide
		__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		int spins = 0; ok = 1;
		while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
			__builtin_amdgcn_s_sleep(1);
			if (++spins > 4000000) { ok = 0; break; }       // never hang the box
		}
		__threadfence();                                    // acquire
	}
	__syncthreads();
	return ok != 0;
}

__global__ void __launch_bounds__(256) k_persistent(unsigned* counter, float4* lines, int rounds, unsigned* errors)
{
	const unsigned nb = gridDim.x, b = blockIdx.x;
	for (int r = 1; r <= rounds; ++r) {
		lines[(size_t)b * 256 + threadIdx.x] = make_float4((float)r, (float)b, (float)threadIdx.x, 0.0f);
		if (!grid_barrier(counter, (unsigned)r * nb)) { if (threadIdx.x == 0) atomicAdd(errors + 1, 1u); return; }
		const unsigned o = (b + 1) % nb;                    // consecutive workgroup ids sit on different XCDs
		const float4 v = lines[(size_t)o * 256 + threadIdx.x];
		if (v.x != (float)r || v.y != (float)o) atomicAdd(errors, 1u);
		// second barrier so nobody overwrites a line before its reader has read it (a solver would need only one per colour)
		if (!grid_barrier(counter + 32, (unsigned)r * nb)) { if (threadIdx.x == 0) atomicAdd(errors + 1, 1u); return; }
	}
}

__global__ void __launch_bounds__(256) k_round(float4* lines, int r, unsigned* errors)
{
	const unsigned nb = gridDim.x, b = blockIdx.x;
	const unsigned o = (b + 1) % nb;
	if (r > 1) { const float4 v = lines[(size_t)o * 256 + threadIdx.x + (size_t)((r - 1) & 1) * nb * 256]; if (v.x != (float)(r - 1) || v.y != (float)o) atomicAdd(errors, 1u); }
	lines[(size_t)b * 256 + threadIdx.x + (size_t)(r & 1) * nb * 256] = make_float4((float)r, (float)b, (float)threadIdx.x, 0.0f);
}

int main(int argc, char** argv)
{
	const int nb = argc > 1 ? atoi(argv[1]) : 256, rounds = argc > 2 ? atoi(argv[2]) : 200;
	unsigned* counter; float4* lines; unsigned* errors;
	CHECK(hipMalloc(&counter, 64 * sizeof(unsigned)));
	CHECK(hipMalloc(&lines, (size_t)nb * 256 * 2 * sizeof(float4)));
	CHECK(hipMalloc(&errors, 2 * sizeof(unsigned)));
	hipStream_t s; CHECK(hipStreamCreate(&s));
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipMemsetAsync(counter, 0, 64 * sizeof(unsigned), s)); CHECK(hipMemsetAsync(errors, 0, 2 * sizeof(unsigned), s));
		CHECK(hipEventRecord(e0, s));
		hipLaunchKernelGGL(k_persistent, dim3(nb), dim3(256), 0, s, counter, lines, rounds, errors);
		CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		unsigned h[2]; CHECK(hipMemcpy(h, errors, sizeof(h), hipMemcpyDeviceToHost));
		printf("persistent: %d workgroups, %d rounds x 2 barriers: %.3f ms -> %.2f us per barrier (data errors %u, timeouts %u)\n", nb, rounds, ms, 1000.0 * ms / (2.0 * rounds), h[0], h[1]);
	}
	// the same exchange as one launch per round, replayed from a graph (what the solver does today)
	hipGraph_t g; hipGraphExec_t ge;
	CHECK(hipMemsetAsync(errors, 0, 2 * sizeof(unsigned), s));
	CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
	for (int r = 1; r <= rounds; ++r) hipLaunchKernelGGL(k_round, dim3(nb), dim3(256), 0, s, lines, r, errors);
	CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
	for (int rep = 0; rep < 3; ++rep) {
		CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		unsigned h[2]; CHECK(hipMemcpy(h, errors, sizeof(h), hipMemcpyDeviceToHost));
		printf("graph of %d launches: %.3f ms -> %.2f us per launch (data errors %u)\n", rounds, ms, 1000.0 * ms / rounds, h[0]);
	}
	return 0;
}
