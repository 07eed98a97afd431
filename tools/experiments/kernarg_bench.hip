// kernarg_bench.hip -- does the size of a by-value kernel argument (the solver passes its ~1 KB `DV` view by value to every launch) change
// what a launch costs inside a replayed hipGraph?  200 dependent launches of a near-empty kernel per replay, argument sizes 8 B ... 2 KB,
// and the alternative: an 8-byte pointer to the same struct in device memory, fields read through it.
//   hipcc --offload-arch=gfx950 -O3 -o kernarg_bench kernarg_bench.hip && ./kernarg_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N> struct Blob { float* out; uint32_t n; uint32_t pad; float v[N]; };

template <int N> __global__ void __launch_bounds__(64) k_val(Blob<N> b) { const uint32_t i = blockIdx.x * 64 + threadIdx.x; if (i < b.n) b.out[i] = b.v[i % N] + b.v[N - 1]; }
template <int N> __global__ void __launch_bounds__(64) k_ptr(const Blob<N>* __restrict__ p) { const Blob<N>& b = *p; const uint32_t i = blockIdx.x * 64 + threadIdx.x; if (i < b.n) b.out[i] = b.v[i % N] + b.v[N - 1]; }

template <int N> static void run(hipStream_t s, float* out, int blocks, int launches, int reps)
{
	Blob<N> h; h.out = out; h.n = blocks * 64; h.pad = 0; for (int i = 0; i < N; ++i) h.v[i] = (float)i;
	Blob<N>* d; CHECK(hipMalloc(&d, sizeof(h))); CHECK(hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice));
	for (int variant = 0; variant < 2; ++variant) {
		hipGraph_t g; hipGraphExec_t ge;
		CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
		for (int k = 0; k < launches; ++k) { if (variant == 0) hipLaunchKernelGGL(k_val<N>, dim3(blocks), dim3(64), 0, s, h); else hipLaunchKernelGGL(k_ptr<N>, dim3(blocks), dim3(64), 0, s, (const Blob<N>*)d); }
		CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
		hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		for (int r = 0; r < 3; ++r) CHECK(hipGraphLaunch(ge, s));
		CHECK(hipStreamSynchronize(s));
		CHECK(hipEventRecord(e0, s));
		for (int r = 0; r < reps; ++r) CHECK(hipGraphLaunch(ge, s));
		CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		printf("  %4zu-byte argument %-22s: %.2f us per launch (%d blocks)\n", variant == 0 ? sizeof(h) : sizeof(void*), variant == 0 ? "by value" : "pointer to device copy", 1000.0f * ms / (reps * launches), blocks);
		CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
	}
	CHECK(hipFree(d));
}

int main()
{
	hipStream_t s; CHECK(hipStreamCreate(&s));
	float* out; CHECK(hipMalloc(&out, 4 * 64 * 4096));
	for (int blocks : { 4, 256, 1600 }) {
		printf("grid of %d one-wave workgroups, 200 dependent launches per graph replay:\n", blocks);
		run<2>(s, out, blocks, 200, 20);
		run<60>(s, out, blocks, 200, 20);
		run<250>(s, out, blocks, 200, 20);
		run<500>(s, out, blocks, 200, 20);
	}
	return 0;
}
