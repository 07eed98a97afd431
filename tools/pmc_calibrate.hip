// pmc_calibrate.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (VERDICT r05 task 2b).
//
// MI355X_MICROARCH.md (HBM section) calibrates FETCH_SIZE only for a 16 B/lane coalesced stream (reads 1/2 of the bytes).  The step's kernels
// are mostly 4-16 B gathers and scattered stores, so this program runs each ACCESS PATTERN the step uses over a buffer far larger than the
// 256 MiB Infinity Cache with a byte count known in advance:
//   stream16 / stream8 / stream4   coalesced reads, 16 / 8 / 4 bytes per lane
//   gather16<8>                    a 16-byte read from a different 128-byte line per lane, every line of the buffer touched exactly once
//   gather16<4>                    a 16-byte read per 64-byte half line, the two halves of a line touched far apart in time
//   gather32                       a 32-byte record (two float4: what a body-record gather is) per 128-byte line
//   write16 / scatter16 / scatter64  coalesced 16 B/lane stores; a 16 B store per 128-byte line; 64 contiguous bytes per lane into its own line
// It prints, per kernel, the bytes REQUESTED by the lanes, the bytes a 64-byte and a 128-byte transfer granule would move, and the time
// (HIP events), so the counters collected around it (tools/collect_pmc.sh calibrate: FETCH_SIZE, WRITE_SIZE and the raw TCC_EA0_RDREQ /
// TCC_BUBBLE / TCC_EA0_RDREQ_32B / TCC_EA0_WRREQ / TCC_EA0_WRREQ_64B) can be turned into a factor per access pattern; the time bounds the
// real traffic from above (nothing moves faster than ~6.3 TB/s).  Build: hipcc --offload-arch=gfx950 -O3 tools/pmc_calibrate.hip -o tools/pmc_calibrate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// every kernel folds what it read into `sink` behind a condition the compiler cannot resolve, so the loads stay
__global__ void __launch_bounds__(256) k_cal_stream16(const float4* p, size_t n, float* sink)
{
	float acc = 0.0f;
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
	if (acc == 123.456f) *sink = acc;
}
__global__ void __launch_bounds__(256) k_cal_stream8(const float2* p, size_t n, float* sink)
{
	float acc = 0.0f;
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) { const float2 v = p[i]; acc += v.x + v.y; }
	if (acc == 123.456f) *sink = acc;
}
__global__ void __launch_bounds__(256) k_cal_stream4(const float* p, size_t n, float* sink)
{
	float acc = 0.0f;
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) acc += p[i];
	if (acc == 123.456f) *sink = acc;
}
// unit u of `units` (a power of two) -> a different unit, far from its neighbours': multiplication by an odd constant is a bijection mod 2^k
__device__ inline size_t scramble(size_t u, size_t units) { return (u * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull) & (units - 1); }
// one 16-byte read per `stride16` float4 (8 = one per 128-byte line, 4 = one per 64-byte half line)
template <int STRIDE16> __global__ void __launch_bounds__(256) k_cal_gather16(const float4* p, size_t units, float* sink)
{
	float acc = 0.0f;
	for (size_t u = blockIdx.x * 256ull + threadIdx.x; u < units; u += gridDim.x * 256ull) { const float4 v = p[scramble(u, units) * (size_t)STRIDE16]; acc += v.x + v.y + v.z + v.w; }
	if (acc == 123.456f) *sink = acc;
}
__global__ void __launch_bounds__(256) k_cal_gather32(const float4* p, size_t units, float* sink)
{
	float acc = 0.0f;
	for (size_t u = blockIdx.x * 256ull + threadIdx.x; u < units; u += gridDim.x * 256ull) {
		const float4* r = p + scramble(u, units) * 8;
		const float4 v = r[0], w = r[1]; acc += v.x + v.y + v.z + v.w + w.x + w.y + w.z + w.w;
	}
	if (acc == 123.456f) *sink = acc;
}
__global__ void __launch_bounds__(256) k_cal_write16(float4* p, size_t n)
{
	for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) p[i] = make_float4((float)i, 1.0f, 2.0f, 3.0f);
}
__global__ void __launch_bounds__(256) k_cal_scatter16(float4* p, size_t units)
{
	for (size_t u = blockIdx.x * 256ull + threadIdx.x; u < units; u += gridDim.x * 256ull) p[scramble(u, units) * 8] = make_float4((float)u, 1.0f, 2.0f, 3.0f);
}
__global__ void __launch_bounds__(256) k_cal_scatter64(float4* p, size_t units)
{
	for (size_t u = blockIdx.x * 256ull + threadIdx.x; u < units; u += gridDim.x * 256ull) {
		float4* r = p + scramble(u, units) * 8;
		const float4 v = make_float4((float)u, 1.0f, 2.0f, 3.0f);
		r[0] = v; r[1] = v; r[2] = v; r[3] = v;
	}
}
// (a full 128-byte line per lane: two of these per manifold are what an AoS cache record costs to write)
__global__ void __launch_bounds__(256) k_cal_scatter128(float4* p, size_t units)
{
	for (size_t u = blockIdx.x * 256ull + threadIdx.x; u < units; u += gridDim.x * 256ull) {
		float4* r = p + scramble(u, units) * 8;
		const float4 v = make_float4((float)u, 1.0f, 2.0f, 3.0f);
#pragma unroll
		for (int k = 0; k < 8; ++k) r[k] = v;
	}
}
__global__ void __launch_bounds__(256) k_cal_gather128(const float4* p, size_t units, float* sink)
{
	float acc = 0.0f;
	for (size_t u = blockIdx.x * 256ull + threadIdx.x; u < units; u += gridDim.x * 256ull) {
		const float4* r = p + scramble(u, units) * 8;
#pragma unroll
		for (int k = 0; k < 8; ++k) { const float4 v = r[k]; acc += v.x + v.y + v.z + v.w; }
	}
	if (acc == 123.456f) *sink = acc;
}

int main(int argc, char** argv)
{
	const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : 2048ull) << 20;      // MiB, a power of two
	const int reps = argc > 2 ? atoi(argv[2]) : 3;
	float4* buf; float* sink;
	CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 4));
	CHECK(hipMemset(buf, 0, bytes));
	const size_t lines = bytes / 128;
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	const dim3 grid(256 * 16), block(256);
	printf("buffer %zu MiB, %zu 128-byte lines, %d launches per kernel (the first is the warm-up)\n", bytes >> 20, lines, reps);
	printf("%-18s %14s %14s %14s %10s %10s\n", "kernel", "requested_B", "granule64_B", "granule128_B", "ms", "req_GB/s");
	struct Row { const char* name; double req, g64, g128; };
	const Row rows[] = {
		{"k_cal_stream16", (double)bytes, (double)bytes, (double)bytes},
		{"k_cal_stream8", (double)bytes, (double)bytes, (double)bytes},
		{"k_cal_stream4", (double)bytes, (double)bytes, (double)bytes},
		{"k_cal_gather16<8>", lines * 16.0, lines * 64.0, lines * 128.0},
		{"k_cal_gather16<4>", lines * 32.0, lines * 128.0, lines * 128.0},      // (a 128-byte granule: the second half comes from cache if the line is still there, else 2 x 128)
		{"k_cal_gather32", lines * 32.0, lines * 64.0, lines * 128.0},
		{"k_cal_gather128", lines * 128.0, lines * 128.0, lines * 128.0},
		{"k_cal_write16", (double)bytes, (double)bytes, (double)bytes},
		{"k_cal_scatter16", lines * 16.0, lines * 64.0, lines * 128.0},
		{"k_cal_scatter64", lines * 64.0, lines * 64.0, lines * 128.0},
		{"k_cal_scatter128", lines * 128.0, lines * 128.0, lines * 128.0},
	};
	for (int k = 0; k < (int)(sizeof(rows) / sizeof(rows[0])); ++k) {
		float best = 1e30f;
		for (int r = 0; r < reps; ++r) {
			CHECK(hipEventRecord(e0, 0));
			switch (k) {
			case 0: hipLaunchKernelGGL(k_cal_stream16, grid, block, 0, 0, (const float4*)buf, bytes / 16, sink); break;
			case 1: hipLaunchKernelGGL(k_cal_stream8, grid, block, 0, 0, (const float2*)buf, bytes / 8, sink); break;
			case 2: hipLaunchKernelGGL(k_cal_stream4, grid, block, 0, 0, (const float*)buf, bytes / 4, sink); break;
			case 3: hipLaunchKernelGGL(k_cal_gather16<8>, grid, block, 0, 0, (const float4*)buf, lines, sink); break;
			case 4: hipLaunchKernelGGL(k_cal_gather16<4>, grid, block, 0, 0, (const float4*)buf, lines * 2, sink); break;
			case 5: hipLaunchKernelGGL(k_cal_gather32, grid, block, 0, 0, (const float4*)buf, lines, sink); break;
			case 6: hipLaunchKernelGGL(k_cal_gather128, grid, block, 0, 0, (const float4*)buf, lines, sink); break;
			case 7: hipLaunchKernelGGL(k_cal_write16, grid, block, 0, 0, buf, bytes / 16); break;
			case 8: hipLaunchKernelGGL(k_cal_scatter16, grid, block, 0, 0, buf, lines); break;
			case 9: hipLaunchKernelGGL(k_cal_scatter64, grid, block, 0, 0, buf, lines); break;
			case 10: hipLaunchKernelGGL(k_cal_scatter128, grid, block, 0, 0, buf, lines); break;
			}
			CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (r > 0 || reps == 1) best = ms < best ? ms : best;
		}
		printf("%-18s %14.0f %14.0f %14.0f %10.3f %10.1f\n", rows[k].name, rows[k].req, rows[k].g64, rows[k].g128, best, rows[k].req / best * 1e-6);
	}
	CHECK(hipFree(buf)); CHECK(hipFree(sink));
	return 0;
}
