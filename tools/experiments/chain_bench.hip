// Micro-benchmark: what does one more level of DEPENDENT global loads cost inside a small kernel on MI355X?
// k_chain<D>: every thread follows D dependent gathers through index arrays (random permutations over `span` elements, each array
// rewritten by a kernel between launches like the solver's velocities are), then writes one value.  64 launches are replayed from a graph.
// Build: hipcc --offload-arch=gfx950 -O3 -o chain_bench chain_bench.hip ; run: ./chain_bench [threads] [span]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Args { uint32_t* idx[6]; uint32_t* out; const uint32_t* table; int write_back; };

template <int D, bool TABLE> __global__ void __launch_bounds__(64) k_chain(Args a, uint32_t n)
{
	uint32_t first = 0;
	if (TABLE) first = a.table[3];                  // one scalar level in front (like the colour table)
	uint32_t i = first + blockIdx.x * 64 + threadIdx.x;
	if (i >= n) return;
#pragma unroll
	for (int k = 0; k < D; ++k) { const uint32_t j = a.idx[k][i]; if (a.write_back) a.idx[k][i] = j; i = j; }      // write_back: the line is dirty at kernel end, like a solved velocity
	a.out[blockIdx.x * 64 + threadIdx.x] = i;
}

__global__ void k_touch(uint32_t* p, uint32_t n) { const uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = p[i]; }

template <int D, bool TABLE> static int run(const char* name, Args a, uint32_t threads, uint32_t span, uint32_t* touch, hipStream_t s)
{
	hipGraph_t g; hipGraphExec_t ge;
	CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
	for (int r = 0; r < 64; ++r) hipLaunchKernelGGL((k_chain<D, TABLE>), dim3((threads + 63) / 64), dim3(64), 0, s, a, span);
	CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	float best = 1e9f;
	for (int rep = 0; rep < 5; ++rep) {
		CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
	}
	printf("%-28s %6.2f us per launch\n", name, 1000.0f * best / 64.0f);
	return 0;
}

int main(int argc, char** argv)
{
	const uint32_t threads = argc > 1 ? atoi(argv[1]) : 1024, span = argc > 2 ? atoi(argv[2]) : (16u << 20);
	const int write_back = argc > 3 ? atoi(argv[3]) : 0;
	std::vector<uint32_t> perm(span); std::iota(perm.begin(), perm.end(), 0u);
	std::mt19937 rng(1);
	Args a;
	for (int k = 0; k < 6; ++k) {
		std::shuffle(perm.begin(), perm.end(), rng);
		uint32_t* p; CHECK(hipMalloc(&p, (size_t)span * 4)); CHECK(hipMemcpy(p, perm.data(), (size_t)span * 4, hipMemcpyHostToDevice));
		a.idx[k] = p;
	}
	a.write_back = write_back;
	uint32_t* out; CHECK(hipMalloc(&out, (size_t)threads * 4 + 256)); a.out = out;
	uint32_t* table; CHECK(hipMalloc(&table, 256)); CHECK(hipMemset(table, 0, 256)); a.table = table;
	hipStream_t s; CHECK(hipStreamCreate(&s));
	printf("%u threads per launch, index arrays of %u elements (%.0f MB each), write back %d\n", threads, span, span * 4.0 / 1e6, write_back);
	if (run<0, false>("no load", a, threads, span, nullptr, s)) return 1;
	if (run<1, false>("1 level", a, threads, span, nullptr, s)) return 1;
	if (run<2, false>("2 dependent levels", a, threads, span, nullptr, s)) return 1;
	if (run<3, false>("3 dependent levels", a, threads, span, nullptr, s)) return 1;
	if (run<4, false>("4 dependent levels", a, threads, span, nullptr, s)) return 1;
	if (run<6, false>("6 dependent levels", a, threads, span, nullptr, s)) return 1;
	if (run<2, true>("table + 2 levels", a, threads, span, nullptr, s)) return 1;
	if (run<3, true>("table + 3 levels", a, threads, span, nullptr, s)) return 1;
	return 0;
}
