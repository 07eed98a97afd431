// dataflow_probe.hip -- what does one hop of a colour-ordered Gauss-Seidel sweep cost on gfx950 when the hand-off between
// constraints is (a) a kernel boundary per colour (what the solver does today) or (b) a tagged write-through record per body
// inside ONE persistent launch (a constraint spins until both of its bodies carry the version it expects)?
//
//   hipcc --offload-arch=gfx950 -O3 -o dataflow_probe dataflow_probe.hip && ./dataflow_probe [bodies_per_edge] [iterations] [alu]
//
// Synthetic but shaped like the real thing: bodies on a 3-D lattice, one constraint per lattice edge (degree <= 6), six colours
// (x-even, x-odd, y-even, ...), constraints colour-sorted and shuffled inside a colour (scattered gathers), a body record of two
// 16-byte halves [v.xyz, tag][w.xyz, tag], an order-sensitive update so that any ordering mistake changes the bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#include <cstring>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

struct Con { uint32_t a, b; uint8_t rank_a, deg_a, rank_b, deg_b; uint32_t colour; };

__device__ __forceinline__ void update(f4& va, f4& wa, f4& vb, f4& wb, int alu)
{
	// order-sensitive, dependent chain of `alu` rounds (each ~8 dependent VALU ops)
	float s = va.x - vb.x + wa.y * 0.25f - wb.z * 0.125f;
	for (int k = 0; k < alu; ++k) { s = s * 0.999f + va.y * 0.001f; s = s - vb.y * 0.0005f; s = s * 1.0001f + 0.0003f; s = fminf(fmaxf(s, -8.0f), 8.0f); }
	const float l = 0.01f * s + 0.001f;
	va.x -= l; va.y += 0.5f * l; va.z -= 0.25f * l; wa.x += l * 0.3f; wa.y -= l * 0.2f; wa.z += l * 0.1f;
	vb.x += l; vb.y -= 0.5f * l; vb.z += 0.25f * l; wb.x -= l * 0.3f; wb.y += l * 0.2f; wb.z -= l * 0.1f;
}

// (a) one launch per (iteration, colour): plain loads and stores, the kernel boundary is the hand-off
__global__ void __launch_bounds__(64) k_phase(f4* body, const Con* con, uint32_t first, uint32_t end, int alu)
{
	for (uint32_t k = first + blockIdx.x * 64 + threadIdx.x; k < end; k += gridDim.x * 64) {
		const Con c = con[k];
		f4 va = body[2 * c.a], wa = body[2 * c.a + 1], vb = body[2 * c.b], wb = body[2 * c.b + 1];
		update(va, wa, vb, wb, alu);
		body[2 * c.a] = va; body[2 * c.a + 1] = wa; body[2 * c.b] = vb; body[2 * c.b + 1] = wb;
	}
}

__device__ __forceinline__ void ld2_sc1(const f4* p, f4& x, f4& y)
{
	asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
	             : "=&v"(x), "=&v"(y) : "v"(p) : "memory");
}
__device__ __forceinline__ void st2_sc1(f4* p, f4 x, f4 y)
{
	asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1\n\ts_nop 1" :: "v"(p), "v"(x), "v"(y) : "memory");
}

// (b) ONE persistent launch: wave g walks the 64-constraint chunks g, g + G, ... of every iteration in order.  The tag (as float bits)
// in .w of both halves of a body record counts the updates the body has received.
__global__ void __launch_bounds__(256) k_dataflow(f4* body, const Con* con, uint32_t n_con, int iterations, int alu, uint32_t* abort_flag, uint32_t* stats)
{
	const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
	const uint32_t n_waves = gridDim.x * 4;
	const uint32_t n_chunks = (n_con + 63) / 64;
	uint32_t polls = 0;
	for (int it = 0; it < iterations; ++it) {
		for (uint32_t ch = wave; ch < n_chunks; ch += n_waves) {
			const uint32_t k = ch * 64 + lane;
			const bool mine = k < n_con;
			Con c; c.a = c.b = 0; c.rank_a = c.deg_a = c.rank_b = c.deg_b = 0;
			if (mine) c = con[k];
			const uint32_t want_a = (uint32_t)it * c.deg_a + c.rank_a, want_b = (uint32_t)it * c.deg_b + c.rank_b;
			f4 va, wa, vb, wb;
			bool ready = !mine;
			uint32_t spins = 0;
			while (true) {
				if (!ready) {
					ld2_sc1(body + 2 * c.a, va, wa);
					ld2_sc1(body + 2 * c.b, vb, wb);
					ready = __float_as_uint(va.w) == want_a && __float_as_uint(wa.w) == want_a && __float_as_uint(vb.w) == want_b && __float_as_uint(wb.w) == want_b;
					++polls;
				}
				if (__all(ready)) break;
				if (++spins > 4000000u) { if (lane == 0) atomicExch(abort_flag, 1u + ch); return; }
				if ((spins & 1023u) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
				__builtin_amdgcn_s_sleep(1);
			}
			if (mine) {
				update(va, wa, vb, wb, alu);
				va.w = wa.w = __uint_as_float(want_a + 1); vb.w = wb.w = __uint_as_float(want_b + 1);
				st2_sc1(body + 2 * c.a, va, wa);
				st2_sc1(body + 2 * c.b, vb, wb);
			}
		}
	}
	if (lane == 0) atomicAdd(stats, polls);
}

int main(int argc, char** argv)
{
	const int E = argc > 1 ? atoi(argv[1]) : 46;          // 46^3 = 97k bodies, ~286k constraints
	const int T = argc > 2 ? atoi(argv[2]) : 10;
	const int alu = argc > 3 ? atoi(argv[3]) : 40;
	const int blocks_per_cu = argc > 4 ? atoi(argv[4]) : 2;
	const uint32_t B = (uint32_t)E * E * E;
	std::vector<Con> cons;
	std::vector<uint8_t> deg(B, 0);
	auto id = [&](int x, int y, int z) { return (uint32_t)((z * E + y) * E + x); };
	for (int axis = 0; axis < 3; ++axis) for (int z = 0; z < E; ++z) for (int y = 0; y < E; ++y) for (int x = 0; x < E; ++x) {
		const int c[3] = { x, y, z };
		if (c[axis] + 1 >= E) continue;
		Con k; k.a = id(x, y, z); k.b = id(x + (axis == 0), y + (axis == 1), z + (axis == 2)); k.colour = axis * 2 + (c[axis] & 1);
		k.rank_a = k.rank_b = k.deg_a = k.deg_b = 0;
		cons.push_back(k);
	}
	std::mt19937 rng(1234);
	std::shuffle(cons.begin(), cons.end(), rng);
	std::stable_sort(cons.begin(), cons.end(), [](const Con& p, const Con& q) { return p.colour < q.colour; });
	// rank of a constraint among its body's constraints by colour (a body has at most one per colour)
	std::vector<uint8_t> mask(B, 0);
	for (auto& k : cons) { mask[k.a] |= 1u << k.colour; mask[k.b] |= 1u << k.colour; }
	for (auto& k : cons) {
		k.deg_a = __builtin_popcount(mask[k.a]); k.deg_b = __builtin_popcount(mask[k.b]);
		k.rank_a = __builtin_popcount(mask[k.a] & ((1u << k.colour) - 1)); k.rank_b = __builtin_popcount(mask[k.b] & ((1u << k.colour) - 1));
	}
	const uint32_t N = (uint32_t)cons.size();
	uint32_t cstart[7] = { 0 };
	for (auto& k : cons) cstart[k.colour + 1]++;
	for (int c = 0; c < 6; ++c) cstart[c + 1] += cstart[c];
	std::vector<float> init(8 * (size_t)B);
	for (uint32_t i = 0; i < B; ++i) { for (int k = 0; k < 8; ++k) init[8 * (size_t)i + k] = 0.001f * (float)((i * 7 + k * 13) % 1000); init[8 * (size_t)i + 3] = 0.0f; init[8 * (size_t)i + 7] = 0.0f; }
	f4 *d_body_a, *d_body_b; Con* d_con; uint32_t *d_abort, *d_stats;
	CHECK(hipMalloc(&d_body_a, 32 * (size_t)B)); CHECK(hipMalloc(&d_body_b, 32 * (size_t)B)); CHECK(hipMalloc(&d_con, sizeof(Con) * N));
	CHECK(hipMalloc(&d_abort, 4)); CHECK(hipMalloc(&d_stats, 4));
	CHECK(hipMemcpy(d_con, cons.data(), sizeof(Con) * N, hipMemcpyHostToDevice));
	hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
	int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_dataflow, 256, 0));
	const int G = prop.multiProcessorCount * std::min(blocks_per_cu, std::max(1, occ - 1));
	printf("bodies %u constraints %u colours 6 iterations %d alu %d | CUs %d occupancy %d blocks/CU -> grid %d blocks (%d waves)\n", B, N, T, alu, prop.multiProcessorCount, occ, G, G * 4);
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	hipStream_t s; CHECK(hipStreamCreate(&s));
	// (a) launches
	float best_a = 1e9f;
	for (int rep = 0; rep < 5; ++rep) {
		CHECK(hipMemcpy(d_body_a, init.data(), 32 * (size_t)B, hipMemcpyHostToDevice));
		CHECK(hipEventRecord(e0, s));
		for (int it = 0; it < T; ++it) for (int c = 0; c < 6; ++c) {
			const uint32_t n = cstart[c + 1] - cstart[c];
			hipLaunchKernelGGL(k_phase, dim3((n + 63) / 64), dim3(64), 0, s, d_body_a, d_con, cstart[c], cstart[c + 1], alu);
		}
		CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best_a = std::min(best_a, ms);
	}
	printf("(a) launch per colour : %8.3f ms total, %6.2f us per hop (%d hops)\n", best_a, 1000.0f * best_a / (6 * T), 6 * T);
	// (b) dataflow
	float best_b = 1e9f; uint32_t polls = 0;
	for (int rep = 0; rep < 5; ++rep) {
		CHECK(hipMemcpy(d_body_b, init.data(), 32 * (size_t)B, hipMemcpyHostToDevice));
		CHECK(hipMemset(d_abort, 0, 4)); CHECK(hipMemset(d_stats, 0, 4));
		CHECK(hipEventRecord(e0, s));
		hipLaunchKernelGGL(k_dataflow, dim3(G), dim3(256), 0, s, d_body_b, d_con, N, T, alu, d_abort, d_stats);
		CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best_b = std::min(best_b, ms);
		uint32_t ab; CHECK(hipMemcpy(&ab, d_abort, 4, hipMemcpyDeviceToHost));
		CHECK(hipMemcpy(&polls, d_stats, 4, hipMemcpyDeviceToHost));
		if (ab) { printf("(b) ABORTED: spin limit hit at chunk %u\n", ab - 1); return 2; }
	}
	printf("(b) persistent dataflow: %8.3f ms total, %6.2f us per hop, %.1f polls per chunk\n", best_b, 1000.0f * best_b / (6 * T), (double)polls / ((double)((N + 63) / 64) * T));
	// same bits?
	std::vector<float> ra(8 * (size_t)B), rb(8 * (size_t)B);
	CHECK(hipMemcpy(ra.data(), d_body_a, 32 * (size_t)B, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(rb.data(), d_body_b, 32 * (size_t)B, hipMemcpyDeviceToHost));
	size_t bad = 0;
	for (uint32_t i = 0; i < B; ++i) for (int k = 0; k < 8; ++k) { if (k == 3 || k == 7) continue; if (memcmp(&ra[8 * (size_t)i + k], &rb[8 * (size_t)i + k], 4)) ++bad; }
	printf("bitwise mismatches between (a) and (b): %zu of %zu values\n", bad, (size_t)B * 6);
	return bad ? 3 : 0;
}
