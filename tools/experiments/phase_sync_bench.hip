// Micro-benchmark for the persistent constraint solver question (task 2): what does ONE colour phase of a
// velocity iteration cost on MI355X when the phases are
//   V0  separate launches replayed from a hipGraph (what the product does today),
//   V1  phases of one resident kernel separated by a flat one-counter grid barrier,
//   V2  the same with the XCD-hierarchical barrier of MI355X_MICROARCH.md (row barrier-xcd: per-XCC counter -> XCD leader release
//       -> top counter -> acquire -> per-XCC generation), bodies in global memory,
//   V3  phases of one resident kernel in which a workgroup owns a spatial tile: interior bodies live in LDS, tile-edge bodies in
//       global memory behind write-through (sc1) stores / sc1 loads, and a workgroup waits only for the epochs of its 8 neighbours.
// The model problem has the shape of config 3's big colours: 102400 bodies on a 320 x 320 lattice, 8 colours that are perfect
// matchings (+x, +y and the two diagonals, even / odd), 2 lanes per constraint (one per body), a 32 B velocity record per body,
// an arithmetic chain of CHAIN dependent FMAs per lane between gather and scatter (0 = pure synchronisation cost).
// Every variant must produce bit-identical velocities (the update is order dependent across colours), which is the check that the
// synchronisation really carries the data.
// Build: hipcc --offload-arch=gfx950 -O3 -o phase_sync_bench phase_sync_bench.hip ;  run: ./phase_sync_bench [passes] [chain]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int TILE = 20, TILES = 16, GRID = TILE * TILES;      // 320 x 320 bodies, 16 x 16 tiles (one workgroup per CU)
constexpr int NCOL = 8, TPB = 512, SPIN_LIMIT = 2000000;

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Rec { float4 v, w; };
constexpr int ROWV = 7;                       // float4s of constraint data per lane and phase (16 B header + 96 B of rows)
struct RowSet { float4 q[ROWV]; };
__device__ __forceinline__ RowSet load_rows(const float4* rows, int c, int tile, int tid)
{
	// lane-major within a (colour, tile) block so that a wave's loads are coalesced 16 B per lane
	const float4* base = rows + ((size_t)(c * TILES * TILES + tile) * ROWV) * TPB + tid;
	RowSet r;
	#pragma unroll
	for (int k = 0; k < ROWV; ++k) r.q[k] = base[(size_t)k * TPB];
	return r;
}
__device__ __forceinline__ float fold_rows(const RowSet& r)
{
	float a = 0.0f;
	#pragma unroll
	for (int k = 0; k < ROWV; ++k) a += r.q[k].x + r.q[k].y - r.q[k].z + r.q[k].w;
	return a;
}

// constraint p (0..199) of colour c in tile (tx, ty): first body (x, y), second body (x + dx, y + dy); false when outside the lattice
__device__ __forceinline__ bool constraint_bodies(int c, int tx, int ty, int p, int& xa, int& ya, int& xb, int& yb)
{
	const int dir = c >> 1, odd = c & 1;
	const int lx = 2 * (p % (TILE / 2)) + odd, ly = p / (TILE / 2);     // every second column
	xa = tx * TILE + lx; ya = ty * TILE + ly;
	const int dx = (dir == 1) ? 0 : 1, dy = (dir == 0) ? 0 : (dir == 3 ? -1 : 1);
	if (dir == 1) {                                                    // +y pairs: every second ROW instead
		const int lx2 = p % TILE, ly2 = 2 * (p / TILE) + odd;
		xa = tx * TILE + lx2; ya = ty * TILE + ly2;
	}
	xb = xa + dx; yb = ya + dy;
	return xb < GRID && yb < GRID && yb >= 0;
}

template <int CHAIN_UNROLL>
__device__ __forceinline__ void solve_pair(Rec& r, int chain, int c, float rowsum)
{
	// exchange with the partner lane, order-dependent update, then a dependent chain standing in for the row arithmetic
	const float px = __shfl_xor(r.v.x, 1), py = __shfl_xor(r.v.y, 1), pz = __shfl_xor(r.w.x, 1);
	float s = 0.25f * (px - r.v.x) + 0.125f * (py - r.v.y) + 0.0625f * (pz - r.w.x) + 0.001f * (float)(c + 1) + 0.0009765625f * rowsum;
	float t = s;
	#pragma unroll 8
	for (int i = 0; i < chain; ++i) t = __builtin_fmaf(t, 0.99951171875f, 0.0001220703125f * s);
	r.v.x += s; r.v.y += 0.5f * t; r.v.z -= 0.25f * s; r.w.x += 0.125f * t; r.w.y -= s; r.w.z += 0.03125f * t;
}

// ------------------------------------------------------------------------------------------------ V0: one launch per phase
__global__ void __launch_bounds__(TPB) k_phase(Rec* vel, const float4* rows, int c, int chain)
{
	const int tx = blockIdx.x % TILES, ty = blockIdx.x / TILES, p = threadIdx.x >> 1, h = threadIdx.x & 1;
	if (p >= TILE * TILE / 2) return;
	int xa, ya, xb, yb;
	const bool ok = constraint_bodies(c, tx, ty, p, xa, ya, xb, yb);
	if (!ok) return;
	const RowSet rs = load_rows(rows, c, blockIdx.x, threadIdx.x);
	const int body = h ? yb * GRID + xb : ya * GRID + xa;
	Rec r = vel[body];
	solve_pair<0>(r, chain, c, fold_rows(rs));
	vel[body] = r;
}

// ------------------------------------------------------------------------------------------------ grid barriers
struct Bar {
	unsigned flat[32];            // V1 counter (own line)
	unsigned census[8 * 32];      // workgroups per XCC
	unsigned xcc_cnt[8 * 32];     // per-XCC arrival counters, one line each
	unsigned xcc_gen[8 * 32];     // per-XCC generation, one line each
	unsigned top[32];
	unsigned nxcc[32];            // number of XCCs that hold workgroups
	unsigned timeouts[32];
};

__device__ __forceinline__ bool spin_ge(gu32* p, unsigned target)
{
	for (int spins = 0; __hip_atomic_load(p, RLX_AGENT) < target; ++spins) {
		__builtin_amdgcn_s_sleep(1);
		if (spins > SPIN_LIMIT) return false;
	}
	return true;
}

__device__ __forceinline__ bool barrier_flat(Bar* b, unsigned gen, int* lds_ok)
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	if (threadIdx.x == 0) {
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__hip_atomic_fetch_add((gu32*)b->flat, 1u, RLX_AGENT);
		*lds_ok = spin_ge((gu32*)b->flat, gen * gridDim.x);
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
	__syncthreads();
	return *lds_ok != 0;
}

__device__ __forceinline__ unsigned xcc_id()
{
	return __builtin_amdgcn_s_getreg((20u) | (0u << 6) | (3u << 11)) & 7u;   // HW_REG_XCC_ID = 20, bits [3:0]
}

// census[xcc] and nxcc are filled in a first phase separated by a flat barrier
template <int FENCE>
__device__ __forceinline__ bool barrier_xcd(Bar* b, unsigned gen, unsigned xcc, unsigned n_here, unsigned n_xcc, int* lds_ok)
{
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's stores have reached the XCD's L2
	__syncthreads();
	if (threadIdx.x == 0) {
		int ok = 1;
		const unsigned old = __hip_atomic_fetch_add((gu32*)&b->xcc_cnt[xcc * 32], 1u, RLX_AGENT);
		if (old == n_here * gen - 1) {                           // last arrival of this XCC: the XCD leader of this generation
			if (FENCE) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // write the XCD L2's dirty lines back
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			}
			__hip_atomic_fetch_add((gu32*)b->top, 1u, RLX_AGENT);
			ok = spin_ge((gu32*)b->top, n_xcc * gen);
			__hip_atomic_store((gu32*)&b->xcc_gen[xcc * 32], gen, RLX_AGENT);
		} else {
			ok = spin_ge((gu32*)&b->xcc_gen[xcc * 32], gen);
		}
		if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
		*lds_ok = ok;
	}
	__syncthreads();
	return *lds_ok != 0;
}

// ------------------------------------------------------------------------------------------------ V1 / V2: resident, grid barrier
__device__ __forceinline__ Rec load_sc1(const Rec* p)
{
	gu64* q = (gu64*)p; Rec r;
	const u64 a = __hip_atomic_load(q + 0, RLX_AGENT), b = __hip_atomic_load(q + 1, RLX_AGENT),
	          c = __hip_atomic_load(q + 2, RLX_AGENT), d = __hip_atomic_load(q + 3, RLX_AGENT);
	r.v.x = __uint_as_float((unsigned)a); r.v.y = __uint_as_float((unsigned)(a >> 32));
	r.v.z = __uint_as_float((unsigned)b); r.v.w = __uint_as_float((unsigned)(b >> 32));
	r.w.x = __uint_as_float((unsigned)c); r.w.y = __uint_as_float((unsigned)(c >> 32));
	r.w.z = __uint_as_float((unsigned)d); r.w.w = __uint_as_float((unsigned)(d >> 32));
	return r;
}
__device__ __forceinline__ void store_sc1(Rec* p, const Rec& r)
{
	gu64* q = (gu64*)p;
	__hip_atomic_store(q + 0, (u64)__float_as_uint(r.v.x) | ((u64)__float_as_uint(r.v.y) << 32), RLX_AGENT);
	__hip_atomic_store(q + 1, (u64)__float_as_uint(r.v.z) | ((u64)__float_as_uint(r.v.w) << 32), RLX_AGENT);
	__hip_atomic_store(q + 2, (u64)__float_as_uint(r.w.x) | ((u64)__float_as_uint(r.w.y) << 32), RLX_AGENT);
	__hip_atomic_store(q + 3, (u64)__float_as_uint(r.w.z) | ((u64)__float_as_uint(r.w.w) << 32), RLX_AGENT);
}
template <int XCD>
__global__ void __launch_bounds__(TPB) k_resident_barrier(Rec* vel, const float4* rows, Bar* bar, int passes, int chain)
{
	__shared__ int lds_ok;
	const int tx = blockIdx.x % TILES, ty = blockIdx.x / TILES, p = threadIdx.x >> 1, h = threadIdx.x & 1;
	unsigned xcc = 0, n_here = 0, n_xcc = 0, gen = 0;
	if (XCD) {
		xcc = xcc_id();
		if (threadIdx.x == 0) {
			const unsigned old = __hip_atomic_fetch_add((gu32*)&bar->census[xcc * 32], 1u, RLX_AGENT);
			if (old == 0) __hip_atomic_fetch_add((gu32*)bar->nxcc, 1u, RLX_AGENT);
		}
		if (!barrier_flat(bar, 1, &lds_ok)) { if (threadIdx.x == 0) atomicAdd(bar->timeouts, 1u); return; }
		n_here = __hip_atomic_load((gu32*)&bar->census[xcc * 32], RLX_AGENT);
		n_xcc = __hip_atomic_load((gu32*)bar->nxcc, RLX_AGENT);
	}
	RowSet rs = load_rows(rows, 0, blockIdx.x, threadIdx.x);      // the next phase's constraint data is fetched BEFORE the barrier
	for (int pass = 0; pass < passes; ++pass)
		for (int c = 0; c < NCOL; ++c) {
			int xa, ya, xb, yb;
			const bool ok = p < TILE * TILE / 2 && constraint_bodies(c, tx, ty, p, xa, ya, xb, yb);
			const float rowsum = fold_rows(rs);
			rs = load_rows(rows, (c + 1) % NCOL, blockIdx.x, threadIdx.x);
			if (ok) {
				const int body = h ? yb * GRID + xb : ya * GRID + xa;
				Rec r = XCD == 2 ? load_sc1(&vel[body]) : vel[body];
				solve_pair<0>(r, chain, c, rowsum);
				if (XCD == 2) store_sc1(&vel[body], r); else vel[body] = r;
			}
			++gen;
			const bool alive = XCD == 2 ? barrier_xcd<0>(bar, gen, xcc, n_here, n_xcc, &lds_ok) : XCD ? barrier_xcd<1>(bar, gen, xcc, n_here, n_xcc, &lds_ok) : barrier_flat(bar, gen, &lds_ok);
			if (!alive) { if (threadIdx.x == 0) atomicAdd(bar->timeouts, 1u); return; }
		}
}

// ------------------------------------------------------------------------------------------------ V3: tiles + neighbour epochs

// epochs[tile * 32]: number of phases the tile has completed (its edge-body stores of those phases are visible)
template <int USE_LDS>
__global__ void __launch_bounds__(TPB) k_resident_tiles(Rec* vel, const float4* rows, unsigned* epochs, unsigned* timeouts, int passes, int chain)
{
	__shared__ Rec lds[TILE * TILE];
	__shared__ int lds_ok;
	const int tx = blockIdx.x % TILES, ty = blockIdx.x / TILES, p = threadIdx.x >> 1, h = threadIdx.x & 1;
	// interior bodies into LDS (edge bodies stay in global memory for the whole kernel)
	for (int i = threadIdx.x; i < TILE * TILE; i += TPB) {
		const int lx = i % TILE, ly = i / TILE;
		lds[i] = vel[(ty * TILE + ly) * GRID + tx * TILE + lx];
	}
	// the neighbour this lane polls (lanes 0..7 of wave 0)
	int nb = -1;
	if (threadIdx.x < 8) {
		const int k = threadIdx.x < 4 ? threadIdx.x : threadIdx.x + 1, nx = tx + k % 3 - 1, ny = ty + k / 3 - 1;
		if (nx >= 0 && nx < TILES && ny >= 0 && ny < TILES) nb = ny * TILES + nx;
	}
	__syncthreads();
	unsigned phase = 0;
	for (int pass = 0; pass < passes; ++pass)
		for (int c = 0; c < NCOL; ++c, ++phase) {
			const RowSet rs = load_rows(rows, c, blockIdx.x, threadIdx.x);   // in flight while wave 0 polls the neighbours
			// wait until every neighbour has completed the previous phase
			if (threadIdx.x < 64) {
				bool ready = nb < 0 || phase == 0;
				int spins = 0, ok = 1;
				while (!__all(ready)) {
					if (!ready) ready = __hip_atomic_load((gu32*)&epochs[nb * 32], RLX_AGENT) >= phase;
					if (!__all(ready)) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_LIMIT) { ok = 0; break; } }
				}
				if (threadIdx.x == 0) lds_ok = ok;
			}
			__syncthreads();
			if (!lds_ok) { if (threadIdx.x == 0) atomicAdd(timeouts, 1u); return; }
			int xa, ya, xb, yb;
			const bool ok = p < TILE * TILE / 2 && constraint_bodies(c, tx, ty, p, xa, ya, xb, yb);
			if (ok) {
				const int x = h ? xb : xa, y = h ? yb : ya, lx = x - tx * TILE, ly = y - ty * TILE;
				const bool mine = lx >= 0 && lx < TILE && ly >= 0 && ly < TILE;
				const bool interior = USE_LDS && mine && lx > 0 && lx < TILE - 1 && ly > 0 && ly < TILE - 1;
				Rec r = interior ? lds[ly * TILE + lx] : load_sc1(&vel[y * GRID + x]);
				solve_pair<0>(r, chain, c, fold_rows(rs));
				if (interior) lds[ly * TILE + lx] = r; else store_sc1(&vel[y * GRID + x], r);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores
			__syncthreads();
			if (threadIdx.x == 0) __hip_atomic_store((gu32*)&epochs[blockIdx.x * 32], phase + 1, RLX_AGENT);
		}
	for (int i = threadIdx.x; i < TILE * TILE; i += TPB) {
		const int lx = i % TILE, ly = i / TILE;
		if (USE_LDS && lx > 0 && lx < TILE - 1 && ly > 0 && ly < TILE - 1) vel[(ty * TILE + ly) * GRID + tx * TILE + lx] = lds[i];
	}
}

// ------------------------------------------------------------------------------------------------ host
static u64 hash_state(const std::vector<Rec>& v)
{
	u64 h = 1469598103934665603ull;
	const unsigned* w = (const unsigned*)v.data();
	for (size_t i = 0; i < v.size() * 8; ++i) { h ^= w[i]; h *= 1099511628211ull; }
	return h;
}

int main(int argc, char** argv)
{
	const int passes = argc > 1 ? atoi(argv[1]) : 10, chain = argc > 2 ? atoi(argv[2]) : 400;
	const int nb = GRID * GRID, nwg = TILES * TILES, phases = passes * NCOL;
	std::vector<Rec> init(nb), out(nb);
	for (int i = 0; i < nb; ++i) {
		const float f = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f);
		init[i].v = make_float4(f, 1.0f - f, 0.5f * f, 0.0f); init[i].w = make_float4(-f, 0.25f, f * f, 0.0f);
	}
	Rec* vel; Bar* bar; unsigned* epochs; float4* rows;
	const size_t nrows = (size_t)NCOL * nwg * ROWV * TPB;
	CHECK(hipMalloc(&rows, nrows * sizeof(float4)));
	{
		std::vector<float4> hr(nrows);
		for (size_t i = 0; i < nrows; ++i) { const float f = (float)((i * 2246822519u) >> 10) * (1.0f / 4194304.0f); hr[i] = make_float4(f, 0.5f * f, f * f, 0.25f); }
		CHECK(hipMemcpy(rows, hr.data(), nrows * sizeof(float4), hipMemcpyHostToDevice));
		printf("constraint data: %.1f MB per pass\n", nrows * 16.0 / 1e6);
	}
	CHECK(hipMalloc(&vel, sizeof(Rec) * nb)); CHECK(hipMalloc(&bar, sizeof(Bar))); CHECK(hipMalloc(&epochs, nwg * 32 * sizeof(unsigned)));
	hipStream_t s; CHECK(hipStreamCreate(&s));
	hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
	printf("%d bodies, %d workgroups x %d threads, %d passes x %d colours = %d phases, chain %d\n", nb, nwg, TPB, passes, NCOL, phases, chain);

	// V0
	hipGraph_t g; hipGraphExec_t ge;
	CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
	for (int pass = 0; pass < passes; ++pass) for (int c = 0; c < NCOL; ++c) hipLaunchKernelGGL(k_phase, dim3(nwg), dim3(TPB), 0, s, vel, rows, c, chain);
	CHECK(hipStreamEndCapture(s, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
	u64 h0 = 0;
	for (int rep = 0; rep < 4; ++rep) {
		CHECK(hipMemcpyAsync(vel, init.data(), sizeof(Rec) * nb, hipMemcpyHostToDevice, s));
		CHECK(hipEventRecord(e0, s)); CHECK(hipGraphLaunch(ge, s)); CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		CHECK(hipMemcpy(out.data(), vel, sizeof(Rec) * nb, hipMemcpyDeviceToHost)); h0 = hash_state(out);
		printf("V0 graph of launches      : %8.3f ms -> %6.2f us per phase   hash %016llx\n", ms, 1000.0 * ms / phases, h0);
	}
	for (int variant = 1; variant <= 5; ++variant)
		for (int rep = 0; rep < 4; ++rep) {
			CHECK(hipMemcpyAsync(vel, init.data(), sizeof(Rec) * nb, hipMemcpyHostToDevice, s));
			CHECK(hipMemsetAsync(bar, 0, sizeof(Bar), s)); CHECK(hipMemsetAsync(epochs, 0, nwg * 32 * sizeof(unsigned), s));
			CHECK(hipEventRecord(e0, s));
			if (variant == 1) hipLaunchKernelGGL(k_resident_barrier<0>, dim3(nwg), dim3(TPB), 0, s, vel, rows, bar, passes, chain);
			if (variant == 2) hipLaunchKernelGGL(k_resident_barrier<1>, dim3(nwg), dim3(TPB), 0, s, vel, rows, bar, passes, chain);
			if (variant == 4) hipLaunchKernelGGL(k_resident_barrier<2>, dim3(nwg), dim3(TPB), 0, s, vel, rows, bar, passes, chain);
			if (variant == 3) hipLaunchKernelGGL(k_resident_tiles<1>, dim3(nwg), dim3(TPB), 0, s, vel, rows, epochs, bar->timeouts, passes, chain);
			if (variant == 5) hipLaunchKernelGGL(k_resident_tiles<0>, dim3(nwg), dim3(TPB), 0, s, vel, rows, epochs, bar->timeouts, passes, chain);
			CHECK(hipEventRecord(e1, s)); CHECK(hipStreamSynchronize(s));
			float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
			CHECK(hipMemcpy(out.data(), vel, sizeof(Rec) * nb, hipMemcpyDeviceToHost));
			Bar hb; CHECK(hipMemcpy(&hb, bar, sizeof(Bar), hipMemcpyDeviceToHost));
			const u64 h = hash_state(out);
			const char* names[] = { "", "V1 resident, flat barrier ", "V2 resident, XCD barrier  ", "V3 resident, tile epochs  ", "V2b XCD barrier, sc1 ld/st", "V3b tile epochs, no LDS   " };
			printf("%s: %8.3f ms -> %6.2f us per phase   hash %016llx %s timeouts %u", names[variant], ms, 1000.0 * ms / phases, h,
			       h == h0 ? "(= V0)" : "(DIFFERS)", hb.timeouts[0]);
			if (variant == 2 || variant == 4) { printf("  census"); for (int x = 0; x < 8; ++x) printf(" %u", hb.census[x * 32]); }
			printf("\n");
		}
	return 0;
}
