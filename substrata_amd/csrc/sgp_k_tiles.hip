// sgp_k_tiles.hip -- (e) tile export and routing, re-tiling histograms, ghost records.
// One of the stage files of the step kernels (stage map: sgp_kernels.h).  Kernels first, their launch wrappers at the end.
#include "sgp_dev_all.h"

// multi-GPU tiles: bodies owned by this tile whose inflated AABB pokes outside [lo,hi)
// Export in ASCENDING BODY ID without a sort: pass 1 counts the qualifying bodies of every 256-body block, pass 2 gives each block the sum
// of the counts before it and each qualifying thread its rank inside the block (wave ballots), so record k of the output is the k-th
// qualifying body.  (The exchange wants a deterministic order; the host used to sort a few thousand 96-byte records every step.)
SGP_DEV bool export_qualifies(const DV& d, uint32_t i, float3 lo, float3 hi, float margin, uint32_t& f_out)
{
	if (i >= d.sp->n_slots) return false;
	const uint32_t f = d.flags[i];
	f_out = f;
	if (!(f & BF_ALIVE) || (f & (BF_GHOST | BF_LARGE)) || f_motion(f) == SGP_MOTION_STATIC) return false;
	const float4 mn = d.aabb_min[i], mx = d.aabb_max[i];
	return mn.x - margin < lo.x || mn.y - margin < lo.y || mn.z - margin < lo.z ||
	       mx.x + margin >= hi.x || mx.y + margin >= hi.y || mx.z + margin >= hi.z;
}

// the part of the record only an ownership migration reads: user data, layer + flags, damping, gravity factor
SGP_DEV void fill_ghost_desc(const DV& d, uint32_t i, uint32_t f, sgp_ghost_record& r)
{
	r.userdata = d.userdata[i];
	const float4 dy = d.dyn[i];
	r.gravity_factor = dy.z; r.linear_damping = dy.x; r.angular_damping = dy.y;
	r.flags = f_layer(f) | ((f & BF_SENSOR) ? SGP_GHOST_FLAG_SENSOR : 0u) | ((f & BF_ALLOW_SLEEP) ? SGP_GHOST_FLAG_ALLOW_SLEEP : 0u) | ((f & BF_ZERO_LIN_DRAG) ? SGP_GHOST_FLAG_ZERO_DRAG : 0u) | ((f & BF_CHASSIS) ? SGP_GHOST_FLAG_CHASSIS : 0u);
	r._pad[0] = 0; r._pad[1] = 0;
}

__global__ void __launch_bounds__(TPB) k_export_count(DV d, float3 lo, float3 hi, float margin)
{
	__shared__ uint32_t wsum[TPB / 64];
	uint32_t f;
	const bool q = export_qualifies(d, blockIdx.x * TPB + threadIdx.x, lo, hi, margin, f);
	const unsigned long long m = __ballot(q);
	if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
	__syncthreads();
	if (threadIdx.x == 0) { uint32_t t = 0; for (int k = 0; k < TPB / 64; ++k) t += wsum[k]; d.export_counts[blockIdx.x] = t; }
}

__global__ void __launch_bounds__(TPB) k_export_boundary(DV d, float3 lo, float3 hi, float margin, sgp_ghost_record* out, uint32_t cap, uint32_t* count)
{
	__shared__ uint32_t part[TPB];
	__shared__ uint32_t wsum[TPB / 64];
	// blocks before this one
	uint32_t acc = 0;
	for (uint32_t b = threadIdx.x; b < blockIdx.x; b += TPB) acc += d.export_counts[b];
	part[threadIdx.x] = acc;
	__syncthreads();
	for (int off = TPB / 2; off > 0; off >>= 1) { if (threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off]; __syncthreads(); }
	const uint32_t base = part[0];
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	uint32_t f = 0;
	const bool q = export_qualifies(d, i, lo, hi, margin, f);
	const unsigned long long m = __ballot(q);
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	if (lane == 0) wsum[wv] = (uint32_t)__popcll(m);
	__syncthreads();
	uint32_t wbase = 0, total = 0;
	for (int k = 0; k < TPB / 64; ++k) { if (k < wv) wbase += wsum[k]; total += wsum[k]; }
	if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *count = base + total;      // the last block knows the grand total
	if (!q) return;
	const uint32_t k = base + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
	if (k >= cap) return;
	sgp_ghost_record r;
	const float4 p = d.pose[POSE_F4 * (size_t)i], qq = d.pose[POSE_F4 * (size_t)i + 1], lv = d.vel[VEL_F4 * (size_t)i], av = d.vel[VEL_F4 * (size_t)i + 1], sh = d.pose[POSE_F4 * (size_t)i + 3];
	r.pos[0] = p.x; r.pos[1] = p.y; r.pos[2] = p.z;
	r.rot[0] = qq.x; r.rot[1] = qq.y; r.rot[2] = qq.z; r.rot[3] = qq.w;
	r.lin_vel[0] = lv.x; r.lin_vel[1] = lv.y; r.lin_vel[2] = lv.z;
	r.ang_vel[0] = av.x; r.ang_vel[1] = av.y; r.ang_vel[2] = av.z;
	r.shape_type = (int32_t)f_shape(f);
	r.shape[0] = sh.x; r.shape[1] = sh.y; r.shape[2] = sh.z; r.shape[3] = 0.0f;
	r.mass = d.torque[i].w; r.friction = sh.w; r.restitution = d.pose[POSE_F4 * (size_t)i + 2].w;
	r.motion_type = f_motion(f);
	r.global_id = i;
	fill_ghost_desc(d, i, f, r);
	out[k] = r;
}

// ---- export with the routing done on the device: one record per (qualifying body, destination tile) --------------------------------
// Same rules as the host statement (sgp_tiles_route): a qualifying body goes to every OTHER tile whose region grown by `pad` contains its
// centre; an owned dynamic body whose centre has left this tile's region emigrates (flagged, listed).  The send buffer is segmented by
// destination (rank order) and ascending in body id inside a segment: counts per (block, destination) -> scan -> write, no sort, no atomics.
SGP_DEV bool tile_in_box(float4 p, const float* lo, const float* hi, float pad)
{
	return p.x >= lo[0] - pad && p.x < hi[0] + pad && p.y >= lo[1] - pad && p.y < hi[1] + pad && p.z >= lo[2] - pad && p.z < hi[2] + pad;
}
SGP_DEV unsigned long long route_mask(const DV& d, uint32_t i, const TileRoute& t, bool& emigrates, uint32_t& f)
{
	emigrates = false;
	const float* mylo = t.boxes + 6 * t.my_rank; const float* myhi = mylo + 3;
	if (!export_qualifies(d, i, make_float3(mylo[0], mylo[1], mylo[2]), make_float3(myhi[0], myhi[1], myhi[2]), t.margin, f)) return 0ull;
	const float4 p = d.pose[POSE_F4 * (size_t)i];
	// an owned dynamic body emigrates only when another tile's own (unpadded) region contains its centre: where the caller's boxes leave a gap
	// nobody would accept the body, so it stays with its current owner instead of vanishing
	// (a vehicle's chassis stays with the tile that holds the vehicle record: SGP_GHOST_FLAG_CHASSIS)
	const bool left = t.n_tiles > 1 && f_motion(f) == SGP_MOTION_DYNAMIC && !(f & BF_CHASSIS) && !tile_in_box(p, mylo, myhi, 0.0f);
	bool taker = false;
	unsigned long long m = 0ull;
	for (uint32_t r = 0; r < t.n_tiles; ++r) {
		if (r == t.my_rank) continue;
		const float* lo = t.boxes + 6 * r;
		if (tile_in_box(p, lo, lo + 3, t.pad)) m |= 1ull << r;
		if (left && tile_in_box(p, lo, lo + 3, 0.0f)) taker = true;
	}
	emigrates = left && taker;
	return m;
}

// Where the owned bodies are, for sgp_tiles_rebalance (a few times a second at most: plain global atomics)
__global__ void __launch_bounds__(TPB) k_tiles_hist(DV d, TilePlanes tp, int level, uint32_t* out)
{
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	if (i >= d.sp->n_slots) return;
	const uint32_t f = d.flags[i];
	if (!(f & BF_ALIVE) || (f & BF_ALIAS) || f_motion(f) != SGP_MOTION_DYNAMIC) return;      // (ghosts are kinematic here: owned bodies only)
	const float4 p = d.pose[POSE_F4 * (size_t)i];
	if (level == 0) {
		int* o = (int*)out;
		atomicMin(&o[0], float_to_ordered(p.x)); atomicMin(&o[1], float_to_ordered(p.y)); atomicMin(&o[2], float_to_ordered(p.z));
		atomicMax(&o[3], float_to_ordered(p.x)); atomicMax(&o[4], float_to_ordered(p.y)); atomicMax(&o[5], float_to_ordered(p.z));
		return;
	}
	uint32_t ix = 0, iy = 0;
	for (uint32_t k = 0; k + 1 < tp.gx; ++k) if (p.x >= tp.xp[k]) ix = k + 1;
	if (level == 3) for (uint32_t k = 0; k + 1 < tp.gy; ++k) if (p.y >= tp.yp[4 * ix + k]) iy = k + 1;
	const int a = level - 1;
	const float c = a == 0 ? p.x : (a == 1 ? p.y : p.z);
	const float w = tp.ghi[a] - tp.glo[a];
	int bin = w > 0.0f ? (int)floorf((c - tp.glo[a]) / w * (float)SGP_TILE_HIST_BINS) : 0;
	bin = min(max(bin, 0), SGP_TILE_HIST_BINS - 1);
	const uint32_t group = level == 1 ? 0u : (level == 2 ? ix : ix + tp.gx * iy);
	// what a body weighs: 1, or (by_contacts) 1 + the contact constraints it was in last step (the colours in its mask): the work of a tile is its
	// constraints more than its bodies, and a pile has them at the bottom
	const uint32_t wgt = tp.by_contacts ? 1u + (uint32_t)__popcll(d.colour_mask[i]) : 1u;
	atomicAdd(&out[group * SGP_TILE_HIST_BINS + (uint32_t)bin], wgt);
}

__global__ void __launch_bounds__(TPB) k_route_count(DV d, TileRoute t, uint32_t* block_counts)
{
	__shared__ uint32_t cnt[SGP_MAX_TILES + 1];
	if (threadIdx.x <= SGP_MAX_TILES) cnt[threadIdx.x] = 0;
	__syncthreads();
	bool emig; uint32_t f;
	const unsigned long long mask = route_mask(d, blockIdx.x * TPB + threadIdx.x, t, emig, f);
	const int lane = threadIdx.x & 63;
	for (uint32_t r = 0; r < t.n_tiles; ++r) { const unsigned long long b = __ballot((mask >> r) & 1ull); if (b && lane == 0) atomicAdd(&cnt[r], (uint32_t)__popcll(b)); }
	{ const unsigned long long b = __ballot(emig); if (b && lane == 0) atomicAdd(&cnt[t.n_tiles], (uint32_t)__popcll(b)); }
	__syncthreads();
	if (threadIdx.x <= t.n_tiles) block_counts[(size_t)blockIdx.x * (t.n_tiles + 1) + threadIdx.x] = cnt[threadIdx.x];
}

// one workgroup: per column (destination, or emigrants) the exclusive scan of the block counts; then the segment starts
__global__ void __launch_bounds__(1024) k_route_scan(const uint32_t* block_counts, uint32_t* block_offsets, uint32_t n_blocks, uint32_t n_tiles, RouteHeader* header,
                                                      uint32_t cap, uint32_t emigrant_cap, uint32_t* gather_row, uint32_t cap_recv, uint32_t host_status)
{
	__shared__ uint32_t wave_sums[16];
	__shared__ uint32_t carry;
	__shared__ uint32_t totals[SGP_MAX_TILES + 1];
	const uint32_t cols = n_tiles + 1;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (uint32_t c = 0; c < cols; ++c) {
		if (threadIdx.x == 0) carry = 0;
		__syncthreads();
		for (uint32_t start = 0; start < n_blocks; start += 1024) {
			const uint32_t b = start + threadIdx.x;
			const uint32_t v = b < n_blocks ? block_counts[(size_t)b * cols + c] : 0u;
			uint32_t x = v;
			for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(x, off, 64); if (lane >= off) x += y; }
			if (lane == 63) wave_sums[wave] = x;
			__syncthreads();
			uint32_t wbase = carry;
			for (int k = 0; k < wave; ++k) wbase += wave_sums[k];
			if (b < n_blocks) block_offsets[(size_t)b * cols + c] = wbase + x - v;
			__syncthreads();
			if (threadIdx.x == 1023) carry = wbase + x;
			__syncthreads();
		}
		if (threadIdx.x == 0) totals[c] = carry;
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		uint32_t acc = 0;
		for (uint32_t r = 0; r < SGP_MAX_TILES; ++r) { const uint32_t n = r < n_tiles ? totals[r] : 0u; header->seg_count[r] = n; header->seg_start[r] = acc; acc += n; }
		header->total = acc; header->n_emigrants = totals[n_tiles]; header->pad[0] = header->pad[1] = 0;
		// this rank's row of the exchange's all-gather: counts per destination, whether it has to route again (or has failed), how many records it can receive
		for (uint32_t r = 0; r < n_tiles; ++r) gather_row[r] = totals[r];
		gather_row[n_tiles] = host_status != ROUTE_OK ? host_status : ((acc > cap || totals[n_tiles] > emigrant_cap) ? ROUTE_REDO : ROUTE_OK);
		gather_row[n_tiles + 1] = cap_recv;
	}
}

__global__ void __launch_bounds__(TPB) k_route_write(DV d, TileRoute t, const uint32_t* block_offsets, const RouteHeader* header, sgp_ghost_record* out, uint32_t cap,
                                                     uint32_t* emigrant_ids, uint32_t emigrant_cap)
{
	__shared__ uint32_t wcnt[TPB / 64][SGP_MAX_TILES + 1];
	const uint32_t i = blockIdx.x * TPB + threadIdx.x;
	bool emig; uint32_t f = 0;
	const unsigned long long mask = route_mask(d, i, t, emig, f);
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const unsigned long long below = (1ull << lane) - 1ull;
	// per wave and column: how many of this wave's lanes write to it
	for (uint32_t r = 0; r <= t.n_tiles; ++r) {
		const unsigned long long b = __ballot(r < t.n_tiles ? ((mask >> r) & 1ull) != 0ull : emig);
		if (lane == 0) wcnt[wv][r] = (uint32_t)__popcll(b);
	}
	__syncthreads();
	if (!mask && !emig) return;
	const uint32_t cols = t.n_tiles + 1;
	sgp_ghost_record r;
	bool built = false;
	for (uint32_t dst = 0; dst < t.n_tiles; ++dst) {
		const unsigned long long b = __ballot(((mask >> dst) & 1ull) != 0ull);      // (every lane that reached this point takes part: the loop bounds are wave-uniform)
		if (!((mask >> dst) & 1ull)) continue;
		uint32_t wbase = 0;
		for (int k = 0; k < wv; ++k) wbase += wcnt[k][dst];
		const uint32_t k = header->seg_start[dst] + block_offsets[(size_t)blockIdx.x * cols + dst] + wbase + (uint32_t)__popcll(b & below);
		if (k >= cap) continue;
		if (!built) {
			const float4 p = d.pose[POSE_F4 * (size_t)i], qq = d.pose[POSE_F4 * (size_t)i + 1], lv = d.vel[VEL_F4 * (size_t)i], av = d.vel[VEL_F4 * (size_t)i + 1], sh = d.pose[POSE_F4 * (size_t)i + 3];
			r.pos[0] = p.x; r.pos[1] = p.y; r.pos[2] = p.z;
			r.rot[0] = qq.x; r.rot[1] = qq.y; r.rot[2] = qq.z; r.rot[3] = qq.w;
			r.lin_vel[0] = lv.x; r.lin_vel[1] = lv.y; r.lin_vel[2] = lv.z;
			r.ang_vel[0] = av.x; r.ang_vel[1] = av.y; r.ang_vel[2] = av.z;
			r.shape_type = (int32_t)f_shape(f);
			r.shape[0] = sh.x; r.shape[1] = sh.y; r.shape[2] = sh.z; r.shape[3] = 0.0f;
			r.mass = d.torque[i].w; r.friction = sh.w; r.restitution = d.pose[POSE_F4 * (size_t)i + 2].w;
			r.motion_type = emig ? (SGP_MOTION_DYNAMIC | SGP_GHOST_TAKE_OWNERSHIP) : f_motion(f);
			r.global_id = (uint64_t)i | ((uint64_t)t.my_rank << 40);
			fill_ghost_desc(d, i, f, r);
			built = true;
		}
		out[k] = r;
	}
	if (emig) {
		uint32_t wbase = 0;
		for (int k = 0; k < wv; ++k) wbase += wcnt[k][t.n_tiles];
		const unsigned long long b = __ballot(emig);
		const uint32_t k = block_offsets[(size_t)blockIdx.x * cols + t.n_tiles] + wbase + (uint32_t)__popcll(b & below);
		if (k < emigrant_cap) emigrant_ids[k] = i;
	}
}

__global__ void __launch_bounds__(TPB) k_ghost_refresh_records(DV d, const sgp_ghost_record* recs, const uint32_t* ids, uint32_t n)
{
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= n) return;
	const uint32_t i = ids[k];
	if (i == SGP_INVALID_ID) return;      // (a record that is no ghost here: an immigrant of the same exchange, a rejected one)
	uint32_t f = d.flags[i];
	if (!(f & BF_ALIVE)) return;
	const sgp_ghost_record& c = recs[k];
	d.pose[POSE_F4 * (size_t)i] = make_float4(c.pos[0], c.pos[1], c.pos[2], d.pose[POSE_F4 * (size_t)i].w);
	d.pose[POSE_F4 * (size_t)i + 1] = make_float4(c.rot[0], c.rot[1], c.rot[2], c.rot[3]);
	if (f_motion(f) != SGP_MOTION_STATIC) {
		d.vel[VEL_F4 * (size_t)i] = make_float4(c.lin_vel[0], c.lin_vel[1], c.lin_vel[2], d.vel[VEL_F4 * (size_t)i].w);
		d.vel[VEL_F4 * (size_t)i + 1] = make_float4(c.ang_vel[0], c.ang_vel[1], c.ang_vel[2], d.vel[VEL_F4 * (size_t)i + 1].w);
	}
	refresh_aabb(d, i, f);
	f = activate_body(d, i, f);
	d.flags[i] = f;
}

// What the host needs of a received record: 16 bytes to decide whether the ghost set changed (global id, ownership flag), and -- round 6 -- everything its
// bookkeeping needs to give a NEWCOMER a body slot without seeing the 128-byte record: the `info` word of the key (is the record valid, does it name a
// primitive shape, its layer / sensor / sleeping / drag bits, does a record that changes owner lie in this tile's region) and 16 more bytes (user data, bounding
// radius, volume) that only come to the host when the set changed.  The body itself is then created ON THE DEVICE from the record (k_create_from_records).
__global__ void __launch_bounds__(TPB) k_pack_ghost_keys(const sgp_ghost_record* recs, uint32_t n, uint4* out, uint4* aux, float3 lo, float3 hi)
{
	const uint32_t k = blockIdx.x * TPB + threadIdx.x;
	if (k >= n) return;
	const sgp_ghost_record& r = recs[k];
	const uint64_t g = r.global_id;
	const int type = r.shape_type;
	bool valid = isfinite(r.pos[0]) && isfinite(r.pos[1]) && isfinite(r.pos[2]) && fabsf(r.pos[0]) <= 1.0e9f && fabsf(r.pos[1]) <= 1.0e9f && fabsf(r.pos[2]) <= 1.0e9f;
	valid = valid && type >= 0 && type <= SGP_SHAPE_MESH;
	const int nparam = type == SGP_SHAPE_BOX ? 3 : (type == SGP_SHAPE_SPHERE ? 1 : ((type == SGP_SHAPE_HULL || type == SGP_SHAPE_MESH) ? 0 : 2));
	for (int i = 0; i < 3; ++i) if (i < nparam) { const float lim = (type == SGP_SHAPE_CAPSULE && i == 1) ? 0.0f : 0.5e-7f; if (!isfinite(r.shape[i]) || r.shape[i] < lim) valid = false; }
	uint32_t info = (valid ? GKEY_VALID : 0u) | (((uint32_t)type & 7u) << GKEY_SHAPE_SHIFT) | ((r.flags & 0x3Fu) << GKEY_FLAGS_SHIFT);
	if (r.pos[0] >= lo.x && r.pos[0] < hi.x && r.pos[1] >= lo.y && r.pos[1] < hi.y && r.pos[2] >= lo.z && r.pos[2] < hi.z) info |= GKEY_IN_REGION;
	out[k] = make_uint4((uint32_t)g, (uint32_t)(g >> 32), r.motion_type, info);
	// bounding radius and volume as the host computes them for a primitive (sgp_world_bodies.hip: bounding_radius, host_shape_volume -- the same expressions)
	float rad = 0.0f, vol = 0.0f;
	const float pi = 3.14159265358979323846f;
	if (type == SGP_SHAPE_SPHERE) { rad = r.shape[0]; vol = (4.0f / 3.0f) * pi * r.shape[0] * r.shape[0] * r.shape[0]; }
	else if (type == SGP_SHAPE_BOX) { rad = sqrtf(r.shape[0] * r.shape[0] + r.shape[1] * r.shape[1] + r.shape[2] * r.shape[2]); vol = 8.0f * r.shape[0] * r.shape[1] * r.shape[2]; }
	else if (type == SGP_SHAPE_CAPSULE) { rad = r.shape[0] + r.shape[1]; vol = pi * r.shape[0] * r.shape[0] * (2.0f * r.shape[1]) + (4.0f / 3.0f) * pi * r.shape[0] * r.shape[0] * r.shape[0]; }
	aux[k] = make_uint4((uint32_t)r.userdata, (uint32_t)(r.userdata >> 32), __float_as_uint(rad), __float_as_uint(vol));
}

// Bodies created straight from received records (round 6): the host chose the slots (in the order the CPU statement chooses them: parity) and mirrored the flags;
// entry = (record index, body slot, flags, 0).  A ghost is a kinematic copy (what make_ghost + CMD_CREATE + CMD_ACTIVATE did through a 160-byte command), a body
// that changes owner arrives as the dynamic body it was, with the mass properties add_one computes for a primitive (the same expressions: mass_properties).
struct CreateDefaults { float gravity_factor, lin_damp, ang_damp, pad; };
__global__ void __launch_bounds__(TPB) k_create_from_records(DV d, const sgp_ghost_record* recs, const uint4* list, uint32_t n, CreateDefaults def)
{
	const uint32_t e = blockIdx.x * TPB + threadIdx.x;
	if (e >= n) return;
	const uint4 en = list[e];
	const sgp_ghost_record& r = recs[en.x];
	const uint32_t i = en.y;
	uint32_t f = en.z | BF_CACHE_INVALID;
	const bool ghost = f & BF_GHOST;
	const float mass = fmaxf(0.001f, r.mass);
	float inv_mass = 0.0f, ii0 = 0.0f, ii1 = 0.0f, ii2 = 0.0f;
	if (f_motion(f) == SGP_MOTION_DYNAMIC) {
		const float* p = r.shape;
		float ix, iy, iz;
		if (r.shape_type == SGP_SHAPE_SPHERE) { const float q = 0.4f * mass * p[0] * p[0]; ix = iy = iz = q; }
		else if (r.shape_type == SGP_SHAPE_BOX) {
			const float sx = 2.0f * p[0], sy = 2.0f * p[1], sz = 2.0f * p[2];
			const float k = mass / 12.0f;
			ix = k * (sy * sy + sz * sz); iy = k * (sx * sx + sz * sz); iz = k * (sx * sx + sy * sy);
		} else {
			const float rr = p[0], H = 2.0f * p[1];
			const float vc = 3.14159265358979323846f * rr * rr * H;
			const float vs = (4.0f / 3.0f) * 3.14159265358979323846f * rr * rr * rr;
			const float mc = mass * vc / (vc + vs), ms = mass * vs / (vc + vs);
			iz = 0.5f * mc * rr * rr + 0.4f * ms * rr * rr;
			ix = mc * (3.0f * rr * rr + H * H) / 12.0f + ms * (0.4f * rr * rr + 0.25f * H * H + 0.375f * H * rr);
			iy = ix;
		}
		inv_mass = 1.0f / mass; ii0 = 1.0f / ix; ii1 = 1.0f / iy; ii2 = 1.0f / iz;
	}
	const float friction = r.friction < 0.0f ? 0.0f : (r.friction > 1.0f ? 1.0f : r.friction), restitution = r.restitution < 0.0f ? 0.0f : (r.restitution > 1.0f ? 1.0f : r.restitution);      // (clamp01 of add_one)
	d.pose[POSE_F4 * (size_t)i] = make_float4(r.pos[0], r.pos[1], r.pos[2], inv_mass);
	d.pose[POSE_F4 * (size_t)i + 1] = make_float4(r.rot[0], r.rot[1], r.rot[2], r.rot[3]);
	d.vel[VEL_F4 * (size_t)i] = make_float4(r.lin_vel[0], r.lin_vel[1], r.lin_vel[2], 0.0f);
	d.vel[VEL_F4 * (size_t)i + 1] = make_float4(r.ang_vel[0], r.ang_vel[1], r.ang_vel[2], 0.0f);
	d.dyn[i] = ghost ? make_float4(def.lin_damp, def.ang_damp, def.gravity_factor, inv_mass) : make_float4(r.linear_damping, r.angular_damping, r.gravity_factor, inv_mass);
	d.force[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
	d.torque[i] = make_float4(0.0f, 0.0f, 0.0f, mass);
	d.pose[POSE_F4 * (size_t)i + 2] = make_float4(ii0, ii1, ii2, restitution);
	d.pose[POSE_F4 * (size_t)i + 3] = make_float4(r.shape[0], r.shape[1], r.shape[2], friction);
	d.submerged[i] = 0.0f;
	d.userdata[i] = r.userdata;
	label_new_body(d, i);
	refresh_aabb(d, i, f);
	reset_sleep(d, i, f_shape(f), d.pose[POSE_F4 * (size_t)i + 3], V3(d.pose[POSE_F4 * (size_t)i]), Q4(d.pose[POSE_F4 * (size_t)i + 1]));
	f = activate_body(d, i, f);      // (both kinds arrive awake: d.activate = 1 in make_ghost and in the take-over)
	d.flags[i] = f;
}
void launch_route_export(const DV& d, uint32_t nb, const TileRoute& t, uint32_t* block_counts, uint32_t* block_offsets, RouteHeader* header,
                         sgp_ghost_record* out, uint32_t cap, uint32_t* emigrant_ids, uint32_t emigrant_cap, uint32_t* gather_row, uint32_t cap_recv, uint32_t host_status, hipStream_t s)
{
	const uint32_t blocks = blocks_for(nb);
	hipLaunchKernelGGL(k_route_count, dim3(blocks), dim3(TPB), 0, s, d, t, block_counts);
	hipLaunchKernelGGL(k_route_scan, dim3(1), dim3(1024), 0, s, (const uint32_t*)block_counts, block_offsets, blocks, t.n_tiles, header, cap, emigrant_cap, gather_row, cap_recv, host_status);
	hipLaunchKernelGGL(k_route_write, dim3(blocks), dim3(TPB), 0, s, d, t, (const uint32_t*)block_offsets, (const RouteHeader*)header, out, cap, emigrant_ids, emigrant_cap);
}
void launch_tiles_hist(const DV& d, uint32_t nb, const TilePlanes& tp, int level, uint32_t* out, hipStream_t s) { hipLaunchKernelGGL(k_tiles_hist, dim3(blocks_for(nb)), dim3(TPB), 0, s, d, tp, level, out); }
void launch_pack_ghost_keys(const sgp_ghost_record* recs, uint32_t n, void* out, void* aux, const float* lo, const float* hi, hipStream_t s)
{
	if (n) hipLaunchKernelGGL(k_pack_ghost_keys, dim3(blocks_for(n)), dim3(TPB), 0, s, recs, n, (uint4*)out, (uint4*)aux, make_float3(lo[0], lo[1], lo[2]), make_float3(hi[0], hi[1], hi[2]));
}
void launch_create_from_records(const DV& d, const sgp_ghost_record* recs, const void* list, uint32_t n, float gravity_factor, float lin_damp, float ang_damp, hipStream_t s)
{
	CreateDefaults def = { gravity_factor, lin_damp, ang_damp, 0.0f };
	if (n) hipLaunchKernelGGL(k_create_from_records, dim3(blocks_for(n)), dim3(TPB), 0, s, d, recs, (const uint4*)list, n, def);
}
void launch_ghost_refresh_records(const DV& d, const sgp_ghost_record* recs, const uint32_t* ids, uint32_t n, hipStream_t s)
{
	if (n) hipLaunchKernelGGL(k_ghost_refresh_records, dim3(blocks_for(n)), dim3(TPB), 0, s, d, recs, ids, n);
}
void launch_export_boundary(const DV& d, uint32_t nb, float3 lo, float3 hi, float margin, sgp_ghost_record* out, uint32_t cap, uint32_t* count, hipStream_t s)
{
	const uint32_t blocks = blocks_for(nb);
	hipLaunchKernelGGL(k_export_count, dim3(blocks), dim3(TPB), 0, s, d, lo, hi, margin);
	hipLaunchKernelGGL(k_export_boundary, dim3(blocks), dim3(TPB), 0, s, d, lo, hi, margin, out, cap, count);
}
