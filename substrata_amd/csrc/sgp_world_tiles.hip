// sgp_world_tiles.hip -- spatial tiles: ghost export / import, the per-step exchange over RCCL (or device copies within one process), re-tiling.
#include "sgp_world_internal.h"

// ---------------------------------------------------------------------------------------------------------------
// multi-GPU tiles

SGP_API int sgp_world_export_boundary(sgp_world* w, const float lo[3], const float hi[3], float margin, sgp_ghost_record* out, uint32_t cap, uint32_t* n_out)
{
	if (!w || !n_out) return fail(SGP_ERR_INVALID, "sgp_world_export_boundary: NULL");
	hipSetDevice(w->device);
	static const bool timing = getenv("SGP_TIMING") != nullptr;
	const auto t0 = std::chrono::steady_clock::now();
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	const uint32_t lim = std::min(cap, w->dv.cap_bodies);
	{ int r = ensure_stage(w, sizeof(sgp_ghost_record) * std::max(lim, 1u)); if (r != SGP_OK) return r; }
	launch_export_boundary(w->dv, w->high, make_float3(lo[0], lo[1], lo[2]), make_float3(hi[0], hi[1], hi[2]), margin,
	                       (sgp_ghost_record*)w->stage_dev, lim, &w->dv.ctr->n_export, w->stream);
	// one sync in the common case: the counters and as many records as the previous call produced (+ 25 %) come back together
	uint32_t guess = out ? std::min(lim, w->last_export + w->last_export / 4 + 64u) : 0u;
	if (guess) HIP_TRY(hipMemcpyAsync(w->stage_host, w->stage_dev, sizeof(sgp_ghost_record) * guess, hipMemcpyDeviceToHost, w->stream));
	{ int r = read_counters(w); if (r != SGP_OK) return r; }
	const uint32_t n = w->h_ctr->n_export, m = std::min(n, lim);
	w->last_export = n;
	const auto t1 = std::chrono::steady_clock::now();
	if (m && out) {
		if (m > guess) {
			HIP_TRY(hipMemcpyAsync((char*)w->stage_host + sizeof(sgp_ghost_record) * guess, (char*)w->stage_dev + sizeof(sgp_ghost_record) * guess,
			                       sizeof(sgp_ghost_record) * (m - guess), hipMemcpyDeviceToHost, w->stream));
			HIP_TRY(hipStreamSynchronize(w->stream));
		}
		// the kernel wrote the records in ascending body id (k_export_count + k_export_boundary): the order of the exchange is deterministic
		memcpy(out, w->stage_host, sizeof(sgp_ghost_record) * m);
	}
	if (timing) { const auto t2 = std::chrono::steady_clock::now(); fprintf(stderr, "[sgp timing] export_boundary: device part %.1f us, sort + copy of %u records %.1f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count(), m, std::chrono::duration<double, std::micro>(t2 - t1).count()); }
	*n_out = n;
	return SGP_OK;
}

// Where the poses of an import come from when the records are already on the device (sgp_tiles_*): the device copy of the records and a
// device array for the local body id of every record (grown here); surviving ghosts are then refreshed by ONE kernel, not by commands.
struct GhostDeviceSource { const sgp_ghost_record* d_recs; uint32_t** d_ids; uint32_t* cap_ids; uint64_t* ids_version; };

// (skip: per RECORD, 1 = not a ghost here (an immigrant of the same exchange); the id array stays aligned with the records, such entries hold "no body")
static int upload_ghost_ids(sgp_world* w, const GhostDeviceSource* dev, const uint8_t* skip = nullptr, uint32_t n_records = 0)
{
	const uint32_t n = skip ? n_records : (uint32_t)w->ghost_seq.size();
	std::vector<uint32_t> ids(n);
	if (skip) { size_t g = 0; for (uint32_t k = 0; k < n; ++k) ids[k] = skip[k] ? SGP_INVALID_ID : w->ghost_seq[g++].second; }
	else for (uint32_t k = 0; k < n; ++k) ids[k] = w->ghost_seq[k].second;
	if (n > *dev->cap_ids) {
		if (*dev->d_ids) { HIP_TRY(hipStreamSynchronize(w->stream)); hipFree(*dev->d_ids); }
		*dev->cap_ids = n + n / 2 + 1024;
		HIP_TRY(hipMalloc((void**)dev->d_ids, sizeof(uint32_t) * (size_t)*dev->cap_ids));
	}
	if (n) { HIP_TRY(hipMemcpyAsync(*dev->d_ids, ids.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, w->stream)); HIP_TRY(hipStreamSynchronize(w->stream)); }      // (`ids` is pageable memory going out of scope)
	*dev->ids_version = skip ? ~0ull : w->ghost_seq_version;      // (an array with holes serves this import only: the next one that finds the set unchanged uploads the plain list)
	return SGP_OK;
}

static int make_ghost(sgp_world* w, const sgp_ghost_record& r, uint32_t* id_out)
{
	sgp_body_desc d; sgp_default_body_desc(&d);
	memcpy(d.pos, r.pos, 12); memcpy(d.rot, r.rot, 16); memcpy(d.lin_vel, r.lin_vel, 12); memcpy(d.ang_vel, r.ang_vel, 12);
	d.shape_type = r.shape_type; memcpy(d.shape, r.shape, 16);
	d.motion_type = SGP_MOTION_KINEMATIC;      // velocity driven, infinite mass for this tile's solve
	// layer and sensor flag of the original: a sensor or a non-collidable body near the border must not become a solid obstacle next door
	d.layer = (int32_t)(r.flags & SGP_GHOST_FLAG_LAYER_MASK);
	if (d.layer == SGP_LAYER_NON_MOVING) d.layer = SGP_LAYER_MOVING;                               // (a kinematic ghost lives on a moving layer)
	if (d.layer == SGP_LAYER_NON_MOVING_NON_COLLIDABLE) d.layer = SGP_LAYER_MOVING_NON_COLLIDABLE;
	d.is_sensor = (r.flags & SGP_GHOST_FLAG_SENSOR) ? 1 : 0;
	d.mass = r.mass; d.friction = r.friction; d.restitution = r.restitution;
	d.activate = 1; d.userdata = r.userdata;       // a ray or an event that meets the ghost names the object, like its owner would
	*id_out = SGP_INVALID_ID;
	return add_one(w, &d, id_out, true);
}

static int import_ghosts_impl(sgp_world* w, const sgp_ghost_record* in_all, uint32_t n_all, const GhostDeviceSource* dev, const uint8_t* skip = nullptr, const uint64_t* gids = nullptr, uint32_t gid_stride = 0, const uint4* keys = nullptr, const uint4* aux = nullptr);
// (skip[k] = 1: record k is no ghost -- an immigrant riding in the same exchange -- and is left out; with a device source the records stay where they are and
// the id array has a hole there)
struct GhostView {      // the ghost records of an import: all of them, or those a mask lets through (by index: nothing is copied)
	const sgp_ghost_record* base; const uint32_t* idx; uint32_t n;
	const uint64_t* gids; uint32_t gid_stride;      // the records' global ids packed (16-byte keys of the exchange), or NULL: the diff then walks 16 bytes per record, not 128
	const uint4* keys; const uint4* aux;            // round 6: the packed keys + user data / radius / volume of every record, or NULL.  With them a newcomer gets its slot from
	                                                // the key alone and the device creates the body from the record (base may then be NULL: no record came to the host)
	uint32_t rec(uint32_t k) const { return idx ? idx[k] : k; }
	const sgp_ghost_record& operator[](uint32_t k) const { return idx ? base[idx[k]] : base[k]; }
	uint64_t gid(uint32_t k) const { const uint32_t r = idx ? idx[k] : k; return gids ? gids[(size_t)r * gid_stride] : base[r].global_id; }
};
static int import_ghosts_view(sgp_world* w, const GhostView& in, uint32_t n, const GhostDeviceSource* dev, const uint8_t* skip, uint32_t n_all);
// A newcomer to the ghost set: through a create command built from its record, or -- when only its key came to the host -- a slot from the key and the body
// created on the device (w->rec_creates; the caller launches k_create_from_records once the import's commands are flushed)
static int make_ghost_at(sgp_world* w, const GhostView& in, uint32_t i, uint32_t* id_out)
{
	if (!in.keys) return make_ghost(w, in[i], id_out);
	*id_out = SGP_INVALID_ID;
	const uint32_t r = in.rec(i);
	const uint4 key = in.keys[r], aux = in.aux[r];
	if (!(key.w & GKEY_VALID)) return SGP_ERR_REJECTED;
	const uint32_t rflags = (key.w >> GKEY_FLAGS_SHIFT) & 0x3Fu, type = (key.w >> GKEY_SHAPE_SHIFT) & 7u;
	uint32_t layer = rflags & SGP_GHOST_FLAG_LAYER_MASK;      // (as make_ghost: a kinematic ghost lives on a moving layer)
	if (layer == SGP_LAYER_NON_MOVING) layer = SGP_LAYER_MOVING;
	if (layer == SGP_LAYER_NON_MOVING_NON_COLLIDABLE) layer = SGP_LAYER_MOVING_NON_COLLIDABLE;
	uint32_t f = BF_ALIVE | SGP_MOTION_KINEMATIC | (layer << BF_LAYER_SHIFT) | (type << BF_SHAPE_SHIFT) | BF_ALLOW_SLEEP | BF_GHOST;      // (allow_sleeping: the default description's)
	if (rflags & SGP_GHOST_FLAG_SENSOR) f |= BF_SENSOR;
	float rad, vol; memcpy(&rad, &aux.z, 4); memcpy(&vol, &aux.w, 4);
	uint32_t id;
	const int rc = book_record_body(w, &f, (uint64_t)aux.x | ((uint64_t)aux.y << 32), rad, vol, true, &id);
	if (rc != SGP_OK) return rc;
	w->rec_creates.push_back(make_uint4(r, id, f, 0u));
	*id_out = id;
	return SGP_OK;
}

static int import_ghosts_impl(sgp_world* w, const sgp_ghost_record* in_all, uint32_t n_all, const GhostDeviceSource* dev, const uint8_t* skip, const uint64_t* gids, uint32_t gid_stride, const uint4* keys, const uint4* aux)
{
	if (!skip) { GhostView v = { in_all, nullptr, n_all, gids, gid_stride, keys, aux }; return import_ghosts_view(w, v, n_all, dev, nullptr, n_all); }
	std::vector<uint32_t> idx; idx.reserve(n_all);
	for (uint32_t k = 0; k < n_all; ++k) if (!skip[k]) idx.push_back(k);
	GhostView v = { in_all, idx.data(), (uint32_t)idx.size(), gids, gid_stride, keys, aux };
	return import_ghosts_view(w, v, v.n, dev, skip, n_all);
}
static int import_ghosts_view(sgp_world* w, const GhostView& in, uint32_t n, const GhostDeviceSource* dev, const uint8_t* skip, uint32_t n_all)
{
	// ghosts keep their local id while they stay in the set, so the contact cache (keyed by body ids) keeps warm-starting.
	// ghost_seq: (global id, local id) of the previous import, in its order
	// 1. the usual case: the same ghosts as in the previous import, in the same order -- no bookkeeping, just refresh their poses
	if (n == w->ghost_seq.size() && n > 0) {
		bool same = true;
		for (uint32_t k = 0; k < n && same; ++k) same = in.gid(k) == w->ghost_seq[k].first && live(w, w->ghost_seq[k].second);
		if (same) {
			if (dev) {
				{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
				if (skip || *dev->ids_version != w->ghost_seq_version) { int r = upload_ghost_ids(w, dev, skip, n_all); if (r != SGP_OK) return r; }      // (the set was last changed by an import that did not come through here; or the records hold immigrants between the ghosts)
				launch_ghost_refresh_records(w->dv, dev->d_recs, *dev->d_ids, n_all, w->stream);
				w->grid_valid = false; w->dirty_since_step = true;
				return SGP_OK;
			}
			// a later import before the next flush supersedes an earlier one: the refresh list holds one record per ghost
			w->ghost_refresh.resize(n);
			for (uint32_t k = 0; k < n; ++k) {
				GhostRefresh& c = w->ghost_refresh[k];
				c.id = w->ghost_seq[k].second;
				memcpy(c.pos, in[k].pos, 12); memcpy(c.rot, in[k].rot, 16); memcpy(c.linv, in[k].lin_vel, 12); memcpy(c.angv, in[k].ang_vel, 12);
			}
			return SGP_OK;
		}
	}
	w->ghost_refresh.clear();
	w->cmds.reserve(w->cmds.size() + n);
	std::vector<std::pair<uint64_t, uint32_t>> seq(n, std::pair<uint64_t, uint32_t>(0, SGP_INVALID_ID));
	std::vector<uint32_t> gone;
	auto refresh_cmd = [&](uint32_t id, uint32_t rec_k) {
		if (dev) return;                       // refreshed from the device copy of the records below
		const sgp_ghost_record& r = in[rec_k];
		BodyCmd c = blank_cmd(id, CMD_SET_POS | CMD_SET_ROT | CMD_SET_VEL | CMD_ACTIVATE);
		memcpy(c.pos, r.pos, 12); memcpy(c.rot, r.rot, 16); memcpy(c.linv, r.lin_vel, 12); memcpy(c.angv, r.ang_vel, 12);
		w->cmds.push_back(c);
	};
	// 2. both the old and the new sequence ascending in global id (what every exchange produces: by source rank, then by the source's body
	//    id): a two-pointer diff finds who stayed, who is new and who left, without hashing.  New ghosts take their slots in record order,
	//    leavers are removed afterwards in ascending id order -- the same allocation order as the general path below.
	bool ascending = true;
	for (uint32_t k = 1; k < n && ascending; ++k) ascending = in.gid(k - 1) < in.gid(k);
	for (size_t k = 1; k < w->ghost_seq.size() && ascending; ++k) ascending = w->ghost_seq[k - 1].first < w->ghost_seq[k].first;
	if (ascending) {
		const std::vector<std::pair<uint64_t, uint32_t>>& old = w->ghost_seq;
		size_t i = 0, j = 0;
		while (i < n || j < old.size()) {
			if (j == old.size() || (i < n && in.gid(i) < old[j].first)) {
				uint32_t id; const int r = make_ghost_at(w, in, (uint32_t)i, &id);
				if (r != SGP_OK && r != SGP_ERR_REJECTED) return r;
				seq[i] = std::make_pair(in.gid(i), r == SGP_OK ? id : SGP_INVALID_ID); ++i;
			} else if (i == n || old[j].first < in.gid(i)) {
				if (old[j].second != SGP_INVALID_ID && live(w, old[j].second)) gone.push_back(old[j].second);
				++j;
			} else {
				uint32_t id = old[j].second;
				if (id != SGP_INVALID_ID && live(w, id)) refresh_cmd(id, (uint32_t)i);
				else { const int r = make_ghost_at(w, in, (uint32_t)i, &id); if (r != SGP_OK && r != SGP_ERR_REJECTED) return r; if (r != SGP_OK) id = SGP_INVALID_ID; }
				seq[i] = std::make_pair(in.gid(i), id); ++i; ++j;
			}
		}
		w->ghost_map_stale = true;
	} else {
		// 3. general: hash map global id -> (generation of the last import that contained it, local id)
		if (w->ghost_map_stale) {
			w->ghost_map.clear();
			for (const auto& e : w->ghost_seq) if (e.second != SGP_INVALID_ID) w->ghost_map[e.first] = ((uint64_t)w->ghost_gen << 32) | e.second;
			w->ghost_map_stale = false;
		}
		const uint32_t gen = ++w->ghost_gen;
		for (uint32_t k = 0; k < n; ++k) {
			seq[k].first = in.gid(k);
			auto it = w->ghost_map.find(in.gid(k));
			if (it != w->ghost_map.end() && live(w, (uint32_t)it->second)) {
				const uint32_t id = (uint32_t)it->second;
				refresh_cmd(id, k);
				it->second = ((uint64_t)gen << 32) | id;
				seq[k].second = id;
				continue;
			}
			uint32_t id; const int r = make_ghost_at(w, in, k, &id);
			if (r == SGP_OK) { w->ghost_map[in.gid(k)] = ((uint64_t)gen << 32) | id; seq[k].second = id; }
			else if (r != SGP_ERR_REJECTED) return r;
		}
		// whatever was not refreshed by this import left the ghost set
		for (auto it = w->ghost_map.begin(); it != w->ghost_map.end();) {
			if ((uint32_t)(it->second >> 32) != gen) { if (live(w, (uint32_t)it->second)) gone.push_back((uint32_t)it->second); it = w->ghost_map.erase(it); }
			else ++it;
		}
	}
	// leavers: removed in ascending id order (deterministic free-list order)
	std::sort(gone.begin(), gone.end());
	for (uint32_t id : gone) sgp_body_remove(w, id);
	w->ghost_seq.swap(seq);
	w->ghost_seq_version++;
	if (dev && n) {
		// new ghosts and removals reach the device first, then ONE kernel gives every ghost of the set its pose from the received records (a rejected
		// record -- non-finite pose ... -- has "no body" in the id array, like an immigrant's: the kernel passes over it)
		{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
		{ int r = upload_ghost_ids(w, dev, skip, n_all); if (r != SGP_OK) return r; }
		launch_ghost_refresh_records(w->dv, dev->d_recs, *dev->d_ids, n_all, w->stream);
		w->grid_valid = false; w->dirty_since_step = true;
	}
	return SGP_OK;
}

SGP_API int sgp_world_import_ghosts(sgp_world* w, const sgp_ghost_record* in, uint32_t n)
{
	if (!w || (!in && n)) return fail(SGP_ERR_INVALID, "sgp_world_import_ghosts: NULL");
	hipSetDevice(w->device);
	return import_ghosts_impl(w, in, n, nullptr);
}

// ---- host-side routing of exported records (tiles.py) -------------------------------------------------------------------------

static inline bool in_box(const float* p, const float* lo, const float* hi, float pad)
{
	return p[0] >= lo[0] - pad && p[0] < hi[0] + pad && p[1] >= lo[1] - pad && p[1] < hi[1] + pad && p[2] >= lo[2] - pad && p[2] < hi[2] + pad;
}

SGP_API int sgp_tiles_route(const sgp_ghost_record* recs, uint32_t n, uint32_t my_rank, const float* boxes, uint32_t n_tiles, float pad,
                            sgp_ghost_record* send_out, uint32_t cap, uint32_t* send_counts,
                            uint32_t* emigrant_ids, uint32_t emigrant_cap, uint32_t* n_emigrants)
{
	if ((!recs && n) || !boxes || !send_counts || !n_emigrants || my_rank >= n_tiles) return fail(SGP_ERR_INVALID, "sgp_tiles_route: bad arguments");
	const float* mylo = boxes + 6 * (size_t)my_rank; const float* myhi = mylo + 3;
	// flags per record: emigrant?
	std::vector<uint8_t> emig(n, 0);
	uint32_t ne = 0;
	for (uint32_t k = 0; k < n; ++k) {
		bool taker = false;      // (same rule as route_mask on the device: without a tile that contains the centre the body stays where it is)
		for (uint32_t r = 0; r < n_tiles && !taker; ++r) if (r != my_rank) taker = in_box(recs[k].pos, boxes + 6 * (size_t)r, boxes + 6 * (size_t)r + 3, 0.0f);
		if (n_tiles > 1 && taker && (recs[k].motion_type & 0xFFu) == SGP_MOTION_DYNAMIC && !(recs[k].flags & SGP_GHOST_FLAG_CHASSIS) && !in_box(recs[k].pos, mylo, myhi, 0.0f)) {
			emig[k] = 1;
			if (ne < emigrant_cap && emigrant_ids) emigrant_ids[ne] = (uint32_t)(recs[k].global_id & 0xFFFFFFFFull);
			++ne;
		}
	}
	*n_emigrants = ne;
	if (ne > emigrant_cap) return fail(SGP_ERR_CAPACITY, "sgp_tiles_route: emigrant list too small");
	uint32_t w = 0;
	for (uint32_t r = 0; r < n_tiles; ++r) {
		send_counts[r] = 0;
		if (r == my_rank) continue;
		const float* lo = boxes + 6 * (size_t)r; const float* hi = lo + 3;
		for (uint32_t k = 0; k < n; ++k) {
			if (!in_box(recs[k].pos, lo, hi, pad)) continue;
			if (w >= cap || !send_out) return fail(SGP_ERR_CAPACITY, "sgp_tiles_route: send buffer too small");
			sgp_ghost_record o = recs[k];
			o.global_id |= (uint64_t)my_rank << 40;
			if (emig[k]) o.motion_type = SGP_MOTION_DYNAMIC | SGP_GHOST_TAKE_OWNERSHIP;
			send_out[w++] = o;
			++send_counts[r];
		}
	}
	return SGP_OK;
}

SGP_API int sgp_tiles_split(const sgp_ghost_record* in, uint32_t n, const float lo[3], const float hi[3],
                            sgp_ghost_record* ghosts_out, uint32_t* n_ghosts, sgp_ghost_record* immigrants_out, uint32_t* n_immigrants)
{
	if ((!in && n) || !lo || !hi || !n_ghosts || !n_immigrants || (n && (!ghosts_out || !immigrants_out))) return fail(SGP_ERR_INVALID, "sgp_tiles_split: bad arguments");
	uint32_t g = 0, m = 0;
	for (uint32_t k = 0; k < n; ++k) {
		if (in[k].motion_type & SGP_GHOST_TAKE_OWNERSHIP) { if (in_box(in[k].pos, lo, hi, 0.0f)) immigrants_out[m++] = in[k]; }
		else ghosts_out[g++] = in[k];
	}
	*n_ghosts = g; *n_immigrants = m;
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------

SGP_API int sgp_world_device_array(sgp_world* w, int which, void** dev_ptr_out, uint32_t* count_out)
{
	if (!w || !dev_ptr_out) return fail(SGP_ERR_INVALID, "sgp_world_device_array: NULL");
	void* p = nullptr;
	switch (which) { case 0: p = w->dv.pose; break; case 1: p = w->dv.vel; break;
	default: return fail(SGP_ERR_INVALID, "sgp_world_device_array: bad index"); }
	*dev_ptr_out = p;
	if (count_out) *count_out = w->high;
	return SGP_OK;
}

SGP_API int sgp_world_stream(sgp_world* w, void** stream_out)
{
	if (!w || !stream_out) return fail(SGP_ERR_INVALID, "sgp_world_stream: NULL");
	ray_server_stop(w);      // (whoever asks for the stream is about to put work on it)
	*stream_out = (void*)w->stream;
	return SGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// sgp_tiles_*: the per-step ghost exchange of the spatial tiles below the C ABI.
//
//   export + ROUTING on the device (k_route_count / scan / write: one record per (body, destination), segmented by destination)
//   -> per-destination counts all-gathered over RCCL (ncclAllGather, device buffers) and read back with ONE small copy (header +
//      counts matrix + emigrant ids)
//   -> the records travel device to device: grouped ncclSend / ncclRecv over xGMI straight out of the send buffer's segments
//      (or, for several tiles driven by one process, plain device-to-device copies)
//   -> the receiving tile refreshes its ghosts.  While the set of ghosts is what it was the step before (the steady state), a kernel
//      applies the poses straight from the received records; only when the set changed (or bodies immigrate) do the records come to
//      the host, which owns the body slots.
// RCCL is bound at run time (dlopen): libsgp.so carries no link-time dependency on it, a single-GPU user never loads it.
#include <dlfcn.h>

namespace {
typedef struct { char internal[128]; } sgp_nccl_unique_id;
typedef void* sgp_nccl_comm;
enum { SGP_NCCL_UINT8 = 1, SGP_NCCL_UINT32 = 3 };          // ncclDataType_t (rccl.h): ncclUint8 = 1, ncclUint32 = 3
struct RcclApi {
	void* lib = nullptr; bool tried = false;
	int (*GetUniqueId)(sgp_nccl_unique_id*) = nullptr;
	int (*CommInitRank)(sgp_nccl_comm*, int, sgp_nccl_unique_id, int) = nullptr;
	int (*CommDestroy)(sgp_nccl_comm) = nullptr;
	int (*AllGather)(const void*, void*, size_t, int, sgp_nccl_comm, hipStream_t) = nullptr;
	int (*Send)(const void*, size_t, int, int, sgp_nccl_comm, hipStream_t) = nullptr;
	int (*Recv)(void*, size_t, int, int, sgp_nccl_comm, hipStream_t) = nullptr;
	int (*GroupStart)() = nullptr;
	int (*GroupEnd)() = nullptr;
	const char* (*GetErrorString)(int) = nullptr;
	int (*CommCount)(sgp_nccl_comm, int*) = nullptr;
};
RcclApi g_rccl;

// The prototypes above are hand-declared so that libsgp.so builds and loads without RCCL.  Where <rccl/rccl.h> is installed at build time
// they are checked against it: same number of parameters, every parameter and the result of the same size and kind (pointer / integer or
// enum / class passed by value), and the enumerators this file passes as integers.  A mismatch is a compile error, not a first-run surprise.
#if __has_include(<rccl/rccl.h>)
}
#include <rccl/rccl.h>
#include <type_traits>
namespace {
template <class A, class B> constexpr bool sgp_abi_same_arg()
{
	return sizeof(A) == sizeof(B) && std::is_pointer<A>::value == std::is_pointer<B>::value && std::is_class<A>::value == std::is_class<B>::value &&
	       (std::is_integral<A>::value || std::is_enum<A>::value) == (std::is_integral<B>::value || std::is_enum<B>::value);
}
template <class F, class G> struct sgp_abi_same : std::false_type {};
template <class R, class... A, class S, class... B> struct sgp_abi_same<R (*)(A...), S (*)(B...)>
{
	template <bool same_arity, class Dummy = void> struct args { static constexpr bool value = false; };
	template <class Dummy> struct args<true, Dummy> { static constexpr bool value = (sgp_abi_same_arg<A, B>() && ... && true); };
	static constexpr bool value = sgp_abi_same_arg<R, S>() && args<sizeof...(A) == sizeof...(B)>::value;
};
#define SGP_CHECK_RCCL(member, fn) static_assert(sgp_abi_same<decltype(RcclApi::member), decltype(&fn)>::value, "hand-declared prototype of " #fn " does not match <rccl/rccl.h>")
SGP_CHECK_RCCL(GetUniqueId, ncclGetUniqueId);
SGP_CHECK_RCCL(CommInitRank, ncclCommInitRank);
SGP_CHECK_RCCL(CommDestroy, ncclCommDestroy);
SGP_CHECK_RCCL(AllGather, ncclAllGather);
SGP_CHECK_RCCL(Send, ncclSend);
SGP_CHECK_RCCL(Recv, ncclRecv);
SGP_CHECK_RCCL(GroupStart, ncclGroupStart);
SGP_CHECK_RCCL(GroupEnd, ncclGroupEnd);
SGP_CHECK_RCCL(GetErrorString, ncclGetErrorString);
SGP_CHECK_RCCL(CommCount, ncclCommCount);
static_assert(sizeof(ncclUniqueId) == sizeof(sgp_nccl_unique_id) && NCCL_UNIQUE_ID_BYTES == SGP_TILES_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
static_assert((int)ncclUint8 == SGP_NCCL_UINT8 && (int)ncclUint32 == SGP_NCCL_UINT32 && (int)ncclSuccess == 0, "ncclDataType_t / ncclResult_t values");
#endif

bool rccl_load()
{
	if (g_rccl.tried) return g_rccl.lib != nullptr;
	g_rccl.tried = true;
	// the copy this process already has (PyTorch brings its own), else the system's
	// SGP_RCCL_LIBRARY: bind this build of the collective library instead (a site's own RCCL; the test-only stand-in of tests/rccl_standin, which
	// lets several processes on ONE GPU run sgp_tiles_exchange itself)
	if (const char* explicit_lib = getenv("SGP_RCCL_LIBRARY")) { if (*explicit_lib) g_rccl.lib = dlopen(explicit_lib, RTLD_NOW | RTLD_GLOBAL); if (*explicit_lib && !g_rccl.lib) return false; }
	const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
	if (!g_rccl.lib) for (const char* n : names) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (g_rccl.lib) break; }
	if (!g_rccl.lib) for (const char* n : names) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.lib) break; }
	if (!g_rccl.lib) return false;
	bool ok = true;
	auto sym = [&](const char* name) { void* p = dlsym(g_rccl.lib, name); if (!p) ok = false; return p; };
	g_rccl.GetUniqueId = (int (*)(sgp_nccl_unique_id*))sym("ncclGetUniqueId");
	g_rccl.CommInitRank = (int (*)(sgp_nccl_comm*, int, sgp_nccl_unique_id, int))sym("ncclCommInitRank");
	g_rccl.CommDestroy = (int (*)(sgp_nccl_comm))sym("ncclCommDestroy");
	g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, sgp_nccl_comm, hipStream_t))sym("ncclAllGather");
	g_rccl.Send = (int (*)(const void*, size_t, int, int, sgp_nccl_comm, hipStream_t))sym("ncclSend");
	g_rccl.Recv = (int (*)(void*, size_t, int, int, sgp_nccl_comm, hipStream_t))sym("ncclRecv");
	g_rccl.GroupStart = (int (*)())sym("ncclGroupStart");
	g_rccl.GroupEnd = (int (*)())sym("ncclGroupEnd");
	g_rccl.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
	if (!ok) { g_rccl.lib = nullptr; return false; }
	g_rccl.CommCount = (int (*)(sgp_nccl_comm, int*))dlsym(g_rccl.lib, "ncclCommCount");      // optional: only reported in sgp_tiles_stats
	return true;
}
int rccl_fail(const char* what, int rc)
{
	g_last_error = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
	return SGP_ERR_HIP;
}
#define RCCL_TRY(call, what) do { const int rc_ = (call); if (rc_ != 0) return rccl_fail(what, rc_); } while (0)
}

#define SGP_TILES_EMIG_INLINE 512          // emigrant ids that come back with the header copy

struct sgp_tiles {
	sgp_world* w = nullptr;
	uint32_t rank = 0, n_tiles = 1;
	TileRoute route;
	sgp_nccl_comm comm = nullptr;
	// device
	uint32_t* d_block_counts = nullptr; uint32_t* d_block_offsets = nullptr; uint32_t cap_blocks = 0;
	char* d_ctl = nullptr;                 // control block, see tiles_ctl_bytes
	uint32_t host_status = ROUTE_OK;       // ROUTE_FAILED once a growth of this rank's buffers failed: told to every rank by the next all-gather
	int test_fail_growth = 0;              // SGP_TILES_TEST_FAIL_RANK names this rank: its next buffer growth fails (tests of the failure path)
	sgp_ghost_record* d_send = nullptr; uint32_t cap_send = 0;
	sgp_ghost_record* d_recv = nullptr; uint32_t cap_recv = 0;
	uint32_t* d_emig = nullptr; uint32_t cap_emig = 0;
	uint32_t* d_seq_ids = nullptr; uint32_t cap_seq = 0; uint64_t ids_version = 0;      // local body id of ghost k of the current ghost set (device copy, for the refresh kernel)
	// host (pinned)
	char* h_ctl = nullptr; sgp_ghost_record* h_recv = nullptr; uint32_t cap_h_recv = 0;
	uint4* d_keys = nullptr; uint32_t cap_keys = 0; void* h_keys = nullptr; uint32_t cap_h_keys = 0;      // (global id, ownership flag, kind of record) of the received records
	uint4* d_aux = nullptr; uint32_t cap_aux = 0; void* h_aux = nullptr;                                  // (user data, bounding radius, volume): to the host only when the set changed
	uint4* d_create = nullptr; uint32_t cap_create = 0; uint4* h_create = nullptr; uint32_t cap_h_create = 0;      // bodies to create on the device from received records
	bool host_records_only = false;        // SGP_TILES_HOST_RECORDS=1: every changed set takes the records to the host (the round-5 path; A/B and tests)
	// last exchange
	std::vector<uint32_t> recv_counts, recv_offsets;
	std::vector<uint64_t> seq_gids;        // global ids of the ghosts of the previous import, in order
	bool seq_valid = false;
	sgp_tiles_stats stats;
	std::vector<sgp_migration> migrations;
	// re-tiling (sgp_tiles_rebalance): this tile's histogram, everybody's (RCCL all-gather), the pinned host copy
	uint32_t* d_hist = nullptr; uint32_t* d_hist_all = nullptr; uint32_t* h_hist = nullptr;
};
// control block (device + pinned host copy): [RouteHeader][matrix: n_tiles rows of n_tiles + 2 words, the all-gathered rows][this rank's row][emigrant ids]
// a row = [records for destination 0 .. n_tiles), status (ROUTE_*), records this rank can receive]
static size_t tiles_row_words(uint32_t n_tiles) { return (size_t)n_tiles + 2; }
static size_t tiles_matrix_off() { return sizeof(RouteHeader); }
static size_t tiles_row_off(uint32_t n_tiles) { return tiles_matrix_off() + sizeof(uint32_t) * (size_t)n_tiles * tiles_row_words(n_tiles); }
static size_t tiles_emig_off(uint32_t n_tiles) { return tiles_row_off(n_tiles) + sizeof(uint32_t) * tiles_row_words(n_tiles); }
static size_t tiles_ctl_bytes(uint32_t n_tiles) { return tiles_emig_off(n_tiles) + sizeof(uint32_t) * SGP_TILES_EMIG_INLINE; }

SGP_API int sgp_tiles_unique_id(uint8_t out[SGP_TILES_UNIQUE_ID_BYTES])
{
	if (!out) return fail(SGP_ERR_INVALID, "sgp_tiles_unique_id: NULL");
	if (!rccl_load()) return fail(SGP_ERR_HIP, "sgp_tiles_unique_id: RCCL (librccl.so) not found");
	sgp_nccl_unique_id id;
	RCCL_TRY(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
	static_assert(sizeof(id) == SGP_TILES_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
	memcpy(out, &id, sizeof(id));
	return SGP_OK;
}

template <typename T> static int tiles_grow(sgp_world* w, T*& p, uint32_t& cap, uint32_t need, bool keep = false)
{
	if (need <= cap) return SGP_OK;
	const uint32_t nc = std::max(need + need / 2, 4096u);
	T* q = nullptr;
	HIP_TRY(hipMalloc((void**)&q, sizeof(T) * (size_t)nc));
	if (p) { HIP_TRY(hipStreamSynchronize(w->stream)); if (keep && cap) HIP_TRY(hipMemcpy(q, p, sizeof(T) * (size_t)cap, hipMemcpyDeviceToDevice)); hipFree(p); }
	p = q; cap = nc;
	return SGP_OK;
}

SGP_API int sgp_tiles_destroy(sgp_tiles* t)
{
	if (!t) return SGP_OK;
	if (t->w) { hipSetDevice(t->w->device); hipStreamSynchronize(t->w->stream); }
	if (t->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(t->comm);
	hipFree(t->d_block_counts); hipFree(t->d_block_offsets); hipFree(t->d_ctl); hipFree(t->d_send); hipFree(t->d_recv); hipFree(t->d_emig); hipFree(t->d_seq_ids);
	if (t->h_ctl) hipHostFree(t->h_ctl);
	if (t->h_recv) hipHostFree(t->h_recv);
	if (t->h_keys) hipHostFree(t->h_keys);
	if (t->h_aux) hipHostFree(t->h_aux);
	if (t->h_create) hipHostFree(t->h_create);
	hipFree(t->d_keys); hipFree(t->d_aux); hipFree(t->d_create);
	hipFree(t->d_hist); hipFree(t->d_hist_all); if (t->h_hist) hipHostFree(t->h_hist);
	delete t;
	return SGP_OK;
}

SGP_API int sgp_tiles_create(sgp_world* w, uint32_t rank, uint32_t n_tiles, const float* boxes, float margin, float radius_pad, const uint8_t* unique_id, sgp_tiles** out)
{
	if (!w || !boxes || !out || n_tiles < 1 || n_tiles > SGP_MAX_TILES || rank >= n_tiles) return fail(SGP_ERR_INVALID, "sgp_tiles_create: bad arguments (1..64 tiles)");
	*out = nullptr;
	hipSetDevice(w->device);
	sgp_tiles* t = new sgp_tiles();
	t->w = w; t->rank = rank; t->n_tiles = n_tiles;
	memset(&t->route, 0, sizeof(t->route));
	memcpy(t->route.boxes, boxes, sizeof(float) * 6 * n_tiles);
	t->route.n_tiles = n_tiles; t->route.my_rank = rank; t->route.margin = margin; t->route.pad = margin + radius_pad;
	memset(&t->stats, 0, sizeof(t->stats));
	if (const char* hr = getenv("SGP_TILES_HOST_RECORDS")) t->host_records_only = *hr && atoi(hr) != 0;
	if (const char* fr = getenv("SGP_TILES_TEST_FAIL_RANK")) t->test_fail_growth = (*fr && (uint32_t)atoi(fr) == rank) ? 1 : 0;      // (debugging aid: DESIGN.md 6)
	const size_t cb = tiles_ctl_bytes(n_tiles);
	if (hipMalloc((void**)&t->d_ctl, cb) != hipSuccess || hipHostMalloc((void**)&t->h_ctl, cb, hipHostMallocDefault) != hipSuccess) { sgp_tiles_destroy(t); return fail(SGP_ERR_HIP, "sgp_tiles_create: allocation"); }
	hipMemset(t->d_ctl, 0, cb); memset(t->h_ctl, 0, cb);
	t->recv_counts.assign(n_tiles, 0); t->recv_offsets.assign(n_tiles, 0);
	if (unique_id) {      // (a one-tile communicator is legal: it lets a single GPU run the whole collective path, bench.py --force-comm)
		if (!rccl_load()) { sgp_tiles_destroy(t); return fail(SGP_ERR_HIP, "sgp_tiles_create: RCCL (librccl.so) not found"); }
		sgp_nccl_unique_id id; memcpy(&id, unique_id, sizeof(id));
		const auto t0 = std::chrono::steady_clock::now();
		const int rc = g_rccl.CommInitRank(&t->comm, (int)n_tiles, id, (int)rank);
		if (rc != 0) { sgp_tiles_destroy(t); return rccl_fail("ncclCommInitRank", rc); }
		t->stats.comm_init_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
		int seen = 0;
		if (g_rccl.CommCount && g_rccl.CommCount(t->comm, &seen) == 0) t->stats.comm_ranks = (uint32_t)seen;
		if (t->stats.comm_ranks && t->stats.comm_ranks != n_tiles) { sgp_tiles_destroy(t); return fail(SGP_ERR_HIP, "sgp_tiles_create: the RCCL communicator does not have one rank per tile"); }
	}
	*out = t;
	return SGP_OK;
}

// phase 1: export + routing kernels (stream order), header and emigrant ids still on the device
static int tiles_launch_route(sgp_tiles* t)
{
	sgp_world* w = t->w;
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	const uint32_t blocks = w->high ? (w->high + 255u) / 256u : 1u;
	const uint32_t cols = t->n_tiles + 1;
	if (blocks * cols > t->cap_blocks) {
		uint32_t c1 = t->cap_blocks, c2 = t->cap_blocks;
		{ int r = tiles_grow(w, t->d_block_counts, c1, blocks * cols); if (r != SGP_OK) return r; }
		{ int r = tiles_grow(w, t->d_block_offsets, c2, blocks * cols); if (r != SGP_OK) return r; }
		t->cap_blocks = std::min(c1, c2);
	}
	if (!t->cap_send) { int r = tiles_grow(w, t->d_send, t->cap_send, 16384u); if (r != SGP_OK) return r; }
	if (!t->cap_emig) { int r = tiles_grow(w, t->d_emig, t->cap_emig, 4096u); if (r != SGP_OK) return r; }
	if (!t->cap_recv) { int r = tiles_grow(w, t->d_recv, t->cap_recv, 16384u); if (r != SGP_OK) return r; }
	launch_route_export(w->dv, w->high, t->route, t->d_block_counts, t->d_block_offsets, (RouteHeader*)t->d_ctl, t->d_send, t->cap_send, t->d_emig, t->cap_emig,
	                    (uint32_t*)(t->d_ctl + tiles_row_off(t->n_tiles)), t->cap_recv, t->host_status, w->stream);
	// the first emigrant ids ride along with the header copy
	HIP_TRY(hipMemcpyAsync(t->d_ctl + tiles_emig_off(t->n_tiles), t->d_emig, sizeof(uint32_t) * std::min<uint32_t>(SGP_TILES_EMIG_INLINE, t->cap_emig), hipMemcpyDeviceToDevice, w->stream));
	return SGP_OK;
}

// phase 2 (after the control block is on the host): does this rank have to grow a buffer and route again?
static bool tiles_needs_redo(const sgp_tiles* t)
{
	const RouteHeader* h = (const RouteHeader*)t->h_ctl;
	return h->total > t->cap_send || h->n_emigrants > t->cap_emig;
}
// grows what the last routing found too small; a failure is remembered (host_status) instead of returned: the caller still has a collective to finish
static void tiles_grow_for_redo(sgp_tiles* t, uint32_t need_recv)
{
	sgp_world* w = t->w;
	const RouteHeader* h = (const RouteHeader*)t->h_ctl;
	int r = SGP_OK;
	if (t->test_fail_growth) { t->test_fail_growth = 0; r = fail(SGP_ERR_HIP, "sgp_tiles_exchange: buffer growth failed (forced by SGP_TILES_TEST_FAIL_RANK)"); }
	if (r == SGP_OK && h->total > t->cap_send) r = tiles_grow(w, t->d_send, t->cap_send, h->total);
	if (r == SGP_OK && h->n_emigrants > t->cap_emig) r = tiles_grow(w, t->d_emig, t->cap_emig, h->n_emigrants);
	if (r == SGP_OK && need_recv > t->cap_recv) r = tiles_grow(w, t->d_recv, t->cap_recv, need_recv);
	if (r != SGP_OK) t->host_status = ROUTE_FAILED;
}
// phase 3b (after the records have left): the emigrants leave this world
static int tiles_remove_emigrants(sgp_tiles* t)
{
	sgp_world* w = t->w;
	const RouteHeader* h = (const RouteHeader*)t->h_ctl;
	t->stats.exported = h->total; t->stats.emigrated = h->n_emigrants;
	if (h->n_emigrants) {
		std::vector<uint32_t> ids(h->n_emigrants);
		const uint32_t inl = std::min<uint32_t>(h->n_emigrants, SGP_TILES_EMIG_INLINE);
		memcpy(ids.data(), t->h_ctl + tiles_emig_off(t->n_tiles), sizeof(uint32_t) * inl);
		if (h->n_emigrants > inl) { HIP_TRY(hipMemcpy(ids.data() + inl, t->d_emig + inl, sizeof(uint32_t) * (h->n_emigrants - inl), hipMemcpyDeviceToHost)); }
		// owned dynamic bodies whose centre has left the tile: removed here, re-created by the tile that contains them (their record is already
		// in the send buffer, flagged SGP_GHOST_TAKE_OWNERSHIP); the caller learns about it through sgp_tiles_drain_migrations
		for (uint32_t id : ids) {
			if (!live(w, id)) continue;
			sgp_migration m; memset(&m, 0, sizeof(m)); m.userdata = w->hb[id].userdata; m.old_id = id; m.new_id = SGP_INVALID_ID; m.direction = SGP_MIGRATION_OUT;
			t->migrations.push_back(m);
			const int r = sgp_body_remove(w, id); if (r != SGP_OK) return r;
		}
	}
	return SGP_OK;
}

// phase 4: what arrived (n records in d_recv, by source rank) becomes this world's ghost set (+ immigrants)
// the bodies the last import queued for creation on the device (w->rec_creates): list up, one kernel over the received records
static int tiles_launch_creates(sgp_tiles* t)
{
	sgp_world* w = t->w;
	const uint32_t n = (uint32_t)w->rec_creates.size();
	if (!n) return SGP_OK;
	// what is still queued comes first: a newcomer may have been given the slot of a body whose removal (an emigrant, a ghost that left) has not reached the
	// device yet -- as a command its creation would have queued up behind that removal
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	{ int r = tiles_grow(w, t->d_create, t->cap_create, n); if (r != SGP_OK) return r; }
	if (n > t->cap_h_create) {
		if (t->h_create) { HIP_TRY(hipStreamSynchronize(w->stream)); hipHostFree(t->h_create); }
		t->cap_h_create = n + n / 2 + 256;
		HIP_TRY(hipHostMalloc((void**)&t->h_create, sizeof(uint4) * (size_t)t->cap_h_create, hipHostMallocDefault));
	}
	HIP_TRY(hipStreamSynchronize(w->stream));      // (the pinned list of the previous import has been read)
	memcpy(t->h_create, w->rec_creates.data(), sizeof(uint4) * n);
	HIP_TRY(hipMemcpyAsync(t->d_create, t->h_create, sizeof(uint4) * n, hipMemcpyHostToDevice, w->stream));
	sgp_body_desc def; sgp_default_body_desc(&def);
	launch_create_from_records(w->dv, t->d_recv, t->d_create, n, def.gravity_factor, def.linear_damping, def.angular_damping, w->stream);
	t->stats.device_creates += n;
	w->rec_creates.clear();
	w->grid_valid = false; w->dirty_since_step = true;
	return SGP_OK;
}

static int tiles_import(sgp_tiles* t, uint32_t n)
{
	sgp_world* w = t->w;
	t->stats.received = n;
	w->rec_creates.clear();
	// steady state: the same ghosts as last step in the same order, nobody immigrating -> poses go from the received records to the bodies
	// on the device; the host only sees 16 bytes per record (global id + ownership flag + what kind of record it is, packed by a kernel), not the 128-byte records
	struct GhostKey { uint64_t global_id; uint32_t motion_type, info; };
	const float* lo = t->route.boxes + 6 * t->rank; const float* hi = lo + 3;
	if (n) {
		{ int r = tiles_grow(w, t->d_keys, t->cap_keys, n); if (r != SGP_OK) return r; }
		{ int r = tiles_grow(w, t->d_aux, t->cap_aux, n); if (r != SGP_OK) return r; }
		if (n > t->cap_h_keys) {
			if (t->h_keys) hipHostFree(t->h_keys);
			if (t->h_aux) hipHostFree(t->h_aux);
			t->cap_h_keys = n + n / 2 + 1024;
			HIP_TRY(hipHostMalloc((void**)&t->h_keys, 16 * (size_t)t->cap_h_keys, hipHostMallocDefault));
			HIP_TRY(hipHostMalloc((void**)&t->h_aux, 16 * (size_t)t->cap_h_keys, hipHostMallocDefault));
		}
		launch_pack_ghost_keys(t->d_recv, n, t->d_keys, t->d_aux, lo, hi, w->stream);
		HIP_TRY(hipMemcpyAsync(t->h_keys, t->d_keys, 16 * (size_t)n, hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		const GhostKey* keys = (const GhostKey*)t->h_keys;
		bool same = n == w->ghost_seq.size();
		for (uint32_t k = 0; k < n && same; ++k) same = !(keys[k].motion_type & SGP_GHOST_TAKE_OWNERSHIP) && keys[k].global_id == w->ghost_seq[k].first && live(w, w->ghost_seq[k].second);
		if (same) {
			{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
			GhostDeviceSource dev = { t->d_recv, &t->d_seq_ids, &t->cap_seq, &t->ids_version };
			if (t->ids_version != w->ghost_seq_version) { int r = upload_ghost_ids(w, &dev); if (r != SGP_OK) return r; }
			launch_ghost_refresh_records(w->dv, t->d_recv, t->d_seq_ids, n, w->stream);
			w->grid_valid = false; w->dirty_since_step = true;
			t->stats.ghosts = n; t->stats.immigrated = 0; t->stats.fast_imports++;
			return SGP_OK;
		}
	}
	// The set changed (or bodies immigrate).  Round 6: the host still owns the body slots -- it hands them out in the order the CPU statement does, from the
	// 16-byte keys --, but the bodies are created on the device straight from the received records: no 128-byte record comes to the host, no 160-byte create
	// command goes back.  Only records that name a hull or a mesh (the host keeps reference counts and mass properties for those shapes) take the old road.
	const GhostKey* keys = (const GhostKey*)t->h_keys;      // (n > 0: packed above)
	bool by_key = !t->host_records_only;
	for (uint32_t k = 0; k < n && by_key; ++k) { const uint32_t type = (keys[k].info >> GKEY_SHAPE_SHIFT) & 7u; by_key = type == SGP_SHAPE_SPHERE || type == SGP_SHAPE_BOX || type == SGP_SHAPE_CAPSULE; }
	if (by_key) {
		if (n) { HIP_TRY(hipMemcpyAsync(t->h_aux, t->d_aux, 16 * (size_t)n, hipMemcpyDeviceToHost, w->stream)); HIP_TRY(hipStreamSynchronize(w->stream)); }
	} else {
		if (n > t->cap_h_recv) {
			if (t->h_recv) hipHostFree(t->h_recv);
			t->cap_h_recv = n + n / 2 + 1024;
			HIP_TRY(hipHostMalloc((void**)&t->h_recv, sizeof(sgp_ghost_record) * (size_t)t->cap_h_recv, hipHostMallocDefault));
		}
		if (n) {
			HIP_TRY(hipMemcpyAsync(t->h_recv, t->d_recv, sizeof(sgp_ghost_record) * (size_t)n, hipMemcpyDeviceToHost, w->stream));
			HIP_TRY(hipStreamSynchronize(w->stream));
		}
		t->stats.slow_imports++;      // (= imports whose records came to the host)
	}
	const sgp_ghost_record* recs = by_key ? nullptr : t->h_recv;
	const uint4* k4 = by_key ? (const uint4*)t->h_keys : nullptr; const uint4* a4 = by_key ? (const uint4*)t->h_aux : nullptr;
	// who is a ghost, who immigrates (flagged records addressed to another tile are dropped)
	bool plain = true;
	for (uint32_t k = 0; k < n && plain; ++k) plain = !(keys[k].motion_type & SGP_GHOST_TAKE_OWNERSHIP);
	const size_t seq_before = w->ghost_seq.size();
	if (plain) {
		// ghosts only: the poses stay on the device -- the host compares global ids (and gives the few bodies that entered the set a slot, removes those that
		// left), one kernel refreshes every ghost from the received records
		bool unchanged = n == seq_before;            // (no ghosts before, none now: nothing for the host to do either)
		for (uint32_t k = 0; k < n && unchanged; ++k) unchanged = keys[k].global_id == w->ghost_seq[k].first;
		GhostDeviceSource dev = { t->d_recv, &t->d_seq_ids, &t->cap_seq, &t->ids_version };
		{ int rc = import_ghosts_impl(w, recs, n, &dev, nullptr, (const uint64_t*)t->h_keys, 2, k4, a4); if (rc != SGP_OK) return rc; }
		{ int rc = tiles_launch_creates(t); if (rc != SGP_OK) return rc; }
		t->stats.ghosts = n; t->stats.immigrated = 0;
		if (unchanged) t->stats.fast_imports++;
		return SGP_OK;
	}
	// bodies immigrate with this exchange: their records sit between the ghosts'.  The ghosts still take the device path (by index: no record is copied, no
	// refresh command is made -- a tile of the collapsing tower holds 25 000 ghosts and receives immigrants in EVERY step: 3.5 MB of records copied and
	// 25 000 commands built, uploaded and applied per step was most of the exchange's 1.2 ms, profiles/r04_tiles_import.md)
	std::vector<uint8_t> skip(n, 0);
	std::vector<uint32_t> immigrants;
	uint32_t n_ghosts = 0;
	for (uint32_t k = 0; k < n; ++k) {
		if (!(keys[k].motion_type & SGP_GHOST_TAKE_OWNERSHIP)) { ++n_ghosts; continue; }
		skip[k] = 1;
		if (by_key ? (keys[k].info & GKEY_IN_REGION) != 0u : in_box(t->h_recv[k].pos, lo, hi, 0.0f)) immigrants.push_back(k);
	}
	{
		GhostDeviceSource dev = { t->d_recv, &t->d_seq_ids, &t->cap_seq, &t->ids_version };
		int rc = import_ghosts_impl(w, recs, n, &dev, skip.data(), (const uint64_t*)t->h_keys, 2, k4, a4); if (rc != SGP_OK) return rc;
	}
	uint32_t n_imm = 0;
	for (uint32_t k : immigrants) {
		uint32_t id = SGP_INVALID_ID;
		int rc;
		uint64_t userdata, gid = keys[k].global_id;
		if (by_key) {
			// the body as it was: dynamic, its own layer and flags; slot from the key, created on the device from the record (mass properties there too)
			const uint4 aux = a4[k];
			userdata = (uint64_t)aux.x | ((uint64_t)aux.y << 32);
			if (!(keys[k].info & GKEY_VALID)) rc = SGP_ERR_REJECTED;
			else {
				const uint32_t rflags = (keys[k].info >> GKEY_FLAGS_SHIFT) & 0x3Fu, type = (keys[k].info >> GKEY_SHAPE_SHIFT) & 7u;
				uint32_t f = BF_ALIVE | SGP_MOTION_DYNAMIC | ((rflags & SGP_GHOST_FLAG_LAYER_MASK) << BF_LAYER_SHIFT) | (type << BF_SHAPE_SHIFT);
				if (rflags & SGP_GHOST_FLAG_SENSOR) f |= BF_SENSOR;
				if (rflags & SGP_GHOST_FLAG_ALLOW_SLEEP) f |= BF_ALLOW_SLEEP;
				if (rflags & SGP_GHOST_FLAG_ZERO_DRAG) f |= BF_ZERO_LIN_DRAG;
				float rad, vol; memcpy(&rad, &aux.z, 4); memcpy(&vol, &aux.w, 4);
				rc = book_record_body(w, &f, userdata, rad, vol, false, &id);
				if (rc == SGP_OK) w->rec_creates.push_back(make_uint4(k, id, f, 0u));
			}
		} else {
			const sgp_ghost_record& r = t->h_recv[k];
			userdata = r.userdata;
			sgp_body_desc d; sgp_default_body_desc(&d);
			memcpy(d.pos, r.pos, 12); memcpy(d.rot, r.rot, 16); memcpy(d.lin_vel, r.lin_vel, 12); memcpy(d.ang_vel, r.ang_vel, 12);
			d.shape_type = r.shape_type; memcpy(d.shape, r.shape, 16);
			d.motion_type = SGP_MOTION_DYNAMIC;
			d.layer = (int32_t)(r.flags & SGP_GHOST_FLAG_LAYER_MASK);
			d.is_sensor = (r.flags & SGP_GHOST_FLAG_SENSOR) ? 1 : 0; d.allow_sleeping = (r.flags & SGP_GHOST_FLAG_ALLOW_SLEEP) ? 1 : 0; d.use_zero_linear_drag = (r.flags & SGP_GHOST_FLAG_ZERO_DRAG) ? 1 : 0;
			d.mass = r.mass; d.friction = r.friction; d.restitution = r.restitution;
			d.gravity_factor = r.gravity_factor; d.linear_damping = r.linear_damping; d.angular_damping = r.angular_damping;
			d.userdata = r.userdata; d.activate = 1;
			rc = add_one(w, &d, &id, false);
		}
		// the previous owner has already let go of the body: failing to take it over must not pass silently
		if (rc != SGP_OK) return fail(rc == SGP_ERR_REJECTED ? SGP_ERR_INVALID : rc, "sgp_tiles_exchange: could not take over a migrating body (raise max_bodies; hull / mesh ids must mean the same shape on every tile)");
		sgp_migration m; memset(&m, 0, sizeof(m)); m.userdata = userdata; m.old_id = (uint32_t)(gid & 0xFFFFFFFFull); m.new_id = id; m.direction = SGP_MIGRATION_IN; m.peer = (uint32_t)(gid >> 40);
		t->migrations.push_back(m);
		++n_imm;
	}
	{ int rc = tiles_launch_creates(t); if (rc != SGP_OK) return rc; }
	t->stats.immigrated = n_imm; t->stats.ghosts = n_ghosts;
	return SGP_OK;
}

SGP_API int sgp_tiles_exchange(sgp_tiles* t)
{
	if (!t || !t->w) return fail(SGP_ERR_INVALID, "sgp_tiles_exchange: NULL");
	sgp_world* w = t->w;
	hipSetDevice(w->device);
	const uint32_t T = t->n_tiles;
	const auto t_begin = std::chrono::steady_clock::now();
	struct Stamp { sgp_tiles* t; std::chrono::steady_clock::time_point t0; ~Stamp() { t->stats.last_exchange_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); t->stats.exchanges++; t->stats.total_exchange_ms += t->stats.last_exchange_ms; } } stamp{ t, t_begin };
	if (T > 1 && !t->comm) return fail(SGP_ERR_INVALID, "sgp_tiles_exchange: created without a communicator (use sgp_tiles_exchange_group for tiles of one process)");
	uint32_t* d_matrix = (uint32_t*)(t->d_ctl + tiles_matrix_off());
	uint32_t* d_row = (uint32_t*)(t->d_ctl + tiles_row_off(T));
	const size_t RW = tiles_row_words(T);
	const uint32_t* matrix = (const uint32_t*)(t->h_ctl + tiles_matrix_off());          // [source][destination .. , status, receive capacity]
	// NO RANK MAY LEAVE ITS PEERS WAITING.  Every collective of an exchange is entered by every rank the same number of times, because every decision
	// that shapes the sequence is taken from the all-gathered rows, which are the same everywhere: a row carries, besides the per-destination counts,
	// the rank's status (decided on the device: do its records fit its send buffers?) and how many records it can receive.  If any rank has to grow a
	// buffer, all ranks route again and gather again (local growth in between; the steady state never gets here: one gather, one host round trip).  A
	// rank whose growth fails does not return: it reports ROUTE_FAILED in the next gather, and then EVERY rank returns an error (SGP_ERR_PEER on the
	// others) without having posted a send or a receive.  Between the last gather and ncclGroupEnd nothing can fail but the collective library itself.
	// (What this cannot cover: a HIP runtime failure of the copies and the synchronisation inside the loop -- a rank whose device stops answering cannot tell
	// anybody, with or without this protocol.)
	uint32_t n_recv = 0;
	bool settled = false;
	for (int attempt = 0; attempt < 4 && !settled; ++attempt) {
		if (t->test_fail_growth && t->host_status == ROUTE_OK) t->host_status = ROUTE_REDO;      // (test hook: ask for a round of growth, which then fails)
		{ int r = tiles_launch_route(t);
		  if (r != SGP_OK) {      // (could not even route: with a communicator the peers are told through the gather, like a failed growth)
			if (!t->comm) return r;
			t->host_status = ROUTE_FAILED;
			std::vector<uint32_t> row(RW, 0u); row[T] = ROUTE_FAILED;
			HIP_TRY(hipMemcpyAsync(d_row, row.data(), sizeof(uint32_t) * RW, hipMemcpyHostToDevice, w->stream));
		  } }
		if (t->comm) RCCL_TRY(g_rccl.AllGather(d_row, d_matrix, RW, SGP_NCCL_UINT32, t->comm, w->stream), "ncclAllGather");
		else HIP_TRY(hipMemcpyAsync(d_matrix, d_row, sizeof(uint32_t) * RW, hipMemcpyDeviceToDevice, w->stream));      // (one tile: its own row)
		HIP_TRY(hipMemcpyAsync(t->h_ctl, t->d_ctl, tiles_ctl_bytes(T), hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
		if (t->host_status == ROUTE_REDO) t->host_status = ROUTE_OK;
		bool any_failed = false, any_redo = false;
		uint32_t my_incoming = 0;
		for (uint32_t r = 0; r < T; ++r) {
			const uint32_t* row = matrix + (size_t)r * RW;
			any_failed = any_failed || row[T] == ROUTE_FAILED;
			any_redo = any_redo || row[T] == ROUTE_REDO;
			uint32_t incoming = 0;
			for (uint32_t src = 0; src < T; ++src) if (src != r) incoming += matrix[(size_t)src * RW + r];
			any_redo = any_redo || incoming > row[T + 1];
			if (r == t->rank) my_incoming = incoming;
		}
		if (any_failed) {
			if (t->host_status == ROUTE_FAILED) { t->host_status = ROUTE_OK; return SGP_ERR_HIP; }      // (sgp_last_error holds what failed here)
			return fail(SGP_ERR_PEER, "sgp_tiles_exchange: another rank could not take part in this exchange (its buffers could not grow); nothing was sent");
		}
		if (!any_redo) { settled = true; n_recv = my_incoming; break; }
		if (attempt == 3) break;
		tiles_grow_for_redo(t, my_incoming);      // (local; a failure travels with the next gather)
		t->stats.route_retries++;
	}
	if (!settled) return fail(SGP_ERR_CAPACITY, "sgp_tiles_exchange: the buffers did not settle in four rounds");      // (every rank gets here together: same rows)
	const RouteHeader* h = (const RouteHeader*)t->h_ctl;
	for (uint32_t r = 0, at = 0; r < T; ++r) { t->recv_counts[r] = (r != t->rank) ? matrix[(size_t)r * RW + t->rank] : 0u; t->recv_offsets[r] = at; at += t->recv_counts[r]; }
	if (T > 1) {
		RCCL_TRY(g_rccl.GroupStart(), "ncclGroupStart");
		for (uint32_t r = 0; r < T; ++r) {
			if (r == t->rank) continue;
			if (h->seg_count[r]) RCCL_TRY(g_rccl.Send(t->d_send + h->seg_start[r], sizeof(sgp_ghost_record) * (size_t)h->seg_count[r], SGP_NCCL_UINT8, (int)r, t->comm, w->stream), "ncclSend");
			if (t->recv_counts[r]) RCCL_TRY(g_rccl.Recv(t->d_recv + t->recv_offsets[r], sizeof(sgp_ghost_record) * (size_t)t->recv_counts[r], SGP_NCCL_UINT8, (int)r, t->comm, w->stream), "ncclRecv");
		}
		RCCL_TRY(g_rccl.GroupEnd(), "ncclGroupEnd");
	}
	{ int r = tiles_remove_emigrants(t); if (r != SGP_OK) return r; }      // (their records are on their way: the new owner creates them in its import)
	t->stats.sent = h->total;
	return tiles_import(t, n_recv);
}

// Several tiles driven by ONE process (one GPU or several): the same exchange with plain device-to-device copies in place of RCCL.
SGP_API int sgp_tiles_exchange_group(sgp_tiles** ts, uint32_t n)
{
	if (!ts || !n) return fail(SGP_ERR_INVALID, "sgp_tiles_exchange_group: NULL");
	for (uint32_t i = 0; i < n; ++i) if (!ts[i] || ts[i]->n_tiles != n || ts[i]->rank != i) return fail(SGP_ERR_INVALID, "sgp_tiles_exchange_group: pass all tiles, in rank order");
	for (int attempt = 0; attempt < 3; ++attempt) {
		for (uint32_t i = 0; i < n; ++i) { hipSetDevice(ts[i]->w->device); int r = tiles_launch_route(ts[i]); if (r != SGP_OK) return r;
			HIP_TRY(hipMemcpyAsync(ts[i]->h_ctl, ts[i]->d_ctl, tiles_ctl_bytes(n), hipMemcpyDeviceToHost, ts[i]->w->stream)); }
		bool any_redo = false;
		for (uint32_t i = 0; i < n; ++i) { hipSetDevice(ts[i]->w->device); HIP_TRY(hipStreamSynchronize(ts[i]->w->stream)); const RouteHeader* h = (const RouteHeader*)ts[i]->h_ctl; if (h->total > ts[i]->cap_send || h->n_emigrants > ts[i]->cap_emig) any_redo = true; }
		if (any_redo) {        // grow whoever was short, route everyone again (nothing has been removed yet)
			for (uint32_t i = 0; i < n; ++i) { const RouteHeader* h = (const RouteHeader*)ts[i]->h_ctl; hipSetDevice(ts[i]->w->device);
				if (h->total > ts[i]->cap_send) { int r = tiles_grow(ts[i]->w, ts[i]->d_send, ts[i]->cap_send, h->total); if (r != SGP_OK) return r; }
				if (h->n_emigrants > ts[i]->cap_emig) { int r = tiles_grow(ts[i]->w, ts[i]->d_emig, ts[i]->cap_emig, h->n_emigrants); if (r != SGP_OK) return r; } }
			if (attempt == 2) return fail(SGP_ERR_CAPACITY, "sgp_tiles_exchange_group: send buffer");
			continue;
		}
		break;
	}
	// the receive buffers grow BEFORE any body is let go of (a growth that failed after the emigrants had been removed lost them)
	for (uint32_t dst = 0; dst < n; ++dst) {
		uint32_t need = 0;
		for (uint32_t src = 0; src < n; ++src) if (src != dst) need += ((const RouteHeader*)ts[src]->h_ctl)->seg_count[dst];
		hipSetDevice(ts[dst]->w->device);
		{ int r = tiles_grow(ts[dst]->w, ts[dst]->d_recv, ts[dst]->cap_recv, std::max(need, 1u)); if (r != SGP_OK) return r; }
	}
	for (uint32_t i = 0; i < n; ++i) { hipSetDevice(ts[i]->w->device); int r = tiles_remove_emigrants(ts[i]); if (r != SGP_OK) return r; }
	for (uint32_t dst = 0; dst < n; ++dst) {
		sgp_tiles* t = ts[dst];
		hipSetDevice(t->w->device);
		uint32_t n_recv = 0;
		for (uint32_t src = 0; src < n; ++src) { const RouteHeader* hs = (const RouteHeader*)ts[src]->h_ctl; t->recv_counts[src] = src == dst ? 0u : hs->seg_count[dst]; t->recv_offsets[src] = n_recv; n_recv += t->recv_counts[src]; }
		{ int r = tiles_grow(t->w, t->d_recv, t->cap_recv, std::max(n_recv, 1u)); if (r != SGP_OK) return r; }
		for (uint32_t src = 0; src < n; ++src) {
			if (!t->recv_counts[src]) continue;
			const RouteHeader* hs = (const RouteHeader*)ts[src]->h_ctl;
			HIP_TRY(hipMemcpyAsync(t->d_recv + t->recv_offsets[src], ts[src]->d_send + hs->seg_start[dst], sizeof(sgp_ghost_record) * (size_t)t->recv_counts[src], hipMemcpyDeviceToDevice, t->w->stream));
		}
		t->stats.sent = ((const RouteHeader*)t->h_ctl)->total;
		{ int r = tiles_import(t, n_recv); if (r != SGP_OK) return r; }
	}
	return SGP_OK;
}

// ---- re-tiling by body count ------------------------------------------------------------------------------------------------------
// A static split of a scene that moves -- BASELINE config 4 is a tower that falls out of its upper tiles -- leaves tiles without work.  The grid keeps
// its topology (gx x gy x gz, tile = ix + gx (iy + gy iz)); its planes move to the quantiles of where the OWNED bodies are: the x planes from all
// bodies, the y planes of every x slab from that slab's bodies, the z planes of every (x, y) column from that column's.  Four small rounds (bounds,
// then one histogram of SGP_TILE_HIST_BINS bins per axis and group), each a kernel + an all-gather of a few KB + one read-back; every rank derives the
// same planes from the same gathered counts.  Bodies then change owner through the ordinary migration of the next exchange.
#define TILE_HIST_MAX_GROUPS 16
static int tiles_hist_buffers(sgp_tiles* t)
{
	const size_t one = sizeof(uint32_t) * TILE_HIST_MAX_GROUPS * SGP_TILE_HIST_BINS;
	if (!t->d_hist) { HIP_TRY(hipMalloc((void**)&t->d_hist, one)); HIP_TRY(hipMalloc((void**)&t->d_hist_all, one * t->n_tiles)); HIP_TRY(hipHostMalloc((void**)&t->h_hist, one * t->n_tiles, hipHostMallocDefault)); }
	return SGP_OK;
}
// one round on the tiles of this process (one with a communicator, or all of a group): sum[k] = counts over every tile (level 0: min / max as ordered ints)
static int tiles_hist_round(sgp_tiles** ts, uint32_t n_local, const TilePlanes& tp, int level, uint32_t len, std::vector<uint64_t>& sum, int bounds[6])
{
	const uint32_t T = ts[0]->n_tiles;
	for (uint32_t i = 0; i < n_local; ++i) {
		sgp_tiles* t = ts[i]; sgp_world* w = t->w;
		hipSetDevice(w->device);
		{ int r = tiles_hist_buffers(t); if (r != SGP_OK) return r; }
		{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
		if (level == 0) { int* h = (int*)t->h_hist; for (int k = 0; k < 6; ++k) h[k] = k < 3 ? 0x7FFFFFFF : (int)0x80000000; h[6] = h[7] = 0; HIP_TRY(hipMemcpyAsync(t->d_hist, h, 32, hipMemcpyHostToDevice, w->stream)); }
		else HIP_TRY(hipMemsetAsync(t->d_hist, 0, sizeof(uint32_t) * len, w->stream));
		if (w->high) launch_tiles_hist(w->dv, w->high, tp, level, t->d_hist, w->stream);
		if (t->comm) {
			RCCL_TRY(g_rccl.AllGather(t->d_hist, t->d_hist_all, len, SGP_NCCL_UINT32, t->comm, w->stream), "ncclAllGather (re-tiling)");
			HIP_TRY(hipMemcpyAsync(t->h_hist, t->d_hist_all, sizeof(uint32_t) * (size_t)len * T, hipMemcpyDeviceToHost, w->stream));
		} else HIP_TRY(hipMemcpyAsync(t->h_hist, t->d_hist, sizeof(uint32_t) * len, hipMemcpyDeviceToHost, w->stream));
	}
	for (uint32_t i = 0; i < n_local; ++i) { hipSetDevice(ts[i]->w->device); HIP_TRY(hipStreamSynchronize(ts[i]->w->stream)); }
	sum.assign(len, 0);
	for (int k = 0; k < 6; ++k) bounds[k] = k < 3 ? 0x7FFFFFFF : (int)0x80000000;
	auto fold = [&](const uint32_t* h) {
		if (level == 0) { const int* b = (const int*)h; for (int k = 0; k < 3; ++k) { bounds[k] = std::min(bounds[k], b[k]); bounds[3 + k] = std::max(bounds[3 + k], b[3 + k]); } }
		else for (uint32_t k = 0; k < len; ++k) sum[k] += h[k];
	};
	if (ts[0]->comm) for (uint32_t r = 0; r < T; ++r) fold(ts[0]->h_hist + (size_t)r * len);
	else for (uint32_t i = 0; i < n_local; ++i) fold(ts[i]->h_hist);
	return SGP_OK;
}
static inline float ordered_int_to_float(int i) { const int v = i >= 0 ? i : i ^ 0x7FFFFFFF; float f; memcpy(&f, &v, 4); return f; }
// the g - 1 planes that cut a histogram into g parts of equal count (linear inside a bin); an empty histogram is cut evenly
static void quantile_planes(const uint64_t* h, float lo, float hi, uint32_t g, float* planes)
{
	uint64_t total = 0; for (uint32_t b = 0; b < SGP_TILE_HIST_BINS; ++b) total += h[b];
	const double bw = ((double)hi - (double)lo) / SGP_TILE_HIST_BINS;
	for (uint32_t k = 1; k < g; ++k) {
		if (!total) { planes[k - 1] = (float)(lo + ((double)hi - lo) * k / g); continue; }
		const double target = (double)total * k / g;
		uint64_t cum = 0; uint32_t b = 0;
		while (b + 1 < SGP_TILE_HIST_BINS && (double)(cum + h[b]) < target) { cum += h[b]; ++b; }
		const double frac = h[b] ? (target - (double)cum) / (double)h[b] : 0.5;
		planes[k - 1] = (float)(lo + (b + std::min(1.0, std::max(0.0, frac))) * bw);
	}
	for (uint32_t k = 1; k + 1 < g; ++k) if (planes[k] < planes[k - 1]) planes[k] = planes[k - 1];
}
static int tiles_rebalance_impl(sgp_tiles** ts, uint32_t n_local, uint32_t gx, uint32_t gy, uint32_t gz, int by_contacts)
{
	const uint32_t T = ts[0]->n_tiles;
	if (!gx || !gy || !gz || gx > 4 || gy > 4 || gz > 4 || gx * gy * gz != T) return fail(SGP_ERR_INVALID, "sgp_tiles_rebalance: the grid must have one cell per tile (at most 4 per axis)");
	TilePlanes tp; memset(&tp, 0, sizeof(tp)); tp.gx = gx; tp.gy = gy; tp.gz = gz; tp.by_contacts = by_contacts ? 1u : 0u;
	std::vector<uint64_t> sum; int bounds[6];
	{ int r = tiles_hist_round(ts, n_local, tp, 0, 8, sum, bounds); if (r != SGP_OK) return r; }
	if (bounds[0] > bounds[3]) return SGP_OK;                         // nobody owns a dynamic body: nothing to balance
	for (int a = 0; a < 3; ++a) { tp.glo[a] = ordered_int_to_float(bounds[a]); tp.ghi[a] = ordered_int_to_float(bounds[3 + a]); const float pad = 1.0e-3f * (1.0f + fabsf(tp.ghi[a] - tp.glo[a])); tp.glo[a] -= pad; tp.ghi[a] += pad; }
	{ int r = tiles_hist_round(ts, n_local, tp, 1, SGP_TILE_HIST_BINS, sum, bounds); if (r != SGP_OK) return r; }
	quantile_planes(sum.data(), tp.glo[0], tp.ghi[0], gx, tp.xp);
	{ int r = tiles_hist_round(ts, n_local, tp, 2, gx * SGP_TILE_HIST_BINS, sum, bounds); if (r != SGP_OK) return r; }
	for (uint32_t ix = 0; ix < gx; ++ix) quantile_planes(sum.data() + (size_t)ix * SGP_TILE_HIST_BINS, tp.glo[1], tp.ghi[1], gy, tp.yp + 4 * ix);
	{ int r = tiles_hist_round(ts, n_local, tp, 3, gx * gy * SGP_TILE_HIST_BINS, sum, bounds); if (r != SGP_OK) return r; }
	float zp[16 * 4]; memset(zp, 0, sizeof(zp));
	for (uint32_t c = 0; c < gx * gy; ++c) quantile_planes(sum.data() + (size_t)c * SGP_TILE_HIST_BINS, tp.glo[2], tp.ghi[2], gz, zp + 4 * c);
	const float big = 1.0e9f;
	float boxes[6 * SGP_MAX_TILES];
	for (uint32_t r = 0; r < T; ++r) {
		const uint32_t ix = r % gx, iy = (r / gx) % gy, iz = r / (gx * gy);
		float* lo = boxes + 6 * r; float* hi = lo + 3;
		lo[0] = ix ? tp.xp[ix - 1] : -big; hi[0] = ix + 1 < gx ? tp.xp[ix] : big;
		lo[1] = iy ? tp.yp[4 * ix + iy - 1] : -big; hi[1] = iy + 1 < gy ? tp.yp[4 * ix + iy] : big;
		lo[2] = iz ? zp[4 * (ix + gx * iy) + iz - 1] : -big; hi[2] = iz + 1 < gz ? zp[4 * (ix + gx * iy) + iz] : big;
	}
	for (uint32_t i = 0; i < n_local; ++i) { memcpy(ts[i]->route.boxes, boxes, sizeof(float) * 6 * T); ts[i]->stats.rebalances++; }
	return SGP_OK;
}
SGP_API int sgp_tiles_rebalance(sgp_tiles* t, uint32_t gx, uint32_t gy, uint32_t gz, int by_contacts)
{
	if (!t || !t->w) return fail(SGP_ERR_INVALID, "sgp_tiles_rebalance: NULL");
	if (t->n_tiles > 1 && !t->comm) return fail(SGP_ERR_INVALID, "sgp_tiles_rebalance: created without a communicator (use sgp_tiles_rebalance_group for tiles of one process)");
	return tiles_rebalance_impl(&t, 1, gx, gy, gz, by_contacts);
}
SGP_API int sgp_tiles_rebalance_group(sgp_tiles** ts, uint32_t n, uint32_t gx, uint32_t gy, uint32_t gz, int by_contacts)
{
	if (!ts || !n) return fail(SGP_ERR_INVALID, "sgp_tiles_rebalance_group: NULL");
	for (uint32_t i = 0; i < n; ++i) if (!ts[i] || ts[i]->n_tiles != n || ts[i]->rank != i || ts[i]->comm) return fail(SGP_ERR_INVALID, "sgp_tiles_rebalance_group: pass all tiles of the (communicator-less) group, in rank order");
	return tiles_rebalance_impl(ts, n, gx, gy, gz, by_contacts);
}
SGP_API int sgp_tiles_get_boxes(sgp_tiles* t, float* boxes_out)
{
	if (!t || !boxes_out) return fail(SGP_ERR_INVALID, "sgp_tiles_get_boxes: NULL");
	memcpy(boxes_out, t->route.boxes, sizeof(float) * 6 * t->n_tiles);
	return SGP_OK;
}

SGP_API int sgp_tiles_get_stats(sgp_tiles* t, sgp_tiles_stats* out)
{
	if (!t || !out) return fail(SGP_ERR_INVALID, "sgp_tiles_get_stats: NULL");
	*out = t->stats;
	return SGP_OK;
}

SGP_API int sgp_tiles_drain_migrations(sgp_tiles* t, sgp_migration* out, uint32_t cap, uint32_t* n_out)
{
	if (!t || !n_out || (!out && cap)) return fail(SGP_ERR_INVALID, "sgp_tiles_drain_migrations: NULL");
	const uint32_t n = (uint32_t)t->migrations.size(), m = std::min(n, cap);
	if (m) memcpy(out, t->migrations.data(), sizeof(sgp_migration) * m);
	*n_out = n;
	t->migrations.erase(t->migrations.begin(), t->migrations.begin() + m);
	return SGP_OK;
}

// Self test of the run-time RCCL binding on ONE GPU (tests/test_tiles_parity_gpu.py): a one-rank communicator, an all-gather, and a grouped
// ncclSend / ncclRecv of `n_records` records from this rank to itself, compared byte for byte.  Not declared in include/sgp.h.
SGP_API int sgp_tiles_selftest_rccl(sgp_world* w, uint32_t n_records)
{
	if (!w || !n_records) return fail(SGP_ERR_INVALID, "sgp_tiles_selftest_rccl: bad arguments");
	hipSetDevice(w->device);
	if (!rccl_load()) return fail(SGP_ERR_HIP, "RCCL (librccl.so) not found");
	sgp_nccl_unique_id id;
	RCCL_TRY(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
	sgp_nccl_comm comm = nullptr;
	RCCL_TRY(g_rccl.CommInitRank(&comm, 1, id, 0), "ncclCommInitRank");
	const size_t bytes = sizeof(sgp_ghost_record) * (size_t)n_records;
	unsigned char *a = nullptr, *b = nullptr; uint32_t *c = nullptr;
	HIP_TRY(hipMalloc((void**)&a, bytes)); HIP_TRY(hipMalloc((void**)&b, bytes)); HIP_TRY(hipMalloc((void**)&c, 64));
	std::vector<unsigned char> src(bytes), dst(bytes, 0);
	for (size_t i = 0; i < bytes; ++i) src[i] = (unsigned char)((i * 2654435761u) >> 13);
	const uint32_t row[4] = { 11, 22, 33, 44 }; uint32_t got[4] = { 0, 0, 0, 0 };
	HIP_TRY(hipMemcpy(a, src.data(), bytes, hipMemcpyHostToDevice)); HIP_TRY(hipMemset(b, 0, bytes)); HIP_TRY(hipMemcpy(c, row, 16, hipMemcpyHostToDevice));
	int rc = g_rccl.AllGather(c, c + 8, 4, SGP_NCCL_UINT32, comm, w->stream);
	if (rc == 0) rc = g_rccl.GroupStart();
	if (rc == 0) rc = g_rccl.Send(a, bytes, SGP_NCCL_UINT8, 0, comm, w->stream);
	if (rc == 0) rc = g_rccl.Recv(b, bytes, SGP_NCCL_UINT8, 0, comm, w->stream);
	if (rc == 0) rc = g_rccl.GroupEnd();
	hipStreamSynchronize(w->stream);
	hipMemcpy(dst.data(), b, bytes, hipMemcpyDeviceToHost); hipMemcpy(got, c + 8, 16, hipMemcpyDeviceToHost);
	hipFree(a); hipFree(b); hipFree(c);
	g_rccl.CommDestroy(comm);
	if (rc != 0) return rccl_fail("RCCL self test", rc);
	if (memcmp(src.data(), dst.data(), bytes) != 0 || memcmp(row, got, 16) != 0) return fail(SGP_ERR_HIP, "RCCL self test: payload mismatch");
	return SGP_OK;
}

