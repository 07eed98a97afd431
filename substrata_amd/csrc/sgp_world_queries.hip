// sgp_world_queries.hip -- ray queries (PhysicsWorld::traceRay, PhysicsWorld.cpp:1668-1725), the character controller's capsule queries and sphere casts.
#include "sgp_world_internal.h"

// ---------------------------------------------------------------------------------------------------------------
// ray queries, PhysicsWorld.cpp:1668-1725

static void finish_hit(sgp_world* w, sgp_hit* h)
{
	h->userdata = h->id != SGP_INVALID_ID ? w->hb[h->id].userdata : 0;
	h->sub_shape = 0;
	if (h->id != SGP_INVALID_ID) h->id = compound_id_of(w, h->id, &h->sub_shape);
}

// One ray through the resident server (RayMailbox, sgp_kernels.h).  Returns 1 when the ray was answered, 0 when the caller should take the launch path
// (server disabled or it could not be reached), < 0 on error.
#define SGP_RAY_SERVER_IDLE_TICKS 30000ull          // 300 us at the 100 MHz wall clock: a caller tracing rays one after the other never lets it idle that long
#define SGP_RAY_SERVER_MAX_TICKS 20000000ull        // 200 ms: a server older than that leaves and is started again (nothing lives on the stream for ever)
static int ray_through_server(sgp_world* w, const sgp_ray* ray, sgp_hit* hit)
{
	if (!w->ray_server_enabled) return 0;
	if (!w->ray_mb) {
		if (hipHostMalloc((void**)&w->ray_mb, sizeof(RayMailbox), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { w->ray_server_enabled = false; (void)hipGetLastError(); return 0; }
		memset(w->ray_mb, 0, sizeof(RayMailbox));      // (generation 0 = nobody: the first server is generation 1)
	}
	RayMailbox* mb = w->ray_mb;
	for (int attempt = 0; attempt < 3; ++attempt) {
		if (!w->ray_server_on) {
			// (the previous server, if any, was told to stop or left on its own; the new one queues behind it on the stream)
			++w->ray_gen;      // (servers of older generations have been told to leave, or have left; what they still write names their generation, not this one)
			RayMailbox* dmb = nullptr;
			if (hipHostGetDevicePointer((void**)&dmb, mb, 0) != hipSuccess) { w->ray_server_enabled = false; (void)hipGetLastError(); return 0; }
			launch_ray_server(w->dv, dmb, w->ray_seq, w->ray_gen, SGP_RAY_SERVER_IDLE_TICKS, SGP_RAY_SERVER_MAX_TICKS, w->stream);
			w->ray_server_on = true; w->ray_server_launches++;
		}
		mb->ray = *ray;
		const uint32_t seq = ++w->ray_seq;
		__atomic_store_n(&mb->req_seq, seq, __ATOMIC_RELEASE);
		const auto t0 = std::chrono::steady_clock::now();
		for (uint32_t spin = 0;; ++spin) {
			if (__atomic_load_n(&mb->done_seq, __ATOMIC_ACQUIRE) == seq && __atomic_load_n(&mb->done_seq2, __ATOMIC_ACQUIRE) == seq) { *hit = mb->hit; w->ray_server_rays++; return 1; }
			if ((spin & 1023u) == 1023u) {
				if (__atomic_load_n(&mb->exited_gen, __ATOMIC_ACQUIRE) == w->ray_gen) break;      // this generation's server left (idle / age) before it saw the request: start another
				if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {      // (never observed: the launch path still answers)
					ray_server_stop(w); HIP_TRY(hipStreamSynchronize(w->stream)); w->ray_server_enabled = false; return 0;
				}
			}
		}
		// the wave is gone; it may still have answered just before leaving
		if (__atomic_load_n(&mb->done_seq, __ATOMIC_ACQUIRE) == seq && __atomic_load_n(&mb->done_seq2, __ATOMIC_ACQUIRE) == seq) { *hit = mb->hit; w->ray_server_on = false; w->ray_server_rays++; return 1; }
		w->ray_server_on = false;
	}
	return 0;
}

SGP_API int sgp_raycast(sgp_world* w, const sgp_ray* rays, uint32_t n, sgp_hit* hits)
{
	if (!w || (!rays && n) || (!hits && n)) return fail(SGP_ERR_INVALID, "sgp_raycast: NULL");
	hipSetDevice(w->device);
	if (n == 1 && w->ray_server_on && w->cmds.empty() && w->ghost_refresh.empty() && !w->large_dirty && !w->large_list_dirty && w->grid_valid) {
		// the caller is tracing rays one by one (PhysicsWorld::traceRay in a loop) and nothing touched the world since the last one: hand it to the resident wave
		const int r = ray_through_server(w, rays, hits);
		if (r < 0) return r;
		if (r == 1) { finish_hit(w, hits); return SGP_OK; }
	}
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	if (!n) return SGP_OK;
	if (!w->grid_valid && w->high) {
		// poses changed since the grid was built (a step integrates after its broad phase; edits move bodies): re-bin
		const DV& d = w->dv; hipStream_t s = w->stream; const uint32_t nb = w->high;
		launch_step_begin(d, *w->h_sp, nb, false, s); w->sp_uploaded = *w->h_sp; w->sp_uploaded_valid = true;
		launch_bp_bounds(d, nb, s); launch_bp_cell(d, nb, s); launch_bp_scan(d, s); launch_bp_scatter(d, nb, s);
		w->grid_valid = true;
	}
	if (n == 1 && w->high) {
		// a single ray: start (or reach) the resident server behind the grid kernels just queued; the next single rays find it running
		const int r = ray_through_server(w, rays, hits);
		if (r < 0) return r;
		if (r == 1) { finish_hit(w, hits); return SGP_OK; }
	}
	const size_t rb = (sizeof(sgp_ray) * n + 15) & ~size_t(15);
	{ int r = ensure_stage(w, rb + sizeof(sgp_hit) * n); if (r != SGP_OK) return r; }
	memcpy(w->stage_host, rays, sizeof(sgp_ray) * n);
	if (n <= 64) {
		// a handful of rays (the facade's traceRay is n = 1): the kernel reads them from, and writes the hits to, the pinned host buffer
		// directly -- one launch and one sync instead of two copies around it
		launch_raycast(w->dv, (const sgp_ray*)w->stage_host, n, (sgp_hit*)((char*)w->stage_host + rb), w->stream);
		HIP_TRY(hipStreamSynchronize(w->stream));
	} else {
		HIP_TRY(hipMemcpyAsync(w->stage_dev, w->stage_host, sizeof(sgp_ray) * n, hipMemcpyHostToDevice, w->stream));
		sgp_hit* dh = (sgp_hit*)((char*)w->stage_dev + rb);
		launch_raycast(w->dv, (const sgp_ray*)w->stage_dev, n, dh, w->stream);
		HIP_TRY(hipMemcpyAsync((char*)w->stage_host + rb, dh, sizeof(sgp_hit) * n, hipMemcpyDeviceToHost, w->stream));
		HIP_TRY(hipStreamSynchronize(w->stream));
	}
	memcpy(hits, (char*)w->stage_host + rb, sizeof(sgp_hit) * n);
	for (uint32_t k = 0; k < n; ++k) finish_hit(w, &hits[k]);
	return SGP_OK;
}

static int ensure_query_grid(sgp_world* w)
{
	if (!w->grid_valid && w->high) {
		// poses changed since the grid was built (a step integrates after its broad phase; edits move bodies): re-bin
		const DV& d = w->dv; hipStream_t s = w->stream; const uint32_t nb = w->high;
		launch_step_begin(d, *w->h_sp, nb, false, s); w->sp_uploaded = *w->h_sp; w->sp_uploaded_valid = true;
		launch_bp_bounds(d, nb, s); launch_bp_cell(d, nb, s); launch_bp_scan(d, s); launch_bp_scatter(d, nb, s);
		w->grid_valid = true;
	}
	return SGP_OK;
}

// CharacterVirtual's CollideShape (PlayerPhysics.cpp:258-353): contacts of capsules with everything within max_separation
SGP_API int sgp_collide_capsules(sgp_world* w, const sgp_capsule_query* qs, uint32_t n, sgp_query_contact* out, uint32_t cap, uint32_t* n_out)
{
	if (!w || (!qs && n) || (!out && cap) || !n_out) return fail(SGP_ERR_INVALID, "sgp_collide_capsules: NULL");
	hipSetDevice(w->device);
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	*n_out = 0;
	if (!n) return SGP_OK;
	ensure_query_grid(w);
	const size_t qb = (sizeof(sgp_capsule_query) * n + 15) & ~size_t(15);
	const size_t ob = sizeof(sgp_query_contact) * std::max(cap, 1u);
	{ int r = ensure_stage(w, qb + ob + 16); if (r != SGP_OK) return r; }
	memcpy(w->stage_host, qs, sizeof(sgp_capsule_query) * n);
	HIP_TRY(hipMemcpyAsync(w->stage_dev, w->stage_host, sizeof(sgp_capsule_query) * n, hipMemcpyHostToDevice, w->stream));
	sgp_query_contact* dout = (sgp_query_contact*)((char*)w->stage_dev + qb);
	uint32_t* dcount = (uint32_t*)((char*)w->stage_dev + qb + ob);
	HIP_TRY(hipMemsetAsync(dcount, 0, sizeof(uint32_t), w->stream));
	launch_collide_capsules(w->dv, (const sgp_capsule_query*)w->stage_dev, n, dout, cap, dcount, w->stream);
	HIP_TRY(hipMemcpyAsync((char*)w->stage_host + qb, dout, ob + 16, hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	const uint32_t cnt = *(const uint32_t*)((char*)w->stage_host + qb + ob);
	const uint32_t m = std::min(cnt, cap);
	sgp_query_contact* h = (sgp_query_contact*)((char*)w->stage_host + qb);
	std::sort(h, h + m, [](const sgp_query_contact& a, const sgp_query_contact& b) {
		if (a.query != b.query) return a.query < b.query;
		if (a.body != b.body) return a.body < b.body;
		return a.sub_shape < b.sub_shape; });         // (the kernel leaves the contact's point index in this field)
	for (uint32_t i = 0; i < m; ++i) { h[i].userdata = w->hb[h[i].body].userdata; h[i].body = compound_id_of(w, h[i].body, &h[i].sub_shape); }
	memcpy(out, h, sizeof(sgp_query_contact) * m);
	*n_out = cnt;
	return SGP_OK;
}

SGP_API int sgp_spherecast(sgp_world* w, const sgp_ray* rays, const float* radii, uint32_t n, sgp_hit* hits)
{
	if (!w || (n && (!rays || !radii || !hits))) return fail(SGP_ERR_INVALID, "sgp_spherecast: NULL");
	hipSetDevice(w->device);
	{ int r = flush_cmds(w); if (r != SGP_OK) return r; }
	if (!n) return SGP_OK;
	ensure_query_grid(w);
	const size_t rb = (sizeof(sgp_ray) * n + 15) & ~size_t(15), fb = (sizeof(float) * n + 15) & ~size_t(15);
	{ int r = ensure_stage(w, rb + fb + sizeof(sgp_hit) * n); if (r != SGP_OK) return r; }
	memcpy(w->stage_host, rays, sizeof(sgp_ray) * n);
	memcpy((char*)w->stage_host + rb, radii, sizeof(float) * n);
	HIP_TRY(hipMemcpyAsync(w->stage_dev, w->stage_host, rb + sizeof(float) * n, hipMemcpyHostToDevice, w->stream));
	sgp_hit* dh = (sgp_hit*)((char*)w->stage_dev + rb + fb);
	launch_spherecast(w->dv, (const sgp_ray*)w->stage_dev, (const float*)((char*)w->stage_dev + rb), n, dh, w->stream);
	HIP_TRY(hipMemcpyAsync((char*)w->stage_host + rb + fb, dh, sizeof(sgp_hit) * n, hipMemcpyDeviceToHost, w->stream));
	HIP_TRY(hipStreamSynchronize(w->stream));
	memcpy(hits, (char*)w->stage_host + rb + fb, sizeof(sgp_hit) * n);
	for (uint32_t k = 0; k < n; ++k) {
		hits[k].userdata = hits[k].id != SGP_INVALID_ID ? w->hb[hits[k].id].userdata : 0;
		hits[k].sub_shape = 0;
		if (hits[k].id != SGP_INVALID_ID) hits[k].id = compound_id_of(w, hits[k].id, &hits[k].sub_shape);
	}
	return SGP_OK;
}

