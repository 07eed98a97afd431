// sgp_world_snapshots.hip -- network snapshots of PhysicsObjects (host only): the wire codec of a physics update and the de-jitter queue with the
// reference's playback rule.
#include "sgp_world_internal.h"

SGP_API int sgp_physics_update_encode(uint64_t uid, const sgp_body_state* st, double client_time, uint8_t out[SGP_PHYSICS_UPDATE_BYTES])
{
	if (!st || !out) return fail(SGP_ERR_INVALID, "sgp_physics_update_encode: NULL");
	uint8_t* p = out;
	memcpy(p, &uid, 8); p += 8;
	for (int i = 0; i < 3; ++i) { const double v = (double)st->pos[i]; memcpy(p, &v, 8); p += 8; }   // Vec3d world_ob->pos
	memcpy(p, st->rot, 16); p += 16;
	memcpy(p, st->lin_vel, 12); p += 12;
	memcpy(p, st->ang_vel, 12); p += 12;
	memcpy(p, &client_time, 8);
	return SGP_OK;
}
SGP_API int sgp_physics_update_decode(const uint8_t in[SGP_PHYSICS_UPDATE_BYTES], uint64_t* uid_out, sgp_pose_vel* rec, double* client_time_out)
{
	if (!in || !rec) return fail(SGP_ERR_INVALID, "sgp_physics_update_decode: NULL");
	const uint8_t* p = in;
	if (uid_out) memcpy(uid_out, p, 8);
	p += 8;
	for (int i = 0; i < 3; ++i) { double v; memcpy(&v, p, 8); p += 8; rec->pos[i] = (float)v; }
	memcpy(rec->rot, p, 16); p += 16;
	memcpy(rec->lin_vel, p, 12); p += 12;
	memcpy(rec->ang_vel, p, 12); p += 12;
	if (client_time_out) memcpy(client_time_out, p, 8);
	for (int i = 0; i < 3; ++i) if (!std::isfinite(rec->pos[i]) || !std::isfinite(rec->lin_vel[i]) || !std::isfinite(rec->ang_vel[i])) return fail(SGP_ERR_REJECTED, "sgp_physics_update_decode: non-finite field");
	return SGP_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Network physics snapshots: the de-jitter ring and the insertion schedule (include/sgp.h has the reference map).  Host-side state only.
#include <map>
struct SnapshotRing {
	struct Entry { sgp_pose_vel rec; double client_time, local_time; };
	Entry slots[SGP_SNAPSHOT_HISTORY];
	uint32_t next_snapshot_i = 0, next_insertable_snapshot_i = 0;
	double transmission_time_offset = 0.0;
	uint32_t idle_expires = 0;      // expire() calls this ring has seen without ever holding a snapshot
};
struct sgp_snapshot_queue { std::map<uint64_t, SnapshotRing> rings; };      // ordered: the playback order is ascending uid, deterministic

SGP_API int sgp_snapshot_queue_create(sgp_snapshot_queue** out)
{
	if (!out) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_create: NULL");
	*out = new sgp_snapshot_queue();
	return SGP_OK;
}
SGP_API int sgp_snapshot_queue_destroy(sgp_snapshot_queue* q) { delete q; return SGP_OK; }

SGP_API int sgp_snapshot_queue_push(sgp_snapshot_queue* q, uint64_t uid, const sgp_pose_vel* rec, double client_time, double local_time)
{
	if (!q || !rec) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_push: NULL");
	REQUIRE_FINITE(finite3(rec->pos) && finite4(rec->rot) && finite3(rec->lin_vel) && finite3(rec->ang_vel), "sgp_snapshot_queue_push");
	SnapshotRing& r = q->rings[uid];
	SnapshotRing::Entry& e = r.slots[r.next_snapshot_i % (uint32_t)SGP_SNAPSHOT_HISTORY];      // the oldest slot is overwritten, pending or not
	e.rec = *rec; e.client_time = client_time; e.local_time = local_time;
	r.next_snapshot_i++;
	return SGP_OK;
}
SGP_API int sgp_snapshot_queue_push_wire(sgp_snapshot_queue* q, const uint8_t msg[SGP_PHYSICS_UPDATE_BYTES], double local_time)
{
	if (!q || !msg) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_push_wire: NULL");
	uint64_t uid = 0; sgp_pose_vel rec; double t = 0.0;
	{ const int r = sgp_physics_update_decode(msg, &uid, &rec, &t); if (r != SGP_OK) return r; }
	return sgp_snapshot_queue_push(q, uid, &rec, t, local_time);
}

SGP_API int sgp_snapshot_queue_ownership(sgp_snapshot_queue* q, uint64_t uid, double global_time_now, double ownership_change_global_time, int renewal)
{
	if (!q) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_ownership: NULL");
	SnapshotRing& r = q->rings[uid];
	const double offset = global_time_now - ownership_change_global_time;      // receiver's clock minus sender's clock at the same event
	if (renewal) { if (r.transmission_time_offset == 0.0) r.transmission_time_offset = offset; }
	else { r.transmission_time_offset = offset; r.next_insertable_snapshot_i = r.next_snapshot_i; }      // a new owner: what the old one queued is void
	return SGP_OK;
}

SGP_API int sgp_snapshot_queue_poll(sgp_snapshot_queue* q, double global_time, double padding_delay, uint64_t* uids_out, sgp_pose_vel* recs_out, uint32_t cap, uint32_t* n_out)
{
	if (!q || !n_out || (cap && (!uids_out || !recs_out))) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_poll: NULL");
	uint32_t n = 0;
	for (auto& kv : q->rings) {
		SnapshotRing& r = kv.second;
		if (!(r.next_insertable_snapshot_i < r.next_snapshot_i)) continue;                       // nothing pending
		const SnapshotRing::Entry& e = r.slots[r.next_insertable_snapshot_i % (uint32_t)SGP_SNAPSHOT_HISTORY];
		const double desired_insertion_time = e.client_time + r.transmission_time_offset + padding_delay;
		if (!(global_time >= desired_insertion_time)) continue;
		if (n < cap) { uids_out[n] = kv.first; recs_out[n] = e.rec; r.next_insertable_snapshot_i++; }      // (beyond cap: stays pending, reported in *n_out)
		++n;
	}
	*n_out = n;
	return SGP_OK;
}

SGP_API int sgp_snapshot_queue_expire(sgp_snapshot_queue* q, double local_time_now, double max_age, uint32_t* n_out)
{
	if (!q) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_expire: NULL");
	for (auto it = q->rings.begin(); it != q->rings.end();) {
		SnapshotRing& r = it->second;
		const bool has_any = r.next_snapshot_i > 0;
		const double last = has_any ? r.slots[(r.next_snapshot_i - 1) % (uint32_t)SGP_SNAPSHOT_HISTORY].local_time : -1.0e300;
		// (a ring that an ownership message created and no transform update ever filled has no time stamp to age by: it goes after 1024 calls -- the caller
		// expires once per frame -- so that the map cannot grow without bound on a long-running client; advisor r03)
		if (has_any ? (local_time_now - last > max_age) : (++it->second.idle_expires > 1024u)) it = q->rings.erase(it); else ++it;
	}
	if (n_out) *n_out = (uint32_t)q->rings.size();
	return SGP_OK;
}

SGP_API int sgp_snapshot_queue_peek(sgp_snapshot_queue* q, uint64_t uid, uint32_t* next_snapshot_i, uint32_t* next_insertable_snapshot_i, double* transmission_time_offset)
{
	if (!q) return fail(SGP_ERR_INVALID, "sgp_snapshot_queue_peek: NULL");
	auto it = q->rings.find(uid);
	if (it == q->rings.end()) return fail(SGP_ERR_BAD_ID, "sgp_snapshot_queue_peek: uid not tracked");
	if (next_snapshot_i) *next_snapshot_i = it->second.next_snapshot_i;
	if (next_insertable_snapshot_i) *next_insertable_snapshot_i = it->second.next_insertable_snapshot_i;
	if (transmission_time_offset) *transmission_time_offset = it->second.transmission_time_offset;
	return SGP_OK;
}

