// Network physics snapshots either side of the physics step: the ObjectPhysicsTransformUpdate wire record
// (/root/reference/gui_client/GUIClient.cpp:7633-7654, shared/Protocol.h:120) and the insertion of received snapshots into the
// physics world with render-smoothing offsets (GUIClient.cpp:7462-7490; ring of 4 and playback delay: shared/WorldObject.h:540-566,
// docs/networked physics.txt).  Host-side helpers over the facade; the batched path costs one upload + one kernel for n objects.
#pragma once
#include "PhysicsWorld.h"
#include "../../include/sgp.h"
#include <vector>
#include <cstring>
#include <algorithm>

struct PhysicsSnapshot     // WorldObject::Snapshot, shared/WorldObject.h:553-561
{
	Vec4f pos;
	Quatf rotation;
	Vec4f linear_vel;
	Vec4f angular_vel;
	double client_time;
	double local_time;
};

// Serialise the state GUIClient sends for an owned dynamic object (uid, Vec3d pos, quat, lin vel, ang vel, client time).
inline void writePhysicsTransformUpdate(uint64 uid, const Vec4f& pos, const Quatf& rot, const Vec4f& lin_vel, const Vec4f& ang_vel, double client_time, uint8_t out[SGP_PHYSICS_UPDATE_BYTES])
{
	sgp_body_state st; memset(&st, 0, sizeof(st));
	for (int i = 0; i < 3; ++i) { st.pos[i] = pos[i]; st.lin_vel[i] = lin_vel[i]; st.ang_vel[i] = ang_vel[i]; }
	for (int i = 0; i < 4; ++i) st.rot[i] = rot.v[i];
	sgp_physics_update_encode(uid, &st, client_time, out);
}

inline bool readPhysicsTransformUpdate(const uint8_t in[SGP_PHYSICS_UPDATE_BYTES], uint64& uid_out, PhysicsSnapshot& snap_out)
{
	sgp_pose_vel r; uint64_t uid = 0; double t = 0;
	if (sgp_physics_update_decode(in, &uid, &r, &t) != SGP_OK) return false;
	uid_out = uid;
	snap_out.pos = Vec4f(r.pos[0], r.pos[1], r.pos[2], 1.f);
	snap_out.rotation = Quatf(r.rot[0], r.rot[1], r.rot[2], r.rot[3]);
	snap_out.linear_vel = Vec4f(r.lin_vel[0], r.lin_vel[1], r.lin_vel[2], 0.f);
	snap_out.angular_vel = Vec4f(r.ang_vel[0], r.ang_vel[1], r.ang_vel[2], 0.f);
	snap_out.client_time = t;
	snap_out.local_time = 0;
	return true;
}

inline Quatf quatMul(const Quatf& a, const Quatf& b)
{
	return Quatf(a.v[3] * b.v[0] + a.v[0] * b.v[3] + a.v[1] * b.v[2] - a.v[2] * b.v[1],
	             a.v[3] * b.v[1] - a.v[0] * b.v[2] + a.v[1] * b.v[3] + a.v[2] * b.v[0],
	             a.v[3] * b.v[2] + a.v[0] * b.v[1] - a.v[1] * b.v[0] + a.v[2] * b.v[3],
	             a.v[3] * b.v[3] - a.v[0] * b.v[0] - a.v[1] * b.v[1] - a.v[2] * b.v[2]);
}

// Insert one snapshot per object (GUIClient.cpp:7474-7484): set pose + velocities in the physics world and compute the
// smoothing offsets that map the snapshot pose back to the pose currently rendered.
inline void insertPhysicsSnapshots(PhysicsWorld& world, const std::vector<PhysicsObject*>& obs, const std::vector<PhysicsSnapshot>& snaps)
{
	std::vector<uint32_t> ids; std::vector<sgp_pose_vel> recs;
	for (size_t i = 0; i < obs.size(); ++i) {
		PhysicsObject& ob = *obs[i];
		const PhysicsSnapshot& s = snaps[i];
		const Vec4f old_effective_pos = ob.smooth_translation + ob.pos;
		const Quatf old_effective_rot = quatMul(ob.smooth_rotation, ob.rot);
		ob.pos = s.pos; ob.rot = s.rotation;                                  // setNewObToWorldTransform, PhysicsWorld.cpp:607-620
		ob.smooth_translation = old_effective_pos - s.pos;
		ob.smooth_rotation = quatMul(old_effective_rot, Quatf(-s.rotation.v[0], -s.rotation.v[1], -s.rotation.v[2], s.rotation.v[3]));
		if (ob.jolt_body_id.IsInvalid()) continue;
		sgp_pose_vel r;
		for (int k = 0; k < 3; ++k) { r.pos[k] = s.pos[k]; r.lin_vel[k] = s.linear_vel[k]; r.ang_vel[k] = s.angular_vel[k]; }
		for (int k = 0; k < 4; ++k) r.rot[k] = s.rotation.v[k];
		ids.push_back(ob.jolt_body_id.GetIndex()); recs.push_back(r);
	}
	if (!ids.empty()) sgp_body_set_pose_vel_batch(world.world, ids.data(), recs.data(), (uint32_t)ids.size());
}

// The de-jitter buffer and its playback schedule for many objects at once: WorldObject::snapshots / next_snapshot_i /
// next_insertable_snapshot_i / transmission_time_offset (shared/WorldObject.h:540-566), filled the way ClientThread does
// (ClientThread.cpp:736-792 receive, :957-975 ownership) and drained the way GUIClient::timerEvent does (GUIClient.cpp:7462-7493:
// global_time >= client_time + transmission_time_offset + 0.1 s, one snapshot per object and frame) -- with all the snapshots that are
// due in a frame going into the physics world through ONE batched call instead of one setNewObToWorldTransform per object.
class PhysicsSnapshotQueue
{
public:
	PhysicsSnapshotQueue() : q(nullptr) { sgp_snapshot_queue_create(&q); }
	~PhysicsSnapshotQueue() { sgp_snapshot_queue_destroy(q); }
	PhysicsSnapshotQueue(const PhysicsSnapshotQueue&) = delete;
	PhysicsSnapshotQueue& operator=(const PhysicsSnapshotQueue&) = delete;

	// ClientThread, Protocol::ObjectPhysicsTransformUpdate: the 80 payload bytes as they came off the wire
	bool receive(const uint8_t msg[SGP_PHYSICS_UPDATE_BYTES], double local_time) { return sgp_snapshot_queue_push_wire(q, msg, local_time) == SGP_OK; }
	// ClientThread, Protocol::ObjectPhysicsOwnershipTaken
	void ownershipTaken(uint64 uid, double global_time_now, double last_physics_ownership_change_global_time, bool renewal)
	{
		sgp_snapshot_queue_ownership(q, uid, global_time_now, last_physics_ownership_change_global_time, renewal ? 1 : 0);
	}
	// GUIClient::timerEvent: everything that is due at global_time goes into the world; `lookup(uid)` returns the object's PhysicsObject* (or
	// null when it has none).  Returns the number of snapshots inserted.
	template <class Lookup> size_t insertDue(PhysicsWorld& world, double global_time, Lookup lookup, double padding_delay = 0.1)
	{
		// first ask how many objects have a snapshot due (cap 0: nothing is consumed), then take exactly those in one poll: a second poll after a short
		// buffer would hand out the NEXT snapshot of the objects the first one served -- two states of one object in a frame, where the reference
		// inserts at most one per object and frame (GUIClient.cpp:7469-7492)
		uint32_t n = 0;
		sgp_snapshot_queue_poll(q, global_time, padding_delay, nullptr, nullptr, 0, &n);
		if (n == 0) return 0;
		if (uids.size() < n) { uids.resize(n); recs.resize(n); }
		uint32_t due = 0;
		sgp_snapshot_queue_poll(q, global_time, padding_delay, uids.data(), recs.data(), (uint32_t)uids.size(), &due);
		n = std::min<uint32_t>(due, (uint32_t)uids.size());
		applyBatch(world, n, lookup);
		return n;
	}
	// objects that have not been heard of for a second leave the active set (GUIClient.cpp:7443-7452)
	uint32_t expire(double local_time_now, double max_age = 1.0) { uint32_t n = 0; sgp_snapshot_queue_expire(q, local_time_now, max_age, &n); return n; }
	sgp_snapshot_queue* handle() const { return q; }

private:
	template <class Lookup> void applyBatch(PhysicsWorld& world, size_t n, Lookup lookup)
	{
		std::vector<PhysicsObject*> obs; std::vector<PhysicsSnapshot> snaps;
		for (size_t i = 0; i < n; ++i) {
			PhysicsObject* ob = lookup(uids[i]);
			if (!ob) continue;
			const sgp_pose_vel& r = recs[i];
			PhysicsSnapshot s;
			s.pos = Vec4f(r.pos[0], r.pos[1], r.pos[2], 1.f); s.rotation = Quatf(r.rot[0], r.rot[1], r.rot[2], r.rot[3]);
			s.linear_vel = Vec4f(r.lin_vel[0], r.lin_vel[1], r.lin_vel[2], 0.f); s.angular_vel = Vec4f(r.ang_vel[0], r.ang_vel[1], r.ang_vel[2], 0.f);
			s.client_time = 0; s.local_time = 0;
			obs.push_back(ob); snaps.push_back(s);
		}
		insertPhysicsSnapshots(world, obs, snaps);
	}
	sgp_snapshot_queue* q;
	std::vector<uint64_t> uids; std::vector<sgp_pose_vel> recs;
};
