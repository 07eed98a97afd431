"""The data format either side of the path: ObjectPhysicsTransformUpdate wire record
(/root/reference/gui_client/GUIClient.cpp:7637-7650) and batched snapshot insertion (:7474-7478)."""
import ctypes as C
import struct

import numpy as np
import pytest

from substrata_amd import abi, build, scenes
from helpers import DT


@pytest.fixture(scope="module")
def lib():
    L = C.CDLL(build.build())
    abi.bind(L, "sgp_")
    return L


def test_wire_record_layout_and_round_trip(lib):
    st = abi.BodyState()
    st.pos[:] = (1.5, -2.25, 3.0)
    st.rot[:] = (0.1, 0.2, 0.3, 0.9273618)
    st.lin_vel[:] = (4.0, 5.0, -6.0)
    st.ang_vel[:] = (0.5, 0.25, -0.125)
    buf = (C.c_uint8 * abi.PHYSICS_UPDATE_BYTES)()
    assert lib.sgp_physics_update_encode(0x1122334455667788, C.byref(st), 1234.5, buf) == abi.OK
    raw = bytes(buf)
    # little endian: uid u64 | pos 3 x f64 | quat 4 x f32 | lin vel 3 x f32 | ang vel 3 x f32 | client time f64
    fields = struct.unpack("<Q3d4f3f3fd", raw)
    assert len(raw) == 80 and fields[0] == 0x1122334455667788
    assert fields[1:4] == (1.5, -2.25, 3.0)
    assert np.allclose(fields[4:8], (0.1, 0.2, 0.3, 0.9273618)) and fields[8:11] == (4.0, 5.0, -6.0)
    assert fields[11:14] == (0.5, 0.25, -0.125) and fields[14] == 1234.5
    uid, rec, t = C.c_uint64(), abi.PoseVel(), C.c_double()
    assert lib.sgp_physics_update_decode(buf, C.byref(uid), C.byref(rec), C.byref(t)) == abi.OK
    assert uid.value == 0x1122334455667788 and t.value == 1234.5
    assert tuple(rec.pos) == (1.5, -2.25, 3.0) and tuple(rec.lin_vel) == (4.0, 5.0, -6.0)
    # a NaN on the wire is rejected, like addObject's finite checks (PhysicsWorld.cpp:1171-1189)
    bad = bytearray(raw)
    bad[8:16] = struct.pack("<d", float("nan"))
    assert lib.sgp_physics_update_decode((C.c_uint8 * 80).from_buffer(bad), None, C.byref(rec), None) == abi.ERR_REJECTED


@pytest.mark.gpu
def test_batched_snapshot_insertion_matches_single_calls(oracle):
    import parity
    descs = scenes.config1_256_boxes()
    tw = parity.make_twin(oracle, max_bodies=512)
    tw.add_batch(descs)
    for _ in range(30):
        tw.step(DT)
    rng = np.random.default_rng(2)
    ids = np.arange(1, 129, dtype=np.uint32)
    recs = np.zeros(len(ids), dtype=abi.pose_vel_dtype)
    recs["pos"] = rng.uniform(-8, 8, (len(ids), 3)).astype(np.float32) + np.float32([0, 0, 12])
    q = rng.standard_normal((len(ids), 4)).astype(np.float32)
    recs["rot"] = q / np.linalg.norm(q, axis=1, keepdims=True)
    recs["lin_vel"] = rng.uniform(-2, 2, (len(ids), 3)).astype(np.float32)
    recs["ang_vel"] = rng.uniform(-1, 1, (len(ids), 3)).astype(np.float32)
    tw.gpu.set_pose_vel_batch(ids, recs)          # one upload + one kernel
    tw.cpu.set_pose_vel_batch(ids, recs)          # falls back to n single calls on the checker
    sg = tw.gpu.get_state(ids)
    assert np.array_equal(sg["pos"], recs["pos"]) and np.array_equal(sg["lin_vel"], recs["lin_vel"])
    for _ in range(30):
        tw.step(DT)
    d = parity.compare(tw, len(descs))
    assert d["pos"] <= 1e-4 and d["lin_vel"] <= 1e-3 and d["active_mismatch"] == 0, d
    tw.close()


# ---- the de-jitter ring and the insertion schedule (shared/WorldObject.h:540-566, ClientThread.cpp:736-792,957-975, GUIClient.cpp:7443-7493)

def wire(uid, pos, t, vel=(0.0, 0.0, 0.0)):
    return struct.pack("<Q3d4f3f3fd", uid, *[float(x) for x in pos], 0.0, 0.0, 0.0, 1.0, *[float(x) for x in vel], 0.0, 0.0, 0.0, float(t))


def test_ring_schedule_and_overflow():
    from substrata_amd.lib import SnapshotQueue
    q = SnapshotQueue()
    uid = 42
    # the owner's clock runs 100 s behind ours: the ownership message was sent at its global time 7.0 and arrives at our 107.0
    q.ownership(uid, 107.0, 7.0, renewal=False)
    assert q.peek(uid) == (0, 0, 100.0)
    # snapshots taken at the owner's times 7.1, 7.2, 7.3 (x = 1, 2, 3) arrive with jitter
    for k, (ct, lt) in enumerate([(7.1, 107.13), (7.2, 107.21), (7.3, 107.38)]):
        q.push_wire(wire(uid, (k + 1.0, 0, 0), ct), lt)
    assert q.peek(uid)[:2] == (3, 0)
    # playback time of snapshot i = client_time + offset + 0.1: 107.2, 107.3, 107.4 -- at most one per poll, in order
    assert q.poll(107.19)[2] == 0
    u, r, n = q.poll(107.2)
    assert n == 1 and int(u[0]) == uid and float(r["pos"][0][0]) == 1.0
    assert q.poll(107.25)[2] == 0                             # the second is not due yet
    u, r, n = q.poll(107.9)                                   # late frame: still ONE snapshot per object and poll (GUIClient.cpp:7469-7492)
    assert n == 1 and float(r["pos"][0][0]) == 2.0
    u, r, n = q.poll(107.9)
    assert n == 1 and float(r["pos"][0][0]) == 3.0
    assert q.poll(200.0)[2] == 0 and q.peek(uid)[:2] == (3, 3)
    # a renewal keeps the offset it has; a new owner replaces it and drops what is queued
    q.push_wire(wire(uid, (4, 0, 0), 7.4), 107.45)
    q.ownership(uid, 150.0, 7.45, renewal=True)
    assert q.peek(uid) == (4, 3, 100.0)
    q.ownership(uid, 108.0, 58.0, renewal=False)
    assert q.peek(uid) == (4, 4, 50.0) and q.poll(1e9)[2] == 0
    # overflow: six snapshots arrive before any is played back; the ring has four slots, so the indices 4, 5 overwrite 0, 1 and playback
    # (which walks next_insertable_snapshot_i from 4 upward and reads slot i % 4) plays x = 14, 15, 12, 13, 14, 15 -- what the reference's
    # indexing yields; restated, not "fixed"
    for k in range(6):
        q.push_wire(wire(uid, (10.0 + k, 0, 0), 60.0 + 0.1 * k), 110.0 + 0.1 * k)
    played = []
    while True:
        u, r, n = q.poll(1e9)
        if n == 0:
            break
        played.append(float(r["pos"][0][0]))
    assert played == [14.0, 15.0, 12.0, 13.0, 14.0, 15.0]
    # out-of-order arrival: slots are filled in ARRIVAL order and played back in that order (the schedule looks at one slot only)
    uid2 = 7
    q.ownership(uid2, 10.0, 10.0, renewal=False)
    q.push_wire(wire(uid2, (2, 0, 0), 10.2), 10.25)
    q.push_wire(wire(uid2, (1, 0, 0), 10.1), 10.26)
    assert q.poll(10.25)[2] == 0                                   # the head (taken at 10.2) is due at 10.3, and it blocks the older one behind it
    u, r, n = q.poll(10.3)
    assert n == 1 and float(r["pos"][0][0]) == 2.0
    u, r, n = q.poll(10.3)
    assert n == 1 and float(r["pos"][0][0]) == 1.0
    # several objects due in one poll come back in ascending uid; expiry drops objects not heard of for a second
    q.push_wire(wire(99, (0, 0, 0), 0.0), 111.0)
    q.push_wire(wire(uid, (0, 0, 0), 0.0), 111.0)
    u, r, n = q.poll(1e9)
    assert [int(x) for x in u] == [42, 99]
    assert q.expire(111.5) == 2                                    # uid2's last arrival (10.26) is older than a second
    assert q.expire(113.0) == 0
    # a non-finite snapshot is refused like any other pose edit
    with pytest.raises(Exception):
        q.push_wire(wire(5, (float("nan"), 0, 0), 1.0), 1.0)
    q.close()


@pytest.mark.gpu
def test_streamed_snapshots_drive_remote_bodies_like_the_oracle(oracle):
    """256 remote-owned boxes whose owner streams an 80-byte ObjectPhysicsTransformUpdate every 0.1 s; the records reach us through a jittery,
    occasionally lossy channel, go through decode -> ring -> timed insertion, and every frame's due snapshots enter the device world through ONE
    sgp_body_set_pose_vel_batch; the oracle world gets the same snapshots through single calls.  Both worlds step 1/60 s in between (the bodies
    also collide with each other and with 64 local boxes), and must agree bit for bit."""
    import parity
    from substrata_amd.lib import SnapshotQueue
    rng = np.random.default_rng(11)
    descs = scenes.config1_256_boxes()
    local = scenes.dynamic_bodies(64)
    local["shape_type"] = abi.SHAPE_BOX; local["shape"][:, :3] = 0.5
    local["pos"] = rng.uniform(-6, 6, (64, 3)).astype(np.float32) + np.float32([0, 0, 14])
    descs = np.concatenate([descs, local])
    tw = parity.make_twin(oracle, max_bodies=512)
    tw.add_batch(descs)
    remote_ids = np.arange(1, 257, dtype=np.uint32)              # body ids of the remote-owned boxes; their uids are 1000 + id
    uid_of = {int(i): 1000 + int(i) for i in remote_ids}
    id_of = {v: k for k, v in uid_of.items()}
    q = SnapshotQueue()
    CLOCK_SKEW = 37.5                                             # our global time minus the owner's
    for i in remote_ids:
        q.ownership(uid_of[int(i)], CLOCK_SKEW + 0.02, 0.0, renewal=False)      # the ownership message took 20 ms
    # the owner's trajectory: each box circles at its own radius and height; a snapshot every 0.1 s of the owner's clock
    phase = rng.uniform(0, 6.28, 256); radius = rng.uniform(2, 9, 256); height = rng.uniform(1, 9, 256)

    def owner_state(k, t):
        a = phase[k] + 0.7 * t
        pos = (radius[k] * np.cos(a), radius[k] * np.sin(a), height[k])
        vel = (-0.7 * radius[k] * np.sin(a), 0.7 * radius[k] * np.cos(a), 0.0)
        return pos, vel
    in_flight = []                                                # (arrival time on our clock, message bytes)
    for n_snap in range(1, 40):
        t_owner = 0.1 * n_snap
        for k, i in enumerate(remote_ids):
            if rng.random() < 0.03:
                continue                                          # lost
            pos, vel = owner_state(k, t_owner)
            delay = 0.02 + float(rng.exponential(0.03))           # jitter; some arrive after their successor was sent
            in_flight.append((t_owner + CLOCK_SKEW + delay, wire(uid_of[int(i)], pos, t_owner, vel)))
    in_flight.sort(key=lambda e: e[0])
    now, cursor, inserted, max_batch = CLOCK_SKEW, 0, 0, 0
    for frame in range(270):
        now += DT
        while cursor < len(in_flight) and in_flight[cursor][0] <= now:
            q.push_wire(in_flight[cursor][1], in_flight[cursor][0]); cursor += 1
        uids, recs, n = q.poll(now)
        assert n == len(uids)
        if n:
            ids = np.array([id_of[int(u)] for u in uids], dtype=np.uint32)
            tw.gpu.set_pose_vel_batch(ids, recs)
            tw.cpu.set_pose_vel_batch(ids, recs)
            inserted += n; max_batch = max(max_batch, n)
        tw.step(DT)
        if frame % 30 == 29:
            d = parity.compare(tw, len(descs))
            assert d["bit_exact"] and d["active_mismatch"] == 0, (frame, d)
    assert inserted > 256 * 30 and max_batch > 64                 # the stream really went through the ring, in batches
    # the remote boxes are where their owner's last played-back snapshots (plus a fraction of a second of free flight) put them
    st = tw.gpu.get_state(remote_ids)
    r_now = np.hypot(st["pos"][:, 0], st["pos"][:, 1])
    assert np.median(np.abs(r_now - radius)) < 1.0
    q.close()
    tw.close()
