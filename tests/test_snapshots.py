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
