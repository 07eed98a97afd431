"""Pinning the oracle against the REAL reference arithmetic: JoltPhysics v5.3.0 behind oracle/jolt_ref/oracle_jolt.cpp (PhysicsWorld's
constructor, addObject and think() restated over the real Jolt API).  Jolt is not vendored in the reference tree and cannot be fetched in
the authoring environment, so without `SGP_JOLT_DIR=... make -C oracle jolt_ref` these tests SKIP and the parity of this repo stays
"unpinned" (DESIGN.md section 5).  With the binary present they are the tolerance statement of BASELINE.json's north_star:

  * before any contact the two integrators must agree to 1e-5 relative (same semi-implicit Euler, same damping, same rotation step);
  * after contacts trajectories of a pile diverge chaotically in any two implementations, so aggregates are compared: mean height,
    deepest penetration, kinetic energy, number of sleeping bodies."""
import numpy as np
import pytest

import jolt_ref_io
from substrata_amd import abi, scenes
from helpers import DT

needs_jolt = pytest.mark.skipif(not jolt_ref_io.available(), reason="PARITY UNPINNED: oracle/_ref/oracle_jolt is not built (needs a JoltPhysics v5.3.0 "
                                "checkout: SGP_JOLT_DIR=... make -C oracle jolt_ref); the oracle is pinned by analytic KATs only")


def test_scene_and_dump_formats_round_trip(tmp_path):
    """The file formats themselves (runs everywhere): what the Jolt program reads is what write_scene writes."""
    d = scenes.config1_256_boxes()
    p = tmp_path / "s.bin"
    jolt_ref_io.write_scene(str(p), d)
    raw = p.read_bytes()
    assert np.frombuffer(raw[:8], np.uint32).tolist() == [0x4A504753, len(d)] and len(raw) == 8 + len(d) * abi.body_desc_dtype.itemsize
    assert np.array_equal(np.frombuffer(raw[8:], dtype=abi.body_desc_dtype)["pos"], d["pos"])
    st = np.zeros(3, dtype=abi.body_state_dtype); st["pos"] = [[1, 2, 3], [4, 5, 6], [7, 8, 9]]
    q = tmp_path / "d.bin"
    q.write_bytes(np.array([0x44504753, 3, 2], np.uint32).tobytes() + np.uint32(10).tobytes() + st.tobytes() + np.uint32(60).tobytes() + st.tobytes())
    got = jolt_ref_io.read_dump(str(q))
    assert sorted(got) == [10, 60] and np.array_equal(got[60]["pos"], st["pos"])


def test_oracle_jolt_driver_still_parses():
    """The real-Jolt driver has never met Jolt here (its sources are absent), so at least keep it from rotting: it must parse against
    declarations of exactly the Jolt names it uses (oracle/jolt_ref/syntax_mock: declarations only, nothing that could be linked or run).
    This proves nothing about parity -- the B1 baseline and the north-star tolerance stay blocked on a JoltPhysics v5.3.0 checkout."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I", os.path.join(root, "oracle", "jolt_ref", "syntax_mock"),
                        os.path.join(root, "oracle", "jolt_ref", "oracle_jolt.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


@needs_jolt
def test_free_flight_matches_jolt_to_1e5(tmp_path, oracle):
    """Bodies far apart, spinning, damped, under gravity: no contact for 40 steps -> the integrators agree to 1e-5 relative."""
    d = scenes.small_mixed(5, 2, seed=11)
    d["pos"][1:, :2] *= 4.0; d["pos"][1:, 2] += 60.0
    rng = np.random.default_rng(3)
    d["ang_vel"][1:] = rng.normal(size=(len(d) - 1, 3)); d["lin_vel"][1:] = rng.normal(size=(len(d) - 1, 3)) * 2
    ref, _ = jolt_ref_io.run(d, 40, [1, 10, 40], str(tmp_path))
    w = oracle.OracleWorld(max_bodies=len(d) + 8); w.add_batch(d)
    for s in range(1, 41):
        w.step(DT)
        if s in ref:
            mine = w.read_states(0, len(d))
            for f in ("pos", "lin_vel", "ang_vel"):
                scale = np.maximum(1.0, np.abs(ref[s][f]))
                assert np.max(np.abs(mine[f] - ref[s][f]) / scale) < 1e-5, (s, f)
            dots = np.abs(np.sum(mine["rot"] * ref[s]["rot"], axis=1))
            assert np.min(dots) > 1.0 - 1e-5, s


@needs_jolt
@pytest.mark.parametrize("config", ["config1", "config2"])
def test_pile_aggregates_match_jolt(tmp_path, oracle, config):
    d = scenes.config1_256_boxes() if config == "config1" else scenes.config2_10k_boxes()
    ref, timing = jolt_ref_io.run(d, 240, [60, 240], str(tmp_path))
    oracle.set_threads(8)
    try:
        w = oracle.OracleWorld(max_bodies=len(d) + 8); w.add_batch(d)
        for s in range(1, 241):
            w.step(DT)
            if s in ref:
                mine = w.read_states(0, len(d))
                zr, zm = ref[s]["pos"][1:, 2], mine["pos"][1:, 2]
                assert abs(zr.mean() - zm.mean()) < 0.05 * max(1.0, abs(zr.mean())), (s, zr.mean(), zm.mean())
                assert zm.min() > zr.min() - 0.05                                  # no deeper into the ground than Jolt lets them
                ker = 0.5 * 50.0 * np.sum(ref[s]["lin_vel"][1:] ** 2); kem = 0.5 * 50.0 * np.sum(mine["lin_vel"][1:] ** 2)
                assert abs(ker - kem) < 0.25 * max(ker, kem) + 5.0 * len(d) * 1e-3, (s, ker, kem)
                if s == 240:
                    assert abs(int((ref[s]["active"][1:] == 0).sum()) - int((mine["active"][1:] == 0).sum())) <= 0.15 * (len(d) - 1) + 4
    finally:
        oracle.set_threads(1)
    print(config, "Jolt:", timing)
