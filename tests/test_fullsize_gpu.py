"""BASELINE-size checks on the GPU: direct oracle parity where the oracle finishes in seconds (config 2 at 10k bodies,
the first steps of config 3 at 100k), and size-independent properties at full size (run-to-run bit reproducibility,
valid colouring, no dropped pairs, energy not created, unit quaternions, nothing below the ground)."""
import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT
import parity

pytestmark = pytest.mark.gpu


def test_config2_10k_parity_with_oracle(oracle):
    descs = scenes.config2_10k_boxes()
    tw = parity.make_twin(oracle, max_bodies=len(descs) + 64)
    tw.add_batch(descs)
    for s in range(1, 121):
        tw.step(DT)
        if s in (1, 10, 25, 40):
            d = parity.compare(tw, len(descs))
            assert d["active_mismatch"] == 0
            assert d["pos"] <= 1e-4 and d["rot"] <= 1e-4 and d["lin_vel"] <= 1e-3 and d["ang_vel"] <= 1e-3, (s, d)
    sg, sc = tw.stats()
    assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points, sg.num_colours, sg.num_colour_rounds) == \
           (sc.num_pairs, sc.num_manifolds, sc.num_contact_points, sc.num_colours, sc.num_colour_rounds)
    assert sg.num_manifolds > 5000
    tw.close()


def test_config3_100k_parity_with_oracle_120_steps(oracle):
    """100k mixed bodies: neighbours already touch at t=0 (scale up to 1.5 on a 1.5 m lattice), so every stage runs at
    full size from the first step.  The oracle runs its order-independent loops on 16 threads (bit-identical to its
    single-thread run, tests/test_oracle_threads.py) so that 120 steps take seconds."""
    import os
    descs = scenes.config3_100k_mixed()
    oracle.set_threads(min(16, os.cpu_count() or 1))
    try:
        tw = parity.make_twin(oracle, max_bodies=len(descs) + 64)
        tw.add_batch(descs)
        for s in range(1, 121):
            tw.step(DT)
            if s in (1, 6, 20, 40, 80, 120):
                d = parity.compare(tw, len(descs))
                assert d["active_mismatch"] == 0
                assert d["pos"] <= 1e-4 and d["rot"] <= 1e-4 and d["lin_vel"] <= 1e-3 and d["ang_vel"] <= 1e-3, (s, d)
                sg, sc = tw.stats()
                assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points, sg.num_colours, sg.num_colour_rounds) == \
                       (sc.num_pairs, sc.num_manifolds, sc.num_contact_points, sc.num_colours, sc.num_colour_rounds), s
        assert sg.num_manifolds > 50000 and sg.pairs_dropped == 0 and sg.manifolds_dropped == 0
        print("config3 100k, 120 steps: bit exact =", d["bit_exact"])
        tw.close()
    finally:
        oracle.set_threads(1)


def test_graph_replay_and_eager_launch_give_identical_bits():
    """The captured hipGraph of a launch plan and the eager launch sequence are the same kernels with the same arguments."""
    import subprocess, sys, os, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import numpy as np\n"
            "from substrata_amd import scenes; from substrata_amd.lib import World\n"
            "d = scenes.config2_10k_boxes(); w = World(max_bodies=len(d) + 8); w.add_batch(d)\n"
            "[w.step(1 / 60) for _ in range(200)]\n"
            "g, e, i = w.launch_counts(); import os\n"
            "print('graph replays', g, 'eager steps', e)\n"
            "assert (g > 40 and e < 160) if os.environ['SGP_NO_GRAPH'] == '0' else (g == 0 and e == 200), (g, e, i)\n"
            "np.save(sys.argv[1], w.read_states(0, len(d)))\n") % root
    outs = []
    with tempfile.TemporaryDirectory() as td:
        for flag in ("0", "1"):
            out = os.path.join(td, f"s{flag}.npy")
            env = dict(os.environ, SGP_NO_GRAPH=flag)
            subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=300)
            outs.append(np.load(out))
    for f in ("pos", "rot", "lin_vel", "ang_vel", "active"):
        assert np.array_equal(outs[0][f], outs[1][f]), f


def test_config5_1k_cars_50k_debris(oracle):
    """BASELINE config 5 at full size: parity with the oracle over the first steps (every wheel cast, the drivetrains and the
    debris contacts), then a longer GPU-only run with physical invariants."""
    import os
    from substrata_amd import abi
    descs, car_ids = scenes.config5_cars_debris()
    nc = len(car_ids)
    oracle.set_threads(min(16, os.cpu_count() or 1))
    try:
        tw = parity.make_twin(oracle, max_bodies=len(descs) + 64)
        d_cpu = descs.copy()
        ig = scenes.use_car_hull(tw.gpu, descs, car_ids); ic = scenes.use_car_hull(tw.cpu, d_cpu, car_ids)
        assert ig.hull_id == ic.hull_id and np.array_equal(descs["pos"], d_cpu["pos"])
        tw.add_batch(descs)
        for b in car_ids:
            tw.vehicle_create(tw.gpu.default_vehicle_desc(int(b)))
        for s in range(1, 25):
            tw.vehicle_set_inputs(0, scenes.config5_inputs(nc, s * DT))
            tw.step(DT)
            if s in (1, 8, 24):
                d = parity.compare(tw, len(descs))
                assert d["active_mismatch"] == 0
                assert d["pos"] <= 1e-4 and d["rot"] <= 1e-4 and d["lin_vel"] <= 1e-3 and d["ang_vel"] <= 1e-3, (s, d)
                vg, vc = tw.vehicle_get_states(0, nc)
                assert np.array_equal(vg["wheels"]["contact_body"], vc["wheels"]["contact_body"])
                assert np.array_equal(vg["wheels"]["angular_velocity"], vc["wheels"]["angular_velocity"])
                assert np.array_equal(vg["engine_rpm"], vc["engine_rpm"]) and np.array_equal(vg["current_gear"], vc["current_gear"])
        print("config5, 24 steps: bit exact =", d["bit_exact"])
        assert (vg["wheels"]["has_contact"] == 1).mean() > 0.9
    finally:
        oracle.set_threads(1)
    w = tw.gpu
    for s in range(25, 300):
        w.vehicle_set_inputs(0, scenes.config5_inputs(nc, s * DT))
        w.step(DT)
    st = w.read_states(0, len(descs))
    cars = st[1:1 + nc]
    assert np.isfinite(st["pos"]).all() and np.isfinite(st["lin_vel"]).all()
    assert (cars["pos"][:, 2] > 0.2).all() and (cars["pos"][:, 2] < 3.0).all()          # on their wheels / on debris / on their side, not through the ground
    # (the debris field is dense -- 0.76 one-metre boxes per m^2 -- so the cars mostly shove boxes around rather than travel)
    moved = np.linalg.norm(cars["pos"][:, :2] - descs["pos"][1:1 + nc, :2], axis=1)
    assert np.median(moved) > 0.5 and moved.max() < 100.0
    assert (st["pos"][1 + nc:, 2] > 0.2).all()                                            # debris rests on the ground
    vs = w.vehicle_get_states(0, nc)
    assert (vs["current_gear"] >= 1).all() and (vs["engine_rpm"] >= 1000).all() and (vs["engine_rpm"] <= 6000).all()
    assert np.median(np.abs(vs["wheels"]["angular_velocity"][:, :2])) > 5.0               # driven wheels turn
    sg = w.stats()
    assert sg.pairs_dropped == 0 and sg.manifolds_dropped == 0
    tw.close()


def mechanical_energy(descs, st):
    m = descs["mass"][1:].astype(np.float64)
    z = st["pos"][1:, 2].astype(np.float64)
    v2 = np.sum(st["lin_vel"][1:].astype(np.float64) ** 2, axis=1)
    return float(np.sum(m * 9.81 * z + 0.5 * m * v2))


def test_config3_100k_properties():
    from substrata_amd.lib import World
    descs = scenes.config3_100k_mixed()
    n = len(descs)
    runs = []
    for rep in range(2):
        w = World(max_bodies=n + 64)
        w.add_batch(descs)
        e0 = mechanical_energy(descs, w.read_states(0, n))
        for _ in range(90):
            w.step(DT)
        st = w.read_states(0, n)
        stats = w.stats()
        cons = w.dump_constraints() if rep == 0 else None
        runs.append(st)
        if rep == 0:
            assert stats.pairs_dropped == 0 and stats.manifolds_dropped == 0
            assert stats.num_manifolds > 150000 and stats.num_colours <= 40 and stats.num_overflow_constraints == 0
            for f in ("pos", "rot", "lin_vel", "ang_vel"):
                assert np.all(np.isfinite(st[f])), f
            assert np.max(np.abs(np.linalg.norm(st["rot"], axis=1) - 1.0)) < 1e-5
            # nothing tunnels through the 1 m thick ground slab.  (Light bodies, 6 kg, under a 10-deep pile of bodies up to
            # 170 kg do get pressed INTO the slab by up to a few dm with 10 PGS iterations -- identical in the oracle.)
            assert st["pos"][1:, 2].min() > -0.5
            assert np.mean(st["pos"][1:, 2] < 0.1) < 0.01
            # no energy is created (restitution < 1, friction, damping); rotational energy only adds to the right-hand side
            assert mechanical_energy(descs, st) <= e0 * (1.0 + 1e-3)
            # valid colouring at full size: constraints of one colour never share a movable body
            movable = st["active"].astype(bool)
            movable[0] = False
            assert parity.check_colouring_valid(cons, movable)
            assert len(cons) == stats.num_manifolds
            # the contact list has no duplicate pair and is sorted
            key = cons["a"].astype(np.uint64) << np.uint64(32) | cons["b"].astype(np.uint64)
            assert np.all(np.diff(key.astype(np.int64)) > 0)
        w.close()
    # run-to-run reproducibility: atomics only ever decide ORDER, never a value
    for f in ("pos", "rot", "lin_vel", "ang_vel"):
        assert np.array_equal(runs[0][f].view(np.uint32), runs[1][f].view(np.uint32)), f


def test_sleeping_at_scale_and_read_active():
    """10k boxes dropped from 5 cm settle and go to sleep; read_active / events agree with the flags."""
    from substrata_amd.lib import World
    g = scenes.ground()
    d, _ = scenes.lattice(100, 100, 1, 1.5, 0.55, seed=5, jitter=0.0, random_rot=False)
    descs = np.concatenate([g, d])
    w = World(max_bodies=len(descs) + 8)
    w.add_batch(descs)
    act = w.drain_events(abi.EVENT_ACTIVATED)
    assert len(act) == 10000
    for _ in range(100):
        w.step(DT)
    st = w.read_states(0, len(descs))
    assert st["active"][1:].sum() == 0
    assert len(w.read_active()) == 0 and len(w.read_active_view()) == 0 and len(w.read_active_poses_view()) == 0
    deact = w.drain_events(abi.EVENT_DEACTIVATED)
    assert len(deact) == 10000 and np.array_equal(np.sort(deact["id"]), np.arange(1, 10001))
    assert np.all(st["lin_vel"] == 0)
    assert abs(float(st["pos"][1:, 2].mean()) - 0.5) < 0.02
    assert w.stats().num_pairs == 0            # a sleeping world generates no work
    w.close()


def test_read_active_copy_and_view_agree_with_read_states():
    """The per-frame read-back of the active bodies: the copying call and the pinned-buffer view return the same records (as a set: the
    compaction order is not part of the contract) and they are the active rows of read_states -- also right after the active count jumped
    (the one-sync path sizes its copy from the previous step's count)."""
    from substrata_amd.lib import World
    g = scenes.ground()
    d, _ = scenes.lattice(40, 40, 2, 1.5, 0.55, seed=7, jitter=0.0, random_rot=False)
    descs = np.concatenate([g, d])
    d["activate"] = 0                                  # a second batch, asleep until something hits it
    d["pos"][:, 0] += 200.0
    w = World(max_bodies=2 * len(descs) + 8)
    w.add_batch(descs)
    w.add_batch(d)
    counts = []
    for s in range(12):
        w.step(DT)
        if s == 6:                                     # wake the sleeping batch: the active count more than doubles between two read-backs
            for i in range(len(descs), len(descs) + len(d)):
                w.activate(i)
        a = w.read_active().copy()
        v = np.array(w.read_active_view())
        st = w.read_states(0, len(descs) + len(d))
        want = st[(st["id"] != abi.INVALID_ID) & (st["active"] != 0)]
        assert len(a) == len(v) == len(want), (s, len(a), len(v), len(want))
        counts.append(len(want))
        for got in (a, v):
            o = np.argsort(got["id"])
            assert np.array_equal(got["id"][o], want["id"])
            for f in ("pos", "rot", "lin_vel", "ang_vel"):
                assert np.array_equal(got[f][o].view(np.uint32), want[f].view(np.uint32)), (s, f)
        pz = np.array(w.read_active_poses_view())          # the 32-byte records: id, position, rotation
        assert len(pz) == len(want)
        o = np.argsort(pz["id"])
        assert np.array_equal(pz["id"][o], want["id"])
        for f in ("pos", "rot"):
            assert np.array_equal(pz[f][o].view(np.uint32), want[f].view(np.uint32)), (s, f)
    assert counts[0] == 3200 and counts[-1] == 6400, counts
    w.close()


def test_raycast_grid_walk_matches_oracle_bruteforce_on_10k(oracle):
    """2048 rays (the ParticleManager cap, ParticleManager.cpp:88-96) through the settled 10k-box pile: the grid-walking
    device kernel returns the same closest hits as the oracle's brute force over all bodies."""
    descs = scenes.config2_10k_boxes()
    tw = parity.make_twin(oracle, max_bodies=len(descs) + 64)
    tw.add_batch(descs)
    for _ in range(30):
        tw.step(DT)
    rng = np.random.default_rng(9)
    n = 2048
    rays = np.zeros(n, dtype=abi.ray_dtype)
    rays["origin"] = rng.uniform(-18, 18, (n, 3)).astype(np.float32) + np.float32([0, 0, 22])
    dirs = rng.standard_normal((n, 3)).astype(np.float32)
    dirs[: n // 2, 2] = -np.abs(dirs[: n // 2, 2]) - 1.0            # half mostly downwards, half anywhere
    rays["dir"] = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    rays["max_t"] = rng.uniform(0.5, 80.0, n).astype(np.float32)
    rays["ignore_id"] = abi.INVALID_ID
    rays["ignore_id"][::7] = 0                                       # some rays ignore the ground quad
    hg, hc = tw.raycast(rays)
    assert np.array_equal(hg["id"], hc["id"])
    assert np.allclose(hg["t"], hc["t"], atol=1e-4) and np.allclose(hg["normal"], hc["normal"], atol=1e-4)
    assert 500 < np.sum(hg["id"] != abi.INVALID_ID) < n
    # edits invalidate the cached grid: move a body into a ray's path and it is hit
    one = np.zeros(1, dtype=abi.ray_dtype)
    one["origin"][0] = (300.0, 300.0, 50.0); one["dir"][0] = (0, 0, -1); one["max_t"] = 100.0; one["ignore_id"] = abi.INVALID_ID
    assert tw.gpu.raycast(one)["id"][0] == 0                         # only the ground out there
    tw.set_pose_vel(5, (300.0, 300.0, 10.0), (0, 0, 0, 1), (0, 0, 0), (0, 0, 0))
    hg, hc = tw.raycast(one)
    assert hg["id"][0] == 5 and hc["id"][0] == 5 and abs(hg["t"][0] - hc["t"][0]) < 1e-4
    tw.close()
