"""PARITY: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Tolerances (fp32, stated per test): both sides are built with -ffp-contract=off and share the constraint ordering
contract, so trajectories agree to rounding until contact-rich chaos amplifies last-bit differences of
libm-free arithmetic -- in practice they stay bit-identical; the asserted bounds are POS_TOL/VEL_TOL below.
"""
import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT, add_ground, dyn, quat_axis_angle
import parity

pytestmark = pytest.mark.gpu

POS_TOL = 1e-4     # metres / quaternion units after <= 240 steps
VEL_TOL = 1e-3     # m/s, rad/s


def run_scene(oracle, descs, steps, check_every=None, **kw):
    tw = parity.make_twin(oracle, max_bodies=max(64, len(descs) + 8), **kw)
    tw.add_batch(descs)
    worst = {"pos": 0.0, "rot": 0.0, "lin_vel": 0.0, "ang_vel": 0.0}
    exact = True
    for s in range(steps):
        tw.step(DT)
        if check_every and (s + 1) % check_every == 0 or s + 1 == steps:
            d = parity.compare(tw, len(descs))
            for k in worst:
                worst[k] = max(worst[k], d[k])
            exact = exact and d["bit_exact"]
            assert d["active_mismatch"] == 0, f"step {s + 1}: sleeping state differs"
    return tw, worst, exact


def assert_close(worst):
    assert worst["pos"] <= POS_TOL and worst["rot"] <= POS_TOL, worst
    assert worst["lin_vel"] <= VEL_TOL and worst["ang_vel"] <= VEL_TOL, worst


def test_free_fall_bit_exact(oracle):
    d = scenes.dynamic_bodies(64)
    rng = np.random.default_rng(0)
    d["pos"] = rng.uniform(-50, 50, (64, 3)).astype(np.float32) + np.float32([0, 0, 200])
    d["ang_vel"] = rng.uniform(-3, 3, (64, 3)).astype(np.float32)
    d["lin_vel"] = rng.uniform(-5, 5, (64, 3)).astype(np.float32)
    d["shape_type"] = np.arange(64) % 3
    d["shape"][:, :2] = (0.3, 0.65)
    tw, worst, exact = run_scene(oracle, d, 60, check_every=10)
    assert exact, worst
    tw.close()


def test_box_on_ground_rest_and_sleep(oracle):
    descs = np.concatenate([scenes.ground(), scenes.dynamic_bodies(1)])
    descs["pos"][1] = (0.3, -0.2, 0.8)
    descs["rot"][1] = quat_axis_angle((1, 2, 3), 0.4)
    tw, worst, exact = run_scene(oracle, descs, 240, check_every=20)
    assert_close(worst)
    sg = tw.gpu.read_states(0, 2)
    assert sg["active"][1] == 0          # asleep on both sides (active_mismatch checked every 20 steps)
    eg, ec = tw.drain_events(abi.EVENT_DEACTIVATED)
    assert np.array_equal(eg["id"], ec["id"])
    tw.close()


def test_config1_256_boxes(oracle):
    """BASELINE config 1 (256 boxes on the ground quad): per-body poses/velocities vs the oracle at steps 1, 10, 60, 240."""
    descs = scenes.config1_256_boxes()
    tw = parity.make_twin(oracle, max_bodies=1024)
    tw.add_batch(descs)
    exact_until = 0
    for s in range(1, 241):
        tw.step(DT)
        if s in (1, 10, 60, 120, 240):
            d = parity.compare(tw, len(descs))
            assert d["active_mismatch"] == 0
            assert d["pos"] <= POS_TOL and d["rot"] <= POS_TOL and d["lin_vel"] <= VEL_TOL and d["ang_vel"] <= VEL_TOL, (s, d)
            if d["bit_exact"] and exact_until == s - 1 or d["bit_exact"]:
                exact_until = s
            sg, sc = tw.stats()
            assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points, sg.num_colours, sg.num_active) == \
                   (sc.num_pairs, sc.num_manifolds, sc.num_contact_points, sc.num_colours, sc.num_active)
            assert sg.num_colour_rounds >= sc.num_colour_rounds    # the device launches rounds in batches
    print("config1 bit-exact through step", exact_until)
    tw.close()


def test_constraints_match(oracle):
    """Same contact set, same colours, same accumulated impulses after a few steps of a pile."""
    descs = scenes.config1_256_boxes()
    tw = parity.make_twin(oracle, max_bodies=1024)
    tw.add_batch(descs)
    for _ in range(45):
        tw.step(DT)
    cg, cc = parity.constraint_sets(tw)
    assert len(cg) == len(cc) and len(cg) > 100
    assert np.array_equal(cg["a"], cc["a"]) and np.array_equal(cg["b"], cc["b"])
    assert np.array_equal(cg["colour"], cc["colour"]) and np.array_equal(cg["np"], cc["np"])
    assert np.allclose(cg["n"], cc["n"], atol=1e-5)
    assert np.allclose(cg["lam_n"], cc["lam_n"], rtol=1e-3, atol=1e-2)
    movable = np.ones(1024, bool)
    movable[0] = False
    assert parity.check_colouring_valid(cg, movable)
    tw.close()


def test_small_mixed_shapes(oracle):
    """Config-3-style mix (box / sphere / capsule, scale 0.5-1.5) small enough for the oracle: 6x6x3 lattice."""
    descs = scenes.small_mixed(6, 3, seed=7)
    tw, worst, exact = run_scene(oracle, descs, 180, check_every=30)
    assert_close(worst)
    tw.close()


def test_restitution_and_friction_kats_on_gpu(oracle):
    tw = parity.make_twin(oracle, max_bodies=64)
    add_ground(tw.gpu, restitution=0.0); add_ground(tw.cpu, restitution=0.0)
    for w in (tw.gpu, tw.cpu):
        dyn(w, abi.SHAPE_SPHERE, (0.5,), pos=(0, 0, 3.0), restitution=0.8, lin_damp=0.0, ang_damp=0.0)
        dyn(w, pos=(5, 0, 0.5), lin_vel=(4, 0, 0), friction=0.5, restitution=0.0, allow_sleeping=0)
    for _ in range(200):
        tw.step(DT)
    d = parity.compare(tw, 3)
    assert d["pos"] <= POS_TOL and d["lin_vel"] <= VEL_TOL, d
    s = tw.gpu.read_states(0, 3)
    assert abs(s["pos"][2][0] - (5 + 16 / (2 * 0.5 * 9.81))) < 0.1      # slid v0^2 / (2 mu g)
    tw.close()


def test_ten_box_stack(oracle):
    descs = np.concatenate([scenes.ground(), scenes.dynamic_bodies(10, restitution=0.0)])
    descs["pos"][1:, 2] = 0.5 + np.arange(10)
    descs["allow_sleeping"][1:] = 0
    tw, worst, exact = run_scene(oracle, descs, 300, check_every=50)
    assert_close(worst)
    tw.close()


def test_buoyancy(oracle):
    tw = parity.make_twin(oracle, max_bodies=64)
    tw.set_water(True, 0.0)
    for w in (tw.gpu, tw.cpu):
        dyn(w, pos=(0, 0, 0.2), mass=510.0, allow_sleeping=0, rot=quat_axis_angle((1, 1, 0), 0.3))
        dyn(w, abi.SHAPE_SPHERE, (0.5,), pos=(3, 0, 0.1), mass=200.0, allow_sleeping=0)
        dyn(w, abi.SHAPE_CAPSULE, (0.3, 0.65), pos=(6, 0, 0.3), mass=100.0, allow_sleeping=0)
        # round 4: a tumbling capsule (ConvexShape's bounding-box stand-in, oriented) and two convex hulls (exact polyhedron under the plane)
        dyn(w, abi.SHAPE_CAPSULE, (0.25, 0.5), pos=(9, 0, 0.4), mass=60.0, allow_sleeping=0, rot=quat_axis_angle((1, 0.2, 0), 1.1), ang_vel=(0.5, 1.0, 0.0))
        rng = np.random.default_rng(5)
        for k in range(2):
            hi = w.hull_create((rng.normal(size=(14, 3)) * (0.5, 0.4, 0.3)).astype(np.float32))
            dyn(w, abi.SHAPE_HULL, (float(hi.hull_id), 0, 0, 0), pos=(12 + 3 * k, 0, 0.3), mass=120.0 + 200.0 * k, allow_sleeping=0, rot=quat_axis_angle((0.3, 1, 0.2), 0.8 * (k + 1)), ang_vel=(0.0, 0.7, 0.3))
    nb = 6
    for s_ in range(240):
        tw.step(DT)
        if s_ in (0, 10, 60):
            d = parity.compare(tw, nb)
            assert d["bit_exact"], (s_, d)
    d = parity.compare(tw, nb)
    assert d["pos"] <= POS_TOL and d["lin_vel"] <= VEL_TOL, d
    assert d["bit_exact"], d
    sg, sc = tw.gpu.read_states(0, nb), tw.cpu.read_states(0, nb)
    assert np.array_equal(sg["underwater"], sc["underwater"]) and sg["underwater"][0] == 1
    assert np.array_equal(sg["submerged_volume"].view(np.uint32), sc["submerged_volume"].view(np.uint32))
    eg, ec = tw.drain_events(abi.EVENT_ENTERED_WATER)
    assert np.array_equal(eg["id"], ec["id"]) and set(eg["id"].tolist()) == set(range(nb))      # (the tumbling capsule leaves and re-enters)
    tw.close()


def test_edits_forces_kinematic_remove(oracle):
    tw = parity.make_twin(oracle, max_bodies=64)
    add_ground(tw.gpu); add_ground(tw.cpu)
    ids = None
    for w in (tw.gpu, tw.cpu):
        a = dyn(w, pos=(0, 0, 0.5), allow_sleeping=0)
        k = dyn(w, pos=(-3, 0, 0.5), motion=abi.MOTION_KINEMATIC)
        b = dyn(w, abi.SHAPE_SPHERE, (0.5,), pos=(4, 4, 0.5))
        c = dyn(w, pos=(8, 0, 3.0))
        ids = (a, k, b, c)
    a, k, b, c = ids
    for s in range(150):
        t = (s + 1) * DT
        tw.move_kinematic(k, (-3 + 1.5 * t, 0, 0.5), (0, 0, 0, 1), DT)
        if s < 30:
            tw.add_force(b, (300.0, 0, 0)); tw.add_torque(b, (0, 0, 20.0)); tw.add_force_at(a, (0, 50.0, 0), (0.5, 0, 0.9))
        if s == 40:
            tw.remove(c)
        if s == 60:
            tw.set_pose_vel(b, (4, 4, 2.0), (0, 0, 0, 1), (0, 0, 1), (0, 0, 0)); tw.activate(b)
        if s == 80:
            tw.set_layer(b, abi.LAYER_MOVING_NON_COLLIDABLE)
        tw.step(DT)
    d = parity.compare(tw, 5)
    assert d["pos"] <= POS_TOL and d["lin_vel"] <= VEL_TOL and d["active_mismatch"] == 0, d
    s = tw.gpu.read_states(0, 5)
    assert s["id"][c] == abi.INVALID_ID            # removed
    assert s["pos"][b][2] < -1.0                   # fell through after the layer change
    assert s["pos"][a][0] > 0.5                    # pushed by the kinematic body
    for kind in (abi.EVENT_ACTIVATED, abi.EVENT_DEACTIVATED):
        eg, ec = tw.drain_events(kind)
        assert np.array_equal(eg["id"], ec["id"])
    # a new body reuses the freed slot on both sides
    ng, nc = tw.add(tw.gpu.default_body_desc())
    assert ng == nc == c
    tw.close()


def test_contact_events_match(oracle):
    descs = scenes.config1_256_boxes()[:66]
    tw = parity.make_twin(oracle, max_bodies=128)
    tw.set_contact_events(True)
    tw.add_batch(descs)
    n_added = n_pers = 0
    for s in range(40):
        tw.step(DT)
        ag, ac = tw.drain_events(abi.EVENT_CONTACT_ADDED)
        pg, pc = tw.drain_events(abi.EVENT_CONTACT_PERSISTED)
        for g, c in ((ag, ac), (pg, pc)):
            assert np.array_equal(g["id1"], c["id1"]) and np.array_equal(g["id2"], c["id2"])
            assert np.array_equal(g["num_points"], c["num_points"])
            assert np.allclose(g["base_offset"], c["base_offset"], atol=1e-4)
            assert np.allclose(g["lin_vel1"], c["lin_vel1"], atol=1e-3) and np.allclose(g["lin_vel2"], c["lin_vel2"], atol=1e-3)
        n_added += len(ag); n_pers += len(pg)
    assert n_added > 20 and n_pers > 100
    tw.close()


def test_raycast_matches(oracle):
    descs = scenes.small_mixed(5, 2, seed=11)
    tw = parity.make_twin(oracle, max_bodies=128)
    tw.add_batch(descs)
    for _ in range(90):
        tw.step(DT)
    rng = np.random.default_rng(5)
    rays = np.zeros(512, dtype=abi.ray_dtype)
    rays["origin"] = rng.uniform(-5, 5, (512, 3)).astype(np.float32) + np.float32([0, 0, 8])
    dirs = rng.standard_normal((512, 3)).astype(np.float32)
    dirs[:, 2] = -np.abs(dirs[:, 2]) - 0.5
    rays["dir"] = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    rays["max_t"] = 50.0
    rays["ignore_id"] = abi.INVALID_ID
    rays["collidable_only"][::2] = 1
    hg, hc = tw.raycast(rays)
    assert np.array_equal(hg["id"], hc["id"])
    assert np.allclose(hg["t"], hc["t"], atol=1e-4) and np.allclose(hg["normal"], hc["normal"], atol=1e-4)
    assert np.sum(hg["id"] != abi.INVALID_ID) > 400
    tw.close()


def test_empty_world_and_rejections(oracle):
    tw = parity.make_twin(oracle, max_bodies=16)
    tw.step(DT)                                   # empty world steps fine
    d = tw.gpu.default_body_desc()
    d.pos[:] = (2e9, 0, 0)
    assert tw.gpu.add(d) == abi.INVALID_ID        # PhysicsWorld.cpp:1178 silent rejection
    d = tw.gpu.default_body_desc()
    d.shape[:] = (0.5, 1e-9, 0.5, 0)
    assert tw.gpu.add(d) == abi.INVALID_ID        # :1184
    assert tw.gpu.num_bodies() == 0
    # capacity
    from substrata_amd.world import SgpError
    for _ in range(16):
        tw.gpu.add(tw.gpu.default_body_desc())
    with pytest.raises(SgpError):
        tw.gpu.add(tw.gpu.default_body_desc())
    # non-finite arguments are refused by every setter (the reference asserts them, PhysicsWorld.cpp:548-556) and leave the world untouched
    nan = float("nan")
    for call in (lambda: tw.gpu.set_pose_vel(0, (nan, 0, 0), (0, 0, 0, 1)), lambda: tw.gpu.set_pose_vel(0, (0, 0, 1), (0, 0, nan, 1)),
                 lambda: tw.gpu.set_vel(0, (0, float("inf"), 0), (0, 0, 0)), lambda: tw.gpu.set_pos(0, (0, nan, 0)),
                 lambda: tw.gpu.add_force(0, (nan, 0, 0)), lambda: tw.gpu.add_torque(0, (0, 0, nan)), lambda: tw.gpu.add_force_at(0, (1, 0, 0), (nan, 0, 0)),
                 lambda: tw.gpu.move_kinematic(0, (0, 0, nan), (0, 0, 0, 1), DT), lambda: tw.gpu.set_pose_shape(0, (0, 0, 1), (0, 0, 0, 1), (nan, 0.5, 0.5, 0))):
        with pytest.raises(SgpError):
            call()
    for _ in range(5):
        tw.gpu.step(DT)
    st = tw.gpu.read_states(0, 16)
    assert np.all(np.isfinite(st["pos"])) and np.all(np.isfinite(st["lin_vel"]))
    tw.close()


def test_incline_roll_sensor_scene(oracle):
    """Tilted gravity (friction cone), a sphere going from slip to roll, a box tipping over, and a sensor volume with its
    contact events: device vs oracle."""
    th = np.radians(35.0)
    tw = parity.make_twin(oracle, max_bodies=64, gravity=(9.81 * np.sin(th), 0.0, -9.81 * np.cos(th)))
    tw.set_contact_events(True)
    for w in (tw.gpu, tw.cpu):
        add_ground(w, friction=0.5, restitution=0.0)
        dyn(w, pos=(0, 0, 0.5), friction=0.5, restitution=0.0, allow_sleeping=0, shape=(1.0, 1.0, 0.5))
        dyn(w, abi.SHAPE_SPHERE, (0.5,), pos=(0, 4, 0.5), lin_vel=(7, 0, 0), friction=0.6, restitution=0.0, allow_sleeping=0)
        dyn(w, shape=(0.5, 0.5, 1.0), pos=(0, 8, 1.2), rot=quat_axis_angle((0, 1, 0), 0.5), friction=1.0, allow_sleeping=0)
        dyn(w, pos=(6, 4, 1.0), motion=abi.MOTION_STATIC, layer=abi.LAYER_NON_MOVING, sensor=1, activate=0, shape=(2.0, 2.0, 1.0))
        dyn(w, abi.SHAPE_CAPSULE, (0.3, 0.65), pos=(2, 12, 1.0), rot=quat_axis_angle((1, 0, 0), 1.2), allow_sleeping=0)
    added = pers = 0
    for _ in range(150):
        tw.step(DT)
        for kind in (abi.EVENT_CONTACT_ADDED, abi.EVENT_CONTACT_PERSISTED):
            eg, ec = tw.drain_events(kind)
            assert np.array_equal(eg["id1"], ec["id1"]) and np.array_equal(eg["id2"], ec["id2"]) and np.array_equal(eg["num_points"], ec["num_points"])
            if kind == abi.EVENT_CONTACT_ADDED:
                added += int(np.sum(eg["id1"] == 4) + np.sum(eg["id2"] == 4))
            else:
                pers += int(np.sum(eg["id1"] == 4) + np.sum(eg["id2"] == 4))
    d = parity.compare(tw, 6)
    assert d["pos"] <= POS_TOL and d["rot"] <= POS_TOL and d["lin_vel"] <= VEL_TOL and d["ang_vel"] <= VEL_TOL, d
    assert added >= 1 and pers >= 5          # the rolling sphere crossed the sensor volume
    tw.close()


@pytest.mark.parametrize("use_cache", [0, 1])
def test_body_pair_contact_cache_on_and_off(oracle, use_cache):
    """The body-pair contact cache (manifolds of resting polytope pairs reused instead of recomputed) on both sides of the comparison, and the
    same scene with the setting off: bit-exact either way, the same number of manifolds taken from the cache, and a teleported box is
    re-collided on both sides in the same step."""
    descs = scenes.config1_256_boxes()
    tw = parity.make_twin(oracle, max_bodies=1024, settings={"use_body_pair_contact_cache": use_cache})
    ig, ic = tw.add_batch(descs)
    seen = 0
    for s in range(1, 201):
        if s == 150:       # a box lifted by 4 mm: its pairs leave the cache for a step
            st = tw.gpu.get_state([int(ig[40])])[0]
            tw.set_pose_vel(int(ig[40]), (float(st["pos"][0]), float(st["pos"][1]), float(st["pos"][2]) + 0.004), tuple(float(x) for x in st["rot"]), (0, 0, 0), (0, 0, 0))
        tw.step(DT)
        sg, sc = tw.stats()
        assert (sg.num_manifolds, sg.num_contact_points, sg.num_cached_manifolds) == (sc.num_manifolds, sc.num_contact_points, sc.num_cached_manifolds), s
        seen = max(seen, sg.num_cached_manifolds)
        if s % 50 == 0:
            d = parity.compare(tw, len(descs))
            assert d["bit_exact"] and d["active_mismatch"] == 0, (s, d)
    assert (seen > 100) if use_cache else (seen == 0)
    tw.close()
