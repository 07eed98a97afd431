"""In-step activation on the device against the CPU statement, bit for bit: what a contact (or a wheel) wakes takes its sleeping island along and
collides in the step that woke it (PhysicsSystem::JobFindCollisions; oracle/sgo_oracle.c find_contacts, k_wake_pairs on the device).
The analytic side -- what the behaviour must be -- is pinned on the oracle in tests/test_oracle_kat.py."""
import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT, add_ground, dyn, add_car
import parity
from test_mesh_parity_gpu import grid_mesh, mesh_body

pytestmark = pytest.mark.gpu


def _exact(tw, n, what):
    d = parity.compare(tw, n)
    assert d["active_mismatch"] == 0 and d["bit_exact"], (what, d)


def _both(tw, fn):
    out = [fn(w) for w in (tw.gpu, tw.cpu)]
    assert out[0] == out[1]
    return out[0]


def test_ball_on_a_sleeping_stack_wakes_all_of_it_at_once(oracle):
    tw = parity.make_twin(oracle, max_bodies=64)
    n = 5
    ids = _both(tw, lambda w: (add_ground(w), [dyn(w, pos=(0.02 * k, 0, 0.5 + 1.0 * k)) for k in range(n)])[1])
    for s in range(400):
        tw.step(DT)
    _exact(tw, n + 1, "settled")
    assert not any(s["active"] for s in tw.gpu.get_state(ids))
    ball = _both(tw, lambda w: dyn(w, abi.SHAPE_SPHERE, (0.25,), pos=(0.1, 0.05, n + 2.0), mass=5.0))
    touched = None
    for s in range(120):
        tw.step(DT)
        sg, sc = tw.gpu.stats(), tw.cpu.stats()
        assert (sg.num_pairs, sg.num_wake_pairs, sg.num_manifolds) == (sc.num_pairs, sc.num_wake_pairs, sc.num_manifolds), s
        _exact(tw, n + 2, f"step {s}")
        if touched is None and tw.gpu.get_state([ids[-1]])[0]["active"]:
            touched = s
            assert all(x["active"] for x in tw.gpu.get_state(ids))       # the whole island, in the step of the first touch
            assert sg.num_wake_pairs >= n and sg.num_manifolds == n + 1          # ball - box, four box - box, box - ground
    assert touched is not None
    tw.close()


def test_sleeping_pile_on_a_mesh_with_hulls_is_woken_by_a_thrown_box(oracle):
    """the second narrow-phase round through all three kernels: primitive pairs, hull pairs and (body, mesh) pairs of the woken bodies"""
    rng = np.random.default_rng(5)
    tw = parity.make_twin(oracle, max_bodies=512)
    V, T = grid_mesh(17, 12.0, lambda x, y: 0.05 * np.sin(0.7 * x) * np.cos(0.6 * y))
    ig, ic = tw.mesh_create(V, T)
    tw.add_batch(mesh_body(ig))
    hg, hc = tw.hull_create(rng.normal(size=(14, 3)) * 0.45)
    n = 60
    d = scenes.dynamic_bodies(n)
    d["pos"] = np.column_stack([rng.uniform(-2.0, 2.0, n), rng.uniform(-2.0, 2.0, n), 0.6 + 0.9 * np.arange(n) / 3.0]).astype(np.float32)
    d["shape_type"] = np.arange(n) % 4
    d["shape"][:, :3] = (0.35, 0.4, 0.3)
    hull = d["shape_type"] == abi.SHAPE_HULL
    d["shape"][hull, 0] = float(hg.hull_id); d["shape"][hull, 1:3] = 0.0
    d["friction"] = 0.8
    d["angular_damping"] = 0.6
    ids_g, ids_c = tw.add_batch(d)
    assert np.array_equal(ids_g, ids_c)
    nb = int(ids_g.max()) + 1
    asleep = False
    for s in range(2400):      # (round 6: the summed warm start changes the pile's chaotic path; it is down to the bodies that rolled off the terrain by step ~1300)
        tw.step(DT)
        if s % 50 == 49 and tw.gpu.stats().num_active <= 6:       # (a sphere or two keep rolling in the terrain's hollows, three fall for ever beside the terrain)
            asleep = True
            break
    _exact(tw, nb, "settled")
    assert asleep, "the pile never went to sleep"
    thrown = _both(tw, lambda w: dyn(w, pos=(-9.0, 0.0, 1.2), mass=200.0))
    tw.set_vel(thrown, (14.0, 0.0, 1.0), (0, 0, 0))
    woke_many = 0
    for s in range(90):
        tw.step(DT)
        sg, sc = tw.gpu.stats(), tw.cpu.stats()
        assert (sg.num_pairs, sg.num_wake_pairs, sg.num_manifolds, sg.num_active) == (sc.num_pairs, sc.num_wake_pairs, sc.num_manifolds, sc.num_active), s
        assert sg.pairs_dropped == 0 and sg.manifolds_dropped == 0
        _exact(tw, nb + 1, f"step {s}")
        woke_many = max(woke_many, sg.num_wake_pairs)
    assert woke_many > 20, woke_many
    tw.close()


def test_a_wheel_wakes_the_island_under_it(oracle):
    """a car rolls onto a sleeping plate that carries sleeping boxes: the wheel wakes the plate (VehicleConstraint::BuildIslands), the plate's island follows in the same step"""
    tw = parity.make_twin(oracle, max_bodies=64)
    recs = []
    for w in (tw.gpu, tw.cpu):
        add_ground(w)
        plate = dyn(w, shape=(3.0, 6.0, 0.1, 0.0), pos=(0, 4.0, 0.1), mass=4000.0, friction=0.8)
        boxes = [dyn(w, shape=(0.3, 0.3, 0.3, 0.0), pos=(1.5, 6.0 + 0.8 * k, 0.5), mass=20.0) for k in range(3)]
        body, vid = add_car(w, pos=(0, -6.0, 0.95))
        recs.append((plate, boxes, body, vid))
    assert recs[0] == recs[1]
    plate, boxes, body, vid = recs[0]
    nb = body + 1
    for s in range(500):
        tw.step(DT)
    _exact(tw, nb, "asleep")
    assert tw.gpu.stats().num_active == 0
    tw.vehicle_set_input(vid, forward=1.0)
    woke = None
    for s in range(420):
        tw.step(DT)
        _exact(tw, nb, f"step {s}")
        if woke is None and tw.gpu.get_state([plate])[0]["active"]:
            woke = s
            assert all(x["active"] for x in tw.gpu.get_state(boxes)), "the boxes on the plate sleep on while the plate is awake"
            assert tw.gpu.stats().num_wake_pairs == tw.cpu.stats().num_wake_pairs > 0
    assert woke is not None, "the car never reached the plate"
    tw.close()


def test_stale_label_after_slot_reuse_matches_the_oracle(oracle):
    """the label generations of round 6 on the device: the scene of tests/test_oracle_kat2.py (a body created in the slot of a removed island root), bit for bit"""
    from test_oracle_kat2 import stale_label_scene
    tw = parity.make_twin(oracle, max_bodies=64)
    out = [stale_label_scene(w) for w in (tw.gpu, tw.cpu)]
    assert out[0] == out[1]
    rest, new = out[0]
    _both(tw, lambda w: dyn(w, abi.SHAPE_SPHERE, (0.2,), pos=(20.0, 0.0, 1.6), mass=5.0))
    for s in range(90):
        tw.step(DT)
        _exact(tw, 8, f"step {s}")
        assert not any(x["active"] for x in tw.gpu.get_state(rest))
    # a ball on the old stack: its boxes' label names a slot that has since gone to another body, so the box the ball touches wakes ALONE -- and must still meet, in the
    # step that wakes it, what was not awake when the step began (the box below, the ground): the device derived "woken" from the label's mark only and left such a body
    # without its resting contacts for a step (tools/fuzz_tiles.py seed 23)
    _both(tw, lambda w: dyn(w, abi.SHAPE_SPHERE, (0.2,), pos=(0.0, 0.0, 2.4), mass=5.0))
    for s in range(60):
        tw.step(DT)
        _exact(tw, 8, f"second ball, step {s}")
        if tw.gpu.stats().num_active == 0:
            continue      # (everything asleep again: the debug views differ in what they show of a step nobody was awake after)
        cg, cc = parity.constraint_sets(tw)
        assert sorted((int(c["a"]), int(c["b"])) for c in cg) == sorted((int(c["a"]), int(c["b"])) for c in cc), f"second ball, step {s}: constraint pairs differ"
    tw.close()


@pytest.mark.parametrize("somebody_awake", [False, True])
def test_a_pile_that_fell_asleep_as_a_whole_wakes_with_its_contacts(oracle, somebody_awake):
    """A step nobody is awake in leaves the contact cache as it found it (StepCounters::any_awake: no table wipe, no rebuild, the buffer parity goes back; host: the
    skipped steps): the scene of tests/test_oracle_kat2.py on the device, bit for bit -- counts, events, states -- through the nudge and 60 steps after it; then the
    same again after edits that wake nobody (real steps with nobody awake: the device's path, not the host's skip).
    somebody_awake: a kinematic body keeps the world awake all along -- the pile's contacts are carried over from step to step (k_cache_build)."""
    from test_oracle_kat2 import sleeping_pile_scene
    tw = parity.make_twin(oracle, max_bodies=16)
    _both(tw, lambda w: w.set_contact_events(True))
    out = [sleeping_pile_scene(w, somebody_awake) for w in (tw.gpu, tw.cpu)]
    assert out[0] == out[1]
    ids = out[0]
    for rnd in range(2):
        for w in (tw.gpu, tw.cpu):
            for k in (abi.EVENT_CONTACT_ADDED, abi.EVENT_CONTACT_PERSISTED):
                w.drain_events(k)
        _both(tw, lambda w: (w.set_vel(ids[2], (0.02, 0.0, 0.0), (0.0, 0.0, 0.0)), w.activate(ids[2])))
        tw.step(DT)
        _exact(tw, 8, f"round {rnd}: the step that wakes the pile")
        sg, sc = tw.gpu.stats(), tw.cpu.stats()
        assert (sg.num_manifolds, sg.num_cached_manifolds) == (sc.num_manifolds, sc.num_cached_manifolds) == (3, 3), (rnd, sg.num_manifolds, sg.num_cached_manifolds, sc.num_manifolds, sc.num_cached_manifolds)
        for k, n in ((abi.EVENT_CONTACT_ADDED, 0), (abi.EVENT_CONTACT_PERSISTED, 3)):
            assert len(tw.gpu.drain_events(k)) == len(tw.cpu.drain_events(k)) == n, (rnd, k)
        for s in range(400):
            tw.step(DT)
            _exact(tw, 8, f"round {rnd}, step {s} after the nudge")
        assert not any(x["active"] for x in tw.gpu.get_state(ids))
        assert tw.gpu.stats().num_manifolds == tw.cpu.stats().num_manifolds == 0
        if rnd == 0:
            # edits that wake nobody (a static box far away, then its removal): real steps, not the host's skipped ones, with nobody awake
            far = [dyn(w, pos=(50.0, 50.0, 5.0), motion=abi.MOTION_STATIC, layer=abi.LAYER_NON_MOVING, activate=0) for w in (tw.gpu, tw.cpu)]
            assert far[0] == far[1]
            for s in range(3):
                tw.step(DT)
            _both(tw, lambda w: w.remove(far[0]))
            for s in range(5):
                tw.step(DT)
    tw.close()


@pytest.mark.parametrize("somebody_awake", [False, True])
def test_kept_contacts_of_a_removed_body_do_not_outlive_it(oracle, somebody_awake):
    """The cache's kept entries are keyed by body slots: the middle box of the sleeping pile is removed and another body is created in its slot, asleep, elsewhere --
    the entries of the old occupant must go (not alive / created since), on both sides alike; then the rest of the pile is woken and everything is compared bit for bit,
    events included."""
    from test_oracle_kat2 import sleeping_pile_scene
    tw = parity.make_twin(oracle, max_bodies=16)
    _both(tw, lambda w: w.set_contact_events(True))
    out = [sleeping_pile_scene(w, somebody_awake) for w in (tw.gpu, tw.cpu)]
    assert out[0] == out[1]
    ids = out[0]
    _both(tw, lambda w: w.remove(ids[1]))
    new = [dyn(w, pos=(6.0, 0.0, 0.5), activate=0) for w in (tw.gpu, tw.cpu)]
    assert new[0] == new[1] == ids[1]                                            # the freed slot is handed out again
    for s in range(3):
        tw.step(DT)
        _exact(tw, 8, f"after the swap, step {s}")
    for w in (tw.gpu, tw.cpu):
        for k in (abi.EVENT_CONTACT_ADDED, abi.EVENT_CONTACT_PERSISTED):
            w.drain_events(k)
    _both(tw, lambda w: (w.activate(ids[2]), w.activate(new[0])))                # the top box falls onto the bottom one; the newcomer wakes where it lies
    for s in range(240):
        tw.step(DT)
        _exact(tw, 8, f"woken, step {s}")
        for k in (abi.EVENT_CONTACT_ADDED, abi.EVENT_CONTACT_PERSISTED):
            eg, ec = tw.gpu.drain_events(k), tw.cpu.drain_events(k)
            assert sorted((int(e["id1"]), int(e["id2"])) for e in eg) == sorted((int(e["id1"]), int(e["id2"])) for e in ec), (s, k)
        sg, sc = tw.gpu.stats(), tw.cpu.stats()
        assert (sg.num_manifolds, sg.num_cached_manifolds) == (sc.num_manifolds, sc.num_cached_manifolds), (s, sg.num_manifolds, sc.num_manifolds, sg.num_cached_manifolds, sc.num_cached_manifolds)
    tw.close()

