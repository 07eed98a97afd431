"""More closed-form answers pinning the oracle's dynamics (friction cone, rotational inertia, impulse direction).
All analytic; none needs the reference (which holds no physics fixtures)."""
import numpy as np
import pytest

from substrata_amd import abi
from helpers import DT, add_ground, dyn, quat_axis_angle

G = 9.81


def tilted_world(oracle, theta):
    """Gravity tilted by theta about y: equivalent to a ground plane inclined by theta."""
    return oracle.OracleWorld(max_bodies=64, gravity=(G * np.sin(theta), 0.0, -G * np.cos(theta)))


@pytest.mark.parametrize("theta_deg,slides", [(20.0, False), (35.0, True)])
def test_friction_cone_on_incline(oracle, theta_deg, slides):
    """mu = 0.5: a box holds on a 20 deg incline (tan 20 = 0.36 < mu) and slides on 35 deg (tan 35 = 0.70 > mu) with
    a = g (sin t - mu cos t)."""
    th = np.radians(theta_deg)
    w = tilted_world(oracle, th)
    add_ground(w, friction=0.5, restitution=0.0)
    i = dyn(w, pos=(0, 0, 0.5), friction=0.5, restitution=0.0, lin_damp=0.0, ang_damp=0.0, allow_sleeping=0, shape=(1.0, 1.0, 0.5))
    n = 90
    for _ in range(n):
        w.step(DT)
    s = w.get_state([i])[0]
    if not slides:
        assert abs(s["pos"][0]) < 0.02 and np.linalg.norm(s["lin_vel"]) < 0.02
    else:
        a = G * (np.sin(th) - 0.5 * np.cos(th))
        t = n * DT
        assert abs(s["lin_vel"][0] - a * t) <= 0.05 * a * t
        assert abs(s["pos"][0] - 0.5 * a * t * t) <= 0.08 * 0.5 * a * t * t
    w.close()


def test_sphere_slip_to_roll_transition(oracle):
    """A sphere launched sliding (no spin) on a rough plane ends up rolling without slipping at v = 5/7 v0
    (I = 2/5 m r^2), independent of mu."""
    w = oracle.OracleWorld(max_bodies=64)
    add_ground(w, friction=0.6, restitution=0.0)
    v0 = 7.0
    i = dyn(w, abi.SHAPE_SPHERE, (0.5,), pos=(0, 0, 0.5), lin_vel=(v0, 0, 0), friction=0.6, restitution=0.0,
            lin_damp=0.0, ang_damp=0.0, allow_sleeping=0)
    for _ in range(150):
        w.step(DT)
    s = w.get_state([i])[0]
    assert abs(s["lin_vel"][0] - 5.0) < 0.1                      # 5/7 * 7
    assert abs(s["ang_vel"][1] - s["lin_vel"][0] / 0.5) < 0.2     # rolling: omega_y = v / r
    w.close()


def test_oblique_elastic_sphere_collision(oracle):
    """Equal spheres, restitution 1, friction 0, zero gravity: the normal components of velocity are exchanged, the
    tangential ones untouched; momentum and kinetic energy are conserved."""
    w = oracle.OracleWorld(max_bodies=16, gravity=(0, 0, 0))
    a = dyn(w, abi.SHAPE_SPHERE, (0.5,), pos=(-2.0, 0.5, 0), lin_vel=(4, 0, 0), restitution=1.0, friction=0.0, lin_damp=0.0, ang_damp=0.0, allow_sleeping=0)
    b = dyn(w, abi.SHAPE_SPHERE, (0.5,), pos=(0.0, 0.0, 0), lin_vel=(0, 0, 0), restitution=1.0, friction=0.0, lin_damp=0.0, ang_damp=0.0, allow_sleeping=0)
    for _ in range(90):
        w.step(DT)
    s = w.get_state([a, b])
    va, vb = s["lin_vel"][0].astype(np.float64), s["lin_vel"][1].astype(np.float64)
    assert np.allclose(va + vb, (4, 0, 0), atol=1e-4)                               # momentum
    assert abs(0.5 * (va @ va + vb @ vb) - 0.5 * 16.0) < 0.02 * 8.0                 # kinetic energy (per unit mass)
    # contact normal at impact: centres 1 m apart with lateral offset 0.5 -> n = (cos 30, -sin 30): b leaves along n
    n = np.array([np.cos(np.radians(30)), -np.sin(np.radians(30)), 0.0])
    assert np.allclose(vb / np.linalg.norm(vb), n, atol=0.03)
    assert abs(np.linalg.norm(vb) - 4 * np.cos(np.radians(30))) < 0.1
    assert abs(va @ vb) < 0.15                                                       # equal masses leave at right angles
    assert np.allclose(s["ang_vel"], 0, atol=1e-5)                                   # frictionless: no spin
    w.close()


@pytest.mark.parametrize("tilt_deg,falls", [(20.0, False), (32.0, True)])
def test_box_tipping_threshold(oracle, tilt_deg, falls):
    """A 1 x 1 x 2 box balanced on an edge falls back if its COM is inside the support (tilt < atan(0.5/1) = 26.6 deg)
    and topples if it is beyond."""
    w = oracle.OracleWorld(max_bodies=16)
    add_ground(w, friction=1.0, restitution=0.0)
    t = np.radians(tilt_deg)
    # rotate about the y axis through the bottom edge at x = +0.5: place the centre accordingly
    hx, hz = 0.5, 1.0
    cx = 0.5 - (hx * np.cos(t) - hz * np.sin(t))
    cz = hx * np.sin(t) + hz * np.cos(t)
    i = dyn(w, shape=(0.5, 0.5, 1.0), pos=(cx, 0, cz + 0.001), rot=quat_axis_angle((0, 1, 0), t), friction=1.0, restitution=0.0, allow_sleeping=0)
    for _ in range(240):
        w.step(DT)
    s = w.get_state([i])[0]
    up_z = 1 - 2 * (s["rot"][0] ** 2 + s["rot"][1] ** 2)          # z component of the body's z axis
    if falls:
        assert abs(up_z) < 0.2 and abs(s["pos"][2] - 0.5) < 0.05   # lying on its long side
    else:
        assert up_z > 0.98 and abs(s["pos"][2] - 1.0) < 0.05       # back upright
    w.close()


def test_kinematic_and_static_do_not_collide(oracle):
    """Pairs without a dynamic body are never generated (Jolt: kinematic vs non-dynamic do not collide by default)."""
    w = oracle.OracleWorld(max_bodies=16)
    add_ground(w)
    k = dyn(w, pos=(0, 0, 0.3), motion=abi.MOTION_KINEMATIC, lin_vel=(0, 0, -1.0))
    for _ in range(30):
        w.step(DT)
    s = w.get_state([k])[0]
    assert abs(s["pos"][2] - (0.3 - 0.5)) < 1e-4 and w.stats().num_pairs == 0
    w.close()


def test_sensor_reports_contacts_without_response(oracle):
    w = oracle.OracleWorld(max_bodies=16)
    w.set_contact_events(True)
    add_ground(w)
    sensor = dyn(w, pos=(0, 0, 1.0), motion=abi.MOTION_STATIC, layer=abi.LAYER_NON_MOVING, sensor=1, activate=0)
    ball = dyn(w, abi.SHAPE_SPHERE, (0.25,), pos=(0, 0, 3.0), allow_sleeping=0)
    hit = pers = 0
    for _ in range(90):
        w.step(DT)
        ev = w.drain_events(abi.EVENT_CONTACT_ADDED)
        hit += int(np.any((ev["id1"] == sensor) & (ev["id2"] == ball)))
        ev = w.drain_events(abi.EVENT_CONTACT_PERSISTED)
        pers += int(np.any((ev["id1"] == sensor) & (ev["id2"] == ball)))
    s = w.get_state([ball])[0]
    # 'added' when the overlap begins (again after the small bounce off the ground takes the ball out of the sensor's
    # speculative margin), 'persisted' on every other step of the overlap
    assert 1 <= hit <= 3 and pers >= 40
    assert abs(s["pos"][2] - 0.25) < 0.03            # fell straight through the sensor and rests on the ground
    w.close()


def test_body_pair_contact_cache_reuses_resting_box_manifolds(oracle):
    """ContactConstraintManager::GetContactsFromCache as restated here: a box resting on a box keeps its manifold while the two have not moved
    relative to each other by more than 1 mm / 2 degrees since it was computed; a nudge beyond that (or switching the setting off) sends the
    pair through the collision test again; sphere contacts are never taken from the cache (polytope pairs only)."""
    def scene(use):
        w = oracle.OracleWorld(max_bodies=64, settings={"use_body_pair_contact_cache": use})
        add_ground(w)
        a = dyn(w, pos=(0, 0, 0.5), restitution=0.0, allow_sleeping=0)
        b = dyn(w, pos=(0.1, 0.05, 1.5), restitution=0.0, allow_sleeping=0)
        s = dyn(w, abi.SHAPE_SPHERE, (0.5,), pos=(3, 0, 0.5), restitution=0.0, allow_sleeping=0)
        return w, a, b, s
    w, a, b, s = scene(1)
    for _ in range(120):
        w.step(DT)
    st = w.stats()
    assert st.num_manifolds == 3 and st.num_cached_manifolds == 2          # ground-box and box-box reused, the sphere's contact recomputed
    # a nudge of 5 mm: the upper pair is re-collided once, then cached again
    p = w.get_state([b])[0]
    w.set_pose_vel(b, (float(p["pos"][0]) + 0.005, float(p["pos"][1]), float(p["pos"][2])), tuple(float(x) for x in p["rot"]), (0, 0, 0), (0, 0, 0))
    w.step(DT)
    assert w.stats().num_cached_manifolds == 1
    for _ in range(30):
        w.step(DT)
    assert w.stats().num_cached_manifolds == 2
    w.close()
    w, a, b, s = scene(0)
    for _ in range(120):
        w.step(DT)
    assert w.stats().num_cached_manifolds == 0 and w.stats().num_manifolds == 3
    w.close()


def _uf_prio(x):
    h = (x * 0x9E3779B1) & 0xFFFFFFFF; h ^= h >> 15; h = (h * 0x85EBCA6B) & 0xFFFFFFFF; h ^= h >> 13
    return h


def stale_label_scene(w):
    """Three boxes stacked, asleep; the member the island is remembered by (lowest uf_prio: the label) sits on TOP and is removed, a new box is created in its
    slot somewhere else and falls asleep there.  Returns (the two boxes left of the stack, the new box)."""
    from helpers import add_ground, dyn
    add_ground(w)
    ids = [dyn(w, pos=(0, 0, 10.0 + k)) for k in range(3)]                     # (parked high up; placed below)
    order = sorted(ids, key=_uf_prio)                                           # order[0] = the island's label
    for z, i in zip((2.5, 1.5, 0.5), order):
        w.set_pose_vel(i, (0.0, 0.0, z), (0, 0, 0, 1))
    for _ in range(300):
        w.step(DT)
    assert not any(s["active"] for s in w.get_state(ids))
    w.remove(order[0])
    new = dyn(w, pos=(20.0, 0.0, 0.5))
    assert new == order[0]                                                      # the freed slot is handed out again
    for _ in range(200):
        w.step(DT)
    st = w.get_state(order[1:] + [new])
    assert not any(s["active"] for s in st)
    return order[1:], new


def test_a_body_created_in_a_removed_island_roots_slot_does_not_wake_that_island(oracle):
    """sleep labels were bare slot ids, so a body created in the slot of a removed island root shared the wake label of the island's
    sleepers -- poking the newcomer woke a stack twenty metres away.  Labels now carry the slot's generation."""
    from helpers import dyn
    from substrata_amd import abi
    w = oracle.OracleWorld(max_bodies=64)
    rest, new = stale_label_scene(w)
    dyn(w, abi.SHAPE_SPHERE, (0.2,), pos=(20.0, 0.0, 1.6), mass=5.0)            # a ball drops on the newcomer
    woke_new = False
    for _ in range(90):
        w.step(DT)
        woke_new = woke_new or bool(w.get_state([new])[0]["active"])
        assert not any(s["active"] for s in w.get_state(rest))                  # the old stack sleeps on
    assert woke_new
    # ... and the cascade itself still works: a ball on the old stack wakes both of its boxes in the step of the touch
    dyn(w, abi.SHAPE_SPHERE, (0.2,), pos=(0.0, 0.0, 2.4), mass=5.0)
    first = None
    for s_ in range(60):
        w.step(DT)
        act = [bool(s["active"]) for s in w.get_state(rest)]
        if first is None and any(act):
            first = s_
            assert all(act)
    assert first is not None
    w.close()


def sleeping_pile_scene(w, somebody_awake=False):
    """Three boxes stacked on the ground: the pile falls asleep as a whole and stays so for a while -- in an otherwise empty world (steps nobody is awake in), or
    while a kinematic body drifts about far away (somebody_awake: every step is a real one).  Returns the box ids, bottom first."""
    from helpers import add_ground, dyn
    add_ground(w)
    ids = [dyn(w, pos=(0.0, 0.0, 0.5 + k)) for k in range(3)]
    if somebody_awake:
        dyn(w, pos=(50.0, 50.0, 5.0), motion=abi.MOTION_KINEMATIC, lin_vel=(0.01, 0.0, 0.0))
    for _ in range(400):
        w.step(DT)
    assert not any(s["active"] for s in w.get_state(ids))
    for _ in range(40):
        w.step(DT)                                                              # steps nobody is awake in
    return ids


@pytest.mark.parametrize("somebody_awake", [False, True])
def test_a_pile_that_fell_asleep_as_a_whole_wakes_with_its_contacts(oracle, somebody_awake):
    """'the first idle step wipes the whole contact cache, so a pile that wakes later starts cold'.  A step nobody is awake in now leaves the cache as it
    found it: when the top box is nudged, the step that wakes the pile (in-step activation: all three collide in it) finds last contacts of every pair -- the body-pair
    cache hands back their manifolds, the events say 'persisted', and the impulses start from what held the pile up.
    somebody_awake: the same while the world never idles -- the contacts of a sleeping pair are carried over from step to step behind the constraints of the step
    (never solved, counted or reported: num_manifolds stays 0 while the pile sleeps)."""
    w = oracle.OracleWorld(max_bodies=16)
    w.set_contact_events(True)
    ids = sleeping_pile_scene(w, somebody_awake)
    assert bool(w.stats().num_active) == somebody_awake and w.stats().num_manifolds == 0
    for k in (abi.EVENT_CONTACT_ADDED, abi.EVENT_CONTACT_PERSISTED):
        w.drain_events(k)
    z0 = [float(s["pos"][2]) for s in w.get_state(ids)]
    w.set_vel(ids[2], (0.02, 0.0, 0.0), (0.0, 0.0, 0.0))
    w.activate(ids[2])
    w.step(DT)
    st = w.stats()
    assert all(s["active"] for s in w.get_state(ids))                           # the nudge woke the box, the box its island, in the same step
    assert st.num_manifolds == 3 and st.num_cached_manifolds == 3, (st.num_manifolds, st.num_cached_manifolds)
    added, persisted = w.drain_events(abi.EVENT_CONTACT_ADDED), w.drain_events(abi.EVENT_CONTACT_PERSISTED)
    assert len(added) == 0 and len(persisted) == 3
    s1 = w.get_state(ids)
    assert max(abs(float(s["pos"][2]) - z) for s, z in zip(s1, z0)) < 2e-4
    w.close()

