"""GPU-vs-oracle parity of the wheeled vehicle path (sgp_vehicle_*, replacing JPH::VehicleConstraint +
WheeledVehicleController as CarPhysics uses them, /root/reference/gui_client/CarPhysics.cpp:94-231): the same cars, inputs and
debris on both sides; chassis / debris states and the drivetrain state must agree to fp32 rounding (in practice bit for bit)."""
import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT, add_car
import parity

pytestmark = pytest.mark.gpu


def vehicle_diff(sg, sc):
    out = {}
    for f in ("engine_rpm", "clutch_friction"):
        out[f] = float(np.max(np.abs(sg[f] - sc[f])))
    assert np.array_equal(sg["current_gear"], sc["current_gear"]) and np.array_equal(sg["active"], sc["active"])
    wg, wc = sg["wheels"], sc["wheels"]
    assert np.array_equal(wg["has_contact"], wc["has_contact"]) and np.array_equal(wg["contact_body"], wc["contact_body"])
    for f in ("suspension_length", "steer_angle", "rotation_angle", "angular_velocity", "contact_position", "contact_normal",
              "suspension_lambda", "longitudinal_lambda", "lateral_lambda", "longitudinal_slip", "lateral_slip"):
        out[f] = float(np.max(np.abs(wg[f] - wc[f])))
    out["inexact_fields"] = [f for f in ("suspension_length", "steer_angle", "rotation_angle", "angular_velocity", "suspension_lambda",
                                         "longitudinal_lambda", "lateral_lambda", "longitudinal_slip", "lateral_slip", "contact_position", "contact_normal")
                             if not np.array_equal(wg[f], wc[f])]              # (value equality: -0.0 == +0.0)
    out["bit_exact"] = not out["inexact_fields"] and out["engine_rpm"] == 0.0
    return out


def drive(tw, ncars, s):
    t = s * DT
    inp = np.zeros(ncars, dtype=abi.vehicle_input_dtype)
    for k in range(ncars):
        inp[k]["forward"] = 1.0 if (s // 120 + k) % 3 != 2 else -1.0
        inp[k]["right"] = np.sin(0.5 * t + k)
        inp[k]["brake"] = 1.0 if (s // 90 + k) % 5 == 4 else 0.0
        inp[k]["hand_brake"] = 1.0 if (s // 150 + k) % 7 == 6 else 0.0
    tw.vehicle_set_inputs(0, inp)


def test_cars_and_debris_match_oracle(oracle):
    tw = parity.make_twin(oracle, max_bodies=2048)
    descs, car_ids = scenes.config5_cars_debris(cars_side=3, n_debris=400, seed=11)
    ncars = len(car_ids)
    descs["pos"][1:1 + ncars, 2] += 0.1 * np.arange(ncars)            # staggered drops
    tw.add_batch(descs)
    for b in car_ids:
        vg, vc = tw.vehicle_create(tw.gpu.default_vehicle_desc(int(b)))
        assert vg == vc
    debris = descs[1 + ncars:]
    # two motorcycles (MotorcycleController: lean spring, lean steering limit) riding through the same field
    from helpers import bike_vehicle_desc
    nbikes = 2
    for k in range(nbikes):
        bd = scenes.dynamic_bodies(1, mass=200.0, friction=0.5, restitution=0.0)
        bd["shape"][0, :3] = (1.7 / 2 * 0.18, 9.0 / 2 * 0.18, 3.2 / 2 * 0.18)
        bd["pos"][0] = (-16.0 + 3.0 * k, -16.0, 0.7)
        ig, ic = tw.add_batch(bd)
        bg = int(ig[0])
        assert bg == int(ic[0])
        vg, vc = tw.vehicle_create(bike_vehicle_desc(tw.gpu, bg))
        assert vg == vc == ncars + k
    ncars_only = ncars
    ncars += nbikes
    n = 1 + ncars + len(debris)                # (bikes were added after the debris: same count)
    for s in range(360):
        if s % 15 == 0:
            drive(tw, ncars, s)
        tw.step(DT)
        if s in (0, 30, 120, 240, 359):
            d = parity.compare(tw, n)
            assert d["active_mismatch"] == 0, (s, d)
            assert d["pos"] <= 2e-4 and d["rot"] <= 2e-4 and d["lin_vel"] <= 2e-3 and d["ang_vel"] <= 2e-3, (s, d)
            sg, sc = tw.vehicle_get_states(0, ncars)
            vd = vehicle_diff(sg, sc)
            assert vd["engine_rpm"] <= 0.5 and vd["angular_velocity"] <= 1e-2 and vd["suspension_length"] <= 1e-4, (s, vd)
    assert vd["bit_exact"] and d["bit_exact"]
    print("cars+debris 360 steps: bodies bit exact =", d["bit_exact"], " vehicles bit exact =", vd["bit_exact"], vd["inexact_fields"],
          {k: v for k, v in vd.items() if isinstance(v, float) and v > 0})
    sg, _ = tw.vehicle_get_states(0, ncars)
    assert (np.abs(sg["wheels"]["angular_velocity"]) > 1.0).any() and (sg["current_gear"] != 0).any()
    # the cars went somewhere
    st = tw.gpu.read_states(1, ncars)
    assert np.max(np.abs(st["lin_vel"])) > 1.0
    tw.close()


def test_vehicle_lifecycle_and_errors():
    from substrata_amd.lib import World
    from substrata_amd.world import SgpError
    w = World(max_bodies=64)
    w.add_batch(scenes.ground())
    body, vid = add_car(w)
    assert vid == 0
    with pytest.raises(SgpError):
        w.vehicle_create(w.default_vehicle_desc(0))               # the ground is not dynamic
    bad = w.default_vehicle_desc(body); bad.num_wheels = 9
    with pytest.raises(SgpError):
        w.vehicle_create(bad)
    b2, v2 = add_car(w, pos=(10, 0, 0.8))
    assert v2 == 1
    w.step(DT)
    w.vehicle_destroy(vid)
    with pytest.raises(SgpError):
        w.vehicle_get_state(vid)
    b3, v3 = add_car(w, pos=(20, 0, 0.8))
    assert v3 == 0                                                # lowest free slot is reused
    w.remove(b2)                                                  # a vehicle does not outlive its chassis
    with pytest.raises(SgpError):
        w.vehicle_set_input(v2, forward=1.0)
    for _ in range(30):
        w.step(DT)
    vs = w.vehicle_get_state(v3)
    assert vs["active"] == 1 and all(x["has_contact"] == 1 for x in vs["wheels"])
    w.vehicle_reset_drivetrain(v3, 0.0, 0.0)
    assert w.vehicle_get_state(v3)["engine_rpm"] == 0.0
    w.close()


# ---- round 4: two-body wheel rows (the dynamic body under a wheel takes the reaction; VehicleConstraint::SetupVelocityConstraint) ----

def _states_exact(tw, n, what):
    d = parity.compare(tw, n)
    assert d["active_mismatch"] == 0 and d["bit_exact"], (what, d)
    return d


def test_wheel_rows_push_back_on_a_floating_plate(oracle):
    """The KAT of tests/test_oracle_vehicle.py on the device: car + free plate conserve momentum (gravity acts on the car only), and the
    device agrees with the oracle bit for bit while the plate sinks away under the wheels and is driven backwards by the tyres."""
    from helpers import dyn
    tw = parity.make_twin(oracle, max_bodies=64)
    ids = []
    for w in (tw.gpu, tw.cpu):
        p = dyn(w, shape=(3.0, 4.0, 0.2, 0.0), pos=(0, 0, -0.2), mass=600.0, gravity_factor=0.0, lin_damp=0.0, ang_damp=0.0, friction=1.0)
        b = dyn(w, shape=(0.9, 2.0, 0.25, 0.0), pos=(0, 0, 0.75), mass=1200.0, friction=0.5, restitution=0.0, lin_damp=0.0, ang_damp=0.0)
        ids.append((p, b, w.vehicle_create(w.default_vehicle_desc(b))))
    assert ids[0] == ids[1]
    plate, body, vid = ids[0]
    n = 45
    for s in range(n):
        tw.step(DT)
    _states_exact(tw, 2, "sinking plate")
    sc, sp = tw.gpu.get_state([body])[0], tw.gpu.get_state([plate])[0]
    pz = 1200.0 * sc["lin_vel"][2] + 600.0 * sp["lin_vel"][2]
    assert abs(pz - (-1200.0 * 9.81 * n * DT)) < 2e-3 * 1200.0 * 9.81 * n * DT
    assert sp["lin_vel"][2] < -1.0
    tw.vehicle_set_input(vid, forward=1.0, right=0.3)
    for s in range(60):
        tw.step(DT)
    _states_exact(tw, 2, "driven plate")
    sg, sc_ = tw.vehicle_get_states(0, 1)
    assert vehicle_diff(sg, sc_)["bit_exact"]
    assert tw.gpu.get_state([plate])[0]["lin_vel"][1] < -0.1
    tw.close()


def test_vehicles_sharing_a_body_are_solved_in_index_order(oracle):
    """Two cars on one dynamic plate and a small car standing on the flat bed of a third: their rows act on a common movable body, so the
    device solves the later ones after the earlier ones (StepCounters::veh_deferred, veh_block_solve) -- same bits as the oracle's loop."""
    from helpers import dyn, add_ground
    tw = parity.make_twin(oracle, max_bodies=128)
    recs = []
    for w in (tw.gpu, tw.cpu):
        add_ground(w)
        plate = dyn(w, shape=(6.0, 5.0, 0.15, 0.0), pos=(0, 0, 0.15), mass=3000.0, friction=0.9)
        a, va = add_car(w, pos=(-2.5, 0, 1.05))
        b, vb = add_car(w, pos=(2.5, 0, 1.05))
        bed = dyn(w, shape=(1.6, 3.2, 0.2, 0.0), pos=(0, 12.0, 0.8), mass=4000.0, friction=0.8, restitution=0.0)
        vbed = w.vehicle_create(w.default_vehicle_desc(bed))
        top = dyn(w, shape=(0.9, 2.0, 0.25, 0.0), pos=(0, 12.0, 1.75), mass=600.0, friction=0.5, restitution=0.0)
        vtop = w.vehicle_create(w.default_vehicle_desc(top))
        for k in range(6):
            dyn(w, pos=(-4.0 + 1.6 * k, 3.0, 0.9), mass=30.0)
        recs.append((plate, a, va, b, vb, bed, vbed, top, vtop))
    assert recs[0] == recs[1]
    plate, a, va, b, vb, bed, vbed, top, vtop = recs[0]
    nb = 12
    deferred_seen = 0
    for s in range(300):
        if s == 60:
            tw.vehicle_set_input(va, forward=1.0, right=0.4)
            tw.vehicle_set_input(vb, forward=-1.0, right=-0.2)
            tw.vehicle_set_input(vtop, forward=0.6)
        if s == 150:
            tw.vehicle_set_input(vbed, forward=1.0)
            tw.vehicle_set_input(va, brake=1.0)
        tw.step(DT)
        stg, stc = tw.gpu.stats(), tw.cpu.stats()
        assert stg.num_deferred_vehicles == stc.num_deferred_vehicles, (s, stg.num_deferred_vehicles, stc.num_deferred_vehicles)
        deferred_seen = max(deferred_seen, stg.num_deferred_vehicles)
        if s in (0, 1, 30, 59, 61, 90, 149, 151, 200, 299):
            _states_exact(tw, nb + 1, s)
            sg, sc = tw.vehicle_get_states(0, 4)
            vd = vehicle_diff(sg, sc)
            assert vd["bit_exact"], (s, vd)
    assert deferred_seen >= 2, deferred_seen          # the second car on the plate and the car on the flat bed
    tw.close()


def test_a_wheel_wakes_the_plate_and_they_sleep_together(oracle):
    from helpers import dyn, add_ground
    tw = parity.make_twin(oracle, max_bodies=64)
    recs = []
    for w in (tw.gpu, tw.cpu):
        add_ground(w)
        plate = dyn(w, shape=(3.0, 4.0, 0.1, 0.0), pos=(0, 0, 0.1), mass=4000.0, friction=0.8)
        body, vid = add_car(w, pos=(0, 0, 0.95))
        recs.append((plate, body, vid))
    plate, body, vid = recs[0]
    for s in range(400):
        tw.step(DT)
    d = _states_exact(tw, 3, "asleep")
    assert tw.gpu.get_state([body])[0]["active"] == 0 and tw.gpu.get_state([plate])[0]["active"] == 0
    tw.vehicle_set_input(vid, forward=0.3)
    tw.step(DT)
    assert tw.gpu.get_state([body])[0]["active"] == 1 and tw.gpu.get_state([plate])[0]["active"] == 1
    for s in range(120):
        tw.step(DT)
    _states_exact(tw, 3, "driving off")
    tw.close()


def test_anti_roll_bias_and_the_sleeping_car_woken_through_a_wheel_match_oracle(oracle):
    """Round 5: (1) a car parked with its left wheels on a plate -- the anti-roll terms are the biases of the suspension rows -- and (2) a parked,
    sleeping car with a ball rolling through the cast of a front wheel (it never touches the chassis): the car must wake in the same step on the device
    as in the oracle, and everything stays bit for bit the same."""
    from helpers import dyn, add_ground
    tw = parity.make_twin(oracle, max_bodies=64)
    for w in (tw.gpu, tw.cpu):
        add_ground(w)
        dyn(w, shape=(0.5, 3.0, 0.03, 0.0), pos=(-0.8, 0.0, 0.03), motion=abi.MOTION_STATIC, layer=abi.LAYER_NON_MOVING)
        add_car(w, pos=(0, 0, 0.85))                                   # car 0: left wheels on the plate
        add_car(w, pos=(20.0, 0, 0.75))                                # car 1: parked on the flat, will be woken through a wheel
    n = 64
    for s in range(420):
        tw.step(DT)
        if s in (0, 5, 60, 419):
            d = parity.compare(tw, n)
            assert d["bit_exact"] and d["active_mismatch"] == 0, (s, d)
            sg, sc = tw.vehicle_get_states(0, 2)
            assert vehicle_diff(sg, sc)["bit_exact"], (s, vehicle_diff(sg, sc))
    sg, sc = tw.vehicle_get_states(0, 2)
    assert sg["active"][1] == 0 and sc["active"][1] == 0               # car 1 sleeps
    lens = sg["wheels"][0]["suspension_length"]
    assert lens[0] != lens[1]                                          # car 0's bars have something to do
    for w in (tw.gpu, tw.cpu):
        dyn(w, shape_type=abi.SHAPE_SPHERE, shape=(0.1, 0, 0, 0), pos=(22.2, 1.42, 0.1), mass=1.0, friction=0.5, lin_vel=(-3.0, 0.0, 0.0))
    woke_at = None
    for s in range(90):
        tw.step(DT)
        ag = int(tw.gpu.read_states(0, n)["active"][3]); ac = int(tw.cpu.read_states(0, n)["active"][3])     # body 3 = car 1's chassis
        assert ag == ac, s
        if woke_at is None and ag:
            woke_at = s
        d = parity.compare(tw, n)
        assert d["bit_exact"] and d["active_mismatch"] == 0, (s, d)
    assert woke_at is not None and woke_at > 3                         # woken when the ball reached the wheel, not by its creation
    sg, sc = tw.vehicle_get_states(0, 2)
    assert vehicle_diff(sg, sc)["bit_exact"]
    tw.close()


def test_wheels_cast_as_cylinders_match_oracle(oracle):
    """VehicleCollisionTesterCastCylinder (BikePhysics.cpp:229; SGP_VEHICLE_TESTER_CYLINDER): the wheel itself is the cast shape -- a rounded disc, found by a
    search over sphere casts along its leading rim.  Two motorcycles and two cars with that tester among boxes, spheres and capsules, over kerbs: the search
    is the same expression sequence on both sides, so wheels, chassis and debris agree bit for bit."""
    from helpers import bike_vehicle_desc, dyn
    tw = parity.make_twin(oracle, max_bodies=1024)
    descs, car_ids = scenes.config5_cars_debris(cars_side=2, n_debris=250, seed=23)
    ncars = 2
    descs = np.concatenate([descs[:1 + ncars], descs[1 + len(car_ids):]])           # ground, two of the cars, the debris
    tw.add_batch(descs)
    nv = 0
    for b in range(1, 1 + ncars):
        vd = tw.gpu.default_vehicle_desc(b)
        vd.collision_tester = abi.VEHICLE_TESTER_CYLINDER
        vg, vc = tw.vehicle_create(vd)
        assert vg == vc == nv
        nv += 1
    # kerbs across the field: what a cast wheel meets with its front and a cast sphere does not
    for k in range(6):
        ids = [dyn(w, shape=(12.0, 0.25, 0.05 + 0.02 * k, 0.0), pos=(0.0, -9.0 + 3.5 * k, 0.05 + 0.02 * k), motion=abi.MOTION_STATIC, layer=abi.LAYER_NON_MOVING) for w in (tw.gpu, tw.cpu)]
        assert ids[0] == ids[1]
    for k in range(2):
        bd = scenes.dynamic_bodies(1, mass=200.0, friction=0.5, restitution=0.0)
        bd["shape"][0, :3] = (1.7 / 2 * 0.18, 9.0 / 2 * 0.18, 3.2 / 2 * 0.18)
        bd["pos"][0] = (-6.0 + 3.0 * k, -12.0, 0.7)
        ig, ic = tw.add_batch(bd)
        assert int(ig[0]) == int(ic[0])
        vd = bike_vehicle_desc(tw.gpu, int(ig[0]))
        vd.collision_tester = abi.VEHICLE_TESTER_CYLINDER
        vg, vc = tw.vehicle_create(vd)
        assert vg == vc == nv
        nv += 1
    n = tw.gpu.num_bodies()
    kerb_hits = 0
    for s in range(300):
        if s % 15 == 0:
            drive(tw, nv, s)
        tw.step(DT)
        if s % 30 == 0 or s == 299:
            d = parity.compare(tw, n + 8)
            assert d["active_mismatch"] == 0 and d["bit_exact"], (s, d)
            sg, sc = tw.vehicle_get_states(0, nv)
            vd_ = vehicle_diff(sg, sc)
            assert vd_["bit_exact"], (s, vd_)
            kerb_hits += int(np.sum((sg["wheels"]["has_contact"] != 0) & (sg["wheels"]["contact_body"] > 1 + ncars + 250 - 1) & (sg["wheels"]["contact_normal"][..., 2] < 0.98)))
    sg, _ = tw.vehicle_get_states(0, nv)
    assert (np.abs(sg["wheels"]["angular_velocity"]) > 1.0).any()
    print("cylinder testers, 300 steps: bit exact; wheel contacts on kerb edges seen at the checks:", kerb_hits)
    tw.close()

