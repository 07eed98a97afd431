"""Known-answer tests for the oracle's wheeled vehicle (oracle/sgo_vehicle.h): the restatement of JPH::VehicleConstraint +
WheeledVehicleController as CarPhysics sets it up (/root/reference/gui_client/CarPhysics.cpp:94-231, defaults
/root/reference/gui_client/Scripting.cpp:315-346).  Jolt is not in the tree ("parity unpinned"), so the pins are physical:
spring statics, traction- and friction-limited acceleration / braking, steering direction, cast geometry."""
import numpy as np
import pytest

from substrata_amd import abi
from helpers import DT, add_ground, add_car, dyn
from test_collide_independent import Shape

G = 9.81


def settle(w, steps=240):
    for _ in range(steps):
        w.step(DT)


def test_suspension_static_equilibrium(oracle):
    """At rest each spring carries m g / 4 with k = m_eff (2 pi f)^2, m_eff = 1 / (1/m + (p x up)^T I^-1 (p x up)) evaluated at the
    mid-travel suspension point (Jolt's frequency/damping spring mode)."""
    w = oracle.OracleWorld(max_bodies=16)
    add_ground(w)
    m = 1200.0
    body, vid = add_car(w, mass=m)
    settle(w, 300)
    vs = w.vehicle_get_state(vid)
    hx, hy, hz = 0.9, 2.0, 0.25
    inv_i = np.array([12.0 / (m * ((2 * hy) ** 2 + (2 * hz) ** 2)), 12.0 / (m * ((2 * hx) ** 2 + (2 * hz) ** 2)),
                      12.0 / (m * ((2 * hx) ** 2 + (2 * hy) ** 2))])
    p = np.array([0.8, 1.3, 0.15]) + 0.5 * (0.2 + 0.5) * np.array([0, 0, -1.0])
    pxu = np.cross(p, [0, 0, -1.0])
    m_eff = 1.0 / (1.0 / m + pxu @ (inv_i * pxu))
    k = m_eff * (2 * np.pi * 2.0) ** 2
    expect = 0.5 - (m * G / 4) / k
    lens = np.array([x["suspension_length"] for x in vs["wheels"]])
    assert np.allclose(lens, expect, atol=2e-3), (lens, expect)
    assert all(x["has_contact"] == 1 for x in vs["wheels"])
    # spring impulses carry the weight
    lam = sum(x["suspension_lambda"] for x in vs["wheels"])
    assert abs(lam - m * G * DT) < 0.03 * m * G * DT          # (asleep by now: the impulses of its last, not quite static, step)
    # chassis height = wheel radius + suspension length - attachment height
    z = w.get_state([body])[0]["pos"][2]
    assert abs(z - (0.42 + expect - 0.15)) < 3e-3
    # and it falls asleep; input wakes it (CarPhysics.cpp:362-363)
    settle(w, 120)
    w.vehicle_set_input(vid, forward=1.0)


def test_traction_limited_launch_and_friction_limited_braking(oracle):
    w = oracle.OracleWorld(max_bodies=16)
    add_ground(w, friction=0.5)
    m = 1200.0
    body, vid = add_car(w, mass=m)
    settle(w, 120)
    w.vehicle_set_input(vid, forward=1.0)
    settle(w, 60)
    v0 = w.get_state([body])[0]["lin_vel"][1]
    settle(w, 60)
    st = w.get_state([body])[0]
    vs = w.vehicle_get_state(vid)
    a = (st["lin_vel"][1] - v0) / (60 * DT)
    # front wheels spin (500 N m * 2.66 * 3.42 >> traction): tyre friction sits on the sliding plateau 1.0, combined with the
    # ground's 0.5 as sqrt(1.0 * 0.5); the drive force is mu * (front axle load), minus the small force spinning up the rear wheels
    front = [vs["wheels"][i] for i in (0, 1)]
    assert all(x["longitudinal_slip"] > 0.2 for x in front)
    mu = np.sqrt(1.0 * 0.5)
    f_drive = sum(mu * x["suspension_lambda"] / DT for x in front)
    f_rear = sum(vs["wheels"][i]["longitudinal_lambda"] / DT for i in (2, 3))          # (negative: spinning the rear wheels up)
    a_expect = (f_drive + f_rear) / m - 0.05 * 0.5 * (v0 + st["lin_vel"][1])             # body linear damping 0.05 / s
    assert abs(a - a_expect) < 0.03 * a, (a, a_expect)
    assert 2.0 < a < mu * G            # weight shifts off the driven axle under acceleration
    assert abs(st["pos"][0]) < 0.05 and abs(st["lin_vel"][0]) < 0.02          # drives straight
    assert vs["current_gear"] == 1 and vs["engine_rpm"] > 4000
    # rear wheels roll without slip
    assert abs(vs["wheels"][2]["angular_velocity"] * 0.42 - st["lin_vel"][1]) < 0.02 * st["lin_vel"][1]
    # brake: the rear wheels lock at once; the fronts lock once the auto box has dropped to neutral (until then the engine, which
    # cannot fall below its idle speed, keeps turning them through the clutch).  From then on every wheel slides on the friction
    # plateau: deceleration = mu g (+ body damping)
    settle(w, 180)
    w.vehicle_set_input(vid, brake=1.0)
    locked_at = None
    for k in range(400):
        w.step(DT)
        vs = w.vehicle_get_state(vid)
        if vs["current_gear"] == 0 and all(abs(x["angular_velocity"]) < 1e-3 for x in vs["wheels"]):
            locked_at = k
            break
    assert locked_at is not None and locked_at < 90
    assert abs(w.vehicle_get_state(vid)["wheels"][2]["angular_velocity"]) < 1e-3
    v1 = float(w.get_state([body])[0]["lin_vel"][1])
    assert v1 > 3.0
    settle(w, 12)
    v2 = float(w.get_state([body])[0]["lin_vel"][1])
    decel = (v1 - v2) / (12 * DT)
    assert abs(decel - (mu * G + 0.05 * 0.5 * (v1 + v2))) < 0.04 * mu * G, (decel, mu * G)
    settle(w, 300)
    assert abs(w.get_state([body])[0]["lin_vel"][1]) < 0.02


def test_steering_turns_towards_the_input(oracle):
    for sign in (1.0, -1.0):
        w = oracle.OracleWorld(max_bodies=16)
        add_ground(w, friction=1.0)
        body, vid = add_car(w)
        settle(w, 60)
        w.vehicle_set_input(vid, forward=0.4, right=0.5 * sign)
        settle(w, 240)
        st = w.get_state([body])[0]
        vs = w.vehicle_get_state(vid)
        assert np.sign(st["pos"][0]) == sign and abs(st["pos"][0]) > 1.0          # "right" is +x when driving along +y
        assert np.sign(st["ang_vel"][2]) == -sign                                   # clockwise seen from above for a right turn
        assert np.isclose(vs["wheels"][0]["steer_angle"], -0.5 * sign * 0.78525, atol=1e-6)
        assert vs["wheels"][2]["steer_angle"] == 0.0
        assert abs(st["pos"][2] - 0.70) < 0.05
        w.close()


def test_wheel_over_obstacle_and_airborne(oracle):
    w = oracle.OracleWorld(max_bodies=16)
    add_ground(w)
    # a static slab under the front-left wheel only
    dyn(w, shape=(0.3, 0.3, 0.05, 0), pos=(-0.8, 1.3, 0.05), motion=abi.MOTION_STATIC, layer=abi.LAYER_NON_MOVING)
    body, vid = add_car(w)
    settle(w, 200)
    vs = w.vehicle_get_state(vid)
    l = [x["suspension_length"] for x in vs["wheels"]]
    assert vs["wheels"][0]["contact_body"] != vs["wheels"][1]["contact_body"]
    assert l[0] < l[1] - 0.03                      # the wheel on the slab is pushed up (the body rolls a little, so less than 0.1)
    assert np.isclose(vs["wheels"][0]["contact_position"][2], 0.1, atol=1e-4)
    assert np.isclose(vs["wheels"][1]["contact_position"][2], 0.0, atol=1e-4)
    # airborne: no contacts, full droop, engine free-revs to the limiter, nothing blows up
    w2 = oracle.OracleWorld(max_bodies=16)
    add_ground(w2)
    b2, v2 = add_car(w2, pos=(0, 0, 30.0))
    w2.vehicle_set_input(v2, forward=1.0)
    settle(w2, 60)
    vs2 = w2.vehicle_get_state(v2)
    assert all(x["has_contact"] == 0 and np.isclose(x["suspension_length"], 0.5) for x in vs2["wheels"])
    assert vs2["engine_rpm"] > 5000 and vs2["wheels"][0]["angular_velocity"] > 20 and vs2["wheels"][2]["angular_velocity"] == 0.0
    st = w2.get_state([b2])[0]
    assert np.isfinite(st["pos"]).all() and abs(st["lin_vel"][2] + G * 1.0) < 0.5


def test_gearbox_shifts_up_with_grip(oracle):
    """With enough grip the wheels stop slipping, the auto box shifts up at 4000 rpm and the revs drop."""
    def grip(vd):
        for i in range(4):
            for k in range(3):
                vd.wheels[i].longitudinal_friction[k][1] *= 4.0
    w = oracle.OracleWorld(max_bodies=16)
    add_ground(w, friction=1.0)
    body, vid = add_car(w, desc_edit=grip)
    settle(w, 60)
    w.vehicle_set_input(vid, forward=1.0)
    gears, rpms = [], []
    for _ in range(600):
        w.step(DT)
        vs = w.vehicle_get_state(vid)
        gears.append(int(vs["current_gear"])); rpms.append(float(vs["engine_rpm"]))
    assert max(gears) >= 3 and gears[0] == 1
    k = gears.index(2)
    assert rpms[k] > 4000 and rpms[k - 1] <= 4000 and min(rpms[k:k + 60]) < rpms[k] - 500
    assert all(b - a in (0, 1) for a, b in zip(gears, gears[1:]))          # one gear at a time, never down while accelerating
    assert w.get_state([body])[0]["lin_vel"][1] > 20.0
    # reverse
    w.vehicle_set_input(vid, brake=1.0)
    settle(w, 400)
    w.vehicle_set_input(vid, forward=-1.0)
    settle(w, 200)
    assert w.vehicle_get_state(vid)["current_gear"] == -1 and w.get_state([body])[0]["lin_vel"][1] < -1.0


def test_sphere_cast_against_brute_force(oracle):
    """sgo_cast_sphere_body vs marching the sphere centre along the ray against an independent signed-distance function."""
    rng = np.random.default_rng(5)
    checked = 0
    for trial in range(300):
        kind = [abi.SHAPE_SPHERE, abi.SHAPE_BOX, abi.SHAPE_CAPSULE][trial % 3]
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        p = (rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8))
        if kind == abi.SHAPE_CAPSULE:
            p = (rng.uniform(0.15, 0.4), rng.uniform(0.2, 0.8), 0.0)
        sh = Shape(kind, p, rng.uniform(-1, 1, size=3), tuple(q))
        o = sh.pos + rng.normal(size=3) * 0.4 + np.array([0, 0, 2.5])
        d = sh.pos + rng.uniform(-0.6, 0.6, size=3) - o; d /= np.linalg.norm(d)
        rs = float(rng.choice([0.0, 0.08, 0.25]))
        max_t = 4.0
        ts = np.linspace(0, max_t, 4001)
        dist = np.array([sh.signed_dist(o + d * t) for t in ts]) - rs
        hit = oracle.cast_sphere(sh.desc(), o, d, max_t, rs)
        if dist[0] <= 0:
            continue
        inside = np.nonzero(dist <= 0)[0]
        if len(inside) == 0:
            assert hit is None or sh.signed_dist(o + d * hit[0]) - rs > -1e-4        # grazing at most
            continue
        lo, hi = ts[inside[0] - 1], ts[inside[0]]
        for _ in range(40):
            mid = 0.5 * (lo + hi)
            if sh.signed_dist(o + d * mid) - rs <= 0: hi = mid
            else: lo = mid
        assert hit is not None, trial
        t, n, pt = hit
        assert abs(t - hi) < 2e-4, (trial, t, hi)
        assert abs(sh.signed_dist(pt)) < 2e-4                                       # touch point lies on the body
        c = o + d * t
        assert np.allclose(c - pt, n * rs, atol=2e-4)                              # sphere centre = touch point + n * rs
        assert abs(np.linalg.norm(n) - 1) < 1e-4 and n @ d < 1e-3
        checked += 1
    assert checked > 150


def roll_deg(q):
    """Roll of the chassis about its forward axis, positive = leaning to its right (x right, y forward, z up)."""
    x, y, z, w = [float(c) for c in q]
    up = np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])
    fw = np.array([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)])
    rt = np.cross(fw, [0, 0, 1.0]); rt /= np.linalg.norm(rt)
    return float(np.degrees(np.arcsin(np.clip(up @ rt, -1, 1))))


def test_motorcycle_lean_controller(oracle):
    """JPH::MotorcycleController as BikePhysics sets it up (BikePhysics.cpp:124-227): the lean spring keeps the two-wheeler upright,
    leans it into a turn (towards the ground reaction), the lean steering limit shrinks the steering angle with speed, and
    without the controller (EnableLeanController(false), :617) the bike falls over."""
    from helpers import add_bike
    w = oracle.OracleWorld(max_bodies=16)
    add_ground(w, friction=1.0)
    b, v = add_bike(w)
    rolls, steer = [], []
    for s in range(600):
        w.vehicle_set_input(v, forward=0.5 if s > 30 else 0.0, right=0.5 if 300 <= s < 480 else 0.0)
        w.step(DT)
        st = w.get_state([b])[0]
        rolls.append(roll_deg(st["rot"])); steer.append(float(w.vehicle_get_state(v)["wheels"][0]["steer_angle"]))
    rolls = np.array(rolls)
    assert np.abs(rolls[:300]).max() < 2.0                                  # upright while accelerating in a straight line
    assert 15.0 < rolls[420:480].mean() < 60.0                              # leans right in the right-hand turn, below the 60 deg cap
    assert abs(rolls[-1]) < 8.0                                             # and comes back up
    st = w.get_state([b])[0]
    assert st["pos"][0] > 20.0 and st["lin_vel"][0] > 5.0                   # turned right
    assert st["pos"][2] > 0.3
    vs = w.vehicle_get_state(v)
    assert vs["current_gear"] >= 2 and all(x["has_contact"] == 1 for x in vs["wheels"][:2])
    # lean steering limit: at ~45 m/s the allowed steering angle is far below 0.5 * 30 deg
    assert 0.0 < -min(steer[300:480]) < 0.05
    # rear-wheel drive: all the engine torque goes to wheel 1 (left_right_split = 1)
    w2 = oracle.OracleWorld(max_bodies=16)
    add_ground(w2, friction=1.0)
    b2, v2 = add_bike(w2)
    w2.vehicle_enable_lean_controller(v2, False)
    for s in range(200):
        w2.vehicle_set_input(v2, forward=0.5 if s > 30 else 0.0)
        w2.step(DT)
    assert abs(roll_deg(w2.get_state([b2])[0]["rot"])) > 60.0               # on its side


# ---- round 4: the wheel rows are two-body constraints (JPH::VehicleConstraint solves them between the chassis and the body under the wheel) ----

def _car_on_floating_plate(oracle, plate_mass=600.0):
    """A car over a free-floating plate (no gravity on the plate, no damping anywhere, no ground): only the wheel rows connect the two."""
    w = oracle.OracleWorld(max_bodies=16)
    plate = dyn(w, shape=(3.0, 4.0, 0.2, 0.0), pos=(0, 0, -0.2), mass=plate_mass, gravity_factor=0.0, lin_damp=0.0, ang_damp=0.0, friction=1.0)
    body = dyn(w, shape=(0.9, 2.0, 0.25, 0.0), pos=(0, 0, 0.75), mass=1200.0, friction=0.5, restitution=0.0, lin_damp=0.0, ang_damp=0.0)
    vid = w.vehicle_create(w.default_vehicle_desc(body))
    return w, plate, body, vid


def test_wheels_push_back_on_the_body_they_stand_on(oracle):
    """Every impulse of a wheel row acts on the chassis and, reversed, on a dynamic body under the wheel: the pair's momentum changes by gravity
    on the car alone.  (Up to round 3 the body under a wheel was kinematic for the rows: the plate would never move.)"""
    w, plate, body, vid = _car_on_floating_plate(oracle)
    n = 45
    settle(w, n)
    sc, sp = w.get_state([body])[0], w.get_state([plate])[0]
    assert all(x["has_contact"] == 1 for x in w.vehicle_get_state(vid)["wheels"])
    pz = 1200.0 * sc["lin_vel"][2] + 600.0 * sp["lin_vel"][2]
    assert abs(pz - (-1200.0 * G * n * DT)) < 2e-3 * 1200.0 * G * n * DT, (pz, -1200.0 * G * n * DT)
    assert sp["lin_vel"][2] < -1.0                                        # the suspension pushes the plate down
    assert sc["lin_vel"][2] > -G * n * DT + 1.0                            # ... and holds the car up against it
    w.close()


def test_tyre_forces_act_on_the_body_under_the_wheels(oracle):
    """Throttle on a free-floating plate: the longitudinal rows push the plate backwards with what they push the car forwards with."""
    w, plate, body, vid = _car_on_floating_plate(oracle, plate_mass=2400.0)
    settle(w, 20)
    w.vehicle_set_input(vid, forward=1.0)
    settle(w, 40)
    sc, sp = w.get_state([body])[0], w.get_state([plate])[0]
    py = 1200.0 * sc["lin_vel"][1] + 2400.0 * sp["lin_vel"][1]
    px = 1200.0 * sc["lin_vel"][0] + 2400.0 * sp["lin_vel"][0]
    assert sc["lin_vel"][1] > 0.5 and sp["lin_vel"][1] < -0.2, (sc["lin_vel"], sp["lin_vel"])
    assert abs(py) < 2e-3 * 1200.0 * abs(sc["lin_vel"][1]) and abs(px) < 1e-2
    w.close()


def test_a_wheel_wakes_the_sleeping_body_under_it(oracle):
    """VehicleConstraint::BuildIslands activates the dynamic bodies the wheels touch and links them with the chassis: car and plate fall asleep
    together, and driver input wakes both in the same step although only the wheels touch the plate."""
    w = oracle.OracleWorld(max_bodies=16)
    add_ground(w)
    plate = dyn(w, shape=(3.0, 4.0, 0.1, 0.0), pos=(0, 0, 0.1), mass=4000.0, friction=0.8)
    body, vid = add_car(w, pos=(0, 0, 0.95))
    settle(w, 400)
    sc, sp = w.get_state([body])[0], w.get_state([plate])[0]
    assert sc["active"] == 0 and sp["active"] == 0
    assert abs(sp["pos"][2] - 0.1) < 0.03                                  # the plate carries the car: it rests on the ground, not pushed through it
    w.vehicle_set_input(vid, forward=0.3)
    w.step(DT)
    w.close()


def test_a_car_on_a_light_box_loads_it(oracle):
    """a car parked with one wheel on a 50 kg box pushes it into the ground with the wheel's load -- seen as the friction that now
    holds the box: a sideways shove that would send a free 50 kg box sliding is resisted by mu * (box weight + wheel load)."""
    def run(with_car):
        w = oracle.OracleWorld(max_bodies=16)
        add_ground(w, friction=1.0)
        box = dyn(w, shape=(0.3, 0.3, 0.1, 0.0), pos=(0.8, 1.3, 0.1), mass=50.0, friction=1.0, allow_sleeping=0)
        if with_car:
            add_car(w, pos=(0, 0, 0.95))
        settle(w, 180)
        x0 = float(w.get_state([box])[0]["pos"][0])
        for _ in range(30):
            w.add_force(box, (900.0, 0.0, 0.0))      # mu m g = 490 N alone; with a quarter of the car on top ~ 3400 N
            w.step(DT)
        x1 = float(w.get_state([box])[0]["pos"][0])
        w.close()
        return x1 - x0
    assert run(False) > 0.3
    assert abs(run(True)) < 0.02


def _spring_constants(m=1200.0):
    """k, c of the default car's suspension rows (frequency 2 Hz, damping ratio 0.5) as test_suspension_static_equilibrium derives them."""
    hx, hy, hz = 0.9, 2.0, 0.25
    inv_i = np.array([12.0 / (m * ((2 * hy) ** 2 + (2 * hz) ** 2)), 12.0 / (m * ((2 * hx) ** 2 + (2 * hz) ** 2)), 12.0 / (m * ((2 * hx) ** 2 + (2 * hy) ** 2))])
    p = np.array([0.8, 1.3, 0.15]) + 0.5 * (0.2 + 0.5) * np.array([0, 0, -1.0])
    pxu = np.cross(p, [0, 0, -1.0])
    m_eff = 1.0 / (1.0 / m + pxu @ (inv_i * pxu))
    om = 2 * np.pi * 2.0
    return m_eff * om * om, 2.0 * m_eff * 0.5 * om


def test_anti_roll_bar_is_the_bias_of_the_suspension_row(oracle):
    """Round 5 (DESIGN 8 'Vehicles'): the anti-roll term a_i = -/+ (len_R - len_L) K dt of a wheel is the velocity bias of its suspension row
    (SpringPart: bias = a + dt k s C with softness s = 1 / (dt (c + dt k))), not an impulse on the chassis.  At rest (J v = 0) a soft row settles at
    lambda = -bias / s, i.e. per wheel   lambda_i = -a_i dt (c + dt k) - dt k C_i   with C_i = len_i - max - preload:
    the bar acts like a spring of K dt (c + dt k) ~ 63 K between the two suspension lengths, far stiffer than the suspension springs themselves.
    A car parked with its left wheels on a 6 cm plate leans off it; the lower side's springs carry more and are shorter by 4 mm without bars, by
    a quarter of that with CarPhysics' bars (K = 1000).  The old statement (impulse a_i on the chassis in the pre-step, K dt
    per metre of difference) left the bars all but inert: K = 1000 gave the picture of K = 0."""
    k, c = _spring_constants()

    def run(K):
        w = oracle.OracleWorld(max_bodies=16)
        add_ground(w)
        dyn(w, shape=(0.5, 3.0, 0.03, 0.0), pos=(-0.8, 0.0, 0.03), motion=abi.MOTION_STATIC, layer=abi.LAYER_NON_MOVING)      # under the two left wheels (x = -0.8)

        def bars(vd):
            for b in range(2):
                vd.anti_roll_bars[b].stiffness = K
        body, vid = add_car(w, pos=(0, 0, 0.85), desc_edit=bars)
        settle(w, 500)
        vs = w.vehicle_get_state(vid)
        lens = np.array([x["suspension_length"] for x in vs["wheels"]], np.float64)
        lam = np.array([x["suspension_lambda"] for x in vs["wheels"]], np.float64)
        assert all(x["has_contact"] == 1 for x in vs["wheels"])
        w.close()
        return lens, lam
    lens0, lam0 = run(0.0)
    lens1, lam1 = run(1000.0)
    for l_i, r_i in ((0, 1), (2, 3)):                                     # default layout: wheels 0 / 2 left, 1 / 3 right
        d0, d1 = lens0[l_i] - lens0[r_i], lens1[l_i] - lens1[r_i]
        assert d0 > 0.003, lens0                                           # no bars: the chassis leans off the plate, the low side carries more (4 mm of spring)
        assert 0.0 < d1 < 0.4 * d0, (lens0, lens1)                         # bars: most of that difference is gone (the old, inert bars left it where it was)
        for lens, lam, K in ((lens0, lam0, 0.0), (lens1, lam1, 1000.0)):
            a = (lens[r_i] - lens[l_i]) * K * DT
            for i, a_i in ((l_i, -a), (r_i, a)):
                expect = -a_i * DT * (c + DT * k) - DT * k * (lens[i] - 0.5)
                assert abs(lam[i] - expect) < 0.03 * abs(expect) + 0.2, (K, i, lam[i], expect)


def test_an_active_body_under_a_wheel_wakes_the_sleeping_car(oracle):
    """Round 5 (DESIGN 8 'Vehicles'): a vehicle constraint is active when its chassis OR a body a wheel touches is active, and building the islands
    then activates the chassis (VehicleConstraint::OnStep / BuildIslands).  A ball rolled through the cast of a wheel of a parked, sleeping car --
    it passes under the chassis without touching it -- wakes the car; the same ball rolling past out of reach of every wheel does not."""
    def run(ball_y):
        w = oracle.OracleWorld(max_bodies=16)
        add_ground(w)
        body, vid = add_car(w)
        settle(w, 420)
        z_chassis_bottom = w.get_state([body])[0]["pos"][2] - 0.25
        ball = dyn(w, shape_type=abi.SHAPE_SPHERE, shape=(0.1, 0, 0, 0), pos=(2.2, ball_y, 0.1), mass=1.0, friction=0.5, lin_vel=(-3.0, 0.0, 0.0))
        assert z_chassis_bottom > 0.25                                        # the ball (top at 0.2) cannot touch the chassis
        woke = False
        for _ in range(90):
            w.step(DT)
            woke = woke or w.get_state([body])[0]["active"] == 1
        w.close()
        return woke
    assert run(1.3 + 0.12)                 # 12 cm beside the front wheels' suspension line (cast radius 0.08 + ball radius 0.1): swept by the cast
    assert not run(0.0)                    # between the axles: no wheel reaches it, and it never touches the chassis


def _signed_dist_many(sh, x):
    """Shape.signed_dist for an array of world points [N, 3] (sphere, box, capsule)."""
    l = (x - sh.pos) @ sh.R
    if sh.kind == abi.SHAPE_SPHERE:
        return np.linalg.norm(l, axis=1) - sh.p[0]
    if sh.kind == abi.SHAPE_BOX:
        q = np.abs(l) - sh.p[:3]
        return np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0.0)
    zc = np.clip(l[:, 2], -sh.p[1], sh.p[1])
    return np.linalg.norm(l - np.stack([np.zeros_like(zc), np.zeros_like(zc), zc], axis=1), axis=1) - sh.p[0]


def test_wheel_cast_against_brute_force(oracle):
    """sgo_cast_disc_body (the cast of VehicleCollisionTesterCastCylinder: a disc rounded by half the wheel's width) vs marching a densely sampled disc -- rim AND
    interior -- along the ray against an independent signed-distance function.  The routine searches the leading half of the rim only; the reference does not know that."""
    rng = np.random.default_rng(11)
    checked = 0
    ang = np.linspace(0, 2 * np.pi, 720, endpoint=False)
    for trial in range(60):
        kind = [abi.SHAPE_SPHERE, abi.SHAPE_BOX, abi.SHAPE_CAPSULE][trial % 3]
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        p = (rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8), rng.uniform(0.2, 0.8))
        if kind == abi.SHAPE_CAPSULE:
            p = (rng.uniform(0.15, 0.4), rng.uniform(0.2, 0.8), 0.0)
        sh = Shape(kind, p, rng.uniform(-1, 1, size=3), tuple(q))
        radius, width = rng.uniform(0.25, 0.45), rng.uniform(0.06, 0.3)
        rho = min(0.5 * width, radius); disc_r = radius - rho
        o = sh.pos + rng.normal(size=3) * 0.5 + np.array([0, 0, 2.2])
        d = sh.pos + rng.uniform(-0.7, 0.7, size=3) - o; d /= np.linalg.norm(d)
        axle = np.cross(d, rng.normal(size=3)); axle /= np.linalg.norm(axle)                 # the cast direction lies in the wheel plane
        e = np.cross(axle, d)
        max_t = 3.0
        rr = np.concatenate([[0.0], np.linspace(0.0, disc_r, 7)[1:]])
        disc = np.concatenate([np.outer(np.cos(ang), e) * r + np.outer(np.sin(ang), d) * r for r in rr])      # points of the flat disc about its centre

        def dist(t):
            return float(_signed_dist_many(sh, disc + (o + d * t)).min()) - rho
        hit = oracle.cast_disc(sh.desc(), o, d, e, d, disc_r, rho, max_t)
        if dist(0.0) <= 0:
            continue
        ts = np.linspace(0, max_t, 1001)
        dd = np.array([dist(t) for t in ts])
        inside = np.nonzero(dd <= 0)[0]
        if len(inside) == 0:
            assert hit is None or dist(hit[0]) > -1e-3, trial                                 # grazing at most
            continue
        lo, hi = ts[inside[0] - 1], ts[inside[0]]
        for _ in range(30):
            mid = 0.5 * (lo + hi)
            if dist(mid) <= 0: hi = mid
            else: lo = mid
        assert hit is not None, trial
        t, n, pt = hit
        assert abs(t - hi) < 5e-4, (trial, t, hi)
        assert abs(sh.signed_dist(pt)) < 3e-4                                                 # the touch point lies on the body
        c = pt + n * rho - (o + d * t)                                                        # the touching sphere's centre, relative to the disc's centre
        assert abs(c @ axle) < 3e-4 and np.linalg.norm(c) < disc_r + 3e-4                     # ... lies on the disc
        checked += 1
    assert checked > 30


def _cylinder(vd):
    vd.collision_tester = abi.VEHICLE_TESTER_CYLINDER


def test_cylinder_tester_equals_the_sphere_tester_on_flat_ground_under_vertical_suspensions(oracle):
    """Wheels whose suspension is perpendicular to flat ground touch it at their lowest point whichever shape is cast: the wheel itself
    (VehicleCollisionTesterCastCylinder) or a sphere of half its width at its bottom (VehicleCollisionTesterCastSphere)."""
    out = []
    for edit in (None, _cylinder):
        w = oracle.OracleWorld(max_bodies=16)
        add_ground(w, friction=1.0)
        body, vid = add_car(w, desc_edit=edit)
        settle(w, 240)
        vs = w.vehicle_get_state(vid)
        assert all(vs["wheels"]["has_contact"][:4]) and all(vs["wheels"]["contact_body"][:4] == 0)
        out.append((np.array(vs["wheels"]["suspension_length"][:4]), np.array(w.get_state([body])[0]["pos"])))
        w.close()
    assert np.max(np.abs(out[0][0] - out[1][0])) < 2e-4 and np.max(np.abs(out[0][1] - out[1][1])) < 2e-4, out
    assert 0.3 < out[1][0].min() and out[1][0].max() < 0.5


def test_cylinder_tester_meets_a_kerb_with_the_wheels_front(oracle):
    """What the cast shape is for: a kerb ahead of the axle, still outside the sphere under it, is met by the wheel's front -- the contact is on the kerb, further forward than the
    axle, and its normal leans back."""
    res = []
    for edit in (None, _cylinder):
        w = oracle.OracleWorld(max_bodies=16)
        add_ground(w, friction=1.0)
        # a kerb 12 cm high whose face stands 20 cm ahead of the front axle (the car looks along +y, its front wheels at y = 1.3): within the wheel's radius
        # (0.42), outside the sphere of half its width (0.08)
        kerb = dyn(w, shape=(2.0, 0.5, 0.06, 0.0), pos=(0.0, 1.3 + 0.20 + 0.5, 0.06), motion=abi.MOTION_STATIC, layer=abi.LAYER_NON_MOVING, friction=1.0)
        body, vid = add_car(w, desc_edit=edit)
        settle(w, 3)
        vs = w.vehicle_get_state(vid)
        front = [i for i in range(4) if vs["wheels"]["contact_position"][i][1] > 0.0]
        res.append((kerb, [int(vs["wheels"]["contact_body"][i]) for i in front], [np.array(vs["wheels"]["contact_normal"][i]) for i in front],
                    [float(vs["wheels"]["contact_position"][i][1]) for i in front]))
        w.close()
    kerb, bodies, normals, ys = res[0]
    assert len(bodies) == 2 and all(b == 0 for b in bodies), res[0]              # the sphere under the axle finds the ground
    kerb, bodies, normals, ys = res[1]
    assert len(bodies) == 2 and all(b == kerb for b in bodies), res[1]           # the wheel finds the kerb
    assert all(y > 1.3 + 0.15 for y in ys) and all(n[1] < -0.2 and n[2] > 0.5 for n in normals), res[1]
