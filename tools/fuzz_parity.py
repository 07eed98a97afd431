#!/usr/bin/env python3
"""Randomised differential test: the HIP path against the oracle on random scenes and random call sequences.

Each seed builds a scene from every shape kind the product knows (boxes, spheres, capsules, convex hulls, optionally a static triangle
terrain and a walled mesh pen), sensors, a kinematic mover, optionally a car, then interleaves steps with random facade-level calls
(teleports with velocities, forces, removals, additions, layer changes, activation, water on/off, contact events) and with queries (rays,
sphere casts, capsule contacts).  After every checkpoint the two worlds must agree bit for bit (states, statistics, query answers).

    python tools/fuzz_parity.py --seeds 0-49 --steps 240        (on the GPU box)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from substrata_amd import abi, scenes          # noqa: E402
from helpers import DT, add_car, add_bike, quat_axis_angle  # noqa: E402
import parity                                  # noqa: E402


def terrain(rng, n=14, size=36.0):
    xs = np.linspace(-size / 2, size / 2, n)
    k = rng.uniform(0.15, 0.4, 2); amp = rng.uniform(0.2, 0.9)
    v = np.array([[x, y, amp * (np.sin(k[0] * x) + np.cos(k[1] * y)) - 0.2] for y in xs for x in xs], np.float32)
    t = []
    for j in range(n - 1):
        for i in range(n - 1):
            a, b, c, d = j * n + i, j * n + i + 1, (j + 1) * n + i, (j + 1) * n + i + 1
            t += [[a, b, d], [a, d, c]]
    return v, np.array(t, np.uint32)


def random_hull_points(rng):
    kind = rng.random()
    if kind < 0.25:      # (round 5) a hull beyond 32 vertices: points on an ellipsoid -- every one a corner -- or a prism / truncated cone over a many-sided polygon
        if rng.random() < 0.6:
            n = int(rng.integers(33, 257))
            p = rng.normal(size=(n, 3)); p /= np.linalg.norm(p, axis=1, keepdims=True)
            return (p * rng.uniform(0.25, 0.6, 3)).astype(np.float32)
        m = int(rng.integers(17, 100)); a = np.linspace(0, 2 * np.pi, m, endpoint=False)
        r0, r1, hh = rng.uniform(0.25, 0.6), rng.uniform(0.25, 0.6), rng.uniform(0.15, 0.5)
        return np.array([(r0 * np.cos(t), r0 * np.sin(t), -hh) for t in a] + [(r1 * np.cos(t), r1 * np.sin(t), hh) for t in a], np.float32)
    n = int(rng.integers(6, 20))
    p = rng.normal(size=(n, 3)) * rng.uniform(0.25, 0.6, 3)
    return p.astype(np.float32)


def run_seed(oracle, seed, steps, verbose=False):
    rng = np.random.default_rng(seed)
    # both launch plans of small worlds get their share (the product reads the switch when the world is created)
    os.environ["SGP_NO_SMALL_WORLD"] = "1" if rng.random() < 0.4 else "0"
    # ... and of the large ones: colours with launches of their own (threshold), the rest by connected component (budget in per mille; 0 = tail kernel)
    os.environ["SGP_ROWS_IN_SMALL_WORLDS"] = "1"      # (takes effect in the seeds that run without the small-world kernel: compact rows, or with the next line none)
    os.environ["SGP_ROWS_MODE2_MIN"] = str(int(rng.choice([0, 65536])))      # (0: the no-rows layout, the default of worlds with 65k+ constraints, in a world of a few hundred)
    os.environ["SGP_TAIL_THRESHOLD"] = str(int(rng.choice([2, 8, 256])))
    os.environ["SGP_HC_BUDGET"] = str(int(rng.choice([0, 160, 400, 1000])))
    os.environ["SGP_HC_MIN_COLOURS"] = str(int(rng.choice([0, 0, 4])))      # (0: the component launch even where it replaces a single colour)
    tw = parity.make_twin(oracle, max_bodies=2048)
    for k in ("SGP_NO_SMALL_WORLD", "SGP_TAIL_THRESHOLD", "SGP_HC_BUDGET", "SGP_HC_MIN_COLOURS"):
        os.environ.pop(k, None)
    use_mesh = rng.random() < 0.6
    use_car = rng.random() < 0.5
    ground = scenes.ground()
    if use_mesh:
        ground["pos"][0, 2] = -3.0          # the mesh is the floor; the quad far below catches what leaves it
    tw.add_batch(ground)
    if use_mesh:
        v, t = terrain(rng)
        if rng.random() < 0.5:      # a pen: four inward-facing walls around the middle of the terrain (corners give several manifolds per body)
            w_, h_ = float(rng.uniform(5.0, 8.0)), 4.0
            base = len(v)
            corners = [(-w_, -w_), (w_, -w_), (w_, w_), (-w_, w_)]
            wv = [(x, y, z) for (x, y) in corners for z in (-1.0, h_)]
            v = np.concatenate([v, np.array(wv, np.float32)])
            tt = []
            for k in range(4):
                a0, a1 = base + 2 * k, base + 2 * k + 1
                b0, b1 = base + 2 * ((k + 1) % 4), base + 2 * ((k + 1) % 4) + 1
                tt += [[a0, a1, b1], [a0, b1, b0]]          # facing the inside of the pen
            t = np.concatenate([t, np.array(tt, np.uint32)])
        mats = rng.integers(0, 5, len(t)).astype(np.uint32) if rng.random() < 0.6 and not os.environ.get("FUZZ_NO_MATS") else None      # material index per triangle (ray hits report it)
        mg, mc = tw.mesh_create(v, t, mats)
        assert mg.mesh_id == mc.mesh_id
        m = scenes.dynamic_bodies(1)
        m["motion_type"] = abi.MOTION_STATIC; m["layer"] = abi.LAYER_NON_MOVING
        m["shape_type"] = abi.SHAPE_MESH; m["shape"][0] = (float(mg.mesh_id), 0, 0, 0)
        m["pos"][0] = (0, 0, 0)
        ig, ic = tw.add_batch(m)
        assert np.array_equal(ig, ic)
    hulls = []
    for _ in range(int(rng.integers(0, 4))):
        pts_ = random_hull_points(rng)
        off_ = tuple(rng.uniform(-0.1, 0.1, 3)) if rng.random() < 0.4 else None
        hg, hc = tw.hull_create(pts_, off_)
        assert hg.hull_id == hc.hull_id
        hulls.append(hg)
    # static compounds (StaticCompoundShape): random children -- boxes, spheres, capsules, a hull, a small mesh -- around the arena
    compounds = []
    for _ in range(0 if os.environ.get("FUZZ_NO_COMPOUND") else int(rng.integers(0, 3))):
        nch = int(rng.integers(1, 5))
        ch = np.zeros(nch, dtype=abi.compound_child_dtype)
        qq = rng.normal(size=(nch, 4)); ch["rot"] = (qq / np.linalg.norm(qq, axis=1, keepdims=True)).astype(np.float32)
        ch["pos"] = rng.uniform([-1.5, -1.5, 0.0], [1.5, 1.5, 2.0], (nch, 3)).astype(np.float32)
        for k in range(nch):
            kk = int(rng.integers(0, 5))
            if kk == 0:
                ch["shape_type"][k] = abi.SHAPE_BOX; ch["shape"][k, :3] = rng.uniform(0.2, 0.9, 3)
            elif kk == 1:
                ch["shape_type"][k] = abi.SHAPE_SPHERE; ch["shape"][k, 0] = rng.uniform(0.2, 0.7)
            elif kk == 2:
                ch["shape_type"][k] = abi.SHAPE_CAPSULE; ch["shape"][k, :2] = (rng.uniform(0.15, 0.4), rng.uniform(0.2, 0.8))
            elif kk == 3 and hulls:
                ch["shape_type"][k] = abi.SHAPE_HULL; ch["shape"][k, 0] = float(hulls[int(rng.integers(len(hulls)))].hull_id)
            else:
                import compound_scene as cs_
                bv, bt = cs_.box_mesh((-0.6, -0.4, 0.0), (0.6, 0.4, float(rng.uniform(0.3, 1.2))))
                cmg, cmc = tw.mesh_create(bv, bt, np.arange(len(bt), dtype=np.uint32) % 3)
                assert cmg.mesh_id == cmc.mesh_id
                ch["shape_type"][k] = abi.SHAPE_MESH; ch["shape"][k, 0] = float(cmg.mesh_id); ch["rot"][k] = (0, 0, 0, 1)
        base = scenes._blank(1)
        base["pos"][0] = tuple(rng.uniform([-7, -7, 0.0], [7, 7, 0.6])); qb = rng.normal(size=4); qb[:2] *= 0.1; base["rot"][0] = qb / np.linalg.norm(qb)
        base["friction"] = 0.6; base["restitution"] = 0.1; base["userdata"] = 5000 + len(compounds)
        cg_, cc_ = tw.add_compound(base, ch)
        assert cg_ == cc_, (seed, "compound id", cg_, cc_)
        if cg_ != abi.INVALID_ID:
            compounds.append(cg_)
    stream = None                                   # a streamed-in static mesh object: added, later removed and its shape destroyed, then another
    big = rng.random() < 0.2                        # one scene in five is crowded: deeper piles, more colours, bodies with many contacts
    n = int(rng.integers(500, 1300)) if big else int(rng.integers(40, 160))
    d = scenes.dynamic_bodies(n)
    d["pos"] = rng.uniform([-8, -8, 1.0], [8, 8, 25.0 if big else 9.0], size=(n, 3)).astype(np.float32)
    q = rng.normal(size=(n, 4)); d["rot"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    kind = rng.integers(0, 4 if hulls else 3, n)
    scale = rng.uniform(0.3, 1.2, n).astype(np.float32)
    for i in range(n):
        if kind[i] == 0:
            d["shape_type"][i] = abi.SHAPE_BOX; d["shape"][i, :3] = scale[i] * rng.uniform(0.4, 1.0, 3)
        elif kind[i] == 1:
            d["shape_type"][i] = abi.SHAPE_SPHERE; d["shape"][i, :3] = (scale[i] * 0.6, 0, 0)
        elif kind[i] == 2:
            d["shape_type"][i] = abi.SHAPE_CAPSULE; d["shape"][i, :3] = (scale[i] * 0.35, scale[i] * 0.6, 0)
        else:
            h = hulls[int(rng.integers(len(hulls)))]
            d["shape_type"][i] = abi.SHAPE_HULL; d["shape"][i] = (float(h.hull_id), 0, 0, 0)
            d["pos"][i] -= 0  # hull bodies are placed by their body frame; fine for a random scene
    for i in rng.choice(n, size=int(rng.integers(0, 3)), replace=False):      # bodies beyond the broad phase's large-body radius
        d["shape_type"][i] = abi.SHAPE_BOX; d["shape"][i, :3] = (float(rng.uniform(3.5, 6.0)), float(rng.uniform(2.0, 5.0)), 0.3); scale[i] = 3.0
        d["pos"][i, 2] = 0.6 + 0.7 * float(rng.random())
        d["rot"][i] = (0, 0, 0, 1)
    d["mass"] = (20.0 * scale ** 3 + 1.0).astype(np.float32)
    d["friction"] = rng.uniform(0.0, 1.0, n).astype(np.float32)
    d["restitution"] = rng.choice([0.0, 0.2, 0.6], n).astype(np.float32)
    d["lin_vel"] = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    d["is_sensor"] = (rng.random(n) < 0.04).astype(d["is_sensor"].dtype)
    d["allow_sleeping"] = (rng.random(n) < 0.9).astype(d["allow_sleeping"].dtype)
    ig, ic = tw.add_batch(d)
    assert np.array_equal(ig, ic)
    live = [int(x) for x in ig if x != abi.INVALID_ID]
    kind_of = {int(x): int(d["shape_type"][k]) for k, x in enumerate(ig) if x != abi.INVALID_ID}
    kin = scenes.dynamic_bodies(1)
    kin["motion_type"] = abi.MOTION_KINEMATIC; kin["shape"][0, :3] = (1.5, 0.4, 0.6); kin["pos"][0] = (-10.0, 0.0, 1.2)
    kg, kc = tw.add_batch(kin); kid = int(kg[0]); assert kid == int(kc[0])
    vid = None
    vids = []
    # every third seed: the vehicles cast their wheels themselves (VehicleCollisionTesterCastCylinder) -- chosen by the seed, not by the generator, so that the scenes
    # of the other seeds stay what they were
    cyl = seed % 3 == 1

    def tester(vd):
        if cyl:
            vd.collision_tester = abi.VEHICLE_TESTER_CYLINDER
        return vd
    if use_car:
        if rng.random() < 0.5:      # the reference's hull chassis with its lowered centre of mass (what config 5 uses)
            cd = scenes.dynamic_bodies(1, mass=1200.0)
            cd["pos"][0] = (9.0, -9.0, 2.0); cd["restitution"] = 0.0
            cdx = cd.copy(); ih2 = scenes.use_car_hull(tw.gpu, cdx, [0]); cdy = cd.copy(); ih3 = scenes.use_car_hull(tw.cpu, cdy, [0])
            assert ih2.hull_id == ih3.hull_id
            bg = tw.gpu.add_batch(cdx); bc = tw.cpu.add_batch(cdy)
            assert np.array_equal(bg, bc)
            cb = int(bg[0])
            vg = tw.gpu.vehicle_create(tester(tw.gpu.default_vehicle_desc(cb))); vc = tw.cpu.vehicle_create(tester(tw.cpu.default_vehicle_desc(cb)))
            assert vg == vc
        else:
            (cb, vg), (cb2, vc) = add_car(tw.gpu, pos=(9.0, -9.0, 2.0), desc_edit=tester), add_car(tw.cpu, pos=(9.0, -9.0, 2.0), desc_edit=tester)
            assert cb == cb2 and vg == vc
        vid = vg; vids.append(vg)
        # round 4 (two-body wheel rows): things for the wheels to stand on -- a loose slab under the car, a second car on the same slab (the
        # device then solves it after the first: StepCounters::veh_deferred), a third one dropped onto the first, flat debris the wheels roll over
        if rng.random() < 0.6 and not os.environ.get("FUZZ_NO_SLAB"):
            sl = scenes.dynamic_bodies(1, mass=float(rng.uniform(300.0, 3000.0)), friction=0.8)
            sl["shape"][0, :3] = (3.2, 5.0, 0.12); sl["pos"][0] = (9.0, -7.5, 0.2)
            ig, ic = tw.add_batch(sl); assert np.array_equal(ig, ic)
            if rng.random() < 0.6:
                (cb3, vg3), (cb4, vc3) = add_car(tw.gpu, pos=(9.0, -4.9, 2.6), desc_edit=tester), add_car(tw.cpu, pos=(9.0, -4.9, 2.6), desc_edit=tester)
                assert cb3 == cb4 and vg3 == vc3
                vids.append(vg3)
            if rng.random() < 0.3:
                (cb5, vg5), (cb6, vc5) = add_car(tw.gpu, pos=(9.0, -9.0, 3.4), mass=400.0, desc_edit=tester), add_car(tw.cpu, pos=(9.0, -9.0, 3.4), mass=400.0, desc_edit=tester)
                assert cb5 == cb6 and vg5 == vc5
                vids.append(vg5)
        if rng.random() < 0.6:
            nf = int(rng.integers(4, 16))
            fl = scenes.dynamic_bodies(nf, mass=float(rng.uniform(5.0, 80.0)))
            fl["shape"][:, :3] = rng.uniform([0.3, 0.3, 0.05], [0.7, 0.7, 0.12], (nf, 3))
            fl["pos"] = rng.uniform([5.0, -13.0, 0.5], [12.0, -2.0, 1.5], (nf, 3)).astype(np.float32)
            ig, ic = tw.add_batch(fl); assert np.array_equal(ig, ic)
    if rng.random() < 0.3:
        (bb, vg), (bb2, vc) = add_bike(tw.gpu, pos=(-9.0, 9.0, 2.0), desc_edit=tester), add_bike(tw.cpu, pos=(-9.0, 9.0, 2.0), desc_edit=tester)
        assert bb == bb2 and vg == vc
        vids.append(vg)
    tw.set_contact_events(int(rng.random() < 0.5))
    water = False
    max_deferred = 0
    wake_pairs = 0
    for s in range(1, steps + 1):
        r = rng.random()
        if r < 0.04 and live:                                   # teleport with velocities
            i = int(rng.choice(live))
            qq = rng.normal(size=4); qq /= np.linalg.norm(qq)
            tw.set_pose_vel(i, tuple(rng.uniform([-6, -6, 2], [6, 6, 8])), tuple(qq), tuple(rng.uniform(-4, 4, 3)), tuple(rng.uniform(-3, 3, 3)))
        elif r < 0.08 and live:                                 # forces
            i = int(rng.choice(live)); tw.activate(i)
            tw.add_force(i, tuple(rng.uniform(-3000, 3000, 3))); tw.add_torque(i, tuple(rng.uniform(-200, 200, 3)))
            tw.add_force_at(i, tuple(rng.uniform(-500, 500, 3)), tuple(rng.uniform(-5, 5, 3)))
        elif r < 0.10 and len(live) > 10:                       # removal
            i = int(rng.choice(live)); live.remove(i); tw.remove(i)
        elif r < 0.13:                                          # additions (slots get reused)
            nb = scenes.dynamic_bodies(2)
            nb["pos"] = rng.uniform([-5, -5, 5], [5, 5, 9], size=(2, 3)).astype(np.float32)
            nb["shape_type"] = [abi.SHAPE_SPHERE, abi.SHAPE_BOX]; nb["shape"][0, :3] = (0.4, 0, 0); nb["shape"][1, :3] = (0.3, 0.5, 0.2)
            ag, ac = tw.add_batch(nb); assert np.array_equal(ag, ac)
            live += [int(x) for x in ag if x != abi.INVALID_ID]
            for k, x in enumerate(ag):
                if x != abi.INVALID_ID:
                    kind_of[int(x)] = int(nb["shape_type"][k])
        elif r < 0.15 and live:                                 # layer change
            i = int(rng.choice(live)); tw.set_layer(i, int(rng.choice([abi.LAYER_MOVING, abi.LAYER_MOVING_NON_COLLIDABLE])))
        elif r < 0.16:
            water = not water; tw.set_water(int(water), float(rng.uniform(0.0, 1.5)))
        elif r < 0.18 and live:                                 # setNewObToWorldTransform with a new scale (primitive shapes)
            i = int(rng.choice(live))
            stt = tw.gpu.get_state([i])[0]
            if kind_of.get(i) == abi.SHAPE_BOX:
                tw.set_pose_shape(i, tuple(stt["pos"]), tuple(stt["rot"]), tuple(rng.uniform(0.2, 0.9, 3)) + (0.0,))
            elif kind_of.get(i) == abi.SHAPE_SPHERE:
                tw.set_pose_shape(i, tuple(stt["pos"]), tuple(stt["rot"]), (float(rng.uniform(0.2, 0.7)), 0.0, 0.0, 0.0))
        elif r < 0.20 and live:
            i = int(rng.choice(live)); tw.set_vel(i, tuple(rng.uniform(-5, 5, 3)), tuple(rng.uniform(-4, 4, 3)))
        elif r < 0.235 and compounds and rng.random() < 0.5:    # a compound is moved (every child follows) or removed
            cid = int(rng.choice(compounds))
            if rng.random() < 0.3:
                compounds.remove(cid); tw.remove(cid)
            else:
                qq = rng.normal(size=4); qq[:2] *= 0.1; qq /= np.linalg.norm(qq)
                tw.set_pose_vel(cid, tuple(rng.uniform([-7, -7, 0.0], [7, 7, 0.8])), tuple(qq), (0, 0, 0), (0, 0, 0))
        elif r < 0.25 and not os.environ.get("FUZZ_NO_STREAM"):   # streaming: a static mesh object comes and goes (shape ids and body slots get reused)
            import compound_scene as cs_
            if stream is None:
                bv, bt = cs_.box_mesh((-1.0, -0.7, 0.0), (1.0, 0.7, float(rng.uniform(0.4, 1.5))))
                smg, smc = tw.mesh_create(bv, bt, np.arange(len(bt), dtype=np.uint32) % 4)
                assert smg.mesh_id == smc.mesh_id
                mb = scenes.dynamic_bodies(1)
                mb["motion_type"] = abi.MOTION_STATIC; mb["layer"] = abi.LAYER_NON_MOVING
                mb["shape_type"] = abi.SHAPE_MESH; mb["shape"][0] = (float(smg.mesh_id), 0, 0, 0)
                mb["pos"][0] = tuple(rng.uniform([-6, -6, 0.0], [6, 6, 0.5]))
                sg_, sc_ = tw.add_batch(mb); assert np.array_equal(sg_, sc_)
                if sg_[0] != abi.INVALID_ID:
                    stream = (int(sg_[0]), int(smg.mesh_id))
                else:
                    tw.mesh_destroy(int(smg.mesh_id))
            else:
                tw.remove(stream[0]); tw.mesh_destroy(stream[1]); stream = None
        elif r < 0.27 and len(live) > 8:                        # a burst of network snapshots (batched setNewObToWorldTransform)
            ids_ = rng.choice(live, size=6, replace=False).astype(np.uint32)
            recs_ = np.zeros(6, dtype=abi.pose_vel_dtype)
            recs_["pos"] = rng.uniform([-6, -6, 1.5], [6, 6, 7], (6, 3)); qq = rng.normal(size=(6, 4)); recs_["rot"] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
            recs_["lin_vel"] = rng.uniform(-3, 3, (6, 3)); recs_["ang_vel"] = rng.uniform(-2, 2, (6, 3))
            tw.set_pose_vel_batch(ids_, recs_)
        tw.move_kinematic(kid, (float(-10.0 + 18.0 * (0.5 - 0.5 * np.cos(s * 0.03))), 0.0, 1.2), quat_axis_angle((0, 0, 1), 0.01 * s), DT)
        for k, v in enumerate(vids):
            inp = dict(forward=float(np.float32(np.sin(0.02 * s + k) > -0.3)), right=float(np.float32(0.5 * np.sin(0.05 * s + 2 * k))), brake=float((s + 40 * k) % 120 > 100))
            tw.vehicle_set_input(v, **inp)
        tw.step(DT)
        if len(vids) > 1:
            dg_, dc_ = tw.gpu.stats().num_deferred_vehicles, tw.cpu.stats().num_deferred_vehicles
            assert dg_ == dc_, (seed, s, "deferred vehicles", dg_, dc_)
            max_deferred = max(max_deferred, dg_)
        wg_, wc_ = tw.gpu.stats().num_wake_pairs, tw.cpu.stats().num_wake_pairs
        assert wg_ == wc_, (seed, s, "pairs of the in-step activation round", wg_, wc_)
        wake_pairs += wg_
        for ev in (abi.EVENT_CONTACT_ADDED, abi.EVENT_CONTACT_PERSISTED, abi.EVENT_ACTIVATED, abi.EVENT_DEACTIVATED, abi.EVENT_ENTERED_WATER):
            eg, ec = tw.drain_events(ev)
            if len(eg) != len(ec):
                def brief(e):
                    return [tuple(int(e[n_][k]) for n_ in e.dtype.names if n_ in ("id", "id1", "id2", "body", "a", "b", "kind", "type")) for k in range(min(len(e), 12))]
                # which constraints differ (pairs present on one side only, or with another point count), and what the bodies are
                dg, dc = tw.gpu.dump_constraints(), tw.cpu.dump_constraints()
                kg = {(int(c["a"]), int(c["b"])): int(c["np"]) for c in dg}; kc = {(int(c["a"]), int(c["b"])): int(c["np"]) for c in dc}
                odd = sorted(k for k in set(kg) | set(kc) if kg.get(k) != kc.get(k))[:6]
                kinds = {i: kind_of.get(i) for k in odd for i in k}
                raise AssertionError((seed, s, "event count", ev, len(eg), len(ec), "constraints that differ (gpu np, oracle np)", [(k, kg.get(k), kc.get(k)) for k in odd], "shape types", kinds, "stream", stream, "compounds", compounds))
            if len(eg):      # the events of THIS step, same payloads (both sides deliver them sorted by body ids)
                for f in eg.dtype.names:
                    if f.startswith("userdata"):
                        continue
                    if not np.array_equal(np.ascontiguousarray(eg[f]).view(np.uint8), np.ascontiguousarray(ec[f]).view(np.uint8)):
                        k = int(np.flatnonzero((np.ascontiguousarray(eg[f]).reshape(len(eg), -1) != np.ascontiguousarray(ec[f]).reshape(len(ec), -1)).any(axis=1))[0])
                        raise AssertionError((seed, s, "event payload", ev, f, k, {n_: np.asarray(eg[n_][k]).tolist() for n_ in eg.dtype.names}, {n_: np.asarray(ec[n_][k]).tolist() for n_ in ec.dtype.names}))
        if s % 40 == 0 or s == steps:
            sg, sc = tw.stats()
            tg = (sg.num_pairs, sg.num_manifolds, sg.num_contact_points, sg.num_colours, sg.num_active, sg.num_overflow_constraints, sg.num_cached_manifolds)
            tc = (sc.num_pairs, sc.num_manifolds, sc.num_contact_points, sc.num_colours, sc.num_active, sc.num_overflow_constraints, sc.num_cached_manifolds)
            assert tg == tc, (seed, s, "stats (pairs, manifolds, points, colours, active, overflow, cached)", tg, tc)
            hi = tw.gpu.num_bodies() + 64
            dd = parity.state_diff(tw.gpu.read_states(0, 2048), tw.cpu.read_states(0, 2048))
            assert dd["bit_exact"] and dd["active_mismatch"] == 0, (seed, s, dd)
            # queries
            rays = np.zeros(24, dtype=abi.ray_dtype)
            rays["origin"] = rng.uniform([-9, -9, 3], [9, 9, 10], (24, 3)); dirv = rng.normal(size=(24, 3)); dirv[:, 2] = -np.abs(dirv[:, 2]) - 0.3
            rays["dir"] = dirv / np.linalg.norm(dirv, axis=1, keepdims=True); rays["max_t"] = 30.0; rays["ignore_id"] = abi.INVALID_ID
            hg, hc = tw.raycast(rays)
            if not (np.array_equal(hg["id"], hc["id"]) and np.array_equal(hg["t"].view(np.uint32), hc["t"].view(np.uint32))):
                badr = np.flatnonzero((hg["id"] != hc["id"]) | (hg["t"].view(np.uint32) != hc["t"].view(np.uint32)))
                raise AssertionError((seed, s, "rays", [(int(k), int(hg["id"][k]), float(hg["t"][k]), int(hc["id"][k]), float(hc["t"][k]), rays["origin"][k].tolist(), rays["dir"][k].tolist()) for k in badr[:3]]))
            for f in ("normal", "triangle", "material", "bary", "sub_shape", "userdata"):      # what traceRay hands back besides the distance
                if not np.array_equal(np.ascontiguousarray(hg[f]).view(np.uint8), np.ascontiguousarray(hc[f]).view(np.uint8)):
                    k = int(np.flatnonzero((np.ascontiguousarray(hg[f]).reshape(len(hg), -1) != np.ascontiguousarray(hc[f]).reshape(len(hc), -1)).any(axis=1))[0])
                    raise AssertionError((seed, s, "ray hit field", f, k, {n_: np.asarray(hg[n_][k]).tolist() for n_ in hg.dtype.names}, {n_: np.asarray(hc[n_][k]).tolist() for n_ in hc.dtype.names}))
            radii = rng.uniform(0.1, 0.5, 24).astype(np.float32)
            cg, cc = tw.spherecast(rays, radii)
            if not (np.array_equal(cg["id"], cc["id"]) and np.array_equal(cg["t"].view(np.uint32), cc["t"].view(np.uint32))):
                bad = np.flatnonzero((cg["id"] != cc["id"]) | (cg["t"].view(np.uint32) != cc["t"].view(np.uint32)))
                info = []
                for k in bad[:4]:
                    ids = [int(cg["id"][k]), int(cc["id"][k])]
                    sts = tw.gpu.get_state([i for i in ids if i != abi.INVALID_ID])
                    info.append((int(k), ids, float(cg["t"][k]), float(cc["t"][k]), float(radii[k]), [int(x) for x in sts["shape_type"]] if "shape_type" in sts.dtype.names else None,
                                 rays["origin"][k].tolist(), rays["dir"][k].tolist()))
                raise AssertionError((seed, s, "casts", info))
            qs = np.zeros(6, dtype=abi.capsule_query_dtype)
            qs["pos"] = rng.uniform([-8, -8, 0.3], [8, 8, 3.0], (6, 3)); qq = rng.normal(size=(6, 4)); qs["rot"] = qq / np.linalg.norm(qq, axis=1, keepdims=True)
            qs["radius"] = 0.3; qs["half_height"] = 0.65; qs["max_separation"] = 0.1; qs["ignore_id"] = abi.INVALID_ID
            kg, kc = tw.collide_capsules(qs)
            assert len(kg) == len(kc) and np.array_equal(kg["body"], kc["body"]) and np.array_equal(kg["distance"].view(np.uint32), kc["distance"].view(np.uint32)) \
                and np.array_equal(kg["normal"].view(np.uint32), kc["normal"].view(np.uint32)) and np.array_equal(kg["sub_shape"], kc["sub_shape"]), (seed, s, "capsule queries")
            for v in vids:
                vg_, vc_ = tw.vehicle_get_state(v)
                if vg_.tobytes() != vc_.tobytes():
                    diffs = []
                    for name in vg_.dtype.names:
                        if name == "wheels":
                            for wi in range(len(vg_["wheels"])):
                                for wn in vg_["wheels"].dtype.names:
                                    a, b = vg_["wheels"][wi][wn], vc_["wheels"][wi][wn]
                                    if np.asarray(a).tobytes() != np.asarray(b).tobytes():
                                        diffs.append((f"wheel{wi}.{wn}", np.asarray(a).tolist(), np.asarray(b).tolist()))
                        elif np.asarray(vg_[name]).tobytes() != np.asarray(vc_[name]).tobytes():
                            diffs.append((name, np.asarray(vg_[name]).tolist(), np.asarray(vc_[name]).tolist()))
                    raise AssertionError((seed, s, "vehicle state", v, diffs[:6]))
    st = tw.gpu.stats()
    if verbose:
        print(f"seed {seed}: mesh {use_mesh} car {use_car} hulls {len(hulls)} bodies {tw.gpu.num_bodies()} manifolds {st.num_manifolds} colours {st.num_colours} vehicles {len(vids)} (deferred <= {max_deferred}, {wake_pairs} pairs of woken bodies): ok")
    tw.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0-19")
    ap.add_argument("--steps", type=int, default=240)
    args = ap.parse_args()
    lo, _, hi = args.seeds.partition("-")
    seeds = range(int(lo), int(hi or lo) + 1)
    from oracle import oracle
    oracle.build()
    failed = []
    for seed in seeds:
        try:
            run_seed(oracle, seed, args.steps, verbose=True)
        except AssertionError as e:
            print(f"seed {seed}: MISMATCH {str(e)[:700]}")
            failed.append(seed)
    print(f"{len(seeds) - len(failed)} of {len(seeds)} seeds bit-exact; failed: {failed}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
