"""GPU-vs-oracle parity with convex hull bodies (sgp_hull_create + SGP_SHAPE_HULL; the role of JPH::ConvexHullShape for dynamic
meshes and vehicle bodies, /root/reference/gui_client/PhysicsWorld.cpp:735-1166, CarPhysics.cpp:66-92): hulls of several kinds
dropped together with boxes, spheres and capsules, rays against them, and a car whose chassis is the reference's 12-point hull."""
import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT, quat_axis_angle
from test_oracle_hull import CAR_HULL
import parity

pytestmark = pytest.mark.gpu


def hull_descs(info, positions, rng, mass):
    d = scenes.dynamic_bodies(len(positions), mass=mass)
    d["shape_type"] = abi.SHAPE_HULL
    d["shape"][:, 0] = float(info.hull_id); d["shape"][:, 1:] = 0
    d["pos"] = positions
    q = rng.normal(size=(len(positions), 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    d["rot"] = q.astype(np.float32)
    return d


def test_hull_bodies_match_oracle(oracle):
    rng = np.random.default_rng(77)
    tw = parity.make_twin(oracle, max_bodies=1024)
    tw.add_batch(scenes.ground())
    infos = []
    for pts in (CAR_HULL, [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1)], rng.normal(size=(12, 3)) * (0.4, 0.6, 0.5), rng.normal(size=(40, 3)) * 0.5,
                [(x, y, z) for x in (-0.5, 0.5) for y in (-0.3, 0.3) for z in (-0.2, 0.2)]):
        ig, ic = tw.hull_create(pts)
        assert (ig.hull_id, ig.num_vertices, ig.num_faces, ig.num_edges) == (ic.hull_id, ic.num_vertices, ic.num_faces, ic.num_edges)
        assert np.array_equal(np.array(ig.com[:]), np.array(ic.com[:])) and np.array_equal(np.array(ig.rot[:]), np.array(ic.rot[:]))
        assert ig.volume == ic.volume and list(ig.unit_inertia) == list(ic.unit_inertia)
        infos.append(ig)
    n_per = 12
    total = 1
    for k, info in enumerate(infos):
        pos = rng.uniform([-4, -4, 1.0], [4, 4, 9.0], size=(n_per, 3)).astype(np.float32)
        tw.add_batch(hull_descs(info, pos, rng, mass=1200.0 if k == 0 else 40.0))
        total += n_per
    mixed = scenes.small_mixed(4, 2, seed=5)[1:]
    mixed["pos"][:, 2] += 6.0
    tw.add_batch(mixed)
    total += len(mixed)
    for s in range(1, 421):
        tw.step(DT)
        if s in (1, 30, 120, 240, 420):
            d = parity.compare(tw, total)
            assert d["active_mismatch"] == 0, (s, d)
            assert d["pos"] <= 2e-4 and d["rot"] <= 2e-4 and d["lin_vel"] <= 2e-3 and d["ang_vel"] <= 2e-3, (s, d)
            sg, sc = tw.stats()
            assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points) == (sc.num_pairs, sc.num_manifolds, sc.num_contact_points), s
    print("hull pile, 420 steps: bit exact =", d["bit_exact"])
    st = tw.gpu.read_states(0, total)
    assert (st["pos"][1:, 2] > 0.05).all() and np.isfinite(st["pos"]).all()
    # rays through the pile
    rays = np.zeros(256, dtype=abi.ray_dtype)
    rays["origin"] = rng.uniform([-5, -5, 6], [5, 5, 8], size=(256, 3)); rays["dir"] = (0, 0, -1); rays["max_t"] = 20.0; rays["ignore_id"] = abi.INVALID_ID
    hg, hc = tw.raycast(rays)
    assert np.array_equal(hg["id"], hc["id"]) and np.max(np.abs(hg["t"] - hc["t"])) <= 1e-5
    assert np.max(np.abs(hg["normal"] - hc["normal"])) <= 1e-5
    tw.close()


def test_car_with_the_reference_hull_as_chassis(oracle):
    """CarPhysics' body: the 12-point hull (model space: y up, z forward) in its centre-of-mass / principal frame; wheels are given in
    the same frame.  Driven over a few boxes; GPU and oracle agree on bodies and drivetrain."""
    from test_oracle_hull import hull_body
    from test_collide_independent import quat_to_mat
    tw = parity.make_twin(oracle, max_bodies=256)
    tw.add_batch(scenes.ground())
    ig, ic = tw.hull_create(CAR_HULL)
    q_obj = quat_axis_angle((1, 0, 0), np.pi / 2)                       # model y-up -> world z-up
    ids = []
    for w, info in ((tw.gpu, ig), (tw.cpu, ic)):
        b = hull_body(w, info, pos_obj=(0, 0, 1.0), rot_obj=q_obj, mass=1200.0, restitution=0.0)
        vd = w.default_vehicle_desc(b)
        # wheel frame data: default desc is z-up / y-forward about the body origin; express it in the hull's body frame
        Rb = quat_to_mat(info.rot[:])                                    # body frame in model space
        M = quat_to_mat(q_obj)                                           # model -> world(z-up) at spawn
        to_body = (M @ Rb).T                                             # world-aligned offsets -> body frame
        com_w = M @ np.array(info.com[:])
        for i in range(4):
            wd = vd.wheels[i]
            p = np.array(wd.position[:]) + np.array([0, 0, 0.25]) - com_w   # the default layout assumes the box chassis' centre; the hull's com sits higher
            wd.position[:] = tuple(to_body @ p)
            for name in ("suspension_dir", "steering_axis", "wheel_up", "wheel_forward"):
                getattr(wd, name)[:] = tuple(to_body @ np.array(getattr(wd, name)[:]))
        vd.up[:] = tuple(to_body @ np.array([0, 0, 1.0])); vd.forward[:] = tuple(to_body @ np.array([0, 1.0, 0]))
        ids.append((b, w.vehicle_create(vd)))
    assert ids[0] == ids[1]
    body, vid = ids[0]
    debris = scenes.dynamic_bodies(24)
    debris["pos"] = np.random.default_rng(3).uniform([-3, 4, 0.5], [3, 30, 0.5], size=(24, 3)).astype(np.float32)
    debris["shape"][:, :3] = 0.25
    tw.add_batch(debris)
    n = 2 + len(debris)
    for s in range(1, 361):
        if s == 60:
            tw.vehicle_set_input(vid, 1.0, 0.0, 0.0, 0.0)
        if s == 200:
            tw.vehicle_set_input(vid, 1.0, 0.3, 0.0, 0.0)
        tw.step(DT)
        if s % 60 == 0:
            d = parity.compare(tw, n)
            assert d["active_mismatch"] == 0 and d["pos"] <= 2e-4 and d["lin_vel"] <= 2e-3, (s, d)
            vg, vc = tw.vehicle_get_states(vid, 1)
            assert np.array_equal(vg["wheels"]["angular_velocity"], vc["wheels"]["angular_velocity"]) and np.array_equal(vg["engine_rpm"], vc["engine_rpm"])
    st = tw.gpu.get_state([body])[0]
    print("hull-chassis car: pos", np.round(st["pos"], 2), "bit exact =", d["bit_exact"])
    assert st["pos"][1] > 8.0 and 0.5 < st["pos"][2] < 1.5
    tw.close()
