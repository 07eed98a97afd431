"""GPU-vs-oracle parity with static triangle-mesh bodies (sgp_mesh_create + SGP_SHAPE_MESH; the role of JPH::MeshShape /
HeightFieldShape for Substrata's static meshes and terrain, /root/reference/gui_client/PhysicsWorld.cpp:735-1166, TerrainSystem.cpp:1300):
a triangulated terrain and a walled room, spheres / boxes / capsules / hulls dropped on them, rays and capsule queries."""
import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT
import parity

pytestmark = pytest.mark.gpu


def grid_mesh(n, size, height_fn):
    xs = np.linspace(-size, size, n)
    V = np.array([(x, y, height_fn(x, y)) for y in xs for x in xs], np.float32)
    T = []
    for j in range(n - 1):
        for i in range(n - 1):
            a = j * n + i; b = a + 1; c = a + n; d = c + 1
            T += [(a, b, d), (a, d, c)]
    return V, np.array(T, np.uint32)


def room_mesh(half, height):
    """An open box seen from inside: floor + four walls (normals pointing inwards)."""
    h, z = half, height
    V = np.array([(-h, -h, 0), (h, -h, 0), (h, h, 0), (-h, h, 0), (-h, -h, z), (h, -h, z), (h, h, z), (-h, h, z)], np.float32)
    T = [(0, 1, 2), (0, 2, 3),                 # floor, +z
         (0, 4, 5), (0, 5, 1),                 # wall y = -h, normal +y
         (1, 5, 6), (1, 6, 2),                 # wall x = +h, normal -x
         (2, 6, 7), (2, 7, 3),                 # wall y = +h, normal -y
         (3, 7, 4), (3, 4, 0)]                 # wall x = -h, normal +x
    return V, np.array(T, np.uint32)


def mesh_body(info, pos=(0, 0, 0), rot=(0, 0, 0, 1)):
    d = scenes._blank(1)
    d["shape_type"] = abi.SHAPE_MESH; d["shape"][0] = 0; d["shape"][0, 0] = float(info.mesh_id)
    d["pos"][0] = pos; d["rot"][0] = rot
    return d


def test_terrain_and_room_match_oracle(oracle):
    rng = np.random.default_rng(21)
    tw = parity.make_twin(oracle, max_bodies=1024)
    V, T = grid_mesh(33, 16.0, lambda x, y: 0.6 * np.sin(0.5 * x) * np.cos(0.4 * y) + 0.01 * (x * x + y * y))
    terrain_mats = (np.arange(len(T), dtype=np.uint32) * 2654435761 >> 7) % 5          # per-triangle material index (user data)
    ig, ic = tw.mesh_create(V, T, materials=terrain_mats)
    assert (ig.mesh_id, ig.num_triangles) == (ic.mesh_id, ic.num_triangles) == (1, 2048)
    mg, mc = tw.add_batch(mesh_body(ig))
    assert int(mg[0]) == int(mc[0]) == 0
    Vr, Tr = room_mesh(3.0, 12.0)
    rg, rc = tw.mesh_create(Vr, Tr)
    q = (0, 0, np.sin(0.2), np.cos(0.2))
    mg2, mc2 = tw.add_batch(mesh_body(rg, pos=(30.0, 0.0, 0.0), rot=q))           # a second, rotated mesh body far from the terrain
    assert int(mg2[0]) == int(mc2[0]) == 3                                       # (ids 1, 2 are the terrain's alias slots)
    hg, hc = tw.hull_create(rng.normal(size=(12, 3)) * 0.5)
    n_dyn = 0
    for centre, count in (((0.0, 0.0), 90), ((30.0, 0.0), 40)):
        d = scenes.dynamic_bodies(count)
        kinds = rng.integers(0, 4, size=count)
        d["shape_type"] = np.where(kinds == 3, abi.SHAPE_HULL, kinds)
        d["shape"][:, :3] = 0.4
        d["shape"][kinds == 2, 1] = 0.5; d["shape"][kinds == 2, 0] = 0.25
        d["shape"][kinds == 3, 0] = float(hg.hull_id); d["shape"][kinds == 3, 1:] = 0
        spread = 9.0 if count == 90 else 2.0
        d["pos"] = np.column_stack([rng.uniform(centre[0] - spread, centre[0] + spread, count), rng.uniform(centre[1] - spread, centre[1] + spread, count), rng.uniform(2.5, 9.0, count)])
        quat = rng.normal(size=(count, 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
        d["rot"] = quat
        d["lin_vel"][:, :2] = rng.uniform(-2, 2, size=(count, 2))
        tw.add_batch(d)
        n_dyn += count
    total = 6 + n_dyn
    for s in range(1, 481):
        tw.step(DT)
        if s in (1, 60, 180, 300, 480):
            d = parity.compare(tw, total)
            assert d["active_mismatch"] == 0, (s, d)
            assert d["pos"] <= 2e-4 and d["rot"] <= 2e-4 and d["lin_vel"] <= 2e-3 and d["ang_vel"] <= 2e-3, (s, d)
            sg, sc = tw.stats()
            assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points) == (sc.num_pairs, sc.num_manifolds, sc.num_contact_points), s
            assert sg.manifolds_dropped == 0
    print("mesh terrain + room, 480 steps: bit exact =", d["bit_exact"])
    st = tw.gpu.read_states(6, n_dyn)
    assert np.isfinite(st["pos"]).all()
    terrain_z = lambda x, y: 0.6 * np.sin(0.5 * x) * np.cos(0.4 * y) + 0.01 * (x * x + y * y)
    on_terrain = st[:90]
    assert (on_terrain["pos"][:, 2] > terrain_z(on_terrain["pos"][:, 0], on_terrain["pos"][:, 1]) - 0.05).all()     # nothing fell through
    in_room = st[90:]
    c, s_ = np.cos(-0.4), np.sin(-0.4)                                                      # room frame
    lx = c * (in_room["pos"][:, 0] - 30.0) - s_ * in_room["pos"][:, 1]; ly = s_ * (in_room["pos"][:, 0] - 30.0) + c * in_room["pos"][:, 1]
    assert (np.abs(lx) < 3.0).all() and (np.abs(ly) < 3.0).all() and (in_room["pos"][:, 2] > 0.1).all()     # the walls held them
    # rays
    rays = np.zeros(512, dtype=abi.ray_dtype)
    rays["origin"] = np.column_stack([rng.uniform(-15, 34, 512), rng.uniform(-15, 15, 512), rng.uniform(5, 12, 512)])
    dd = rng.normal(size=(512, 3)) * (0.6, 0.6, 0.2) + (0, 0, -1.0); dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    rays["dir"] = dd; rays["max_t"] = 40.0; rays["ignore_id"] = abi.INVALID_ID
    rg_, rc_ = tw.raycast(rays)
    assert np.array_equal(rg_["id"], rc_["id"]) and (rg_["id"] == 0).sum() > 100 and (rg_["id"] == 3).sum() > 5
    assert np.max(np.abs(rg_["t"] - rc_["t"])) <= 1e-4 and np.max(np.abs(rg_["normal"] - rc_["normal"])) <= 1e-5
    # which triangle was hit, its material (MeshShape::GetTriangleUserData -> RayTraceResult::hit_mat_index) and the barycentrics: bit for bit
    on_terrain = rg_["id"] == 0
    assert np.array_equal(rg_["triangle"], rc_["triangle"]) and np.array_equal(rg_["material"], rc_["material"])
    assert np.array_equal(rg_["bary"].view(np.uint32), rc_["bary"].view(np.uint32))
    assert np.array_equal(rg_["material"][on_terrain], terrain_mats[rg_["triangle"][on_terrain]]) and len(np.unique(rg_["material"][on_terrain])) == 5
    assert np.all(rg_["triangle"][(rg_["id"] != 0) & (rg_["id"] != 3)] == abi.INVALID_ID)
    # the hit point rebuilt from the barycentrics lies on the ray
    tri = T[rg_["triangle"][on_terrain]]; u = rg_["bary"][on_terrain][:, :1]; v = rg_["bary"][on_terrain][:, 1:]
    pt = (1 - u - v) * V[tri[:, 0]] + u * V[tri[:, 1]] + v * V[tri[:, 2]]
    assert np.max(np.abs(pt - (rays["origin"][on_terrain] + rays["dir"][on_terrain] * rg_["t"][on_terrain][:, None]))) < 2e-4
    # sphere casts (wheel tester / character sweep) against the meshes and the bodies lying on them
    radii = rng.choice([0.0, 0.08, 0.3], size=512).astype(np.float32)
    rays["max_t"] = rng.uniform(2.0, 25.0, size=512)
    sg_, sc_ = tw.spherecast(rays, radii)
    assert np.array_equal(sg_["id"], sc_["id"]) and (sg_["id"] == 0).sum() > 60
    assert np.max(np.abs(sg_["t"] - sc_["t"])) <= 1e-4 and np.max(np.abs(sg_["normal"] - sc_["normal"])) <= 1e-4
    # capsule queries against the meshes (the character controller's CollideShape)
    qy = np.zeros(128, dtype=abi.capsule_query_dtype)
    px = rng.uniform(-12, 12, 128); py = rng.uniform(-12, 12, 128)
    qy["pos"] = np.column_stack([px, py, terrain_z(px, py) + rng.uniform(0.85, 1.1, 128)])
    qy["pos"][:16] = [(30.0 + 2.6 * np.cos(a), 2.6 * np.sin(a), 0.97) for a in np.linspace(0, 6.2, 16)]     # along the room's walls
    qy["rot"] = (0, 0, 0, 1); qy["radius"] = 0.3; qy["half_height"] = 0.65; qy["max_separation"] = 0.12; qy["ignore_id"] = abi.INVALID_ID; qy["collidable_only"] = 1
    cg, cc = tw.collide_capsules(qy)
    assert len(cg) == len(cc) and len(cg) > 60
    assert np.array_equal(cg["query"], cc["query"]) and np.array_equal(cg["body"], cc["body"])
    for f in ("point", "normal", "distance"):
        assert np.max(np.abs(cg[f] - cc[f])) <= 1e-5, f
    tw.close()


def test_car_on_mesh_terrain_matches_oracle(oracle):
    """A car (hull chassis, four wheel casts per step) driving over a triangulated terrain: wheels find the mesh, GPU == oracle."""
    from helpers import add_car
    tw = parity.make_twin(oracle, max_bodies=256)
    V, T = grid_mesh(41, 40.0, lambda x, y: 0.4 * np.sin(0.3 * x) * np.sin(0.25 * y))
    ig, ic = tw.mesh_create(V, T)
    tw.add_batch(mesh_body(ig))
    ids = []
    for w in (tw.gpu, tw.cpu):
        ids.append(add_car(w, pos=(0.0, -20.0, 1.3)))
    assert ids[0] == ids[1]
    body, vid = ids[0]
    for s in range(1, 421):
        if s == 60:
            tw.vehicle_set_input(vid, 1.0, 0.0, 0.0, 0.0)
        if s == 240:
            tw.vehicle_set_input(vid, 1.0, 0.4, 0.0, 0.0)
        tw.step(DT)
        if s % 60 == 0:
            d = parity.compare(tw, body + 1)
            assert d["active_mismatch"] == 0 and d["pos"] <= 2e-4 and d["lin_vel"] <= 2e-3, (s, d)
            vg, vc = tw.vehicle_get_states(vid, 1)
            assert np.array_equal(vg["wheels"]["contact_body"], vc["wheels"]["contact_body"]) and np.array_equal(vg["wheels"]["angular_velocity"], vc["wheels"]["angular_velocity"])
    st = tw.gpu.get_state([body])[0]
    vs = tw.gpu.vehicle_get_state(vid)
    print("car on terrain: pos", np.round(st["pos"], 2), "bit exact =", d["bit_exact"])
    assert st["pos"][1] > -10.0 and 0.2 < st["pos"][2] < 2.0 and (vs["wheels"]["contact_body"][:4] == 0).sum() >= 2
    tw.close()


def test_car_with_cylinder_tester_on_mesh_terrain_matches_oracle(oracle):
    """The wheel itself as the cast shape (SGP_VEHICLE_TESTER_CYLINDER) against triangles: the search runs per triangle the swept wheel can reach."""
    from helpers import add_car

    def cyl(vd):
        vd.collision_tester = abi.VEHICLE_TESTER_CYLINDER
    tw = parity.make_twin(oracle, max_bodies=256)
    V, T = grid_mesh(41, 40.0, lambda x, y: 0.4 * np.sin(0.3 * x) * np.sin(0.25 * y) + 0.15 * np.sin(1.7 * y))
    ig, ic = tw.mesh_create(V, T)
    tw.add_batch(mesh_body(ig))
    ids = []
    for w in (tw.gpu, tw.cpu):
        ids.append(add_car(w, pos=(0.0, -20.0, 1.5), desc_edit=cyl))
    assert ids[0] == ids[1]
    body, vid = ids[0]
    for s in range(1, 301):
        if s == 60:
            tw.vehicle_set_input(vid, 1.0, 0.0, 0.0, 0.0)
        if s == 200:
            tw.vehicle_set_input(vid, 1.0, 0.4, 0.0, 0.0)
        tw.step(DT)
        if s % 30 == 0:
            d = parity.compare(tw, body + 1)
            assert d["active_mismatch"] == 0 and d["bit_exact"], (s, d)
            vg, vc = tw.vehicle_get_states(vid, 1)
            for f in ("contact_body", "angular_velocity", "suspension_length", "contact_normal", "contact_position"):
                assert np.array_equal(vg["wheels"][f], vc["wheels"][f]), (s, f)
    st = tw.gpu.get_state([body])[0]
    vs = tw.gpu.vehicle_get_state(vid)
    print("car with cast wheels on terrain: pos", np.round(st["pos"], 2))
    assert st["pos"][1] > -15.0 and 0.2 < st["pos"][2] < 2.5 and (vs["wheels"]["contact_body"][:4] == 0).sum() >= 2
    tw.close()


def test_mesh_added_after_a_large_dynamic_body_matches_oracle(oracle):
    """A static mesh body streamed in AFTER (so: with a higher id than) a dynamic body that is itself beyond the large-body radius, the big box
    lying across it.  Both go through the large-body pair kernel; the mesh body's two alias slots must not pair with the box (found by the
    fuzzer in round 2: with the mesh at the higher id the large-large rule let the aliases through, three manifolds became up to nine)."""
    import compound_scene as cs
    tw = parity.make_twin(oracle, max_bodies=256)
    tw.add_batch(scenes.ground())
    big = scenes.dynamic_bodies(1, mass=500.0)
    big["shape_type"] = abi.SHAPE_BOX; big["shape"][0, :3] = (5.0, 3.0, 0.3); big["pos"][0] = (0.0, 0.0, 2.2); big["restitution"] = 0.0
    small = scenes.dynamic_bodies(6)
    small["pos"] = [(-3 + 1.2 * k, 4.5, 1.0) for k in range(6)]
    ig, ic = tw.add_batch(np.concatenate([big, small])); assert np.array_equal(ig, ic)
    tw.set_contact_events(1)
    for _ in range(5):
        tw.step(DT)
    V, T = cs.box_mesh((-1.0, -0.7, 0.0), (1.0, 0.7, 1.4))
    mg, mc = tw.mesh_create(V, T, np.arange(len(T), dtype=np.uint32) % 3); assert mg.mesh_id == mc.mesh_id
    mb = scenes.dynamic_bodies(1)
    mb["motion_type"] = abi.MOTION_STATIC; mb["layer"] = abi.LAYER_NON_MOVING
    mb["shape_type"] = abi.SHAPE_MESH; mb["shape"][0] = (float(mg.mesh_id), 0, 0, 0); mb["pos"][0] = (0.5, 0.2, 0.0)
    jg, jc = tw.add_batch(mb); assert np.array_equal(jg, jc) and int(jg[0]) > int(ig[0])
    touched = False
    for s in range(1, 181):
        tw.step(DT)
        sg, sc = tw.stats()
        assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points) == (sc.num_pairs, sc.num_manifolds, sc.num_contact_points), s
        for ev in (abi.EVENT_CONTACT_ADDED, abi.EVENT_CONTACT_PERSISTED):
            eg, ec = tw.drain_events(ev)
            assert len(eg) == len(ec), (s, ev)
            touched |= any(int(e["id2"]) == int(jg[0]) or int(e["id1"]) == int(jg[0]) for e in eg)
        if s % 30 == 0:
            d = parity.state_diff(tw.gpu.read_states(0, 64), tw.cpu.read_states(0, 64))
            assert d["bit_exact"] and d["active_mismatch"] == 0, (s, d)
    assert touched                                          # the big box did come to rest on the mesh
    tw.close()


def test_large_boxes_on_a_fine_mesh_match_oracle(oracle):
    """Bodies that span many triangles: a car-sized slab on 0.5 m triangles has 100-250 candidate triangles, which the mesh kernel hands from its
    eight-lanes-per-pair launch to the wave-per-pair launch (level-by-level tree walk, 64 triangle tests per round, ordered merge).  The result
    is that of the sequential walk: the oracle's."""
    rng = np.random.default_rng(8)
    tw = parity.make_twin(oracle, max_bodies=256)
    V, T = grid_mesh(65, 16.0, lambda x, y: 0.04 * np.sin(0.9 * x) * np.cos(0.8 * y))       # 0.5 m triangles
    ig, ic = tw.mesh_create(V, T)
    tw.add_batch(mesh_body(ig))
    n = 9
    d = scenes.dynamic_bodies(n, mass=800.0)
    d["shape_type"] = abi.SHAPE_BOX
    d["shape"][:, 0] = 2.2; d["shape"][:, 1] = 1.0; d["shape"][:, 2] = 0.5           # bounds of up to 4.9 m: 200+ candidates for the turned ones (the wave-per-pair launch holds 1024)
    d["shape"][6:, :3] = 0.3                                               # three small ones stay with the eight-lane launch
    gx, gy = np.meshgrid(np.arange(3), np.arange(3))
    d["pos"] = np.column_stack([(gx.ravel() - 1) * 7.0, (gy.ravel() - 1) * 5.0, rng.uniform(0.9, 1.6, n)])
    a = rng.uniform(0, np.pi, n); d["rot"] = np.column_stack([0.05 * rng.normal(size=n), 0.05 * rng.normal(size=n), np.sin(a / 2), np.cos(a / 2)])
    d["rot"] /= np.linalg.norm(d["rot"], axis=1, keepdims=True)
    tw.add_batch(d)
    total = 3 + n
    for s in range(1, 181):
        tw.step(DT)
        if s in (1, 20, 60, 120, 180):
            c = parity.compare(tw, total)
            assert c["active_mismatch"] == 0, (s, c)
            assert c["pos"] <= 2e-4 and c["rot"] <= 2e-4 and c["lin_vel"] <= 2e-3 and c["ang_vel"] <= 2e-3, (s, c)
            sg, sc = tw.stats()
            assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points) == (sc.num_pairs, sc.num_manifolds, sc.num_contact_points), s
            assert sg.manifolds_dropped == sc.manifolds_dropped == 0
    print("large boxes on a fine mesh, 180 steps: bit exact =", c["bit_exact"])
    st = tw.gpu.read_states(3, n)
    assert (st["pos"][:, 2] > 0.2).all() and (st["pos"][:, 2] < 1.0).all()      # resting on the floor
    tw.close()


def test_many_static_meshes_go_through_their_grid(oracle):
    """A parcel grid in small: 100 static mesh buildings (beyond the handful at which the static large bodies get a grid of their own: LargeGrid),
    bodies falling among them, rays, swept spheres and capsule queries through the grid; one building is moved and one removed half way (the grid
    is rebuilt).  The oracle walks every body for everything: same pairs, same hits, same states."""
    rng = np.random.default_rng(17)
    tw = parity.make_twin(oracle, max_bodies=1024)
    tw.add_batch(scenes.ground())
    hx = 3.0
    V = np.array([(-hx, -hx, 0), (hx, -hx, 0), (hx, hx, 0), (-hx, hx, 0), (-hx, -hx, 4), (hx, -hx, 4), (hx, hx, 4), (-hx, hx, 4)], np.float32)
    T = np.array([(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (1, 2, 6), (1, 6, 5), (2, 3, 7), (2, 7, 6), (3, 0, 4), (3, 4, 7)], np.uint32)
    ig, ic = tw.mesh_create(V, T)
    side = 10
    d = scenes._blank(side * side)
    d["shape_type"] = abi.SHAPE_MESH; d["shape"][:] = 0; d["shape"][:, 0] = float(ig.mesh_id)
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    d["pos"] = np.column_stack([(gx.ravel() - side / 2) * 11.0, (gy.ravel() - side / 2) * 11.0, np.zeros(side * side)])
    a = rng.uniform(0, np.pi, side * side); d["rot"] = np.column_stack([np.zeros_like(a), np.zeros_like(a), np.sin(a / 2), np.cos(a / 2)])
    mg, mc = tw.add_batch(d)
    assert np.array_equal(mg, mc)
    n_dyn = 300
    b = scenes.dynamic_bodies(n_dyn)
    b["shape_type"] = rng.integers(0, 3, n_dyn)
    b["shape"][:, :3] = 0.4; b["shape"][b["shape_type"] == 2, 0] = 0.25
    b["pos"] = np.column_stack([rng.uniform(-52, 52, n_dyn), rng.uniform(-52, 52, n_dyn), rng.uniform(5.0, 9.0, n_dyn)])
    tw.add_batch(b)
    total = 1 + 3 * side * side + n_dyn

    def queries():
        rays = np.zeros(512, dtype=abi.ray_dtype)
        rays["origin"] = np.column_stack([rng.uniform(-55, 55, 512), rng.uniform(-55, 55, 512), rng.uniform(1.0, 12.0, 512)])
        dd = rng.normal(size=(512, 3)) * (1.0, 1.0, 0.3); rays["dir"] = dd / np.linalg.norm(dd, axis=1, keepdims=True)
        rays["max_t"] = rng.uniform(5.0, 80.0, 512); rays["ignore_id"] = abi.INVALID_ID
        rg_, rc_ = tw.raycast(rays)
        assert np.array_equal(rg_["id"], rc_["id"]) and np.array_equal(rg_["triangle"], rc_["triangle"]) and (rg_["id"] != abi.INVALID_ID).sum() > 100
        assert np.max(np.abs(rg_["t"] - rc_["t"])) <= 1e-4
        radii = rng.choice([0.0, 0.1, 0.3], size=512).astype(np.float32)
        rays["max_t"] = rng.uniform(1.0, 6.0, 512)
        sg_, sc_ = tw.spherecast(rays, radii)
        assert np.array_equal(sg_["id"], sc_["id"]) and np.max(np.abs(sg_["t"] - sc_["t"])) <= 1e-4
        qy = np.zeros(64, dtype=abi.capsule_query_dtype)
        k = rng.integers(0, side * side, 64); ang = rng.uniform(0, 2 * np.pi, 64)
        qy["pos"] = np.column_stack([d["pos"][k, 0] + 3.9 * np.cos(ang), d["pos"][k, 1] + 3.9 * np.sin(ang), np.full(64, 0.97)])     # around the buildings' walls
        qy["rot"] = (0, 0, 0, 1); qy["radius"] = 0.3; qy["half_height"] = 0.65; qy["max_separation"] = 0.12; qy["ignore_id"] = abi.INVALID_ID; qy["collidable_only"] = 1
        cg, cc = tw.collide_capsules(qy)
        assert len(cg) == len(cc) and len(cg) >= 64                            # the ground at least
        assert np.array_equal(cg["query"], cc["query"]) and np.array_equal(cg["body"], cc["body"])
        assert np.max(np.abs(cg["point"] - cc["point"])) <= 1e-5

    for s in range(1, 241):
        tw.step(DT)
        if s == 120:
            for w in (tw.gpu, tw.cpu):
                w.set_pose_shape(int(mg[37]), (3.0, -2.0, 0.0), (0, 0, 0, 1), (float(ig.mesh_id), 0, 0, 0))      # a building moves ...
                w.remove(int(mg[58]))                                                                              # ... and one is torn down
        if s in (1, 60, 119, 121, 180, 240):
            c = parity.compare(tw, total)
            assert c["active_mismatch"] == 0 and c["pos"] <= 2e-4 and c["lin_vel"] <= 2e-3, (s, c)
            sg, sc = tw.stats()
            assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points) == (sc.num_pairs, sc.num_manifolds, sc.num_contact_points), s
            queries()
    print("100 static meshes through the large bodies' grid, 240 steps: bit exact =", c["bit_exact"])
    tw.close()


def test_kinematic_mesh_platform_carries_bodies(oracle):
    """A scripted object with a mesh shape is a KINEMATIC mesh body (the reference builds a MeshShape for everything that is not dynamic and
    moves it with MoveKinematic: PhysicsWorld.cpp:706-731, 1290): a platform mesh that rises and slides carries the boxes on it by friction,
    a swinging door mesh pushes a ball.  The alias slots behind a mesh body follow its pose and velocities.  GPU == oracle."""
    rng = np.random.default_rng(4)
    tw = parity.make_twin(oracle, max_bodies=256)
    tw.add_batch(scenes.ground())
    V, T = room_mesh(2.5, 0.6)                              # a tray: floor + low rim
    ig, ic = tw.mesh_create(V, T)
    plat = mesh_body(ig, pos=(0.0, 0.0, 0.5))
    plat["motion_type"] = abi.MOTION_KINEMATIC; plat["layer"] = abi.LAYER_MOVING; plat["activate"] = 1
    pg, pc = tw.add_batch(plat)
    Vd = np.array([(-0.1, -1.5, 0), (0.1, -1.5, 0), (0.1, 1.5, 0), (-0.1, 1.5, 0), (-0.1, -1.5, 2.5), (0.1, -1.5, 2.5), (0.1, 1.5, 2.5), (-0.1, 1.5, 2.5)], np.float32)
    Td = np.array([(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (1, 2, 6), (1, 6, 5), (2, 3, 7), (2, 7, 6), (3, 0, 4), (3, 4, 7)], np.uint32)
    dg, dc = tw.mesh_create(Vd, Td)
    door = mesh_body(dg, pos=(12.0, 0.0, 0.0))
    door["motion_type"] = abi.MOTION_KINEMATIC; door["layer"] = abi.LAYER_MOVING; door["activate"] = 1
    dgid, dcid = tw.add_batch(door)
    assert int(pg[0]) == int(pc[0]) and int(dgid[0]) == int(dcid[0])
    n = 8
    b = scenes.dynamic_bodies(n, mass=10.0, friction=0.8)
    b["shape_type"] = np.arange(n) % 3
    b["shape"][:, :3] = 0.3; b["shape"][np.arange(n) % 3 == 2, 0] = 0.2
    b["pos"] = np.column_stack([rng.uniform(-1.5, 1.5, n), rng.uniform(-1.5, 1.5, n), rng.uniform(1.2, 2.0, n)])
    tw.add_batch(b)
    ball = scenes.dynamic_bodies(1, mass=5.0); ball["shape_type"] = abi.SHAPE_SPHERE; ball["shape"][0] = (0.4, 0, 0, 0); ball["pos"][0] = (12.9, 0.8, 0.4)
    tw.add_batch(ball)
    total = 1 + 3 + 3 + n + 1
    for s in range(1, 301):
        t = s * DT
        ppos = (0.8 * np.sin(0.8 * t), 0.0, 0.5 + 0.5 * min(t, 2.0))                   # rises for two seconds while sliding to and fro
        a = 0.6 * np.sin(1.2 * t)                                                        # the door swings about z
        for w in (tw.gpu, tw.cpu):
            w.move_kinematic(int(pg[0]), ppos, (0, 0, 0, 1), DT)
            w.move_kinematic(int(dgid[0]), (12.0, 0.0, 0.0), (0, 0, np.sin(a / 2), np.cos(a / 2)), DT)
        tw.step(DT)
        if s in (1, 30, 90, 150, 240, 300):
            c = parity.compare(tw, total)
            assert c["active_mismatch"] == 0 and c["pos"] <= 2e-4 and c["rot"] <= 2e-4 and c["lin_vel"] <= 2e-3, (s, c)
            sg, sc = tw.stats()
            assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points) == (sc.num_pairs, sc.num_manifolds, sc.num_contact_points), s
    print("kinematic mesh platform + door, 300 steps: bit exact =", c["bit_exact"])
    st = tw.gpu.read_states(7, n)
    plat_now = tw.gpu.get_state([int(pg[0])])[0]
    assert abs(float(plat_now["pos"][2]) - 1.5) < 1e-3                                   # the platform went where it was told
    assert (st["pos"][:, 2] > 1.5).all() and (np.abs(st["pos"][:, 0] - plat_now["pos"][0]) < 2.6).all()     # ... and took its load along
    ballst = tw.gpu.read_states(7 + n, 1)[0]
    assert np.linalg.norm(ballst["pos"][:2] - np.float32([12.9, 0.8])) > 0.3             # the door pushed the ball
    tw.close()


def test_capsules_around_small_meshes_same_constraints(oracle):
    """Capsule against triangle without a search (closest approach to the three edges, the ends, the piercing point): 200 small box meshes at random
    orientations with five capsules around each, one step -- the constraint lists of device and oracle are the same, entry for entry (a bogus
    candidate -- the axis crossing a triangle's plane beside the triangle -- once produced hundreds of contacts half a metre from anything)."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__)))
    import compound_scene as cs_
    rng = np.random.default_rng(1)
    tw = parity.make_twin(oracle, max_bodies=4096)
    NM, PER = 200, 5
    bv, bt = cs_.box_mesh((-0.6, -0.4, 0.0), (0.6, 0.4, 0.9))
    ig, ic = tw.mesh_create(bv, bt)
    d = scenes._blank(NM); d["shape_type"] = abi.SHAPE_MESH; d["shape"][:] = 0; d["shape"][:, 0] = float(ig.mesh_id)
    cen = np.column_stack([(np.arange(NM) % 15) * 6.0, (np.arange(NM) // 15) * 6.0, np.full(NM, 3.0)])
    d["pos"] = cen
    q = rng.normal(size=(NM, 4)); d["rot"] = q / np.linalg.norm(q, axis=1, keepdims=True)
    tw.add_batch(d)
    b = scenes.dynamic_bodies(NM * PER)
    b["shape_type"] = abi.SHAPE_CAPSULE; b["shape"][:, 0] = rng.uniform(0.15, 0.4, NM * PER); b["shape"][:, 1] = rng.uniform(0.2, 0.8, NM * PER)
    b["pos"] = np.repeat(cen, PER, axis=0) + rng.normal(size=(NM * PER, 3)) * 0.55
    q = rng.normal(size=(NM * PER, 4)); b["rot"] = q / np.linalg.norm(q, axis=1, keepdims=True)
    b["gravity_factor"] = 0.0
    tw.add_batch(b)
    tw.step(DT)
    cg, cc = tw.gpu.dump_constraints(), tw.cpu.dump_constraints()
    og, oc = np.lexsort((cg["b"], cg["a"])), np.lexsort((cc["b"], cc["a"]))
    assert len(cg) == len(cc) and 300 < len(cg) < 2500
    for f in ("a", "b", "np", "n", "bias"):
        assert np.array_equal(np.ascontiguousarray(cg[f][og]).view(np.uint8), np.ascontiguousarray(cc[f][oc]).view(np.uint8)), f
    # every contact is a contact: the capsule is within reach of its mesh (no constraint between bodies half a metre apart)
    st = tw.gpu.read_states(0, 3 * NM + NM * PER)
    for c in cg[:: max(1, len(cg) // 200)]:
        m, x = int(c["a"]), int(c["b"])
        if m >= 3 * NM:
            continue                                        # (two capsules)
        assert np.linalg.norm(st[x]["pos"] - st[m - m % 3]["pos"]) < 0.9 + 0.8 + 0.4 + 0.75 + 0.1      # box half diagonal + half height + radius + offset of the box centre
    tw.close()


def test_large_dynamic_bodies_on_grid_resident_static_meshes_pair_once(oracle):
    """Moving bodies beyond the large-body radius stay on the large bodies' linear list, static meshes (>= 32 of them) sit in the LargeGrid: a
    slab that rests across several buildings meets each of them through BOTH routes of k_bp_large (the building's thread walking the list, the
    slab's own query of the grid), and the pair must come out once whichever of the two has the lower id (round 3's advisor found it emitted
    twice when the static body's id was the higher one).  One slab is created before the buildings, one after."""
    tw = parity.make_twin(oracle, max_bodies=512)
    tw.add_batch(scenes.ground())

    def slab(x):
        b = scenes.dynamic_bodies(1, mass=900.0)
        b["shape_type"] = abi.SHAPE_BOX; b["shape"][0, :3] = (6.0, 4.5, 0.3); b["pos"][0] = (x, 0.0, 5.2); b["restitution"] = 0.0
        return b
    s0g, s0c = tw.add_batch(slab(-16.5)); assert np.array_equal(s0g, s0c)
    hx = 2.0
    V = np.array([(-hx, -hx, 0), (hx, -hx, 0), (hx, hx, 0), (-hx, hx, 0), (-hx, -hx, 4), (hx, -hx, 4), (hx, hx, 4), (-hx, hx, 4)], np.float32)
    T = np.array([(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (1, 2, 6), (1, 6, 5), (2, 3, 7), (2, 7, 6), (3, 0, 4), (3, 4, 7)], np.uint32)
    ig, ic = tw.mesh_create(V, T)
    side = 6
    d = scenes._blank(side * side)
    d["shape_type"] = abi.SHAPE_MESH; d["shape"][:] = 0; d["shape"][:, 0] = float(ig.mesh_id)
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    d["pos"] = np.column_stack([(gx.ravel() - side / 2) * 5.5, (gy.ravel() - side / 2) * 5.5, np.zeros(side * side)])
    mg, mc = tw.add_batch(d); assert np.array_equal(mg, mc)
    s1g, s1c = tw.add_batch(slab(8.25)); assert np.array_equal(s1g, s1c)
    assert int(s0g[0]) < int(mg[0]) < int(s1g[0])
    small = scenes.dynamic_bodies(24)
    small["pos"] = [(-14.0 + 1.3 * (k % 12), -2.0 + 3.0 * (k // 12), 7.0) for k in range(24)]
    tw.add_batch(small)
    total = 1 + 1 + 3 * side * side + 1 + 24
    on_mesh = set()
    tw.set_contact_events(1)
    for s in range(1, 151):
        tw.step(DT)
        sg, sc = tw.stats()
        assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points) == (sc.num_pairs, sc.num_manifolds, sc.num_contact_points), s
        for ev in (abi.EVENT_CONTACT_ADDED, abi.EVENT_CONTACT_PERSISTED):
            eg, ec = tw.drain_events(ev)
            assert len(eg) == len(ec), (s, ev)
            for e in eg:
                a, b = int(e["id1"]), int(e["id2"])
                for slab_id in (int(s0g[0]), int(s1g[0])):
                    if slab_id in (a, b): on_mesh.add((slab_id, a + b - slab_id))
        if s % 30 == 0:
            c = parity.compare(tw, total)
            assert c["bit_exact"] and c["active_mismatch"] == 0, (s, c)
    mesh_ids = set(int(m) for m in mg)
    assert any(o in mesh_ids for (sl, o) in on_mesh if sl == int(s0g[0])) and any(o in mesh_ids for (sl, o) in on_mesh if sl == int(s1g[0]))
    tw.close()


def test_active_edges_flags_and_sliding_bodies_match_oracle(oracle):
    """Round 4: JPH::MeshShape's active edges.  The flags the product computes for a rolling terrain, a flat floor with a step and a room are the
    sequential CPU statement's bit for bit; bodies of every kind sliding across the seams of those meshes (contacts on inactive edges take the
    triangle's normal: ActiveEdges::FixNormal) stay bit-identical; and on the flat part nothing is slowed down by a seam."""
    import ctypes as C
    rng = np.random.default_rng(9)
    tw = parity.make_twin(oracle, max_bodies=512)
    V, T = grid_mesh(25, 12.0, lambda x, y: 0.0 if x < 4.0 else 0.35 * (x - 4.0) + 0.25 * np.sin(0.9 * y))      # flat floor running into rolling ground
    ig, ic = tw.mesh_create(V, T)
    fg = np.zeros(len(T), np.uint8); fc = np.zeros(len(T), np.uint8)
    assert tw.gpu._lib.sgp_mesh_edge_flags(tw.gpu._h, ig.mesh_id, fg.ctypes.data, len(T)) == 0
    f = oracle.lib().sgo_mesh_edge_flags; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    assert f(tw.cpu._h, ic.mesh_id, fc.ctypes.data, len(T)) == 0
    assert np.array_equal(fg, fc)
    assert (fg == 7).sum() < len(T) // 4 and (fg == 0).sum() > len(T) // 4          # mostly seams of the flat part, active edges on the rolling part and the rim
    mg, mc = tw.add_batch(mesh_body(ig))
    hg, hc = tw.hull_create(rng.normal(size=(14, 3)) * 0.4)
    n = 60
    d = scenes.dynamic_bodies(n)
    d["pos"][:, 0] = rng.uniform(-10, -2, n); d["pos"][:, 1] = rng.uniform(-10, 10, n); d["pos"][:, 2] = rng.uniform(0.6, 1.2, n)
    d["lin_vel"][:, 0] = rng.uniform(2.0, 6.0, n); d["lin_vel"][:, 1] = rng.uniform(-1.5, 1.5, n)
    d["friction"] = 0.05
    kind = np.arange(n) % 4
    for i in range(n):
        if kind[i] == 0: d["shape_type"][i] = abi.SHAPE_BOX; d["shape"][i, :3] = rng.uniform(0.25, 0.5, 3)
        elif kind[i] == 1: d["shape_type"][i] = abi.SHAPE_SPHERE; d["shape"][i, :3] = (rng.uniform(0.25, 0.45), 0, 0)
        elif kind[i] == 2: d["shape_type"][i] = abi.SHAPE_CAPSULE; d["shape"][i, :3] = (0.25, 0.45, 0)
        else: d["shape_type"][i] = abi.SHAPE_HULL; d["shape"][i] = (float(hg.hull_id), 0, 0, 0)
    # one frictionless sphere that stays on the flat part: the seams must not slow it down
    probe = scenes.dynamic_bodies(1); probe["shape_type"] = abi.SHAPE_SPHERE; probe["shape"][0, :3] = (0.4, 0, 0)
    probe["pos"][0] = (-11.0, 0.3, 0.4); probe["lin_vel"][0] = (3.0, 0.0, 0.0); probe["friction"] = 0.0; probe["linear_damping"] = 0.0; probe["angular_damping"] = 0.0
    d = np.concatenate([d, probe])
    ig_, ic_ = tw.add_batch(d)
    assert np.array_equal(ig_, ic_)
    nb = 3 + n + 1
    for s in range(240):
        tw.step(DT)
        if s in (0, 20, 60, 120, 239):
            dd = parity.compare(tw, nb)
            assert dd["bit_exact"] and dd["active_mismatch"] == 0, (s, dd)
        if s == 150:
            pv = tw.gpu.get_state([int(ig_[-1])])[0]
            assert abs(pv["lin_vel"][0] - 3.0) < 1e-3 and abs(pv["pos"][2] - 0.4) < 0.03, pv      # ~30 seams crossed, none felt (it rides the penetration slop deep)
    tw.close()


def test_static_meshes_streaming_in_and_out_keep_the_grid_exact(oracle):
    """Round 4: a client that streams parcel objects in and out.  While a grid of the static large bodies stands, a new building waits on the linear
    list and a removed one leaves a dead entry behind (the grid is rebuilt only every 64 newcomers, when a quarter of it is dead, or when a dead
    entry's body slot is handed out again) -- pairs, rays, casts, capsule contacts and states stay the sequential CPU statement's."""
    rng = np.random.default_rng(23)
    tw = parity.make_twin(oracle, max_bodies=2048)
    tw.add_batch(scenes.ground())
    hx = 3.0
    V = np.array([(-hx, -hx, 0), (hx, -hx, 0), (hx, hx, 0), (-hx, hx, 0), (-hx, -hx, 4), (hx, -hx, 4), (hx, hx, 4), (-hx, hx, 4)], np.float32)
    T = np.array([(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (1, 2, 6), (1, 6, 5), (2, 3, 7), (2, 7, 6), (3, 0, 4), (3, 4, 7)], np.uint32)
    ig, ic = tw.mesh_create(V, T)
    side = 8

    def building(x, y, ang):
        d = scenes._blank(1)
        d["shape_type"] = abi.SHAPE_MESH; d["shape"][:] = 0; d["shape"][:, 0] = float(ig.mesh_id)
        d["pos"][0] = (x, y, 0.0); d["rot"][0] = (0, 0, np.sin(ang / 2), np.cos(ang / 2))
        return d
    live = {}
    for gy in range(side):
        for gx in range(side):
            mg, mc = tw.add_batch(building((gx - side / 2) * 11.0, (gy - side / 2) * 11.0, float(rng.uniform(0, np.pi))))
            assert int(mg[0]) == int(mc[0])
            live[int(mg[0])] = True
    n_dyn = 200
    b = scenes.dynamic_bodies(n_dyn)
    b["shape_type"] = rng.integers(0, 3, n_dyn)
    b["shape"][:, :3] = 0.4; b["shape"][b["shape_type"] == 2, 0] = 0.25
    b["pos"] = np.column_stack([rng.uniform(-50, 50, n_dyn), rng.uniform(-50, 50, n_dyn), rng.uniform(5.0, 9.0, n_dyn)])
    b["lin_vel"][:, :2] = rng.uniform(-3, 3, (n_dyn, 2))
    tw.add_batch(b)
    added = removed = 0
    for s in range(1, 301):
        if s % 3 == 0:                                   # a building appears (often in the body slots one that left has freed) ...
            mg, mc = tw.add_batch(building(float(rng.uniform(-60, 60)), float(rng.uniform(-60, 60)), float(rng.uniform(0, np.pi))))
            assert int(mg[0]) == int(mc[0]) and int(mg[0]) != abi.INVALID_ID
            live[int(mg[0])] = True; added += 1
        if s % 4 == 0 and len(live) > 20:                # ... and one goes
            k = int(rng.choice(sorted(live)))
            tw.remove(k); del live[k]; removed += 1
        tw.step(DT)
        if s % 30 == 0 or s in (3, 4, 5, 13):
            total = tw.gpu.num_bodies()
            assert total == tw.cpu.num_bodies()
            c = parity.state_diff(tw.gpu.read_states(0, 2048), tw.cpu.read_states(0, 2048))
            assert c["active_mismatch"] == 0 and c["bit_exact"], (s, c)
            sg, sc = tw.stats()
            assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points) == (sc.num_pairs, sc.num_manifolds, sc.num_contact_points), s
            rays = np.zeros(256, dtype=abi.ray_dtype)
            rays["origin"] = np.column_stack([rng.uniform(-60, 60, 256), rng.uniform(-60, 60, 256), rng.uniform(1.0, 12.0, 256)])
            dd = rng.normal(size=(256, 3)) * (1.0, 1.0, 0.3); rays["dir"] = dd / np.linalg.norm(dd, axis=1, keepdims=True)
            rays["max_t"] = rng.uniform(5.0, 80.0, 256); rays["ignore_id"] = abi.INVALID_ID
            rg_, rc_ = tw.raycast(rays)
            assert np.array_equal(rg_["id"], rc_["id"]) and np.array_equal(rg_["triangle"], rc_["triangle"])
            radii = rng.choice([0.0, 0.1, 0.3], size=256).astype(np.float32)
            rays["max_t"] = rng.uniform(1.0, 6.0, 256)
            sg_, sc_ = tw.spherecast(rays, radii)
            assert np.array_equal(sg_["id"], sc_["id"])
    assert added >= 90 and removed >= 70
    tw.close()


def test_axis_aligned_boxes_on_axis_aligned_triangles_bit_exact(oracle):
    """The degenerate corners of the closed-form box - triangle search (sgd_tri_box_sat): boxes whose edges run exactly along the triangles' edges (cross
    products that are exactly zero: the parallel test), box centres exactly above a triangle's edge or centroid (an axis exactly perpendicular to the
    centre offset: neither sense of a cube edge is turned, the two senses are opposite axes), identity / half-turn rotations (matrix entries that are
    exactly 0 and +-1: zero components whose sign the sums decide).  Flat floors of axis-aligned and of diagonal triangles, a ramp.  Bit for bit."""
    tw = parity.make_twin(oracle, max_bodies=512)
    V, T = grid_mesh(9, 8.0, lambda x, y: 0.0)                      # 2 m squares cut along one diagonal: edges along x, along y and along (1, 1)
    ig, ic = tw.mesh_create(V, T)
    tw.add_batch(mesh_body(ig))
    Vr = np.array([(20, -4, 0), (28, -4, 0), (28, 4, 2), (20, 4, 2)], np.float32)      # a ramp of two triangles, rising along y
    ir, _ = tw.mesh_create(Vr, np.array([(0, 1, 2), (0, 2, 3)], np.uint32))
    tw.add_batch(mesh_body(ir))
    rots = [(0, 0, 0, 1), (0, 0, 1, 0), (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, np.sqrt(0.5), np.sqrt(0.5))]
    pos = []
    for ix in range(-3, 4):
        for iy in range(-3, 4):
            pos.append((ix * 2.0, iy * 2.0, 0.5 + 0.01 * ((ix + iy) % 3)))          # centres on the grid's vertices (where six triangles meet) ...
    for ix in range(-3, 3):
        pos.append((ix * 2.0 + 1.0, ix * 2.0 + 1.0, 0.75))                      # ... and on the diagonals' midpoints
    for k in range(6):
        pos.append((21.0 + k, -3.0 + k, 1.6 + 0.25 * k))                        # over the ramp
    n = len(pos)
    d = scenes.dynamic_bodies(n)
    d["shape_type"] = abi.SHAPE_BOX
    d["shape"][:, 0] = 0.5; d["shape"][:, 1] = 0.25; d["shape"][:, 2] = 0.5
    d["shape"][::3, :3] = 0.5
    d["pos"] = np.array(pos, np.float32)
    d["rot"] = np.array([rots[k % len(rots)] for k in range(n)], np.float32)
    tw.add_batch(d)
    total = 6 + n
    for s in range(1, 241):
        tw.step(DT)
        if s in (1, 2, 5, 30, 120, 240):
            c = parity.compare(tw, total)
            assert c["bit_exact"], (s, c)
            sg, sc = tw.stats()
            assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points) == (sc.num_pairs, sc.num_manifolds, sc.num_contact_points), s
    st = tw.gpu.read_states(6, n)
    assert (st["pos"][:49, 2] > 0.2).all() and (st["pos"][:49, 2] < 0.8).all()      # the boxes over the floor rest on it
    tw.close()
