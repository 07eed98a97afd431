"""The two batched world queries of the character controller (sgp_collide_capsules, sgp_spherecast; what JPH::CharacterVirtual asks
the world, /root/reference/gui_client/PlayerPhysics.cpp:258-353,477-481): GPU against the oracle on a settled mixed pile with hulls."""
import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT
import parity

pytestmark = pytest.mark.gpu


def test_capsule_contacts_and_sphere_casts_match_oracle(oracle):
    rng = np.random.default_rng(8)
    tw = parity.make_twin(oracle, max_bodies=1024)
    descs = scenes.small_mixed(8, 3, seed=12)
    tw.add_batch(descs)
    ig, ic = tw.hull_create(rng.normal(size=(14, 3)) * 0.6)
    hd = scenes.dynamic_bodies(20)
    hd["shape_type"] = abi.SHAPE_HULL; hd["shape"][:, 0] = float(ig.hull_id); hd["shape"][:, 1:] = 0
    hd["pos"] = rng.uniform([-4, -4, 1], [4, 4, 5], size=(20, 3))
    tw.add_batch(hd)
    sens = scenes.dynamic_bodies(1); sens["is_sensor"] = 1; sens["motion_type"] = abi.MOTION_STATIC; sens["pos"][0] = (0, 0, 1.0); sens["shape"][0, :3] = 1.0
    tw.add_batch(sens)
    for _ in range(200):
        tw.step(DT)
    n = 256
    q = np.zeros(n, dtype=abi.capsule_query_dtype)
    q["pos"] = rng.uniform([-5, -5, 0.2], [5, 5, 3.0], size=(n, 3))
    quat = rng.normal(size=(n, 4)); quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    q["rot"] = quat; q["rot"][: n // 2] = (0, 0, 0, 1)                       # half of them upright like a player
    q["radius"] = 0.3; q["half_height"] = 0.65; q["max_separation"] = 0.12
    q["ignore_id"] = abi.INVALID_ID; q["ignore_id"][::7] = 5
    q["collidable_only"] = 1
    cg, cc = tw.collide_capsules(q)
    assert len(cg) == len(cc) and len(cg) > 100
    assert np.array_equal(cg["query"], cc["query"]) and np.array_equal(cg["body"], cc["body"])
    for f in ("point", "normal", "distance", "point_velocity", "inv_mass"):
        assert np.max(np.abs(cg[f] - cc[f])) <= 1e-5, f
    assert np.array_equal(cg["motion_type"], cc["motion_type"]) and np.array_equal(cg["is_sensor"], cc["is_sensor"])
    assert (cg["distance"] <= 0.12 + 1e-5).all() and (cg["is_sensor"] == 1).any() and (cg["distance"] < 0).any()
    assert not ((cg["body"] == 5) & (q["ignore_id"][cg["query"]] == 5)).any()
    # sphere casts: downwards from above the pile, sideways through it
    rays = np.zeros(n, dtype=abi.ray_dtype)
    rays["origin"] = rng.uniform([-5, -5, 3.0], [5, 5, 6.0], size=(n, 3))
    d = rng.normal(size=(n, 3)) * (0.5, 0.5, 0.2) + (0, 0, -1.0); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["dir"] = d; rays["max_t"] = rng.uniform(0.5, 8.0, size=n); rays["ignore_id"] = abi.INVALID_ID; rays["collidable_only"] = 1
    radii = rng.choice([0.0, 0.1, 0.3], size=n).astype(np.float32)
    hg, hc = tw.spherecast(rays, radii)
    assert np.array_equal(hg["id"], hc["id"]) and (hg["id"] != abi.INVALID_ID).sum() > 100
    assert np.max(np.abs(hg["t"] - hc["t"])) <= 1e-5 and np.max(np.abs(hg["normal"] - hc["normal"])) <= 1e-5
    # a thicker cast never travels further than a thinner one along the same ray
    h0, _ = tw.spherecast(rays, np.zeros(n, np.float32)); h3, _ = tw.spherecast(rays, np.full(n, 0.3, np.float32))
    both = (h0["id"] != abi.INVALID_ID) & (h3["id"] != abi.INVALID_ID)
    assert (h3["t"][both] <= h0["t"][both] + 1e-4).all()
    tw.close()


def test_capsule_queries_on_a_mesh_with_active_edges_match_oracle(oracle):
    """what CharacterVirtual::GetContactsAtPosition asks (CollideOnlyWithActive + the direction of travel), on a terrain mesh: the same contacts on both sides,
    and a query standing beside a seam of a flat stretch gets the flat normal with the flag and a leaning one without"""
    from test_mesh_parity_gpu import grid_mesh, mesh_body
    rng = np.random.default_rng(4)
    tw = parity.make_twin(oracle, max_bodies=64)
    V, T = grid_mesh(25, 12.0, lambda x, y: 0.0 if abs(x) < 4 and abs(y) < 4 else 0.35 * np.sin(0.9 * x) * np.cos(0.8 * y))
    ig, ic = tw.mesh_create(V, T)
    tw.add_batch(mesh_body(ig))
    n = 512
    q = np.zeros(n, dtype=abi.capsule_query_dtype)
    xy = rng.uniform(-10, 10, size=(n, 2))
    h = np.where((np.abs(xy[:, 0]) < 4) & (np.abs(xy[:, 1]) < 4), 0.0, 0.35 * np.sin(0.9 * xy[:, 0]) * np.cos(0.8 * xy[:, 1]))
    q["pos"] = np.column_stack([xy, h + 0.3 + 0.65 + rng.uniform(-0.05, 0.04, n)])
    q["rot"] = (0, 0, 0, 1)
    q["radius"] = 0.3; q["half_height"] = 0.65; q["max_separation"] = 0.08; q["ignore_id"] = abi.INVALID_ID; q["collidable_only"] = 1
    mv = rng.normal(size=(n, 3)) * (1, 1, 0.2); mv /= np.linalg.norm(mv, axis=1, keepdims=True)
    q["movement"] = mv; q["active_edges"] = 1
    q["active_edges"][::5] = 0
    cg, cc = tw.collide_capsules(q)
    assert len(cg) == len(cc) and len(cg) > 300
    assert np.array_equal(cg["query"], cc["query"]) and np.array_equal(cg["body"], cc["body"])
    for f in ("point", "normal", "distance"):
        assert np.array_equal(cg[f].view(np.uint32), cc[f].view(np.uint32)), f
    # on the flat middle (vertices at whole metres: the triangles between 3 and 4 m already slope): without the flag some queries beside a seam see a
    # leaning normal -- a slope that is not there; walking INTO that slope with the flag set, the contact carries the floor's normal
    flat = (np.abs(q["pos"][cg["query"], 0]) < 2.5) & (np.abs(q["pos"][cg["query"], 1]) < 2.5)
    q2 = q.copy(); q2["active_edges"] = 0
    pg, pc = tw.collide_capsules(q2)
    flat2 = (np.abs(q2["pos"][pg["query"], 0]) < 2.5) & (np.abs(q2["pos"][pg["query"], 1]) < 2.5)
    leaning = flat2 & (pg["normal"][:, 2] < 0.99)
    assert flat.sum() > 30 and leaning.sum() >= 5
    q3 = q2[pg["query"][leaning]].copy()
    nxy = pg["normal"][leaning].copy(); nxy[:, 2] = 0; nxy /= np.linalg.norm(nxy, axis=1, keepdims=True)
    q3["movement"] = -nxy; q3["active_edges"] = 1
    fg, fc = tw.collide_capsules(q3)
    assert np.array_equal(fg["normal"].view(np.uint32), fc["normal"].view(np.uint32))
    # (other seams around the same capsule may still answer with their own normals when the walk does not run into them: what must be gone is every normal
    # that leans AGAINST the direction of travel -- the floor is flat)
    against = np.einsum("ij,ij->i", fg["normal"][:, :2], q3["movement"][fg["query"], :2])
    assert (against > -1e-4).all(), against.min()
    before = np.einsum("ij,ij->i", pg["normal"][leaning][:, :2], -nxy[:, :2])
    assert (before < -0.1).all()
    tw.close()


def test_single_rays_through_the_resident_server_equal_the_batched_answers():
    """PhysicsWorld::traceRay is called one ray at a time by unchanged callers (ParticleManager.cpp:164, HoverCarPhysics.cpp:348): such rays go to a
    resident wave through a host-mapped mailbox (round 5) that traces each with all 64 lanes.  The answers must be those of the batched kernel, bit for
    bit, on a pile with hulls over a terrain mesh -- also when the world is edited or stepped between two rays (the server is told to leave, the next
    ray starts another)."""
    from substrata_amd.lib import World
    rng = np.random.default_rng(21)
    w = World(max_bodies=2048)
    descs = scenes.small_mixed(8, 3, seed=5)
    w.add_batch(descs)
    # a bumpy terrain under and around the pile (rays that miss the bodies hit its triangles: triangle index, material, barycentrics travel too)
    gx = np.linspace(-12, 12, 25); gy = np.linspace(-12, 12, 25)
    vx, vy = np.meshgrid(gx, gy, indexing="ij")
    verts = np.stack([vx.ravel(), vy.ravel(), 0.25 + 0.15 * np.sin(vx.ravel()) * np.cos(0.7 * vy.ravel())], axis=1).astype(np.float32)
    tris = []
    for i in range(24):
        for j in range(24):
            a = i * 25 + j; tris += [(a, a + 25, a + 1), (a + 1, a + 25, a + 26)]
    mi = w.mesh_create(verts, np.array(tris, np.uint32), materials=np.arange(len(tris)) % 7)
    md = scenes.dynamic_bodies(1); md["motion_type"] = abi.MOTION_STATIC; md["layer"] = abi.LAYER_NON_MOVING
    md["shape_type"] = abi.SHAPE_MESH; md["shape"][0] = (float(mi.mesh_id), 0, 0, 0); md["pos"][0] = (0, 0, 0)
    w.add_batch(md)
    hi = w.hull_create(rng.normal(size=(16, 3)) * 0.5)
    hd = scenes.dynamic_bodies(24)
    hd["shape_type"] = abi.SHAPE_HULL; hd["shape"][:, 0] = float(hi.hull_id); hd["shape"][:, 1:] = 0
    hd["pos"] = rng.uniform([-4, -4, 1], [4, 4, 5], size=(24, 3))
    w.add_batch(hd)
    for _ in range(120):
        w.step(DT)
    n = 600
    rays = np.zeros(n, dtype=abi.ray_dtype)
    rays["origin"] = rng.uniform([-6, -6, 0.2], [6, 6, 6.0], size=(n, 3))
    d = rng.normal(size=(n, 3)) + (0, 0, -0.8); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["dir"] = d; rays["max_t"] = rng.uniform(0.05, 9.0, size=n); rays["ignore_id"] = abi.INVALID_ID; rays["ignore_id"][::9] = 7
    rays["collidable_only"] = rng.integers(0, 2, size=n)
    batched = w.raycast(rays)
    assert (batched["id"] != abi.INVALID_ID).sum() > 200 and (batched["triangle"] != abi.INVALID_ID).sum() > 20
    single = np.concatenate([w.raycast(rays[k:k + 1]) for k in range(n)])
    assert single.tobytes() == batched.tobytes()
    # edits and a step between single rays: each answer equals the batched answer in the state it was asked in
    for k in range(0, 120, 3):
        if k % 2 == 0:
            w.set_pos(1 + (k % 40), (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)), 2.5))
        else:
            w.step(DT)
        one = w.raycast(rays[k:k + 1]); two = w.raycast(rays[k + 1:k + 2])
        ref = w.raycast(rays[k:k + 2])
        assert one.tobytes() == ref[:1].tobytes() and two.tobytes() == ref[1:].tobytes(), k
    w.close()
