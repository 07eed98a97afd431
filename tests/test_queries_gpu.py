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
