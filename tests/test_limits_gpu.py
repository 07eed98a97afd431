"""Edge cases of the world's extent and capacities, HIP path against the oracle (and, where the outcome cannot be bit-exact, sanity)."""
import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT
import parity

pytestmark = pytest.mark.gpu


def two_piles(offset):
    a, _ = scenes.lattice(5, 5, 3, 1.2, 0.8, seed=21)
    b, _ = scenes.lattice(5, 5, 3, 1.2, 0.8, seed=22)
    b["pos"][:, 0] += offset[0]; b["pos"][:, 1] += offset[1]; b["pos"][:, 2] += offset[2]
    return np.concatenate([scenes.ground(width=2000.0), a, b])


def test_two_piles_far_apart_and_a_body_falling_away(oracle):
    """The broad-phase grid spans the bounds of all small bodies: two piles 900 m apart (both on the 2000 m ground quad) plus one body
    that has left the ground and keeps falling stretch it over hundreds of metres, so the cell size must coarsen without losing pairs."""
    descs = two_piles((900.0, -300.0, 0.0))
    runaway = scenes.dynamic_bodies(1)
    runaway["pos"][0] = (1500.0, 0.0, 5.0)          # beyond the ground quad: falls forever
    runaway["allow_sleeping"] = 0
    descs = np.concatenate([descs, runaway])
    tw = parity.make_twin(oracle, max_bodies=512)
    tw.add_batch(descs)
    for s in range(1, 301):
        tw.step(DT)
        if s % 50 == 0:
            sg, sc = tw.gpu.stats(), tw.cpu.stats()
            assert (sg.num_pairs, sg.num_manifolds, sg.num_contact_points) == (sc.num_pairs, sc.num_manifolds, sc.num_contact_points), s
            assert sg.pairs_dropped == 0 and sg.manifolds_dropped == 0
            d = parity.compare(tw, len(descs))
            assert d["bit_exact"] and d["active_mismatch"] == 0, (s, d)
    st = tw.gpu.read_states(0, len(descs))
    assert st["pos"][-1, 2] < -100.0                                  # the runaway is far below by now
    assert np.all(st["pos"][1:-1, 2] > 0.2)                           # both piles rest on the ground
    tw.close()


def test_pair_and_manifold_capacity_overflow_is_counted_not_fatal():
    """More contacts than max_manifolds: the excess is dropped and counted (SURVEY: the reference's cMaxContactConstraints behaves the same
    way -- Jolt drops contacts beyond its buffer), the step neither crashes nor corrupts the rest."""
    from substrata_amd.lib import World
    descs = scenes.config2_10k_boxes()
    w = World(max_bodies=len(descs) + 16, max_manifolds=4096, max_body_pairs=200000)
    w.add_batch(descs)
    dropped = 0
    for _ in range(150):
        w.step(DT)
        st = w.stats()
        dropped = max(dropped, st.manifolds_dropped)
        assert st.num_manifolds <= 4096
    assert dropped > 0
    s = w.read_states(0, len(descs))
    assert np.all(np.isfinite(s["pos"])) and np.all(np.isfinite(s["lin_vel"]))
    w.close()
    # the same with the pair buffer as the bottleneck
    w = World(max_bodies=len(descs) + 16, max_body_pairs=8192)
    w.add_batch(descs)
    dropped = 0
    for _ in range(100):
        w.step(DT)
        dropped = max(dropped, w.stats().pairs_dropped)
    assert dropped > 0
    s = w.read_states(0, len(descs))
    assert np.all(np.isfinite(s["pos"]))
    w.close()


def test_streaming_static_meshes_through_a_small_world(oracle):
    """Mesh bodies streamed in and out of a world with room for only a handful (Substrata's normal life): slot triples, mesh ids and mesh
    storage are reused -- the HIP world hands out the same ids as the oracle for hundreds of cycles, bodies keep colliding with whatever
    mesh currently sits in a reused slot, and device memory stops growing."""
    from test_mesh_parity_gpu import grid_mesh, mesh_body
    rng = np.random.default_rng(9)
    tw = parity.make_twin(oracle, max_bodies=64)
    tw.add_batch(scenes.ground())
    balls = scenes.dynamic_bodies(6); balls["shape_type"] = abi.SHAPE_SPHERE; balls["shape"][:, 0] = 0.3; balls["shape"][:, 1:] = 0
    balls["pos"] = [(x, y, 4.0) for x in (-2, 0, 2) for y in (-1, 1)]; balls["restitution"] = 0.6
    tw.add_batch(balls)
    live = []
    bytes_after_warmup = None
    for cycle in range(150):
        n = int(rng.integers(9, 21))
        V, T = grid_mesh(n, 4.0, lambda x, y, c=cycle: 1.0 + 0.3 * np.sin(0.7 * x + c) * np.cos(0.5 * y))
        ig, ic = tw.mesh_create(V, T, materials=np.arange(len(T), dtype=np.uint32) % 7)
        assert ig.mesh_id == ic.mesh_id
        bg, bc = tw.add_batch(mesh_body(ig, pos=(0.0, 0.0, 0.2 * (cycle % 3))))
        assert int(bg[0]) == int(bc[0])
        live.append((int(bg[0]), ig.mesh_id))
        for _ in range(4):
            tw.step(DT)
        if len(live) > 3:                     # stream the oldest out
            body, mesh = live.pop(0)
            tw.remove(body); tw.mesh_destroy(mesh)
        if cycle % 25 == 24:
            d = parity.compare(tw, 64)
            assert d["bit_exact"] and d["active_mismatch"] == 0, (cycle, d)
            if cycle == 49:
                bytes_after_warmup = tw.gpu.stats().device_bytes
    assert max(b for b, _ in live) < 7 + 4 * 3                         # slots were reused: ground + 6 balls + at most 4 live triples
    assert tw.gpu.stats().device_bytes == bytes_after_warmup           # the mesh pools reached their steady size
    st = tw.gpu.read_states(1, 6)
    assert np.all(st["pos"][:, 2] > 0.29) and np.sum(st["pos"][:, 2] > 0.9) >= 2      # balls still ride the current terrain (some rolled off its edge onto the ground)
    tw.close()
