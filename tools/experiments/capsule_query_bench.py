"""Latency of the character controller's shape query (sgp_collide_capsules: JPH::CharacterVirtual's CollideShape) for a player standing on a
terrain mesh / next to a finely tessellated wall.   PYTHONPATH=. python tools/experiments/capsule_query_bench.py"""
import time
import numpy as np
from substrata_amd import abi, scenes
from substrata_amd.lib import World

def grid_mesh(n, size, height_fn):
    xs = np.linspace(-size, size, n).astype(np.float32)
    X, Y = np.meshgrid(xs, xs)
    V = np.column_stack([X.ravel(), Y.ravel(), height_fn(X.ravel(), Y.ravel())]).astype(np.float32)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1))
    a = (j * n + i).ravel(); b = a + 1; c = a + n; d = c + 1
    return V, np.concatenate([np.column_stack([a, b, d]), np.column_stack([a, d, c])]).astype(np.uint32)

def mesh_body(mesh_id):
    d = scenes._blank(1)
    d["shape_type"] = abi.SHAPE_MESH; d["shape"][0] = 0; d["shape"][0, 0] = float(mesh_id)
    return d

for tri, label in ((1.25, "1.25 m triangles"), (0.25, "0.25 m triangles")):
    n = int(round(160.0 / tri)) + 1
    w = World(max_bodies=64)
    V, T = grid_mesh(n, 80.0, lambda x, y: 0.3 * np.sin(0.2 * x) * np.cos(0.17 * y))
    info = w.mesh_create(V, T)
    w.add_batch(mesh_body(info.mesh_id))
    w.step(1 / 60)
    for nq in (1, 64):
        q = np.zeros(nq, dtype=abi.capsule_query_dtype)
        rng = np.random.default_rng(1)
        px, py = rng.uniform(-40, 40, nq), rng.uniform(-40, 40, nq)
        q["pos"] = np.column_stack([px, py, 0.3 * np.sin(0.2 * px) * np.cos(0.17 * py) + 0.93])
        q["rot"] = (0, 0, 0, 1); q["radius"] = 0.3; q["half_height"] = 0.65; q["max_separation"] = 0.12; q["ignore_id"] = abi.INVALID_ID; q["collidable_only"] = 1
        for _ in range(5): c = w.collide_capsules(q)
        t0 = time.perf_counter()
        for _ in range(50): c = w.collide_capsules(q)
        ms = (time.perf_counter() - t0) * 1e3 / 50
        print(f"{label}, {len(T)} triangles: {nq} capsule quer{'y' if nq == 1 else 'ies'}: {ms:.3f} ms per call, {len(c)} contacts", flush=True)
    w.close()
