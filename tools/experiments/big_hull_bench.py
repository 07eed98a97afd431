"""What a step costs when the bodies are convex hulls of many vertices (round 5: up to 256, JPH::ConvexHullShape::cMaxPointsInHull -- the hull of a
dynamic mesh object, PhysicsWorld.cpp:1062-1080): a pile of N hulls with V vertices each on the ground plane, for V = 12 (rounds 1-4), 64, 128, 256,
and the same pile on a triangulated terrain.
    PYTHONPATH=.:tools/experiments python tools/experiments/big_hull_bench.py [n_bodies [vertex counts, e.g. 64,256 [floors: 0 = plane, 1 = terrain, e.g. 0,1]]]"""
import sys, time
import numpy as np
from substrata_amd import abi, scenes
from substrata_amd.lib import World
from mesh_terrain_bench import grid_mesh, mesh_body

DT = 1.0 / 60.0


def pile(n, nv, terrain):
    rng = np.random.default_rng(11)
    w = World(max_bodies=n + 16)
    if terrain:
        V, T = grid_mesh(129, 40.0, lambda x, y: 0.5 * np.sin(0.3 * x) * np.cos(0.25 * y))
        w.add_batch(mesh_body(w.mesh_create(V, T).mesh_id))
    else:
        w.add_batch(scenes.ground())
    ids = []
    for k in range(8):                                                     # eight different clouds on ellipsoids: every point is a corner
        p = rng.normal(size=(nv, 3)); p /= np.linalg.norm(p, axis=1, keepdims=True)
        info = w.hull_create((p * rng.uniform(0.3, 0.5, size=3)).astype(np.float32))
        assert info.num_vertices == nv
        ids.append(info.hull_id)
    d = scenes.dynamic_bodies(n)
    d["shape_type"] = abi.SHAPE_HULL; d["shape"][:] = 0
    d["shape"][:, 0] = np.array(ids, np.float32)[np.arange(n) % 8]
    side = int(np.ceil((n / 4) ** 0.5))
    gx, gy, gz = np.meshgrid(np.arange(side), np.arange(side), np.arange(4))
    d["pos"] = (np.column_stack([gx.ravel(), gy.ravel(), gz.ravel()])[:n] * (1.1, 1.1, 1.1) + (-side * 0.55, -side * 0.55, 1.5)).astype(np.float32)
    q = rng.normal(size=(n, 4)); d["rot"] = q / np.linalg.norm(q, axis=1, keepdims=True)
    w.add_batch(d)
    for _ in range(90): w.step(DT)
    t0 = time.perf_counter()
    for _ in range(60): w.step(DT)
    ms = (time.perf_counter() - t0) * 1e3 / 60
    st = w.stats()
    prof = w.step_profiled(DT)
    names = w.kernel_class_names(); km = list(prof.kernel_ms)
    print(f"{n} hulls of {nv} vertices on {'a 32k-triangle terrain' if terrain else 'the ground plane'}: {ms:.3f} ms/step, pairs {st.num_pairs}, manifolds {st.num_manifolds}, active {st.num_active}, dropped {st.manifolds_dropped}", flush=True)
    print("   kernel classes over 0.03 ms:", {(names[i] if names else i): round(km[i], 3) for i in range(len(km)) if km[i] > 0.03}, flush=True)
    w.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    nvs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [12, 64, 128, 256]
    floors = [bool(int(x)) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [False, True]
    for terrain in floors:
        for nv in nvs:
            pile(n, nv, terrain)
