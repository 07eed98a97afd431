"""Bodies on triangle meshes: what a step costs when the pairs are (body, static mesh) pairs -- Substrata's usual case (every building and
the terrain are MeshShapes; PhysicsWorld.cpp:735-1166).  Two scenes: many small bodies on a terrain mesh (a few candidate triangles per
pair) and a few hundred large boxes on a finely tessellated floor (100+ candidates per pair).
    python tools/experiments/mesh_terrain_bench.py [n_small] [n_large]"""
import sys, time
import numpy as np
from substrata_amd import abi, scenes
from substrata_amd.lib import World

DT = 1.0 / 60.0


def grid_mesh(n, size, height_fn):
    xs = np.linspace(-size, size, n).astype(np.float32)
    X, Y = np.meshgrid(xs, xs)
    V = np.column_stack([X.ravel(), Y.ravel(), height_fn(X.ravel(), Y.ravel())]).astype(np.float32)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1))
    a = (j * n + i).ravel(); b = a + 1; c = a + n; d = c + 1
    T = np.concatenate([np.column_stack([a, b, d]), np.column_stack([a, d, c])]).astype(np.uint32)
    return V, T


def mesh_body(mesh_id):
    d = scenes._blank(1)
    d["shape_type"] = abi.SHAPE_MESH; d["shape"][0] = 0; d["shape"][0, 0] = float(mesh_id)
    return d


def run(name, w, steps, warm):
    for _ in range(warm): w.step(DT)
    t0 = time.perf_counter()
    for _ in range(steps): w.step(DT)
    ms = (time.perf_counter() - t0) * 1e3 / steps
    st = w.stats()
    prof = w.step_profiled(DT)
    print(f"{name}: {ms:.3f} ms/step, manifolds {st.num_manifolds}, active {st.num_active}, dropped {st.manifolds_dropped}", flush=True)
    names = w.kernel_class_names()
    km = list(prof.kernel_ms)
    print("   kernel classes over 0.02 ms:", {(names[i] if names else i): round(km[i], 3) for i in range(len(km)) if km[i] > 0.02}, flush=True)


def small_bodies_on_terrain(n):
    rng = np.random.default_rng(5)
    w = World(max_bodies=n + 16)
    V, T = grid_mesh(257, 160.0, lambda x, y: 1.5 * np.sin(0.08 * x) * np.cos(0.07 * y))
    info = w.mesh_create(V, T)
    w.add_batch(mesh_body(info.mesh_id))
    d = scenes.dynamic_bodies(n)
    kinds = rng.integers(0, 3, size=n)
    import os
    if os.environ.get("MESH_BENCH_KIND"): kinds[:] = int(os.environ["MESH_BENCH_KIND"])      # all spheres (0) / boxes (1) / capsules (2) / convex hulls (3): what each kind costs
    d["shape_type"] = kinds
    d["shape"][:, :3] = 0.4
    d["shape"][kinds == 2, 1] = 0.5; d["shape"][kinds == 2, 0] = 0.25
    if (kinds == 3).any():                                    # convex hulls: eight different 14-point clouds of the boxes' size
        hull_ids = []
        for k in range(8):
            pts = rng.normal(size=(14, 3)); pts = (pts / np.linalg.norm(pts, axis=1, keepdims=True) * rng.uniform(0.3, 0.45, (14, 1))).astype(np.float32)
            hull_ids.append(w.hull_create(pts).hull_id)
        sel = np.flatnonzero(kinds == 3)
        d["shape"][sel] = 0
        d["shape"][sel, 0] = np.array(hull_ids, np.float32)[np.arange(len(sel)) % 8]
    side = int(np.ceil(np.sqrt(n)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    xy = (np.column_stack([gx.ravel(), gy.ravel()])[:n] - side / 2) * (300.0 / side)
    d["pos"] = np.column_stack([xy[:, 0], xy[:, 1], rng.uniform(2.5, 3.5, n)])
    q = rng.normal(size=(n, 4)); d["rot"] = q / np.linalg.norm(q, axis=1, keepdims=True)
    w.add_batch(d)
    run(f"{n} small bodies on a 131k-triangle terrain", w, 120, 60)


def large_boxes_on_fine_floor(n):
    rng = np.random.default_rng(6)
    w = World(max_bodies=n + 16)
    V, T = grid_mesh(513, 128.0, lambda x, y: 0.05 * np.sin(0.9 * x) * np.cos(0.8 * y))      # 0.5 m triangles
    info = w.mesh_create(V, T)
    w.add_batch(mesh_body(info.mesh_id))
    d = scenes.dynamic_bodies(n, mass=800.0)
    d["shape_type"] = abi.SHAPE_BOX
    d["shape"][:, 0] = 2.2; d["shape"][:, 1] = 1.0; d["shape"][:, 2] = 0.5            # a car-sized slab: ~ 4.4 x 2 m footprint = ~ 70 triangles + margin
    side = int(np.ceil(np.sqrt(n)))
    gx, gy = np.meshgrid(np.arange(side), np.arange(side))
    xy = (np.column_stack([gx.ravel(), gy.ravel()])[:n] - side / 2) * (230.0 / side)
    d["pos"] = np.column_stack([xy[:, 0], xy[:, 1], np.full(n, 0.9)])
    a = rng.uniform(0, np.pi, n); d["rot"] = np.column_stack([np.zeros(n), np.zeros(n), np.sin(a / 2), np.cos(a / 2)])
    w.add_batch(d)
    run(f"{n} car-sized boxes on a 524k-triangle floor", w, 120, 60)


if __name__ == "__main__":
    small_bodies_on_terrain(int(sys.argv[1]) if len(sys.argv) > 1 else 20000)
    large_boxes_on_fine_floor(int(sys.argv[2]) if len(sys.argv) > 2 else 400)
