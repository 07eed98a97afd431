"""What adding / removing ONE static mesh body costs a world that already holds n of them (the static large bodies' grid is rebuilt).
    PYTHONPATH=. python tools/experiments/large_grid_rebuild_cost.py"""
import time, numpy as np
from substrata_amd import abi, scenes
from substrata_amd.lib import World
from many_meshes_bench import box_mesh
for n_side in (32, 128, 256):
    n = n_side * n_side
    w = World(max_bodies=3 * n + 4096)
    w.add_batch(scenes.ground())
    V, T = box_mesh(3.0, 3.0, 2.0); info = w.mesh_create(V, T)
    d = scenes._blank(n); d["shape_type"] = abi.SHAPE_MESH; d["shape"][:] = 0; d["shape"][:, 0] = float(info.mesh_id)
    gx, gy = np.meshgrid(np.arange(n_side), np.arange(n_side)); d["pos"] = np.column_stack([(gx.ravel() - n_side / 2) * 12.0, (gy.ravel() - n_side / 2) * 12.0, np.zeros(n)])
    w.add_batch(d)
    b = scenes.dynamic_bodies(64); b["pos"] = np.column_stack([np.arange(64) * 1.5, np.full(64, 6.0), np.full(64, 3.0)]); w.add_batch(b)
    for _ in range(10): w.step(1 / 60)
    t0 = time.perf_counter()
    for _ in range(20): w.step(1 / 60)
    base = (time.perf_counter() - t0) / 20
    one = d[:1].copy()
    ts = []
    for k in range(20):
        one["pos"][0] = (1000.0 + 10 * k, 0, 0)
        t0 = time.perf_counter(); ids = w.add_batch(one); w.step(1 / 60); ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); w.remove(int(ids[0])); w.step(1 / 60); tr = time.perf_counter() - t0
    print(f"{n} static meshes: step {base*1e3:.3f} ms; add one + step {np.median(ts)*1e3:.3f} ms; remove one + step {tr*1e3:.3f} ms", flush=True)
    w.close()
