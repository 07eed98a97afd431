"""What it costs to stream ONE static mesh object in or out of a world that holds thousands (a client loading parcels): host time of the call + the
flush that follows it (a step would flush it too), with the grid of the static large bodies standing.
    PYTHONPATH=. python tools/experiments/mesh_streaming_cost.py [n_side, default 64 -> 4096 buildings]"""
import sys, time
import numpy as np
from substrata_amd import abi, scenes
from substrata_amd.lib import World
from many_meshes_bench import box_mesh

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_mesh = n_side * n_side
w = World(max_bodies=3 * n_mesh + 4096)
w.add_batch(scenes.ground())
V, T = box_mesh(3.0, 3.0, 2.0)
info = w.mesh_create(V, T)
d = scenes._blank(n_mesh)
d["shape_type"] = abi.SHAPE_MESH; d["shape"][:] = 0; d["shape"][:, 0] = float(info.mesh_id)
gx, gy = np.meshgrid(np.arange(n_side), np.arange(n_side))
d["pos"] = np.column_stack([(gx.ravel() - n_side / 2) * 12.0, (gy.ravel() - n_side / 2) * 12.0, np.zeros(n_mesh)])
ids = w.add_batch(d)
b = scenes.dynamic_bodies(512); b["pos"] = np.column_stack([np.linspace(-50, 50, 512), np.zeros(512), np.full(512, 6.0)])
w.add_batch(b)
for _ in range(30): w.step(1 / 60)
ray = np.zeros(1, dtype=abi.ray_dtype); ray["origin"] = (0.5, 0.5, 30.0); ray["dir"] = (0, 0, -1); ray["max_t"] = 50.0; ray["ignore_id"] = abi.INVALID_ID
one = scenes._blank(1); one["shape_type"] = abi.SHAPE_MESH; one["shape"][:] = 0; one["shape"][:, 0] = float(info.mesh_id)
t_add, t_rem, new_ids = [], [], []
for k in range(200):
    one["pos"][0] = (float((k % 20) * 12.0 + 6.0), float((k // 20) * 12.0 + 6.0), 0.0)
    t0 = time.perf_counter(); i = w.add_batch(one); w.raycast(ray); t_add.append(time.perf_counter() - t0)       # (the ray forces the flush, and costs ~40 us itself)
    new_ids.append(int(i[0]))
    if k % 2:
        t0 = time.perf_counter(); w.remove(int(ids[3 + 7 * k])); w.raycast(ray); t_rem.append(time.perf_counter() - t0)
t0 = time.perf_counter(); w.raycast(ray); t_ray = time.perf_counter() - t0
t_add = np.array(t_add) * 1e6; t_rem = np.array(t_rem) * 1e6
print(f"{n_mesh} static mesh bodies: add one + flush: median {np.median(t_add):.0f} us, 90th percentile {np.percentile(t_add, 90):.0f} us, max {t_add.max():.0f} us (the full rebuilds, every 64 newcomers); "
      f"remove one + flush: median {np.median(t_rem):.0f} us, max {t_rem.max():.0f} us; the ray that forces the flush alone: {1e6 * t_ray:.0f} us")
