"""A world shaped like a Substrata parcel grid: thousands of static mesh objects (every building is a MeshShape body) and some thousand dynamic
bodies among them.  What a step and a batch of rays cost as the number of static meshes grows.
    PYTHONPATH=. python tools/experiments/many_meshes_bench.py [n_side ...]"""
import sys, time
import numpy as np
from substrata_amd import abi, scenes
from substrata_amd.lib import World

DT = 1.0 / 60.0

def box_mesh(hx, hy, hz):
    V = np.array([(-hx, -hy, 0), (hx, -hy, 0), (hx, hy, 0), (-hx, hy, 0), (-hx, -hy, 2 * hz), (hx, -hy, 2 * hz), (hx, hy, 2 * hz), (-hx, hy, 2 * hz)], np.float32)
    T = np.array([(0, 2, 1), (0, 3, 2), (4, 5, 6), (4, 6, 7), (0, 1, 5), (0, 5, 4), (1, 2, 6), (1, 6, 5), (2, 3, 7), (2, 7, 6), (3, 0, 4), (3, 4, 7)], np.uint32)
    return V, T

def run(n_side, n_dyn=8192):
    rng = np.random.default_rng(3)
    n_mesh = n_side * n_side
    w = World(max_bodies=n_dyn + 3 * n_mesh + 64)
    w.add_batch(scenes.ground())
    V, T = box_mesh(3.0, 3.0, 2.0)
    info = w.mesh_create(V, T)
    d = scenes._blank(n_mesh)
    d["shape_type"] = abi.SHAPE_MESH; d["shape"][:] = 0; d["shape"][:, 0] = float(info.mesh_id)
    gx, gy = np.meshgrid(np.arange(n_side), np.arange(n_side))
    pitch = 12.0
    d["pos"] = np.column_stack([(gx.ravel() - n_side / 2) * pitch, (gy.ravel() - n_side / 2) * pitch, np.zeros(n_mesh)])
    t0 = time.perf_counter(); w.add_batch(d); t_add = time.perf_counter() - t0
    b = scenes.dynamic_bodies(n_dyn)
    b["shape_type"] = rng.integers(0, 3, n_dyn)
    b["shape"][:, :3] = 0.4; b["shape"][b["shape_type"] == 2, 0] = 0.25
    ext = n_side * pitch / 2 - 2
    b["pos"] = np.column_stack([rng.uniform(-ext, ext, n_dyn), rng.uniform(-ext, ext, n_dyn), rng.uniform(5.0, 9.0, n_dyn)])
    w.add_batch(b)
    for _ in range(60): w.step(DT)
    t0 = time.perf_counter()
    for _ in range(60): w.step(DT)
    ms = (time.perf_counter() - t0) * 1e3 / 60
    prof = w.step_profiled(DT); names = w.kernel_class_names(); km = list(prof.kernel_ms)
    rays = np.zeros(2048, dtype=abi.ray_dtype)
    rays["origin"] = np.column_stack([rng.uniform(-ext, ext, 2048), rng.uniform(-ext, ext, 2048), np.full(2048, 12.0)])
    dd = rng.normal(size=(2048, 3)) * (0.5, 0.5, 0.1) + (0, 0, -1.0); rays["dir"] = dd / np.linalg.norm(dd, axis=1, keepdims=True)
    rays["max_t"] = 60.0; rays["ignore_id"] = abi.INVALID_ID
    h = w.raycast(rays)
    t0 = time.perf_counter()
    for _ in range(10): h = w.raycast(rays)
    ray_ms = (time.perf_counter() - t0) * 1e3 / 10
    q = np.zeros(1, dtype=abi.capsule_query_dtype)
    q["pos"] = (1.0, 3.4, 0.95); q["rot"] = (0, 0, 0, 1); q["radius"] = 0.3; q["half_height"] = 0.65; q["max_separation"] = 0.12; q["ignore_id"] = abi.INVALID_ID; q["collidable_only"] = 1
    c = w.collide_capsules(q)
    t0 = time.perf_counter()
    for _ in range(20): c = w.collide_capsules(q)
    cap_ms = (time.perf_counter() - t0) * 1e3 / 20
    st = w.stats()
    print(f"{n_mesh} static meshes + {n_dyn} dynamic bodies: add {t_add:.2f} s, {ms:.3f} ms/step (pairs {st.num_pairs}, manifolds {st.num_manifolds}), "
          f"2048 rays {ray_ms:.3f} ms ({int((h['id'] != abi.INVALID_ID).sum())} hits, id checksum {int(h['id'].astype(np.uint64).sum())}), 1 capsule query {cap_ms:.3f} ms ({len(c)} contacts)", flush=True)
    print("   kernel classes over 0.05 ms:", {names[i]: round(km[i], 3) for i in range(len(km)) if km[i] > 0.05}, flush=True)
    w.close()

if __name__ == "__main__":
    for n in ([int(a) for a in sys.argv[1:]] or [8, 32, 64]):
        run(n)
