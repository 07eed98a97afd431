import time, numpy as np
from substrata_amd import abi
from substrata_amd.lib import World
def grid_mesh(n, size):
    xs = np.linspace(-size, size, n).astype(np.float32); X, Y = np.meshgrid(xs, xs)
    V = np.column_stack([X.ravel(), Y.ravel(), np.sin(X.ravel())]).astype(np.float32)
    i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1)); a = (j * n + i).ravel(); b = a + 1; c = a + n; d = c + 1
    return V, np.concatenate([np.column_stack([a, b, d]), np.column_stack([a, d, c])]).astype(np.uint32)
w = World(max_bodies=64)
for n in (33, 129, 257, 513):
    V, T = grid_mesh(n, 50.0)
    t0 = time.perf_counter(); info = w.mesh_create(V, T); dt = time.perf_counter() - t0
    print(f"mesh_create {len(T)} triangles: {dt*1e3:.1f} ms", flush=True)
rng = np.random.default_rng(0)
for n in (12, 100, 2000):
    P = rng.normal(size=(n, 3))
    t0 = time.perf_counter(); h = w.hull_create(P); dt = time.perf_counter() - t0
    print(f"hull_create {n} points: {dt*1e3:.2f} ms", flush=True)
