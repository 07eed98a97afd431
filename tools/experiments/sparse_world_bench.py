"""The small bodies' cell grid is dense over the bounds of all small bodies and coarsens until it fits its table.  What a world costs whose
bodies form piles far apart (clusters over a 2 km square) against the same piles side by side.
    PYTHONPATH=. python tools/experiments/sparse_world_bench.py"""
import time, numpy as np
from substrata_amd import abi, scenes
from substrata_amd.lib import World
DT = 1 / 60
def run(spread, n_clusters=40, per=500):
    rng = np.random.default_rng(2)
    w = World(max_bodies=n_clusters * per + 64)
    w.add_batch(scenes.ground())
    for c in range(n_clusters):
        d, _ = scenes.lattice(10, 10, per // 100, 1.1, 0.6, seed=c, jitter=0.05, random_rot=True, origin_centered=False)
        d["pos"][:, 0] += (c % 8) * spread; d["pos"][:, 1] += (c // 8) * spread
        w.add_batch(d)
    for _ in range(90): w.step(DT)
    t0 = time.perf_counter()
    for _ in range(60): w.step(DT)
    ms = (time.perf_counter() - t0) * 1e3 / 60
    prof = w.step_profiled(DT); names = w.kernel_class_names(); km = list(prof.kernel_ms); st = w.stats()
    print(f"clusters {spread:.0f} m apart: {ms:.3f} ms/step, pairs {st.num_pairs}, manifolds {st.num_manifolds}, active {st.num_active}; ", {names[i]: round(km[i], 3) for i in range(len(km)) if km[i] > 0.05}, flush=True)
    w.close()
for s in (15.0, 100.0, 400.0, 2000.0):
    run(s)
