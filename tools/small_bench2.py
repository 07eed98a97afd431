"""Developer probe: where a small world's step time goes (config 1 while active, 256 boxes kept awake)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from substrata_amd import scenes
from substrata_amd.lib import World
layers = int(os.environ.get("SMALL_LAYERS", "4"))      # 8 x 8 x layers boxes (4 = BASELINE config 1)
descs = np.concatenate([scenes.ground(), scenes.lattice(8, 8, layers, 1.5, 1.0, seed=1)[0]])
descs["allow_sleeping"] = 0
for graphs in ("0", "1"):
    os.environ["SGP_NO_GRAPH"] = graphs
    w = World(max_bodies=len(descs) + 64)
    w.add_batch(descs)
    for _ in range(200):
        w.step(1 / 60)
    t = time.perf_counter(); n = 500
    for _ in range(n):
        w.step(1 / 60)
    el = time.perf_counter() - t
    st = w.stats()
    p = w.step_profiled(1 / 60)
    names = w.kernel_class_names()
    nl = sum(p.kernel_launches[k] for k in range(len(names)))
    print(f"SGP_NO_GRAPH={graphs}: {1000 * el / n:.3f} ms/step wall; active {st.num_active} manifolds {st.num_manifolds} colours {st.num_colours}; launches {nl}; GPU span of a profiled step {p.total_ms:.3f} ms")
    print("   ", {names[k]: (round(p.kernel_ms[k], 3), p.kernel_launches[k]) for k in range(len(names)) if p.kernel_launches[k]})
    w.close()
# the CPU port (oracle, NOT Jolt) on the same awake scene, one thread (more threads lose at this size): B2 beside the awake config 1
if "--no-cpu" not in sys.argv:
    from oracle import oracle
    oracle.set_threads(1)
    c = oracle.OracleWorld(max_bodies=len(descs) + 64)
    c.add_batch(descs)
    for _ in range(200):
        c.step(1 / 60)
    t = time.perf_counter(); n = 300
    for _ in range(n):
        c.step(1 / 60)
    el = time.perf_counter() - t
    cst = c.stats()
    print(f"cpu port (oracle, not Jolt), 1 thread: {1000 * el / n:.3f} ms/step; active {cst.num_active} manifolds {cst.num_manifolds} colours {cst.num_colours}")
    c.close()
