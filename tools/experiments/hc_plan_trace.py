"""How the launch plan's choice of the first component colour behaves over a long run of config 3 (from the lattice: falling, piling up,
settling): steps in which a component went to the serial catch-all (and what such a step costs), per SGP_HC_BUDGET."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes
from substrata_amd.lib import World, init
init()
d = scenes.config3_100k_mixed()
for budget in sys.argv[1:] or ["160", "300"]:
    os.environ["SGP_HC_BUDGET"] = budget
    w = World(max_bodies=len(d) + 32768); w.add_batch(d)
    slow = []; times = []
    for s in range(600):
        t0 = time.perf_counter(); w.step(1 / 60); dt = time.perf_counter() - t0
        st = w.stats()
        times.append(dt)
        if st.num_catch_all_constraints: slow.append((s, st.num_catch_all_constraints, st.num_component_constraints, round(dt * 1e3, 2)))
    t = np.array(times) * 1e3
    print(f"budget {budget}: 600 steps, total {t.sum():.0f} ms, median {np.median(t):.2f} ms, steps with a catch-all: {len(slow)}, their time {sum(x[3] for x in slow):.0f} ms")
    print("   (step, catch-all constraints, component constraints, ms):", slow[:40])
    w.close()
