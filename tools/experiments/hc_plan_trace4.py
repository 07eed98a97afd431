"""hc_plan_trace.py for BASELINE config 4 (1M boxes): per step time, constraints by component / through the catch-all."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes
from substrata_amd.lib import World, init
init()
d = scenes.config4_1m_boxes()
for budget in sys.argv[1:] or ["160"]:
    os.environ["SGP_HC_BUDGET"] = budget
    w = World(max_bodies=len(d) + 65536); w.add_batch(d)
    rows = []
    for s in range(200):
        t0 = time.perf_counter(); w.step(1 / 60); dt = time.perf_counter() - t0
        st = w.stats()
        rows.append((s, round(dt * 1e3, 1), st.num_manifolds, st.num_colours, st.num_component_constraints, st.num_catch_all_constraints))
    t = np.array([r[1] for r in rows])
    print(f"budget {budget}: 200 steps total {t.sum():.0f} ms; steps 140-199 mean {t[140:].mean():.2f} ms; catch-all steps: {[r for r in rows if r[5]]}")
    print("   every 20th:", rows[::20])
    w.close()
