"""What a step of BASELINE config 4 costs on ONE GPU as the tower comes down and settles: ms per step and the step statistics every 10 steps.
    python tools/experiments/config4_collapse_trace.py [lattice edge, default 100] [steps, default 400] [stop when a step takes more than ms, default 400]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes                          # noqa: E402
from substrata_amd.lib import World, init                 # noqa: E402
init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
stop_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 400.0
w = World(max_bodies=n ** 3 + 1024)
w.add_batch(scenes.config4_1m_boxes(n))
print("| step | ms per step | active | pairs | constraints | points | colours | overflow constraints | component constraints | catch-all constraints | cached manifolds | dropped |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
acc = 0.0
for s in range(1, steps + 1):
    t0 = time.perf_counter(); w.step(1 / 60); dt = time.perf_counter() - t0
    acc += dt
    if s % 10 == 0 or 1e3 * dt > stop_ms:
        st = w.stats()
        print(f"| {s} | {1e3 * acc / (10 if s % 10 == 0 else s % 10):.2f} | {st.num_active} | {st.num_pairs} | {st.num_manifolds} | {st.num_contact_points} | {st.num_colours} | {st.num_overflow_constraints} | "
              f"{st.num_component_constraints} | {st.num_catch_all_constraints} | {st.num_cached_manifolds} | {st.pairs_dropped + st.manifolds_dropped} |", flush=True)
        acc = 0.0
    if 1e3 * dt > stop_ms:
        print(f"stopped: step {s} took {1e3 * dt:.0f} ms"); break
