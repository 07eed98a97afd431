"""BASELINE config 2 (10k boxes): steps/s and per-kernel-class time once the pile has formed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes
from substrata_amd.lib import World, init
init()
descs = scenes.config2_10k_boxes()
descs["allow_sleeping"] = 0
w = World(max_bodies=len(descs) + 64); w.add_batch(descs)
for _ in range(300): w.step(1 / 60)
t = time.perf_counter(); n = 300
for _ in range(n): w.step(1 / 60)
el = time.perf_counter() - t
st = w.stats(); names = w.kernel_class_names()
p = w.step_profiled(1 / 60)
print(f"{n / el:.0f} steps/s ({1000 * el / n:.3f} ms/step); manifolds {st.num_manifolds} points {st.num_contact_points} colours {st.num_colours}; launches {sum(p.kernel_launches[k] for k in range(len(names)))}")
print({names[k]: (round(p.kernel_ms[k], 3), p.kernel_launches[k]) for k in range(len(names)) if p.kernel_launches[k]})
