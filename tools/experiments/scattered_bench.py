"""A world of scattered objects (the usual Substrata scene, scaled up): 20k boxes lying apart on the ground and 3k stacks of three.  Every
component of its contact graph is tiny, so the probes take every colour to the component launch: one solve launch per pass."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import abi, scenes
from substrata_amd.lib import World, init
init()
n_single, n_stacks = 20000, 3000
d = scenes.dynamic_bodies(n_single + 3 * n_stacks, mass=20.0)
d["shape"][:, :3] = 0.4
i = np.arange(n_single)
d["pos"][:n_single, 0] = (i % 200) * 1.7
d["pos"][:n_single, 1] = (i // 200) * 1.7
d["pos"][:n_single, 2] = 0.41
j = np.arange(3 * n_stacks)
d["pos"][n_single:, 0] = ((j // 3) % 60) * 2.5 + 400.0
d["pos"][n_single:, 1] = ((j // 3) // 60) * 2.5
d["pos"][n_single:, 2] = 0.41 + (j % 3) * 0.805
d["allow_sleeping"] = 0
descs = np.concatenate([scenes.ground(), d])
for budget in sys.argv[1:] or ["0", "160"]:
    os.environ["SGP_HC_BUDGET"] = budget
    w = World(max_bodies=len(descs) + 64); w.add_batch(descs)
    for _ in range(400): w.step(1 / 60)
    t0 = time.perf_counter()
    for _ in range(200): w.step(1 / 60)
    dt = (time.perf_counter() - t0) / 200
    st = w.stats()
    p = w.step_profiled(1 / 60)
    launches = sum(p.kernel_launches)
    print(f"budget {budget}: {1 / dt:.0f} steps/s ({dt * 1e3:.3f} ms); active {st.num_active} manifolds {st.num_manifolds} colours {st.num_colours} by component {st.num_component_constraints}; launches per step {launches}")
    w.close()
