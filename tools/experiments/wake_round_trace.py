#!/usr/bin/env python3
"""What the first steps of a leg of bench.py look like, step by step (config 3: settle 240 -> snapshot -> fresh world): wall time, manifolds, pairs of the
in-step activation round, activations, the component launch's share, eager / graph.  Run with and without SGP_NO_WAKE_ROUND=1."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from substrata_amd import scenes          # noqa: E402
from substrata_amd.lib import World       # noqa: E402

DT = 1.0 / 60.0
descs = scenes.config3_100k_mixed(100, 100, 10, seed=3)
w = World(max_bodies=len(descs) + 32768)
w.add_batch(descs)
for _ in range(240):
    w.step(DT)
S = w.read_states(0, len(descs))
snap = descs.copy()
for k in ("pos", "rot", "lin_vel", "ang_vel"):
    snap[k] = S[k]
snap["activate"] = (S["active"] != 0).astype(np.int32)
print("asleep in the snapshot:", int((S["active"] == 0).sum()) - 1)
w.close()
w = World(max_bodies=len(snap) + 32768)
w.add_batch(snap)
g0 = e0 = 0
for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 70):
    t0 = time.perf_counter()
    w.step(DT)
    ms = (time.perf_counter() - t0) * 1e3
    st = w.stats()
    g, e, _ = w.launch_counts()
    print(f"step {s:3d}: {ms:6.3f} ms  manifolds {st.num_manifolds}  colours {st.num_colours}  wake pairs {st.num_wake_pairs:5d}  activated {st.num_activated:4d}  deactivated {st.num_deactivated:4d}  active {st.num_active}"
          f"  components {st.num_component_constraints:6d} catch-all {st.num_catch_all_constraints:5d}  {'graph' if g > g0 else 'eager'}")
    g0, e0 = g, e
w.close()
