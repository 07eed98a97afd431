"""How often does a body fall asleep in the bench pile?  (sizing an optimisation of k_cache_build's carry walk: it could skip the step's own constraints when nobody did)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401
from substrata_amd import scenes
from substrata_amd.lib import World
descs = scenes.config3_100k_mixed(100, 100, 10, seed=3)
w = World(max_bodies=len(descs) + 64)
w.add_batch(descs)
for s in range(400):
    w.step(1.0 / 60.0)
    if s >= 240 and s % 10 == 0:
        st = w.stats()
        print(s, "active", st.num_active, "activated", st.num_activated, "deactivated", st.num_deactivated, "manifolds", st.num_manifolds, flush=True)
