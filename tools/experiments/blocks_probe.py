"""Config 3 settled for 240 steps with the normal code path, then 30 more steps (the window the timing tools look at).  With SGP_DEBUG_FLAGS set the
physics of the whole run is garbage; only kernel durations are of interest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes
from substrata_amd.lib import World, init
init()
d = scenes.config3_100k_mixed()
w = World(max_bodies=len(d) + 32768); w.add_batch(d)
for _ in range(270): w.step(1 / 60)
st = w.stats()
print("constraints", st.num_manifolds, "pairs", st.num_pairs, "cached", st.num_cached_manifolds, "colours", st.num_colours)
