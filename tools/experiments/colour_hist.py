import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes
from substrata_amd.lib import World, init
init()
descs = scenes.config3_100k_mixed()
w = World(max_bodies=len(descs) + 64); w.add_batch(descs)
for _ in range(420): w.step(1/60)
c = w.dump_constraints()
h = np.bincount(c["colour"].astype(int), minlength=64)
print([int(x) for x in h[:32]], int(h[63]))
