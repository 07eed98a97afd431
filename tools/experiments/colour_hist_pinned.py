"""Colour histogram of the bench's pinned state (config 3 after 240 settle steps, re-imported, 44 more steps): constraints per colour,
points per colour, and what the tail kernel has to chew through."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes
from substrata_amd.lib import World, init
init()
d = scenes.config3_100k_mixed()
w = World(max_bodies=len(d) + 32768); w.add_batch(d)
for _ in range(240): w.step(1 / 60)
S = w.read_states(0, len(d)); snap = d.copy()
for f in ("pos", "rot", "lin_vel", "ang_vel"): snap[f] = S[f]
snap["activate"] = (S["active"] != 0).astype(np.int32)
w.close()
w = World(max_bodies=len(d) + 32768); w.add_batch(snap)
for _ in range(44): w.step(1 / 60)
c = w.dump_constraints(cap=2_000_000)
col = c["colour"]; npts = c["np"]
h = np.bincount(col, minlength=64)
print("colours used:", int((h > 0).sum()), "constraints:", len(c))
tf = 0
while tf < 63 and h[tf] > 256: tf += 1
print("tail_first =", tf, " tail constraints =", int(h[tf:63].sum()), " overflow =", int(h[63]))
for k in range(64):
    if h[k]: print(f"  colour {k:2d}: {h[k]:6d} constraints, mean points {npts[col == k].mean():.2f}, 4-point share {np.mean(npts[col == k] == 4):.2f}")
deg = np.bincount(np.concatenate([c["a"], c["b"]]), minlength=len(d))
print("body degree: max", int(deg[1:].max()), " 99.9th pct", int(np.percentile(deg[1:], 99.9)), " mean", float(deg[1:].mean()))
