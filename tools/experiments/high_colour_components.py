"""Connected components of the sub-graph formed by the constraints of colour >= K (pinned config 3 state): if they are small, every
component can be solved by one wave in colour order, all of them in ONE launch per pass instead of one launch per colour + the tail."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes
from substrata_amd.lib import World, init
init()
d = scenes.config3_100k_mixed()
w = World(max_bodies=len(d) + 32768); w.add_batch(d)
for _ in range(284): w.step(1 / 60)
c = w.dump_constraints(cap=2_000_000)
S = w.read_states(0, len(d))
movable = (S["active"] != 0) & (d["motion_type"] == 2)
print("constraints", len(c))
for K in (6, 8, 9, 10, 11, 12):
    h = c[c["colour"] >= K]
    a, b = h["a"].astype(np.int64), h["b"].astype(np.int64)
    parent = np.arange(len(d))
    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]; x = parent[x]
        return x
    for x, y in zip(a, b):
        if movable[x] and movable[y]:
            rx, ry = find(x), find(y)
            if rx != ry: parent[max(rx, ry)] = min(rx, ry)
    root = np.array([find(x if movable[x] else y) for x, y in zip(a, b)])
    _, cnt = np.unique(root, return_counts=True)
    # longest chain = distinct colours in a component (phases)
    ph = {}
    for r, col in zip(root, h["colour"]): ph.setdefault(r, set()).add(int(col))
    mp = max(len(v) for v in ph.values())
    print(f"K={K}: {len(h)} constraints in {len(cnt)} components; size max {cnt.max()}, 99.9 pct {np.percentile(cnt, 99.9):.0f}, mean {cnt.mean():.1f}; > 32: {(cnt > 32).sum()}, > 384: {(cnt > 384).sum()}; max phases {mp}")
