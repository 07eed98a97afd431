"""What a block decomposition of the solver would look like on the bench's pinned state (config 3): bodies are binned into world-anchored
cubes of edge L by AABB centre; a constraint is INTERIOR when all of its movable bodies sit in one cube, BOUNDARY otherwise.  Prints, per L:
blocks, bodies per block, interior / boundary split, the degree bounds that drive the number of colours in each set."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes, abi
from substrata_amd.lib import World, init
init()
d = scenes.config3_100k_mixed()
w = World(max_bodies=len(d) + 32768); w.add_batch(d)
for _ in range(240 + 44): w.step(1 / 60)
S = w.read_states(0, len(d))
c = w.dump_constraints(cap=2_000_000)
print("constraints", len(c), "fields", c.dtype.names)
pos = S["pos"][:, :3].astype(np.float64)
movable = (S["active"] != 0) & (d["motion_type"] == abi.MOTION_DYNAMIC) if "motion_type" in d.dtype.names else (S["active"] != 0)
print("bodies", len(d), "movable", int(movable.sum()), " extent", pos[movable].min(0), pos[movable].max(0))
a, b = c["a"].astype(np.int64), c["b"].astype(np.int64)
ma, mb = movable[a], movable[b]
for L in (3.0, 4.0, 5.0, 6.0, 8.0):
    bc = np.floor(pos / L).astype(np.int64)
    key = (bc[:, 0] + 4096) + ((bc[:, 1] + 4096) << 13) + ((bc[:, 2] + 4096) << 26)
    uk, inv = np.unique(key[movable], return_inverse=True)
    per = np.bincount(inv)
    ka, kb = key[a], key[b]
    interior = np.where(ma & mb, ka == kb, True)
    nb = int((~interior).sum())
    # degree of each body in each set
    def deg(sel):
        dd = np.zeros(len(d), np.int64)
        np.add.at(dd, a[sel & ma], 1); np.add.at(dd, b[sel & mb], 1)
        return dd
    di, db = deg(interior), deg(~interior)
    # interior constraints per block
    blk = np.where(ma, ka, kb)[interior]
    _, cnt = np.unique(blk, return_counts=True)
    print(f"L={L}: blocks {len(uk)}, bodies/block mean {per.mean():.0f} max {per.max()}, interior {int(interior.sum())} ({interior.mean():.3f}), boundary {nb}; "
          f"interior constraints/block mean {cnt.mean():.0f} max {cnt.max()}; max degree interior {di.max()} boundary {db.max()} (99.9 pct {np.percentile(db[db > 0], 99.9):.0f})")
