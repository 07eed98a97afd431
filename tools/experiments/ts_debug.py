"""Tile solver vs colour launches, step by step on the device: first step whose results differ, how many bodies, which kind."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes
from substrata_amd.lib import World, init
init()
DT = 1 / 60
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 50
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 10
descs = scenes.config3_100k_mixed(nx, nx, nz, seed=5)
os.environ["SGP_TS_MIN_CONSTRAINTS"] = "0"
os.environ["SGP_TILE_SOLVER"] = "0"; a = World(max_bodies=len(descs) + 64); a.add_batch(descs)
os.environ["SGP_TILE_SOLVER"] = os.environ.get("TS_MODE", "1"); b = World(max_bodies=len(descs) + 64); b.add_batch(descs)
b2 = World(max_bodies=len(descs) + 64); b2.add_batch(descs)
for s in range(1, 200):
    a.step(DT); b.step(DT); b2.step(DT)
    sa, sb = a.read_states(0, len(descs)), b.read_states(0, len(descs))
    ts = b.stats().tile_solver
    bad = np.where((sa["lin_vel"] != sb["lin_vel"]).any(axis=1) | (sa["ang_vel"] != sb["ang_vel"]).any(axis=1) | (sa["pos"] != sb["pos"]).any(axis=1))[0]
    if s % 10 == 0 or len(bad):
        print(f"step {s}: tile_solver {ts} constraints {b.stats().num_manifolds} colours {b.stats().num_colours} differing bodies {len(bad)}", flush=True)
    if len(bad):
        cons = b.dump_constraints()
        deg = np.zeros(len(descs) + 64, int)
        for c in cons:
            deg[int(c["a"])] += 1; deg[int(c["b"])] += 1
        print("first differing ids", bad[:20], "their degrees", deg[bad[:20]])
        print("max |dv|", float(np.abs(sa["lin_vel"] - sb["lin_vel"]).max()), "pos of first", sb["pos"][bad[:5]])
        xs = sb["pos"][bad][:, :2]
        print("xy range of differing bodies", xs.min(axis=0), xs.max(axis=0), "world xy range", sb["pos"][1:, :2].min(axis=0), sb["pos"][1:, :2].max(axis=0))
        sb2 = b2.read_states(0, len(descs))
        print("tile solver run twice: identical", bool((sb["lin_vel"] == sb2["lin_vel"]).all() and (sb["pos"] == sb2["pos"]).all()))
        for i in bad[:3]:
            print("body", i, "type", int(descs["shape_type"][i]), "dv", sa["lin_vel"][i] - sb["lin_vel"][i], "dw", sa["ang_vel"][i] - sb["ang_vel"][i], "dpos", sa["pos"][i] - sb["pos"][i])
            for c in cons:
                if int(c["a"]) == i or int(c["b"]) == i:
                    o = int(c["b"]) if int(c["a"]) == i else int(c["a"])
                    print("   constraint a", int(c["a"]), "b", int(c["b"]), "colour", int(c["colour"]), "np", int(c["np"]), "other pos", sb["pos"][o], "other type", int(descs["shape_type"][o]), "other deg", deg[o])
        break
