"""Cost of taking in ghosts every step: sgp_world_import_ghosts (host bookkeeping + queued body commands) and the command flush at the
start of the next step, for a tile that receives ~n ghost records per step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes, tiles, abi
from substrata_amd.lib import World, init
init()
descs = scenes.config3_100k_mixed()
src = World(max_bodies=len(descs) + 64); src.add_batch(descs)
for _ in range(200): src.step(1 / 60)
lo = np.array([float(descs["pos"][1:, 0].min()) - 0.2, float(descs["pos"][1:, 1].min()) - 0.2, -1e9], np.float32); hi = np.array([1e9, 1e9, 1e9], np.float32)
recs = src.export_boundary(lo, hi, 2.0)
print(len(recs), "records")
dst = World(max_bodies=len(descs) + 32768); dst.add_batch(descs)
for _ in range(200): dst.step(1 / 60)
def T(): return time.perf_counter()
n = 60
t = T()
for _ in range(n): dst.step(1 / 60)
base = (T() - t) / n
moved = recs.copy(); moved["pos"][:, 0] -= 400.0     # ghosts far away from the pile: no extra contacts, only the bookkeeping
dst.import_ghosts(moved); dst.step(1 / 60)
ti = ts = 0.0
for k in range(n):
    moved["pos"][:, 2] += 1e-4
    t = T(); dst.import_ghosts(moved); ti += T() - t
    t = T(); dst.step(1 / 60); ts += T() - t
print(f"step alone {1e3 * base:.3f} ms; with {len(recs)} ghosts refreshed every step: import {1e6 * ti / n:.0f} us + step {1e3 * ts / n:.3f} ms (flush and ghost bodies: +{1e6 * (ts / n - base):.0f} us)")
