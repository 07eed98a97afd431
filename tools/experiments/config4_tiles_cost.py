"""BASELINE config 4 cut into 8 tiles (2 x 2 x 2) that all live in THIS process on one GPU (sgp_tiles_exchange_group: the routing kernels,
header read-back and import of the RCCL path, device-to-device copies in place of ncclSend / ncclRecv): what the exchange costs next to
the steps while the tower comes down -- bodies fall through the z faces by the thousand, so the imports go through the host.

    python tools/experiments/config4_tiles_cost.py [lattice edge, default 60] [steps, default 240]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                             # noqa: E402
from substrata_amd import scenes, tiles                   # noqa: E402
from substrata_amd.lib import World, init                 # noqa: E402

init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 240
T = 8
worlds, boxes = [], []
for r in range(T):
    d, lo, hi = scenes.config4_tile_descs(r, T, n=n)
    w = World(max_bodies=2 * n ** 3 // 4 + 65536)
    w.add_batch(d)
    worlds.append(w); boxes.append(np.concatenate([lo, hi]))
boxes = np.array(boxes, np.float32)
nts = [tiles.NativeTiles(worlds[r], r, T, boxes, 2.0) for r in range(T)]
print(f"config 4 with a {n}^3 lattice = {n ** 3} boxes in {T} tiles on one GPU")
print("| steps | ms per step (8 tiles stepped one after the other) | of which exchange (all 8 tiles) | emigrants per step | ghosts held | imports through the host / on the device |")
print("|---|---|---|---|---|---|")
for w0 in range(0, steps, 40):
    t_ex = t_all = 0.0; emig = 0
    s0 = [t.stats() for t in nts]
    for _ in range(40):
        torch.cuda.synchronize(); a = time.perf_counter()
        tiles.NativeTiles.exchange_group(nts)
        torch.cuda.synchronize(); b = time.perf_counter()
        for w in worlds:
            w.step(1 / 60)
        torch.cuda.synchronize(); c = time.perf_counter()
        t_ex += b - a; t_all += c - a
        emig += sum(t.stats().emigrated for t in nts)
    s1 = [t.stats() for t in nts]
    slow = sum(b.slow_imports - a.slow_imports for a, b in zip(s0, s1)); fast = sum(b.fast_imports - a.fast_imports for a, b in zip(s0, s1))
    print(f"| {w0 + 1}-{w0 + 40} | {1e3 * t_all / 40:.2f} | {1e3 * t_ex / 40:.2f} | {emig / 40:.0f} | {sum(s.ghosts for s in s1)} | {slow} / {fast} |", flush=True)
owned = [w.num_bodies() - 1 - t.stats().ghosts for w, t in zip(worlds, nts)]
print("owned bodies per tile at the end:", owned, "sum", sum(owned))
