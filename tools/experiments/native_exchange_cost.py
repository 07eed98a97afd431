"""What one ghost exchange costs per step on the NATIVE path (sgp_tiles_*): config 3 (100k bodies) cut into two tiles through the middle of
the settled pile, both tiles in this process on one GPU (sgp_tiles_exchange_group: the same routing kernels, header read-back and import as
the RCCL path, device-to-device copies in place of ncclSend / ncclRecv).  Reports the per-tile cost of an exchange next to the step.

    python tools/experiments/native_exchange_cost.py            (on the GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                             # noqa: E402
from substrata_amd import scenes, tiles                   # noqa: E402
from substrata_amd.lib import World, init                 # noqa: E402

init()
descs = scenes.config3_100k_mixed()
boxes = np.array([[-1e9, -1e9, -1e9, 0.0, 1e9, 1e9], [0.0, -1e9, -1e9, 1e9, 1e9, 1e9]], np.float32)      # the cut x = 0: a 150 m long border
worlds, nts = [], []
for r in range(2):
    mine = np.concatenate([[True], (descs["pos"][1:, 0] >= boxes[r, 0]) & (descs["pos"][1:, 0] < boxes[r, 3])])
    w = World(max_bodies=int(mine.sum()) + 32768)
    w.add_batch(descs[mine])
    worlds.append(w)
nts = [tiles.NativeTiles(worlds[r], r, 2, boxes, 2.0) for r in range(2)]
for _ in range(240):
    tiles.NativeTiles.exchange_group(nts)
    for w in worlds:
        w.step(1 / 60)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    for w in worlds:
        w.step(1 / 60)
torch.cuda.synchronize()
t_step = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    tiles.NativeTiles.exchange_group(nts)
    for w in worlds:
        w.step(1 / 60)
torch.cuda.synchronize()
t_both = (time.perf_counter() - t0) / n
st = [t.stats() for t in nts]
print(f"two 50k tiles: steps {1e3 * t_step:.3f} ms, steps + exchange {1e3 * t_both:.3f} ms -> exchange {1e3 * (t_both - t_step) / 2:.3f} ms per tile")
for r, s in enumerate(st):
    print(f"  tile {r}: {s.exported} records sent, {s.ghosts} ghosts held, imports on the device {s.fast_imports} / through the host {s.slow_imports}")
# the exchange alone, back to back (no steps in between: the ghost set cannot change, so every import takes the device path)
t0 = time.perf_counter()
for _ in range(n):
    tiles.NativeTiles.exchange_group(nts)
torch.cuda.synchronize()
print(f"exchange alone, steady ghost set: {1e3 * (time.perf_counter() - t0) / n / 2:.3f} ms per tile")
