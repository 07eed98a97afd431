"""What one ghost exchange costs per step on the RCCL path with a single rank (collectives with self): the host-side part of
GhostExchange.exchange() -- export kernel, read-backs, numpy filtering, two all-gathers, import."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from substrata_amd import scenes, tiles
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
import ghost_exchange
from substrata_amd.lib import World, init
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
init()
descs = scenes.config3_100k_mixed()
w = World(max_bodies=len(descs) + 4096); w.add_batch(descs)
for _ in range(200): w.step(1 / 60)
# a tile boundary through the pile on two sides (like an interior tile of a 4x2 grid has)
lo = np.array([float(descs["pos"][1:, 0].min()) - 0.2, float(descs["pos"][1:, 1].min()) - 0.2, -1e9], np.float32); hi = np.array([1e9, 1e9, 1e9], np.float32)
ex = ghost_exchange.GhostExchange(w, 0, 1, lo, hi, margin=2.0, dist=dist, device=torch.device("cuda", 0))
for _ in range(10): ex.exchange(); w.step(1 / 60)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 100
for _ in range(n): w.step(1 / 60)
torch.cuda.synchronize(); t_step = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n): ex.exchange(); w.step(1 / 60)
torch.cuda.synchronize(); t_both = (time.perf_counter() - t0) / n
print(f"step {1e3 * t_step:.3f} ms, step + exchange {1e3 * t_both:.3f} ms -> exchange {1e3 * (t_both - t_step):.3f} ms with {ex.last_exported} records exported")
dist.destroy_process_group()


# ---- the pieces of one exchange as an interior tile of a 4 x 2 grid would see them: route to 3 neighbours, receive as much ----
import time as _t
def T(): torch.cuda.synchronize(); return _t.perf_counter()
recs = w.export_boundary(lo, hi, 2.0, cap=1 << 20)
boxes = np.array([[-1e9, -1e9, -1e9, lo[0], 1e9, 1e9], [lo[0], lo[1], -1e9, 1e9, 1e9, 1e9], [lo[0], -1e9, -1e9, 1e9, lo[1], 1e9]], np.float32)
acc = {}
def add(k, dt): acc[k] = acc.get(k, 0.0) + dt
for _ in range(50):
    t = T(); recs = w.export_boundary(lo, hi, 2.0, cap=1 << 20); add("export_boundary", T() - t)
    t = T(); send, counts, emig = tiles.route(recs, 1, boxes, 3.5); add("route (C)", T() - t)
    t = T(); g, im = tiles.split(send, lo, hi); add("split (C)", T() - t)
    t = T(); w.import_ghosts(g[:0]); add("import_ghosts(0)", T() - t)
print(f"  {len(recs)} exported -> {len(send)} routed")
for k, v in acc.items():
    print(f"  {k}: {1e6 * v / 50:.0f} us")
