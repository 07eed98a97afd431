"""Where a single traceRay's 6 us go: the same loop of single-ray calls (the resident ray server, sgp_world_queries.hip) with rays that cost the
server nothing to trace (max_t ~ 0: no cell visited) against real ones.  The difference is the wave's tracing; the rest is the mailbox round trip +
the caller's overhead (ctypes here: measured beside it with a call that does nothing).
    PYTHONPATH=. python tools/experiments/ray_server_breakdown.py"""
import time, ctypes as C
import numpy as np
from substrata_amd import abi, scenes
from substrata_amd.lib import World

w = World(max_bodies=4096)
w.add_batch(scenes.ground())
d = scenes.small_mixed(8, 4, seed=3)[1:]
w.add_batch(d)
for _ in range(120): w.step(1 / 60)
rng = np.random.default_rng(1)
N = 4000
def loop(rays):
    hits = np.zeros(1, dtype=abi.hit_dtype)
    fn = w._fn("raycast"); h = w._h
    rp = [rays[i:i + 1].ctypes.data for i in range(len(rays))]; hp = hits.ctypes.data
    fn(h, rp[0], 1, hp)                      # starts the server
    t0 = time.perf_counter()
    for p in rp: fn(h, p, 1, hp)
    return (time.perf_counter() - t0) / len(rays) * 1e6
rays = np.zeros(N, dtype=abi.ray_dtype)
rays["origin"] = rng.uniform([-4, -4, 6], [4, 4, 8], size=(N, 3)); rays["dir"] = (0, 0, -1); rays["max_t"] = 20.0; rays["ignore_id"] = abi.INVALID_ID
print(f"rays down onto a pile (max_t 20): {loop(rays):.2f} us per call")
r2 = rays.copy(); r2["max_t"] = 1e-6
print(f"rays of length 1e-6 (nothing to trace): {loop(r2):.2f} us per call")
r3 = rays.copy(); r3["origin"][:, 2] = 500.0; r3["dir"] = (0, 0, 1); r3["max_t"] = 5.0
print(f"rays in empty space far above (max_t 5): {loop(r3):.2f} us per call")
fn = w._fn("world_num_bodies") if hasattr(w, "_fn") else None
try:
    t0 = time.perf_counter()
    for _ in range(N): w.num_bodies()
    print(f"a call that does nothing on the device (num_bodies through the same binding): {(time.perf_counter() - t0) / N * 1e6:.2f} us")
except Exception as e:
    print("no-op call:", e)
w.close()
