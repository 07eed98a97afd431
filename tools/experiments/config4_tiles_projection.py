"""BASELINE config 4 (100^3 = 1M boxes, the collapsing tower) cut into T tiles that all live in THIS process on one GPU (sgp_tiles_exchange_group),
with the tile regions re-balanced every K steps (sgp_tiles_rebalance_group; K = 0: the static split).  Every tile's step is timed on its own (a world's
step is a blocking call), so the table projects what T GPUs -- one tile each, stepping at the same time -- would do:

    projected step = max over tiles (tile step) + exchange per tile (+ re-balancing, amortised)        projected steps/s = 1 / that

against the same scene on one GPU without tiles (bench.py --workload config4 on one GPU: `profiles/r03n_bench_config4_1gpu.log`, 58.7 steps/s in
the same window of the collapse).  What the projection cannot see: RCCL's own latency over xGMI instead of device-to-device copies (the record
volume is a few MB per step), and ranks waiting for each other inside the exchange (the max over tiles stands in for that).

    python tools/experiments/config4_tiles_projection.py [tiles, default 8] [re-balance every K steps, default 16] [steps, default 600] [lattice edge, default 100]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                             # noqa: E402
from substrata_amd import scenes, tiles                   # noqa: E402
from substrata_amd.lib import World, init                 # noqa: E402

init()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 600
n = int(sys.argv[4]) if len(sys.argv) > 4 else 100
WIN = 50
BY_CONTACTS = bool(int(os.environ.get("PROJ_BY_CONTACTS", "0")))
grid = tiles.tile_grid(T)
worlds, boxes = [], []
for r in range(T):
    d, lo, hi = scenes.config4_tile_descs(r, T, n=n)
    w = World(max_bodies=int(2.2 * n ** 3 / T) + 98304)
    w.add_batch(d)
    worlds.append(w); boxes.append(np.concatenate([lo, hi]))
boxes = np.array(boxes, np.float32)
nts = [tiles.NativeTiles(worlds[r], r, T, boxes, 2.0) for r in range(T)]
total = n ** 3
print(f"# config 4, {n}^3 = {total} boxes in {T} tiles ({grid[0]} x {grid[1]} x {grid[2]}) on one GPU, regions re-balanced every {K} steps by {'bodies + contacts' if BY_CONTACTS else 'body count'}" if K else
      f"# config 4, {n}^3 = {total} boxes in {T} tiles ({grid[0]} x {grid[1]} x {grid[2]}) on one GPU, static regions")
print()
print("(catch-all: constraints a tile's component launch had to solve serially because the plan, made from the previous step, put a colour too many into it -- summed over tiles and steps)")
print("| steps | max tile step ms | mean tile step ms | exchange ms per tile | re-balance ms per step | projected step ms | projected steps/s | owned bodies per tile: min / max (share of all) | emigrants per step (all tiles) | imports through the host / on the device | catch-all constraints |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
min_share_ever = 1.0
for w0 in range(0, steps, WIN):
    t_ex = t_rb = 0.0; emig = 0; catch_all = 0
    tile_ms = np.zeros(T); max_ms = 0.0
    s0 = [t.stats() for t in nts]
    lo_share, hi_share = 1.0, 0.0
    for s in range(w0, min(w0 + WIN, steps)):
        torch.cuda.synchronize(); a = time.perf_counter()
        if K and s % K == 0:
            tiles.NativeTiles.rebalance_group(nts, grid, by_contacts=BY_CONTACTS)
            torch.cuda.synchronize()
        b = time.perf_counter()
        tiles.NativeTiles.exchange_group(nts)
        torch.cuda.synchronize(); c = time.perf_counter()
        per = np.zeros(T)
        for r, w in enumerate(worlds):
            t1 = time.perf_counter(); w.step(1 / 60); per[r] = time.perf_counter() - t1
        if os.environ.get("PROJ_STOP_MS") and s > int(os.environ.get("PROJ_STOP_AFTER", "0")) and 1e3 * per.max() > float(os.environ["PROJ_STOP_MS"]):
            r = int(per.argmax()); st = worlds[r].stats()
            print(f"step {s}: tile {r} took {1e3 * per.max():.0f} ms: bodies {worlds[r].num_bodies()} active {st.num_active} pairs {st.num_pairs} constraints {st.num_manifolds} points {st.num_contact_points} colours {st.num_colours} "
                  f"overflow {st.num_overflow_constraints} component {st.num_component_constraints} catch-all {st.num_catch_all_constraints} dropped {st.pairs_dropped} + {st.manifolds_dropped} ghosts {nts[r].stats().ghosts}", flush=True)
            if os.environ.get("SGP_TIMING_ONE"):
                pr = worlds[r].step_profiled(1 / 60)
                print({worlds[r]._lib.sgp_kernel_class_name(k).decode(): round(pr.kernel_ms[k], 2) for k in range(32) if pr.kernel_launches[k]}, flush=True)
            sys.exit(0)
        t_rb += b - a; t_ex += c - b
        tile_ms += per; max_ms += per.max()
        emig += sum(t.stats().emigrated for t in nts)
        catch_all += sum(w.stats().num_catch_all_constraints for w in worlds)
        owned = np.array([w.num_bodies() - 1 - t.stats().ghosts for w, t in zip(worlds, nts)], dtype=np.float64)
        lo_share = min(lo_share, owned.min() / total); hi_share = max(hi_share, owned.max() / total)
    m = min(WIN, steps - w0)
    s1 = [t.stats() for t in nts]
    slow = sum(y.slow_imports - x.slow_imports for x, y in zip(s0, s1)); fast = sum(y.fast_imports - x.fast_imports for x, y in zip(s0, s1))
    proj = 1e3 * (max_ms / m + t_ex / m / T + t_rb / m / T)
    min_share_ever = min(min_share_ever, lo_share)
    print(f"| {w0 + 1}-{w0 + m} | {1e3 * max_ms / m:.2f} | {1e3 * tile_ms.sum() / m / T:.2f} | {1e3 * t_ex / m / T:.3f} | {1e3 * t_rb / m / T:.3f} | {proj:.2f} | {1e3 / proj:.1f} | "
          f"{100 * lo_share:.1f} % / {100 * hi_share:.1f} % | {emig / m:.0f} | {slow} / {fast} | {catch_all} |", flush=True)
owned = [w.num_bodies() - 1 - t.stats().ghosts for w, t in zip(worlds, nts)]
print()
print("owned bodies per tile at the end:", owned, "sum", sum(owned), f"; smallest share of a tile at any step: {100 * min_share_ever:.1f} %")
print("regions at the end (lo xyz, hi xyz):")
for r, b in enumerate(nts[0].boxes()):
    print(f"  tile {r}: " + " ".join(f"{v:9.2f}" if abs(v) < 1e8 else ("     -inf" if v < 0 else "     +inf") for v in b))
