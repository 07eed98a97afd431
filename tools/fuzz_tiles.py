#!/usr/bin/env python3
"""Randomised differential test of the tile path: N adjacent tiles in ONE process (1 x 2 or 2 x 2), ghosts handed over by direct calls
(export -> route -> split -> import, what GhostExchange does across ranks), HIP worlds against oracle worlds, bit for bit.  Random piles
of primitive bodies straddle the tile borders, some are thrown across them (ownership migrates), some are removed or teleported.

    python tools/fuzz_tiles.py --seeds 0-49 --steps 240        (on the GPU box)
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from substrata_amd import abi, scenes, tiles   # noqa: E402
from helpers import DT                         # noqa: E402
import parity                                  # noqa: E402

TILE_W = 10.0


def exchange(worlds, boxes, margin, log):
    n = len(worlds)
    sent = []
    for r, w in enumerate(worlds):
        recs = w.export_boundary(boxes[r, :3], boxes[r, 3:], margin)
        send, counts, emig = tiles.route(recs, r, boxes, margin + 1.5)
        for i in emig:
            w.remove(int(i))
        off = [0] + [int(x) for x in np.cumsum(counts)]
        sent.append([send[off[d]:off[d + 1]] for d in range(n)])
        log.append(("export", r, len(recs), [int(c) for c in counts], len(emig)))
    for r, w in enumerate(worlds):
        arrived = np.concatenate([sent[src][r] for src in range(n)]) if n > 1 else sent[0][0][:0]
        ghosts, immigrants = tiles.split(arrived, boxes[r, :3], boxes[r, 3:])
        w.import_ghosts(ghosts)
        if len(immigrants):
            w.add_batch(tiles.records_to_descs(immigrants))
        log.append(("import", r, len(ghosts), len(immigrants)))


def run_seed(oracle, seed, steps, verbose=False):
    from substrata_amd.lib import World
    rng = np.random.default_rng(seed)
    n_tiles = int(rng.choice([2, 4]))
    boxes = np.array([np.concatenate(tiles.tile_bounds(r, n_tiles, TILE_W, TILE_W)[:2]) for r in range(n_tiles)], np.float32)
    span = TILE_W * (2 if n_tiles >= 2 else 1)
    gpu = [World(max_bodies=2048) for _ in range(n_tiles)]
    cpu = [oracle.OracleWorld(max_bodies=2048) for _ in range(n_tiles)]
    total = 0
    for r in range(n_tiles):
        lo, hi, origin = tiles.tile_bounds(r, n_tiles, TILE_W, TILE_W)
        n = int(rng.integers(30, 120))
        d = scenes.dynamic_bodies(n)
        d["pos"][:, 0] = origin[0] + rng.uniform(0.3, TILE_W - 0.3, n)
        d["pos"][:, 1] = origin[1] + rng.uniform(0.3, TILE_W - 0.3, n)
        d["pos"][:, 2] = rng.uniform(0.6, 6.0, n)
        q = rng.normal(size=(n, 4)); d["rot"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
        kind = rng.integers(0, 3, n); sc = rng.uniform(0.3, 0.9, n)
        for i in range(n):
            if kind[i] == 0: d["shape_type"][i] = abi.SHAPE_BOX; d["shape"][i, :3] = sc[i] * rng.uniform(0.5, 1.0, 3)
            elif kind[i] == 1: d["shape_type"][i] = abi.SHAPE_SPHERE; d["shape"][i, :3] = (sc[i] * 0.6, 0, 0)
            else: d["shape_type"][i] = abi.SHAPE_CAPSULE; d["shape"][i, :3] = (sc[i] * 0.35, sc[i] * 0.6, 0)
        d["lin_vel"] = rng.uniform(-4, 4, (n, 3)).astype(np.float32)          # plenty of border crossings
        d["mass"] = (20 * sc ** 3 + 1).astype(np.float32)
        descs = np.concatenate([scenes.ground(), d])
        ig = gpu[r].add_batch(descs); ic = cpu[r].add_batch(descs)
        assert np.array_equal(ig, ic)
        total += n
    migrated = 0
    for s in range(1, steps + 1):
        lg, lc = [], []
        exchange(gpu, boxes, 1.5, lg)
        exchange(cpu, boxes, 1.5, lc)
        assert lg == lc, (seed, s, "exchange logs differ", [a for a, b in zip(lg, lc) if a != b][:3], [b for a, b in zip(lg, lc) if a != b][:3])
        migrated += sum(e[4] for e in lg if e[0] == "export")
        for r in range(n_tiles):
            gpu[r].step(DT); cpu[r].step(DT)
        if s % 30 == 0 or s == steps:
            for r in range(n_tiles):
                dd = parity.state_diff(gpu[r].read_states(0, 2048), cpu[r].read_states(0, 2048))
                assert dd["bit_exact"] and dd["active_mismatch"] == 0, (seed, s, r, dd)
            # nothing lost or duplicated: owned dynamic bodies over all tiles
            owned = sum(gpu[r].num_bodies() - 1 - [e for e in lg if e[0] == "import" and e[1] == r][0][2] for r in range(n_tiles))
            assert owned == total, (seed, s, "owned bodies", owned, total)
    if verbose:
        print(f"seed {seed}: {n_tiles} tiles, {total} bodies, {migrated} migrations: ok")
    for w in gpu + cpu:
        w.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0-19")
    ap.add_argument("--steps", type=int, default=240)
    args = ap.parse_args()
    lo, _, hi = args.seeds.partition("-")
    seeds = range(int(lo), int(hi or lo) + 1)
    import torch  # noqa: F401  (torch first, see tests/conftest.py)
    from oracle import oracle
    oracle.build()
    failed = []
    for seed in seeds:
        try:
            run_seed(oracle, seed, args.steps, verbose=True)
        except AssertionError as e:
            print(f"seed {seed}: MISMATCH {str(e)[:500]}")
            failed.append(seed)
    print(f"{len(seeds) - len(failed)} of {len(seeds)} seeds bit-exact; failed: {failed}")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
