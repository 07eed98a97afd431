#!/usr/bin/env python3
"""Randomised differential test of the tile path: N adjacent tiles in ONE process (1 x 2 or 2 x 2), ghosts handed over by direct calls
(export -> route -> split -> import, what GhostExchange does across ranks), HIP worlds against oracle worlds, bit for bit.  Random piles
of primitive bodies straddle the tile borders, some are thrown across them (ownership migrates), some are removed, teleported or kicked in
mid-run.  Odd seeds exchange the HIP worlds through the native path (sgp_tiles_exchange_group: routing kernels, device-to-device copies,
device-side ghost refresh) while the oracle worlds go through the Python statement of the rules; every third seed stacks the tiles in z.

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


def exchange(worlds, boxes, margin, log, moved=None):
    """moved (optional): per tile, [ids that left, ids that arrived] -- how the caller keeps track of which bodies a tile owns."""
    n = len(worlds)
    sent = []
    for r, w in enumerate(worlds):
        recs = w.export_boundary(boxes[r, :3], boxes[r, 3:], margin)
        send, counts, emig = tiles.route(recs, r, boxes, margin + 1.5)
        for i in emig:
            w.remove(int(i))
        if moved is not None:
            moved[r][0] += [int(i) for i in emig]
        off = [0] + [int(x) for x in np.cumsum(counts)]
        sent.append([send[off[d]:off[d + 1]] for d in range(n)])
        log.append(("export", r, len(recs), [int(c) for c in counts], len(emig)))
    for r, w in enumerate(worlds):
        arrived = np.concatenate([sent[src][r] for src in range(n)]) if n > 1 else sent[0][0][:0]
        ghosts, immigrants = tiles.split(arrived, boxes[r, :3], boxes[r, 3:])
        w.import_ghosts(ghosts)
        if len(immigrants):
            new_ids = w.add_batch(tiles.records_to_descs(immigrants))
            if moved is not None:
                moved[r][1] += [int(i) for i in new_ids if i != abi.INVALID_ID]
        log.append(("import", r, len(ghosts), len(immigrants)))


CHECK_EVERY = int(os.environ.get("FUZZ_TILES_CHECK_EVERY", "30"))
TRACE_FROM = int(os.environ.get("FUZZ_TILES_TRACE_FROM", "0"))


def run_seed(oracle, seed, steps, verbose=False):
    from substrata_amd.lib import World
    rng = np.random.default_rng(seed)
    n_tiles = int(rng.choice([2, 4]))
    native = seed % 2 == 1
    grid = (1, 1, 2) if (seed % 3 == 2 and n_tiles == 2) else ((2, 1, 2) if (seed % 3 == 2) else None)      # tiles stacked in z (faces at z = 3)
    tb = lambda r: tiles.tile_bounds(r, n_tiles, TILE_W, TILE_W, 3.0 if grid else None, grid=grid)      # noqa: E731
    boxes = np.array([np.concatenate(tb(r)[:2]) for r in range(n_tiles)], np.float32)
    span = TILE_W * (2 if n_tiles >= 2 else 1)
    gpu = [World(max_bodies=2048) for _ in range(n_tiles)]
    cpu = [oracle.OracleWorld(max_bodies=2048) for _ in range(n_tiles)]
    total = 0
    own = []
    for r in range(n_tiles):
        lo, hi, origin = tb(r)
        n = int(rng.integers(30, 120))
        d = scenes.dynamic_bodies(n)
        d["pos"][:, 0] = origin[0] + rng.uniform(0.3, TILE_W - 0.3, n)
        d["pos"][:, 1] = origin[1] + rng.uniform(0.3, TILE_W - 0.3, n)
        d["pos"][:, 2] = rng.uniform(max(float(lo[2]), 0.0) + 0.6, min(float(hi[2]), 6.2) - 0.2, n)
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
        own.append(set(int(i) for i in ig[1:] if i != abi.INVALID_ID))
        total += n
    nt = [tiles.NativeTiles(gpu[r], r, n_tiles, boxes, 1.5) for r in range(n_tiles)] if native else None
    migrated = 0
    # round 4: re-tiling -- every few steps the native tiles move their split planes to the body-count quantiles (sgp_tiles_rebalance_group) and the
    # oracle tiles take over the regions: bodies then change owner by the dozen in the next exchange
    retile_every = int(rng.integers(5, 25)) if (native and rng.random() < 0.6 and not os.environ.get("FUZZ_NO_RETILE")) else 0
    retile_grid = grid if grid else tiles.tile_grid(n_tiles)
    retiles = 0
    for s in range(1, steps + 1):
        if retile_every and s % retile_every == 0:
            tiles.NativeTiles.rebalance_group(nt, retile_grid, by_contacts=bool(rng.integers(2)))
            boxes = nt[0].boxes()
            retiles += 1
        # edits between steps, applied to both worlds of a tile: removal, teleport inside the tile, a kick
        if rng.random() < 0.25:
            r = int(rng.integers(n_tiles))
            if len(own[r]) > 5:
                i = int(rng.choice(sorted(own[r])))
                what = rng.random()
                if TRACE_FROM and s >= TRACE_FROM: print(f"   [trace] step {s}: edit on tile {r} body {i}: {'remove' if what < 0.3 else ('teleport' if what < 0.6 else 'kick')}", flush=True)
                if what < 0.3:
                    gpu[r].remove(i); cpu[r].remove(i); own[r].discard(i); total -= 1
                elif what < 0.6:
                    lo_, hi_ = boxes[r, :3], boxes[r, 3:]
                    xl, xh = max(float(lo_[0]), 0.0) + 1.0, min(float(hi_[0]), span) - 1.0
                    yl, yh = max(float(lo_[1]), 0.0) + 1.0, min(float(hi_[1]), span) - 1.0
                    zl, zh = max(float(lo_[2]), 0.0) + 0.8, min(float(hi_[2]), 6.0) - 0.3
                    p_ = (float(rng.uniform(xl, max(xl + 0.1, xh))), float(rng.uniform(yl, max(yl + 0.1, yh))), float(rng.uniform(zl, max(zl + 0.1, zh))))
                    qq = rng.normal(size=4); qq /= np.linalg.norm(qq)
                    v_ = tuple(float(x) for x in rng.uniform(-3, 3, 3)); w_ = tuple(float(x) for x in rng.uniform(-2, 2, 3))
                    gpu[r].set_pose_vel(i, p_, tuple(qq), v_, w_); cpu[r].set_pose_vel(i, p_, tuple(qq), v_, w_)
                else:
                    v_ = tuple(float(x) for x in rng.uniform(-6, 6, 3)); w_ = tuple(float(x) for x in rng.uniform(-3, 3, 3))
                    gpu[r].set_vel(i, v_, w_); cpu[r].set_vel(i, v_, w_)
        lg, lc = [], []
        moved = [[[], []] for _ in range(n_tiles)]
        exchange(cpu, boxes, 1.5, lc, moved)
        if native:
            tiles.NativeTiles.exchange_group(nt)
            for r in range(n_tiles):
                st = nt[r].stats()
                exp = [e for e in lc if e[0] == "export" and e[1] == r][0]; imp = [e for e in lc if e[0] == "import" and e[1] == r][0]
                assert (st.exported, st.emigrated, st.ghosts, st.immigrated) == (sum(exp[3]), exp[4], imp[2], imp[3]), (seed, s, r, "native exchange counts",
                                                                                                                         (st.exported, st.emigrated, st.ghosts, st.immigrated), exp, imp)
                mg = nt[r].drain_migrations()
                assert sorted(int(m["old_id"]) for m in mg if m["direction"] == 0) == sorted(moved[r][0]), (seed, s, r, "emigrant ids")
                assert sorted(int(m["new_id"]) for m in mg if m["direction"] == 1) == sorted(moved[r][1]), (seed, s, r, "immigrant ids")
            lg = lc
        else:
            exchange(gpu, boxes, 1.5, lg)
        assert lg == lc, (seed, s, "exchange logs differ", [a for a, b in zip(lg, lc) if a != b][:3], [b for a, b in zip(lg, lc) if a != b][:3])
        for r in range(n_tiles):
            own[r] -= set(moved[r][0]); own[r] |= set(moved[r][1])
            if TRACE_FROM and s >= TRACE_FROM and (moved[r][0] or moved[r][1]): print(f"   [trace] step {s}: tile {r} emigrants {sorted(moved[r][0])} immigrants {sorted(moved[r][1])}", flush=True)
        if TRACE_FROM and s >= TRACE_FROM:
            for r in range(n_tiles):
                ag_, ac_ = gpu[r].read_states(0, 2048)["active"], cpu[r].read_states(0, 2048)["active"]
                print(f"   [trace] step {s} before stepping: tile {r} awake gpu {int(ag_.sum())} cpu {int(ac_.sum())} alive {gpu[r].num_bodies()}", flush=True)
        migrated += sum(e[4] for e in lg if e[0] == "export")
        for r in range(n_tiles):
            gpu[r].step(DT); cpu[r].step(DT)
        if TRACE_FROM and s >= TRACE_FROM and os.environ.get("FUZZ_TILES_TRACE_BODY"):
            tb_ = int(os.environ["FUZZ_TILES_TRACE_BODY"]); tr_ = int(os.environ.get("FUZZ_TILES_TRACE_TILE", "0"))
            for nm_, w_ in (("gpu", gpu[tr_]), ("cpu", cpu[tr_])):
                st_ = w_.stats(); cs_ = [c for c in w_.dump_constraints() if tb_ in (int(c["a"]), int(c["b"]))]
                print(f"   [trace] after step {s} {nm_}: body {tb_} active {int(w_.read_states(0, 2048)['active'][tb_])} constraints {[(int(c['a']), int(c['b']), int(c['colour']), int(c['np'])) for c in cs_]} cached {st_.num_cached_manifolds} wake pairs {st_.num_wake_pairs} activated {st_.num_activated} deactivated {st_.num_deactivated} rounds {st_.num_colour_rounds}", flush=True)
        if s % CHECK_EVERY == 0 or s == steps:
            for r in range(n_tiles):
                sg_, sc_ = gpu[r].read_states(0, 2048), cpu[r].read_states(0, 2048)
                dd = parity.state_diff(sg_, sc_)
                if not (dd["bit_exact"] and dd["active_mismatch"] == 0) and os.environ.get("FUZZ_TILES_DETAIL"):
                    # which bodies differ (debugging aid: FUZZ_TILES_CHECK_EVERY=1 FUZZ_TILES_DETAIL=1)
                    bad = [int(i) for i in range(2048) if any(not np.array_equal(sg_[f][i], sc_[f][i]) for f in ("pos", "rot", "lin_vel", "ang_vel"))]
                    print(f"seed {seed} step {s} tile {r}: bodies that differ {bad[:12]}; owned here {sorted(own[r])[:0]}", flush=True)
                    cg_, cc_ = gpu[r].dump_constraints(), cpu[r].dump_constraints()
                    kg = {(int(c["a"]), int(c["b"])): c for c in cg_}; kc = {(int(c["a"]), int(c["b"])): c for c in cc_}
                    print("   constraints gpu/cpu:", len(kg), len(kc), "only gpu", sorted(set(kg) - set(kc))[:6], "only cpu", sorted(set(kc) - set(kg))[:6], flush=True)
                    for k_ in sorted(set(kg) & set(kc)):
                        a_, b_ = kg[k_], kc[k_]
                        if any(not np.array_equal(a_[f], b_[f]) for f in a_.dtype.names):
                            print("   constraint", k_, "gpu", {f: a_[f].tolist() for f in a_.dtype.names if not np.array_equal(a_[f], b_[f])}, "cpu", {f: b_[f].tolist() for f in a_.dtype.names if not np.array_equal(a_[f], b_[f])}, flush=True)
                    sa, sb = gpu[r].stats(), cpu[r].stats()
                    print("   stats gpu/cpu:", {f[0]: (getattr(sa, f[0]), getattr(sb, f[0])) for f in sa._fields_ if isinstance(getattr(sa, f[0]), int) and getattr(sa, f[0]) != getattr(sb, f[0])}, flush=True)
                    for i in bad[:4]:
                        print("   body", i, "owned" if i in own[r] else "ghost/other", "gpu", sg_["pos"][i], sg_["lin_vel"][i], "cpu", sc_["pos"][i], sc_["lin_vel"][i], "active", sg_["active"][i], sc_["active"][i], flush=True)
                assert dd["bit_exact"] and dd["active_mismatch"] == 0, (seed, s, r, dd)
            # nothing lost or duplicated: owned dynamic bodies over all tiles
            owned = sum(gpu[r].num_bodies() - 1 - [e for e in lg if e[0] == "import" and e[1] == r][0][2] for r in range(n_tiles))
            assert owned == total, (seed, s, "owned bodies", owned, total)
    if verbose:
        print(f"seed {seed}: {n_tiles} tiles{' stacked in z' if grid else ''}, {'native' if native else 'python'} exchange, {total} bodies, {migrated} migrations, {retiles} re-tilings: ok")
    if nt:
        for t in nt:
            t.close()
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
