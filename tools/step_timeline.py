#!/usr/bin/env python3
"""Where a step's wall time goes: kernel durations vs the gaps between consecutive kernels, from a rocprofv3 kernel trace (rocpd .db).
Steps are cut at k_step_begin; the last `--steps` complete steps are averaged per kernel name.
Usage: python tools/step_timeline.py <results.db> [--steps 40]"""
import argparse
import sqlite3
from collections import OrderedDict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--skip-last", type=int, default=12, help="ignore the last N steps of the trace (bench.py ends with eager, event-bracketed profiled steps)")
    a = ap.parse_args()
    db = sqlite3.connect(a.db)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select start, end, {name_col} from kernels order by start").fetchall()
    rows = [(s, e, n.split("(")[0]) for s, e, n in rows]
    begins = [i for i, r in enumerate(rows) if r[2].startswith("k_step_begin")]
    if len(begins) < a.steps + a.skip_last + 2:
        raise SystemExit("trace too short")
    last = len(begins) - 1 - a.skip_last
    sel = begins[last - a.steps:last + 1]
    dur = OrderedDict(); gap = OrderedDict(); cnt = OrderedDict()
    by_pass = {}      # velocity pass (1 .. 10) -> [sum of the colour launches' durations, launches]: is the first pass, which meets the constraint rows cold, slower?
    span = kern = gaps = 0.0
    n_launch = 0
    for k in range(a.steps):
        lo, hi = sel[k], sel[k + 1]
        step = rows[lo:hi]
        # the step ends with k_cache_build, whose first workgroup copies the counters out (k_step_end until round 3); what follows (host sync, next step's enqueue) is not part of it
        ends = [i for i, r in enumerate(step) if r[2].startswith("k_step_end")] or [i for i, r in enumerate(step) if r[2].startswith("k_cache_build")]
        end_i = max(ends)
        step = step[:end_i + 1]
        span += step[-1][1] - step[0][0]
        vpass = 1
        for i, (s, e, n) in enumerate(step):
            if n.startswith("void k_solve_colour<1"): bp = by_pass.setdefault(vpass, [0.0, 0]); bp[0] += e - s; bp[1] += 1
            if n.startswith("void k_solve_hc<1") or n.startswith("void k_solve_tail_vel"): vpass += 1
            dur[n] = dur.get(n, 0.0) + (e - s); cnt[n] = cnt.get(n, 0) + 1
            kern += e - s
            if i:
                g = max(0, s - step[i - 1][1])
                gap[n] = gap.get(n, 0.0) + g; gaps += g
            n_launch += 1
    S = a.steps
    print(f"averaged over {S} steps: {n_launch / S:.1f} launches per step; first launch to end of last kernel {span / S / 1e3:.1f} us = "
          f"kernels {kern / S / 1e3:.1f} us + gaps {gaps / S / 1e3:.1f} us (mean gap {gaps / max(n_launch - S, 1) / 1e3:.2f} us)\n")
    print("| kernel | launches/step | us/step in kernel | mean duration us | us/step in the gap before it | mean gap us |")
    print("|---|---|---|---|---|---|")
    for n in sorted(dur, key=lambda x: -(dur[x] + gap.get(x, 0.0))):
        c = cnt[n]
        print(f"| {n} | {c / S:.1f} | {dur[n] / S / 1e3:.1f} | {dur[n] / c / 1e3:.2f} | {gap.get(n, 0.0) / S / 1e3:.1f} | {gap.get(n, 0.0) / c / 1e3:.2f} |")
    if by_pass:
        print("\nvelocity colour launches by pass (mean duration us): " + ", ".join(f"{k}: {v[0] / max(v[1], 1) / 1e3:.2f}" for k, v in sorted(by_pass.items())))


if __name__ == "__main__":
    main()
