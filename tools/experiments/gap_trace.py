#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace, launch by launch: gap before the launch, duration, grid, workgroup, LDS, scratch (rocpd .db).
Usage: python tools/experiments/gap_trace.py <results.db> [step index from the end, default 80]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print("# columns:", cols)
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
extra = [c for c in ("grid_x", "workgroup_x", "lds_size", "scratch_size", "grid_size_x", "workgroup_size_x", "lds_block_size", "private_segment_size") if c in cols]
rows = cur.execute(f"select start, end, {name_col}, {', '.join(extra) if extra else '0'} from kernels order by start").fetchall()
begins = [i for i, r in enumerate(rows) if r[2].startswith("k_step_begin")]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 80
lo, hi = begins[-k - 1], begins[-k]
prev_end = None
for r in rows[lo:hi]:
    gap = (r[0] - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{gap:7.2f} {(r[1] - r[0]) / 1e3:8.2f}  {r[2].split('(')[0][:40]:40s} {r[3:]}")
    prev_end = r[1]
