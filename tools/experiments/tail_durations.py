"""Durations of every k_solve_tail launch of one late step, in launch order (rocpd .db from rocprofv3 --kernel-trace)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = [(s, e, n.split("(")[0]) for s, e, n in cur.execute(f"select start, end, {name_col} from kernels order by start")]
begins = [i for i, r in enumerate(rows) if r[2].startswith("k_step_begin")]
lo, hi = begins[-20], begins[-19]
step = rows[lo:hi]
print("step with", len(step), "launches")
for i, (s, e, n) in enumerate(step):
    if "tail" in n or "k_solve_colour" in n and i % 15 == 0:
        print(f"  #{i:3d} {n:28s} {1e-3 * (e - s):8.2f} us   (previous: {step[i - 1][2]})")
