"""Mean duration of one kernel per window of steps over a whole rocprofv3 trace (rocpd .db): shows which part of a bench run a number came from.
Usage: python tools/experiments/kernel_windows.py <results.db> <kernel name prefix> [window]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = [(s, e, n.split("(")[0]) for s, e, n in cur.execute(f"select start, end, {name_col} from kernels order by start")]
pref = sys.argv[2]; win = int(sys.argv[3]) if len(sys.argv) > 3 else 25
step = -1; per = {}
for s, e, n in rows:
    if n.startswith("k_step_begin"): step += 1
    if n.startswith(pref) and step >= 0:
        per.setdefault(step // win, []).append((e - s) / 1e3)
    if n.startswith("k_step_begin"):
        per.setdefault(("span", step // win), []).append(s)
print("steps in trace:", step + 1)
for k in sorted(k for k in per if not isinstance(k, tuple)):
    v = per[k]; sp = per[("span", k)]
    print(f"steps {k * win:4d}-{k * win + win - 1:4d}: {len(v) / win:5.1f} launches/step, mean {sum(v) / len(v):8.2f} us, step period {(sp[-1] - sp[0]) / max(len(sp) - 1, 1) / 1e3:8.1f} us")
