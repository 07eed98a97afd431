"""Summarise a rocprofv3 rocpd sqlite (.db) kernel trace into a per-kernel stats table (markdown/CSV-ish text).
Usage: python tools/rocpd_summary.py <results.db> [out.txt] [--json kernel_time.json --source "text"]
(--json: the average duration per kernel as bench.py reads it for roofline.frac_kernel_time, profiles/kernel_time.json)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    med = {}
    for n, dur in cur.execute(f"select {name_col}, end-start from kernels"):
        med.setdefault(n, []).append(dur)
    lines = ["| kernel | calls | total_ms | avg_us | median_us | min_us | max_us | % |", "|---|---|---|---|---|---|---|---|"]
    for n, c, s, a, mn, mx in rows:
        short = n.split("(")[0]
        m = sorted(med[n])[len(med[n]) // 2]
        lines.append(f"| {short} | {c} | {s / 1e6:.3f} | {a / 1e3:.2f} | {m / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * s / total:.1f} |")
    txt = "\n".join(lines)
    args = sys.argv[2:]
    if "--json" in args:
        import json
        jpath = args[args.index("--json") + 1]
        source = args[args.index("--source") + 1] if "--source" in args else "rocprofv3 --kernel-trace"
        avg = {}
        for n, c, s, a, mn, mx in rows:
            short = n.split("(")[0].replace("void ", "").strip()
            avg[short] = round(a / 1e3, 3)
        json.dump({"source": source, "avg_us": avg}, open(jpath, "w"), indent=1, sort_keys=True)
        args = [x for i, x in enumerate(args) if x not in ("--json", "--source") and (i == 0 or args[i - 1] not in ("--json", "--source"))]
    if args:
        open(args[0], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
