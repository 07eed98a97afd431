"""Per-kernel means of the counters tools/collect_sq.sh collected (second half of each kernel's launches), one row per kernel.
Usage: python tools/sq_summary.py <dir with pass*/> <out.md>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1:3]
    vals = defaultdict(lambda: defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            vals[row["Kernel_Name"].split("(")[0].replace("void ", "").strip()][row["Counter_Name"]].append(float(row["Counter_Value"]))
    cols = sorted({c for k in vals for c in vals[k]})
    mean = {k: {c: (sum(v[len(v) // 2:]) / max(1, len(v) - len(v) // 2)) for c, v in vals[k].items()} for k in vals}
    calls = {k: max(len(v) for v in vals[k].values()) for k in vals}
    tot = sum(mean[k].get("SQ_BUSY_CYCLES", 0.0) * calls[k] for k in vals) or 1.0
    lines = ["| kernel | launches | " + " | ".join(cols) + " | wait frac | waves |", "|" + "---|" * (len(cols) + 4)]
    for k in sorted(vals, key=lambda k: -mean[k].get("SQ_BUSY_CYCLES", 0.0) * calls[k]):
        if mean[k].get("SQ_BUSY_CYCLES", 0.0) * calls[k] < 0.005 * tot:
            continue
        m = mean[k]
        wf = m.get("SQ_WAIT_INST_ANY", 0.0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else float("nan")
        lines.append(f"| {k} | {calls[k]} | " + " | ".join(f"{m.get(c, float('nan')):.4g}" for c in cols) + f" | {wf:.2f} | {m.get('SQ_WAVES', float('nan')):.0f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
