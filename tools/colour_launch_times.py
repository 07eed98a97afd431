#!/usr/bin/env python3
"""Per-colour launch durations of the velocity iterations from a rocprofv3 kernel trace (csv).
Usage: python tools/colour_launch_times.py <kernel_trace.csv>
Within one velocity pass the launches are k_solve_colour<1> for colours 0..tail_first-1 followed by one k_solve_tail; this groups the
trace of the last steps by position inside the pass and prints mean duration and mean gap to the previous kernel."""
import csv
import sys
from collections import defaultdict


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    rows = rows[len(rows) // 2:]
    dur = defaultdict(list); gap = defaultdict(list)
    pos = None
    for i in range(1, len(rows)):
        s, e, n = rows[i]
        if "k_solve_colour<1>" in n:
            pos = 0 if "k_solve_colour<1>" not in rows[i - 1][2] else pos + 1
            dur[pos].append(e - s); gap[pos].append(s - rows[i - 1][1])
        elif "k_solve_tail" in n and "k_solve_colour<1>" in rows[i - 1][2]:
            dur["tail"].append(e - s); gap["tail"].append(s - rows[i - 1][1])
    print("| position in pass | launches | mean duration us | mean gap before us |")
    print("|---|---|---|---|")
    for k in sorted(dur, key=lambda x: (isinstance(x, str), x)):
        print(f"| {k} | {len(dur[k])} | {sum(dur[k]) / len(dur[k]) / 1e3:.2f} | {sum(gap[k]) / len(gap[k]) / 1e3:.2f} |")


if __name__ == "__main__":
    main()
