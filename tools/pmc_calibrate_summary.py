"""Turns the passes of tools/pmc_calibrate.sh into a table: per calibration kernel the counters of its LAST launch next to the bytes the lanes
requested and the bytes a 64- / 128-byte granule would move, and the factor FETCH_SIZE / WRITE_SIZE (KiB x 1024) must be multiplied by to give the
granule-128 bytes (what the memory side moves if every miss is a 128-byte line) -- per access pattern, not one factor for everything.
Usage: python tools/pmc_calibrate_summary.py <dir with pass*/ and plain.txt> <out.md> <out.json>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, out_md, out_json = sys.argv[1:4]
    plain = {}
    for line in open(os.path.join(d, "plain.txt")):
        p = line.split()
        if len(p) == 6 and p[0].startswith("k_cal_"):
            plain[p[0]] = dict(requested=float(p[1]), granule64=float(p[2]), granule128=float(p[3]), ms=float(p[4]))
    ctr = defaultdict(dict)
    for f in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(list)
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            per[(name, row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (name, c), v in per.items():
            ctr[name][c] = v[-1]
    cols = ["FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum", "TCC_BUBBLE_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum",
            "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_sum"]
    lines = ["| kernel | requested B | granule-64 B | granule-128 B | ms | requested GB/s | granule-128 GB/s | " + " | ".join(cols) + " | FETCH KiB x 1024 / requested | WRITE KiB x 1024 / requested |",
             "|" + "---|" * (9 + len(cols))]
    out = {}
    for k, p in plain.items():
        c = ctr.get(k, {})
        fb, wb = c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
        lines.append(f"| {k} | {p['requested']:.0f} | {p['granule64']:.0f} | {p['granule128']:.0f} | {p['ms']:.3f} | {p['requested'] / p['ms'] * 1e-6:.0f} | {p['granule128'] / p['ms'] * 1e-6:.0f} | "
                     + " | ".join(f"{c.get(x, float('nan')):.0f}" for x in cols) + f" | {fb / p['requested']:.3f} | {wb / p['requested']:.3f} |")
        out[k] = dict(p, fetch_bytes_reported=fb, write_bytes_reported=wb, counters=c)
    open(out_md, "w").write("\n".join(lines) + "\n")
    json.dump(out, open(out_json, "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
