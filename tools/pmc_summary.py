"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean counter value per launch (last steps only).
Usage: python tools/pmc_summary.py <fetch_csv> <write_csv> [out.md]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (x1024 = bytes).  Per MI355X_MICROARCH.md (HBM section), on gfx950
FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams, i.e. reads exactly 1/2 of a 16 B/lane stream: the
`read_x2` column applies that correction; WRITE_SIZE is uncalibrated and reported as is."""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    per = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            per[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return per


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    lines = ["| kernel | launches | FETCH_SIZE KiB/launch | read bytes/launch (x2 gfx950 correction) | WRITE_SIZE KiB/launch | write bytes/launch |",
             "|---|---|---|---|---|---|"]
    for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
        fv = fetch[k][len(fetch[k]) // 2:]      # second half of the run = fully loaded steps
        wv = write.get(k, [0.0])
        wv = wv[len(wv) // 2:]
        fm, wm = sum(fv) / len(fv), sum(wv) / max(len(wv), 1)
        lines.append(f"| {k} | {len(fetch[k])} | {fm:.1f} | {fm * 1024 * 2:.0f} | {wm:.1f} | {wm * 1024:.0f} |")
    txt = "\n".join(lines)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
