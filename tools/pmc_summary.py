"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean counter value per launch (last steps only).
Usage: python tools/pmc_summary.py <fetch_csv> <write_csv> [out.md] [out.json bodies]
(out.json: the two figures bench.py puts into roofline.traffic / roofline_solver.traffic -- bytes per body of the three body-sweep kernels,
bytes per launch of a velocity-iteration colour launch; `bodies` = body slots the sweep covers)
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (x1024 = bytes).  Per MI355X_MICROARCH.md (HBM section), on gfx950
FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams, i.e. reads exactly 1/2 of a 16 B/lane stream.  Round 6 calibrated the
other access patterns of the step on this part (tools/pmc_calibrate.sh, profiles/r06_pmc_calibration.md): TCC_EA0_RDREQ counts ONE request per
128-byte line touched for 16 / 8 / 4 B-per-lane streams and for scattered 16 / 32 / 128-byte gathers alike, TCC_BUBBLE (the counter FETCH_SIZE takes its
128-byte requests from) reads 0, and a scattered 16-byte gather takes the time of a full line -- so the x2 correction holds for EVERY read pattern and the
`read x2` column is the bytes the L2s requested from the memory side, in whole 128-byte lines (Infinity-Cache hits included: a kernel whose working set
partly sits in the 256 MiB cache can show more than the 6.3 TB/s HBM delivers).  WRITE_SIZE is exact for 64-byte and wider stores and counts 32 bytes
for a scattered 16-byte store (which also costs 2.2x the time of a 64-byte one): reported as is."""
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
    if len(sys.argv) > 5:
        import json
        bodies = int(sys.argv[5])

        def per_launch(name):
            # (the solver kernels carry template arguments after the mode: "void k_solve_colour<1, 0>" -- match by the name up to the first argument)
            full = [k for k in fetch if k == name or (name.endswith(">") and k.startswith(name[:-1] + ","))]
            # the template variant with the most launches = the one the step really runs (round 5 took the first in dict order: k_solve_colour<1, 0, 1>,
            # a layout only the first steps of a world use, 480 launches against 28k)
            name = max(full, key=lambda k: len(fetch[k])) if full else name
            fv = fetch.get(name, [0.0]); wv = write.get(name, [0.0])
            fv = fv[len(fv) // 2:]; wv = wv[len(wv) // 2:]
            return sum(fv) / len(fv) * 1024 * 2 + sum(wv) / len(wv) * 1024
        sweep = sum(per_launch(k) for k in ("k_pre_solve", "k_integrate_pose", "k_finalize"))
        def variant(name):
            full = [k for k in fetch if k == name or (name.endswith(">") and k.startswith(name[:-1] + ","))]
            return max(full, key=lambda k: len(fetch[k])) if full else None
        out = {"sweep_bytes_per_body": sweep / bodies, "solve_velocity_bytes_per_launch": per_launch("void k_solve_colour<1>"),
               "solve_velocity_kernel": variant("void k_solve_colour<1>"), "solve_components_kernel": variant("void k_solve_hc<1>"),
               "per_kernel_bytes_per_launch": {k: per_launch(k) for k in fetch if len(fetch[k]) >= 8},
               "calibration": "profiles/r06_pmc_calibration.md: one TCC_EA0_RDREQ per 128-byte line for streams and gathers alike (FETCH_SIZE x 2 = lines x 128 B); WRITE_SIZE exact from 64 B up",
               # the one launch per pass that takes every colour from the plan's hc_first on (0 if the run never used it)
               "solve_components_bytes_per_launch": per_launch("void k_solve_hc<1>") if any(k.startswith("void k_solve_hc<1") for k in fetch) else 0.0,
               "bodies": bodies,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, eager launches), tools/collect_pmc.sh + tools/pmc_summary.py: "
                         "FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) + WRITE_SIZE KiB x 1024, mean over the second half of each kernel's launches"}
        json.dump(out, open(sys.argv[4], "w"), indent=1)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
