"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel mean counter value per launch (last steps only).
Usage: python tools/pmc_summary.py <fetch_csv> <write_csv> [out.md] [out.json bodies]
(out.json: the two figures bench.py puts into roofline.traffic / roofline_solver.traffic -- bytes per body of the three body-sweep kernels,
bytes per launch of a velocity-iteration colour launch; `bodies` = body slots the sweep covers)
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
    if len(sys.argv) > 5:
        import json
        bodies = int(sys.argv[5])

        def per_launch(name):
            # (the solver kernels carry template arguments after the mode: "void k_solve_colour<1, 0>" -- match by the name up to the first argument)
            full = [k for k in fetch if k == name or (name.endswith(">") and k.startswith(name[:-1] + ","))]
            name = full[0] if full else name
            fv = fetch.get(name, [0.0]); wv = write.get(name, [0.0])
            fv = fv[len(fv) // 2:]; wv = wv[len(wv) // 2:]
            return sum(fv) / len(fv) * 1024 * 2 + sum(wv) / len(wv) * 1024
        sweep = sum(per_launch(k) for k in ("k_pre_solve", "k_integrate_pose", "k_finalize"))
        out = {"sweep_bytes_per_body": sweep / bodies, "solve_velocity_bytes_per_launch": per_launch("void k_solve_colour<1>"),
               # the one launch per pass that takes every colour from the plan's hc_first on (0 if the run never used it)
               "solve_components_bytes_per_launch": per_launch("void k_solve_hc<1>") if any(k.startswith("void k_solve_hc<1") for k in fetch) else 0.0,
               "bodies": bodies,
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, eager launches), tools/collect_pmc.sh + tools/pmc_summary.py: "
                         "FETCH_SIZE KiB x 1024 x 2 (gfx950 correction) + WRITE_SIZE KiB x 1024, mean over the second half of each kernel's launches"}
        json.dump(out, open(sys.argv[4], "w"), indent=1)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
