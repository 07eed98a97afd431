"""The evidence tools themselves (pure Python, no GPU): tools/pmc_summary.py must take a kernel's DOMINANT template variant (round 5 took the
first in dict order -- k_solve_colour<1, 0, 1>, 480 launches of a layout only a world's first steps use -- and reported 205 MB per launch for a kernel that moves 17 MB)."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, counter, rows):
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
        for name, values in rows.items():
            for v in values:
                w.writerow([name, counter, v])


def test_pmc_summary_takes_the_variant_with_the_most_launches(tmp_path):
    fetch = {"void k_solve_colour<1, 0, 1>(DV)": [100000.0] * 8,           # listed first, rare, huge
             "void k_solve_colour<1, 2, 1>(DV)": [7000.0] * 400,           # what the step really runs
             "void k_solve_hc<1, 1>(DV)": [9000.0] * 6, "void k_solve_hc<1, 2>(DV)": [8000.0] * 60,
             "k_pre_solve(DV)": [5000.0] * 40, "k_integrate_pose(DV)": [3000.0] * 40, "k_finalize(DV)": [6000.0] * 40}
    write = {k: [v[0] / 4] * len(v) for k, v in fetch.items()}
    f, w = str(tmp_path / "f.csv"), str(tmp_path / "w.csv")
    _write(f, "FETCH_SIZE", fetch); _write(w, "WRITE_SIZE", write)
    out_md, out_json = str(tmp_path / "o.md"), str(tmp_path / "o.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f, w, out_md, out_json, "100001"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    j = json.load(open(out_json))
    assert j["solve_velocity_kernel"] == "void k_solve_colour<1, 2, 1>" and j["solve_components_kernel"] == "void k_solve_hc<1, 2>"
    assert abs(j["solve_velocity_bytes_per_launch"] - (7000.0 * 1024 * 2 + 1750.0 * 1024)) < 1.0          # FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024
    assert abs(j["sweep_bytes_per_body"] - ((5000 + 3000 + 6000) * 1024 * 2 + (1250 + 750 + 1500) * 1024) / 100001) < 1e-6


def test_committed_pmc_traffic_is_the_dominant_solver_kernel():
    """what bench.py will print as roofline_solver.traffic: the committed table names the kernel it is about, within 3 x of the launch's algorithmic bytes"""
    j = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert j["solve_velocity_kernel"] == "void k_solve_colour<1, 2, 1>"
    assert 5e6 < j["solve_velocity_bytes_per_launch"] < 3 * 9.2e6
