#!/usr/bin/env python3
"""How the body-array sweep (k_apply_forces + k_integrate_pose + k_finalize, the kernels `roofline` in bench.py is computed on)
scales with the body count: bodies in free fall (no ground, no contacts), HIP-event timings from sgp_world_step_profiled.

At BASELINE's 100k bodies each of the three launches moves ~6 MB and is bound by launch + memory latency, not by bandwidth; this
script shows where the same kernels land when a launch is large enough to fill the chip.  Usage (on the GPU box):
    python tools/sweep_scaling.py [--sizes 100000,400000,1600000,6400000]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from substrata_amd import scenes          # noqa: E402
from substrata_amd.lib import World, init  # noqa: E402

DT = 1.0 / 60.0
ALGO_BYTES = 188
HBM_PEAK = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="100000,400000,1600000,6400000")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    init()
    print("| bodies | sweep ms (3 launches) | algorithmic GB/s | frac of 8 TB/s | apply_forces us | integrate_pose us | finalize us |")
    print("|---|---|---|---|---|---|---|")
    for n in [int(x) for x in args.sizes.split(",")]:
        side = int(np.ceil(n ** (1.0 / 3.0)))
        d = scenes.dynamic_bodies(n)
        idx = np.arange(n)
        d["pos"][:, 0] = (idx % side) * 3.0
        d["pos"][:, 1] = ((idx // side) % side) * 3.0
        d["pos"][:, 2] = (idx // (side * side)) * 3.0 + 10.0
        d["shape_type"] = idx % 3
        d["shape"][:, :2] = (0.3, 0.65)
        d["shape"][idx % 3 == 0, :3] = 0.5
        d["ang_vel"] = np.float32([0.3, 0.2, 0.1])
        w = World(max_bodies=n + 64)
        w.add_batch(d)
        for _ in range(5):
            w.step(DT)
        names = w.kernel_class_names()
        k = {nm: i for i, nm in enumerate(names)}
        acc = np.zeros(len(names))
        for _ in range(args.steps):
            p = w.step_profiled(DT)
            acc += np.array([p.kernel_ms[i] for i in range(len(names))])
        acc /= args.steps
        sweep = acc[k["apply_forces"]] + acc[k["integrate_pose"]] + acc[k["finalize"]]
        gbs = ALGO_BYTES * n / (sweep * 1e-3) / 1e9
        print(f"| {n} | {sweep:.4f} | {gbs:.0f} | {gbs / HBM_PEAK:.3f} | {1e3 * acc[k['apply_forces']]:.1f} | {1e3 * acc[k['integrate_pose']]:.1f} | {1e3 * acc[k['finalize']]:.1f} |")
        w.close()


if __name__ == "__main__":
    main()
