#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the ORACLE (oracle/sgo_oracle.c).  Run from the repository root:

    python tools/make_golden.py [scenario ...]

"full" fixtures hold every body's pose / velocity / active flag at steps 1, 10, 60, 240; "digest" fixtures (10k bodies) hold the
SHA-256 of the same arrays, the first 64 bodies verbatim and three aggregates.  See tests/golden_scenes.py for the scenarios and
tests/test_golden.py for the checks (oracle on the CPU, the HIP path on the GPU, both bit-exact).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_scenes as gs          # noqa: E402
from oracle import oracle           # noqa: E402


def main():
    oracle.build()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    only = set(sys.argv[1:])                         # (names given: those fixtures only, the others stay as committed)
    for name, (fn, kind) in gs.SCENARIOS.items():
        if only and name not in only:
            continue
        arrays = {}
        for step, st in fn(lambda **kw: oracle.OracleWorld(**kw)):
            tag = f"s{step}_"
            if kind == "full":
                for k, v in st.items():
                    arrays[tag + k] = v
            else:
                arrays[tag + "sha256"] = np.frombuffer(bytes.fromhex(gs.digest(st)), dtype=np.uint8)
                for k, v in st.items():
                    arrays[tag + k] = v[:gs.HEAD]
                arrays[tag + "n_active"] = np.int64(st["active"].sum())
                arrays[tag + "mean_z"] = np.float64(st["pos"][:, 2].astype(np.float64).mean())
                arrays[tag + "sum_v2"] = np.float64((st["lin_vel"].astype(np.float64) ** 2).sum())
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **arrays)
        print(f"{name}: {len(arrays)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
