"""Builds substrata_amd/libsgp.so (HIP, gfx950 only) in-tree with hipcc.  Cross-compiles without a GPU.

-ffp-contract=off: fp32 expressions round exactly as written (no FMA contraction), the contract that lets the parity
tests compare the device path with the CPU oracle to rounding.  Correctly rounded fp32 divide/sqrt is hipcc's default.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsgp.so")
SOURCES = ["sgp_kernels.hip", "sgp_world.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("experiments", f) for f in sorted(os.listdir(os.path.join(CSRC, "experiments")))] + [os.path.join("..", "..", "include", "sgp.h")]      # every header: several are generated
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + list(extra) + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libsgp.so")
    if verbose and (r.stdout or r.stderr):
        print(r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    # --experiments: also compile csrc/experiments/* (the resident tile solver of round 3, the solver probe) into the library -- measured negatives and
    # timing aids that the product does not carry (SGP_TILE_SOLVER, tools/solve_probe.py, tests/test_tile_solver_gpu.py need such a build)
    build(force="--force" in sys.argv or "--experiments" in sys.argv, verbose=True, extra=("-DSGP_EXPERIMENTS",) if "--experiments" in sys.argv else ())
