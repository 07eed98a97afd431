"""Builds substrata_amd/libsgp.so (HIP, gfx950 only) in-tree with hipcc.  Cross-compiles without a GPU.

One object per stage file of csrc/ (sgp_k_*.hip: the kernels; sgp_world*.hip: the host side of the C ABI), compiled side by side and
linked; an object is rebuilt only when its source, a header or this script is newer (objects live in csrc/.obj/, git-ignored).

-ffp-contract=off: fp32 expressions round exactly as written (no FMA contraction), the contract that lets the parity
tests compare the device path with the CPU oracle to rounding.  Correctly rounded fp32 divide/sqrt is hipcc's default.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, ".obj")
LIB = os.path.join(HERE, "libsgp.so")
KERNEL_SOURCES = sorted(f for f in os.listdir(CSRC) if f.startswith("sgp_k_") and f.endswith(".hip"))
HOST_SOURCES = sorted(f for f in os.listdir(CSRC) if f.startswith("sgp_world") and f.endswith(".hip"))
UNITY = "sgp_kernels_experiments.hip"      # every stage file + csrc/experiments/* as one translation unit (--experiments)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "sgp.h")]      # every header: several are generated
EXPERIMENT_FILES = [os.path.join("experiments", f) for f in sorted(os.listdir(os.path.join(CSRC, "experiments")))]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
         "-fvisibility=hidden", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-Wno-unused-value"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(deps, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _common_deps(experiments):
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if experiments:
        deps += [os.path.join(CSRC, f) for f in EXPERIMENT_FILES + KERNEL_SOURCES]
    return deps


def _sources(experiments):
    return ([UNITY] if experiments else KERNEL_SOURCES) + HOST_SOURCES


def _obj(src, experiments):
    return os.path.join(OBJ, src[:-4] + (".x.o" if experiments else ".o"))


def _tag_path():
    return os.path.join(HERE, "libsgp.tag")      # what the library was linked from: "product" or "experiments" (travels with it to the GPU box; the objects do not)


def needs_build(experiments=False):
    tag = "experiments" if experiments else "product"
    if not os.path.exists(LIB) or not os.path.exists(_tag_path()) or open(_tag_path()).read().strip() != tag:
        return True
    return _newer(_common_deps(experiments) + [os.path.join(CSRC, s) for s in _sources(experiments)], LIB)


def build(force=False, verbose=False, extra=()):
    experiments = "-DSGP_EXPERIMENTS" in extra
    if not force and not needs_build(experiments):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    common = _common_deps(experiments)
    todo = [s for s in _sources(experiments) if force or _newer(common + [os.path.join(CSRC, s)], _obj(s, experiments))]

    def compile_one(src):
        cmd = [hipcc()] + FLAGS + list(extra) + ["-c", os.path.join(CSRC, src), "-o", _obj(src, experiments)]
        if verbose:
            print(" ".join(cmd), flush=True)
        return src, subprocess.run(cmd, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4) or 1) as pool:
        results = list(pool.map(compile_one, todo))
    failed = [(s, r) for s, r in results if r.returncode != 0]
    for s, r in results:
        if r.returncode != 0 or (verbose and (r.stdout or r.stderr)):
            sys.stderr.write(f"---- {s}\n{r.stdout}{r.stderr}")
    if failed:
        raise RuntimeError("hipcc failed building " + ", ".join(s for s, _ in failed))
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + [_obj(s, experiments) for s in _sources(experiments)] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libsgp.so")
    with open(_tag_path(), "w") as f:
        f.write("experiments" if experiments else "product")
    return LIB


if __name__ == "__main__":
    # --experiments: also compile csrc/experiments/* (the resident tile solver of round 3, the solver probe) into the library -- measured negatives and
    # timing aids that the product does not carry (SGP_TILE_SOLVER, tools/solve_probe.py, tests/test_tile_solver_gpu.py need such a build)
    build(force="--force" in sys.argv, verbose=True, extra=("-DSGP_EXPERIMENTS",) if "--experiments" in sys.argv else ())
