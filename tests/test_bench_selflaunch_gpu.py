"""`python3 bench.py --gpus N` exactly as the driver invokes it (no launcher around it, VERDICT r05 task 3): bench.py must start its N ranks itself
(torch.distributed.run on 127.0.0.1) and end its stdout with rank 0's JSON line.

A one-GPU box cannot give two ranks a GPU each, so the -m gpu test sets the test switch SGP_BENCH_SHARE_GPU=1: both ranks on cuda:0, gloo between the
processes, the product's sgp_tiles_exchange carried by the test-only collective library (tests/rccl_standin, through SGP_RCCL_LIBRARY).  The line
says "transport": "test stand-in"; nothing here is a scaling number.  The CPU test checks the launch itself: without a GPU every rank must stop
with the product's "needs a GPU" message -- not with a usage error of bench.py."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_direct_invocation_starts_the_ranks_cpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("the CPU statement of the launch test")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "must be launched through" not in r.stderr + r.stdout
    assert (r.stderr + r.stdout).count("bench.py needs a GPU") >= 1      # the ranks were started and each said why it cannot run here


@pytest.mark.gpu
def test_driver_command_n2_shared_gpu():
    from test_tiles_multirank_gpu import build_standin
    env = dict(os.environ)
    env["SGP_BENCH_SHARE_GPU"] = "1"
    env["SGP_RCCL_LIBRARY"] = build_standin()
    env.setdefault("SGP_RCCL_STANDIN_TIMEOUT_S", "120")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    # the literal driver command (python3 bench.py --gpus N --steps K --warmup W) on the scaled-down tower of BASELINE config 4
    r = subprocess.run(["python3", "bench.py", "--gpus", "2", "--steps", "12", "--warmup", "4", "--lattice", "12", "--profile-steps", "2"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    last = r.stdout.strip().splitlines()[-1]
    out = json.loads(last)                                   # the JSON line is the LAST line of the parent's stdout
    assert out["n_gpus"] == 2 and out["steps"] == 12 and out["warmup"] == 4
    assert out["scaling"] == "strong" and out["transport"] == "test stand-in"
    assert out["value"] > 0 and out["config"]["tiles"] == 2
    assert out["config"]["rccl_ranks_seen_per_tile"] == [2, 2]
    assert out["config"]["dropped_pairs_or_manifolds"] == 0
    assert sum(out["config"]["owned_bodies_per_tile"]) == 12 ** 3
