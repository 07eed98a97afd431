"""Committed regression fixtures (tests/golden/*.npz, made by tools/make_golden.py from the oracle): the oracle must still
reproduce them bit for bit (CPU), and so must the HIP path through the C ABI (GPU, no oracle in the loop)."""
import os

import numpy as np
import pytest

import golden_scenes as gs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def check(name, make_world):
    fn, kind = gs.SCENARIOS[name]
    want = np.load(os.path.join(GOLDEN, name + ".npz"))
    seen = 0
    for step, st in fn(make_world):
        tag = f"s{step}_"
        if kind == "full":
            for k, v in st.items():
                assert np.array_equal(v.view(np.uint8), want[tag + k].view(np.uint8)), f"{name} step {step}: {k} differs from the fixture"
        else:
            for k, v in st.items():
                assert np.array_equal(v[:gs.HEAD].view(np.uint8), want[tag + k].view(np.uint8)), f"{name} step {step}: {k}[:{gs.HEAD}]"
            assert int(st["active"].sum()) == int(want[tag + "n_active"])
            assert gs.digest(st) == bytes(want[tag + "sha256"]).hex(), f"{name} step {step}: state digest differs from the fixture"
        seen += 1
    assert seen == len(gs.CHECKPOINTS)


@pytest.mark.parametrize("name", ["config1", "mixed", "car", "hulls_on_terrain", "big_hulls", "config2"])
def test_oracle_reproduces_golden(oracle, name):
    check(name, lambda **kw: oracle.OracleWorld(**kw))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["config1", "mixed", "car", "hulls_on_terrain", "big_hulls", "config2"])
def test_hip_path_reproduces_golden(name):
    from substrata_amd.lib import World
    check(name, lambda **kw: World(**kw))
