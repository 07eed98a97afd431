"""High colours solved by connected component (k_hc_* / k_solve_hc: one launch per pass for every colour from the plan's `hc_first` on).
Which colours go that way is a launch-plan knob and must never show in the results: a pile deep enough for a dozen colours, stepped on
the HIP path against the oracle bit for bit, with the knob at "off" (tail kernel), at its default, and at "every colour" (where the
pile is one component far too large for a workgroup, so the serial catch-all of the same launch does the work)."""
import os

import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT
import parity

pytestmark = pytest.mark.gpu

ENV = ("SGP_NO_SMALL_WORLD", "SGP_TAIL_THRESHOLD", "SGP_HC_BUDGET", "SGP_HC_MIN_COLOURS")


def pile_scene(n_side=6, layers=14):
    """A tight column of boxes, spheres and capsules: every body ends up with many neighbours (many colours)."""
    n = n_side * n_side * layers
    d = scenes.dynamic_bodies(n, mass=10.0)
    i = np.arange(n)
    d["pos"][:, 0] = (i % n_side - (n_side - 1) / 2) * 0.72 + 0.05 * ((i // (n_side * n_side)) % 2)
    d["pos"][:, 1] = ((i // n_side) % n_side - (n_side - 1) / 2) * 0.72 + 0.03 * ((i // (n_side * n_side)) % 3)
    d["pos"][:, 2] = 0.5 + (i // (n_side * n_side)) * 0.75
    sph, cap = i % 5 == 0, (i % 7 == 0) & (i % 5 != 0)
    d["shape"][:, :3] = 0.35
    d["shape_type"][:] = abi.SHAPE_BOX
    d["shape_type"][sph] = abi.SHAPE_SPHERE
    d["shape"][sph, 1:] = 0
    d["shape_type"][cap] = abi.SHAPE_CAPSULE
    d["shape"][cap, 0] = 0.25
    d["shape"][cap, 1] = 0.3
    d["shape"][cap, 2:] = 0
    return np.concatenate([scenes.ground(), d])


def make_twin_with_env(oracle, env, **kw):
    old = {k: os.environ.get(k) for k in ENV}
    try:
        for k in ENV:
            os.environ.pop(k, None)
        os.environ.update(env)
        return parity.make_twin(oracle, **kw)      # (the product reads the switches when the world is created)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("budget", ["0", "160", "1000"])
def test_pile_by_component_matches_oracle(oracle, budget):
    descs = pile_scene()
    tw = make_twin_with_env(oracle, {"SGP_NO_SMALL_WORLD": "1", "SGP_TAIL_THRESHOLD": "24", "SGP_HC_BUDGET": budget}, max_bodies=1024)
    tw.add_batch(descs)
    by_component = catch_all = colours = 0
    for s in range(1, 101):
        tw.step(DT)
        sg = tw.gpu.stats()
        by_component = max(by_component, sg.num_component_constraints)
        catch_all = max(catch_all, sg.num_catch_all_constraints)
        colours = max(colours, sg.num_colours)
        if s % 10 == 0 or s <= 3:
            sc = tw.cpu.stats()
            assert (sg.num_manifolds, sg.num_contact_points, sg.num_colours) == (sc.num_manifolds, sc.num_contact_points, sc.num_colours), s
            d = parity.compare(tw, len(descs))
            assert d["bit_exact"] and d["active_mismatch"] == 0, (s, budget, d)
    assert colours >= 8, colours
    if budget == "0":
        assert by_component == 0
    else:
        assert by_component > 100, by_component
    if budget == "1000":
        assert catch_all > 256, catch_all          # the whole pile is one component: the serial catch-all ran
    tw.close()


def test_compact_rows_give_the_same_bits(monkeypatch):
    """SGP_COMPACT_ROWS_MIN / SGP_ROWS_MODE: from that many constraints on the velocity iterations read no precomputed rows at all (mode 2: every
    lane rebuilds r x axis and I (r x axis) from its body's lever arm and records) or only r x axis (mode 1) -- what a million-body world does to
    cut the bytes it streams per pass.  The same functions of the same operands, so the same bits as the full rows."""
    from substrata_amd.lib import World
    descs = scenes.small_mixed(20, 6, seed=23)          # (more than 2048 body slots: the small-world kernels never use compact rows)
    out = []
    for thr, mode in (("4000000000", "2"), ("0", "2"), ("0", "1")):
        monkeypatch.setenv("SGP_COMPACT_ROWS_MIN", thr)
        monkeypatch.setenv("SGP_ROWS_MODE", mode)
        monkeypatch.setenv("SGP_NO_SMALL_WORLD", "1")
        w = World(max_bodies=4096)
        w.add_batch(descs)
        for _ in range(150):
            w.step(DT)
        out.append(w.read_states(0, len(descs)))
        w.close()
    assert parity.state_diff(out[0], out[1])["bit_exact"] and parity.state_diff(out[0], out[2])["bit_exact"]
