"""The tile solver (k_ts_solve: all velocity iterations of a step in one resident launch, tiles synchronised through neighbour epochs) against
the oracle, bit for bit.  It only re-schedules the colour order per body, so it must reproduce the colour launches exactly -- on a pile that
spans many tiles (shared rim bodies, private interiors), with the sparse high colours, with an overflow colour (a body with more than 63
contacts, solved serially by tile 0), and with bodies large enough to be touched by more than four tiles (all tiles become neighbours)."""
import os

import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT
import parity

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_the_experiments_build():
    """The tile solver is a measured negative (profiles/r03_tile_solver.md) and no longer part of the product library: these tests run where
    libsgp.so was built with `python -m substrata_amd.build --experiments`."""
    from substrata_amd.lib import load
    lib = load()
    if not hasattr(lib, "sgp_debug_has_experiments") or lib.sgp_debug_has_experiments() == 0:
        pytest.skip("libsgp.so was built without csrc/experiments (python -m substrata_amd.build --experiments)")


@pytest.fixture
def ts_env(monkeypatch):
    monkeypatch.setenv("SGP_TILE_SOLVER", "1")
    monkeypatch.setenv("SGP_TS_MIN_CONSTRAINTS", "0")


def run(oracle, descs, steps, check_every, max_bodies):
    from substrata_amd.lib import World
    g = World(max_bodies=max_bodies); c = oracle.OracleWorld(max_bodies=max_bodies)
    g.add_batch(descs); c.add_batch(descs)
    used = 0; kinds = set()
    for s in range(1, steps + 1):
        g.step(DT); c.step(DT)
        t = g.stats().tile_solver
        used += 1 if t else 0; kinds.add(int(t))
        if s % check_every == 0:
            d = parity.state_diff(g.read_states(0, len(descs)), c.read_states(0, len(descs)))
            assert d["bit_exact"] and d["active_mismatch"] == 0, (s, d)
            sg, sc = g.stats(), c.stats()
            assert (sg.num_manifolds, sg.num_contact_points, sg.num_colours) == (sc.num_manifolds, sc.num_contact_points, sc.num_colours), s
    g.close(); c.close()
    return used, kinds


def test_pile_bit_exact_with_the_tile_solver(oracle, ts_env):
    # a quarter of config 3: 25k mixed bodies falling into a pile (enough body slots for the tile solver to apply: 64 colours x 256 tiles)
    descs = scenes.config3_100k_mixed(50, 50, 10, seed=5)
    used, kinds = run(oracle, descs, 150, 30, len(descs) + 64)
    assert used >= 100 and 1 in kinds              # (the first steps have no contacts yet: the plan keeps the ordinary launches)


def test_overflow_colour_and_wide_bodies(oracle, ts_env):
    # the same pile with two 16 m plates thrown in: each collects far more than 63 contacts (overflow colour: tile 0's serial phase) and spans
    # more than four tiles' worth of neighbours (every tile becomes every tile's neighbour for those steps)
    descs = scenes.config3_100k_mixed(50, 50, 7, seed=6)
    plates = scenes.dynamic_bodies(2)
    plates["shape_type"] = abi.SHAPE_BOX
    plates["shape"][:, :3] = (8.0, 8.0, 0.25)
    plates["pos"][0] = (12.0, 12.0, 0.3); plates["pos"][1] = (-14.0, -8.0, 13.5)      # one lies on the ground under the pile, one lands on top
    plates["mass"] = 3000.0
    descs = np.concatenate([descs, plates])
    from substrata_amd.lib import World
    g = World(max_bodies=len(descs) + 64); c = oracle.OracleWorld(max_bodies=len(descs) + 64)
    g.add_batch(descs); c.add_batch(descs)
    overflow = 0; kinds = set()
    for s in range(1, 181):
        g.step(DT); c.step(DT)
        st = g.stats()
        kinds.add(int(st.tile_solver))
        if st.tile_solver:
            overflow = max(overflow, int(st.num_overflow_constraints))
        if s % 30 == 0:
            d = parity.state_diff(g.read_states(0, len(descs)), c.read_states(0, len(descs)))
            assert d["bit_exact"] and d["active_mismatch"] == 0, (s, d)
    assert overflow > 0 and 2 in kinds, (overflow, kinds)
    g.close(); c.close()


def test_on_and_off_give_the_same_bits(ts_env, monkeypatch):
    from substrata_amd.lib import World
    descs = scenes.config3_100k_mixed(60, 60, 6, seed=9)
    out = []
    for on in ("1", "0"):
        monkeypatch.setenv("SGP_TILE_SOLVER", on)
        w = World(max_bodies=len(descs) + 64)
        w.add_batch(descs)
        ran = 0
        for _ in range(120):
            w.step(DT); ran += 1 if w.stats().tile_solver else 0
        assert (ran > 60) == (on == "1")
        out.append(w.read_states(0, len(descs)))
        w.close()
    d = parity.state_diff(out[0], out[1])
    assert d["bit_exact"], d
