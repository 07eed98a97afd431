"""The overflow colour: a movable body with more than 63 contacts against other movable bodies (Jolt's non-parallel split).  Its excess
constraints are solved serially in ascending priority after the regular colours, in the warm start (k_warm_bodies + tail), the velocity
and position iterations (tail / small-world kernel).  HIP path against the oracle, bit for bit, in both launch plans."""
import os

import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT
import parity

pytestmark = pytest.mark.gpu


def plate_scene():
    """A 14 x 14 x 0.6 m dynamic plate lying on the ground with 12 x 12 small boxes and spheres dropped onto it."""
    plate = scenes.dynamic_bodies(1, mass=4000.0)
    plate["shape"][0] = (7.0, 7.0, 0.3, 0.0)
    plate["pos"][0] = (0.0, 0.0, 0.31)
    n = 12
    d = scenes.dynamic_bodies(n * n, mass=20.0)
    ij = np.arange(n * n)
    d["pos"][:, 0] = (ij % n - (n - 1) / 2) * 1.1
    d["pos"][:, 1] = (ij // n - (n - 1) / 2) * 1.1
    d["pos"][:, 2] = 1.0 + 0.02 * (ij % 7)
    d["shape"][:, :3] = 0.35
    d["shape_type"] = np.where(ij % 3 == 0, abi.SHAPE_SPHERE, abi.SHAPE_BOX)
    return np.concatenate([scenes.ground(), plate, d])


@pytest.mark.parametrize("small_world", [True, False])
def test_plate_with_more_than_63_contacts(oracle, small_world):
    descs = plate_scene()
    old = os.environ.get("SGP_NO_SMALL_WORLD")
    os.environ["SGP_NO_SMALL_WORLD"] = "0" if small_world else "1"
    try:
        tw = parity.make_twin(oracle, max_bodies=512)
    finally:
        if old is None:
            os.environ.pop("SGP_NO_SMALL_WORLD", None)
        else:
            os.environ["SGP_NO_SMALL_WORLD"] = old
    tw.add_batch(descs)
    seen_overflow = by_component = 0
    for s in range(1, 181):
        tw.step(DT)
        if s % 20 == 0 or s == 1:
            sg, sc = tw.gpu.stats(), tw.cpu.stats()
            assert (sg.num_manifolds, sg.num_contact_points, sg.num_colours, sg.num_overflow_constraints) == \
                   (sc.num_manifolds, sc.num_contact_points, sc.num_colours, sc.num_overflow_constraints), s
            seen_overflow = max(seen_overflow, sg.num_overflow_constraints)
            by_component = max(by_component, sg.num_component_constraints)
            d = parity.compare(tw, len(descs))
            assert d["bit_exact"] and d["active_mismatch"] == 0, (s, d)
    assert seen_overflow > 0, "the scene never produced an overflow constraint"
    # the general launch plan solves the overflow colour in the catch-all of the component launch, the small-world kernel on its own
    assert (by_component > 0) == (not small_world), by_component
    # the pile has settled on the plate
    st = tw.gpu.read_states(0, len(descs))
    assert np.all(st["pos"][2:, 2] > 0.6) and np.all(np.abs(st["lin_vel"][1:]).max(axis=1) < 1.0)
    tw.close()
