"""Long mixed-use run, GPU vs oracle: bodies settle and fall asleep, get kicked awake, are added, removed, teleported, pushed
with forces, a kinematic platform moves through them and the water plane is switched on -- 1500 steps of the facade's whole call
surface (PhysicsWorld.cpp:546-722,1169-1353,1356-1443) with both sides compared every 100 steps."""
import numpy as np
import pytest

from substrata_amd import abi, scenes
from helpers import DT, quat_axis_angle
import parity

pytestmark = pytest.mark.gpu


def test_1500_steps_of_mixed_calls_stay_in_lockstep(oracle):
    rng = np.random.default_rng(123)
    tw = parity.make_twin(oracle, max_bodies=1024)
    descs = scenes.small_mixed(7, 3, seed=31)
    tw.add_batch(descs)
    n0 = len(descs)
    # a kinematic platform
    plat = scenes.dynamic_bodies(1)
    plat["motion_type"] = abi.MOTION_KINEMATIC
    plat["shape"][0, :3] = (2.0, 2.0, 0.2); plat["pos"][0] = (12.0, 0.0, 0.2)
    pg, pc = tw.add_batch(plat)
    pid = int(pg[0]); assert pid == int(pc[0])
    tw.set_contact_events(1)
    live = list(range(1, n0))
    bit_exact_all = True
    slept = woke = 0
    for s in range(1, 1501):
        if s % 97 == 0 and live:                      # kick a random body (setNewObToWorldTransform with velocities)
            i = int(rng.choice(live))
            st = tw.gpu.get_state([i])[0]
            tw.set_pose_vel(i, tuple(st["pos"]), tuple(st["rot"]), tuple(rng.uniform(-3, 3, 3)), tuple(rng.uniform(-2, 2, 3)))
        if s % 211 == 0 and len(live) > 20:           # remove one
            i = int(rng.choice(live)); live.remove(i)
            tw.remove(i)
        if s % 173 == 0:                              # add a few new ones above the pile (slots of removed bodies get reused)
            nb = scenes.dynamic_bodies(3)
            nb["pos"] = rng.uniform([-3, -3, 4], [3, 3, 7], size=(3, 3)).astype(np.float32)
            nb["shape_type"] = [abi.SHAPE_BOX, abi.SHAPE_SPHERE, abi.SHAPE_CAPSULE]
            nb["shape"][1, :3] = (0.4, 0, 0); nb["shape"][2, :3] = (0.3, 0.5, 0)
            ig, ic = tw.add_batch(nb)
            assert np.array_equal(ig, ic)
            live += [int(x) for x in ig]
        if 300 <= s < 700:                            # the platform sweeps through the scene and back (moveKinematicObject)
            x = 12.0 - 20.0 * np.sin((s - 300) / 400.0 * np.pi)
            tw.move_kinematic(pid, (float(x), 0.0, 0.2), quat_axis_angle((0, 0, 1), 0.002 * (s - 300)), DT)
        if s % 50 == 0 and live:                      # controller-style forces (BodyInterface::AddForce / AddTorque)
            i = int(rng.choice(live))
            tw.activate(i)
            tw.add_force(i, (0.0, 0.0, 4000.0)); tw.add_torque(i, (0.0, 50.0, 0.0))
        if s == 900:
            tw.set_water(1, 0.8)
        if s == 1200:
            tw.set_water(0, 0.0)
        tw.step(DT)
        sg, sc = tw.stats()
        slept += sg.num_deactivated; woke += sg.num_activated
        assert (sg.num_activated, sg.num_deactivated) == (sc.num_activated, sc.num_deactivated), s
        if s % 100 == 0:
            hi = max(live + [pid]) + 1
            d = parity.compare(tw, hi)
            assert d["active_mismatch"] == 0, (s, d)
            assert d["pos"] <= 2e-4 and d["rot"] <= 2e-4 and d["lin_vel"] <= 2e-3 and d["ang_vel"] <= 2e-3, (s, d)
            bit_exact_all = bit_exact_all and d["bit_exact"]
            eg = tw.gpu.drain_events(abi.EVENT_CONTACT_ADDED); ec = tw.cpu.drain_events(abi.EVENT_CONTACT_ADDED)
            assert len(eg) == len(ec)
            tw.gpu.drain_events(abi.EVENT_CONTACT_PERSISTED); tw.cpu.drain_events(abi.EVENT_CONTACT_PERSISTED)
    print("1500 mixed steps: bit exact at every checkpoint =", bit_exact_all, "| deactivations", slept, "activations", woke)
    assert slept > 20 and woke > 10
    tw.close()
