"""Two adjacent tiles in ONE process, ghosts handed over by direct calls (export -> route -> split -> import, no process group): the
HIP worlds and the oracle worlds go through the same sequence and must agree bit for bit -- the tile path (ghost import as kinematic
bodies, persistent ghost ids, emigration / immigration) measured against the oracle like every other part of the step."""
import numpy as np
import pytest

from substrata_amd import abi, scenes, tiles
from helpers import DT
import parity

pytestmark = pytest.mark.gpu

TILE_W = 12.0


def tile_scene(rank):
    lo, hi, origin = tiles.tile_bounds(rank, 2, TILE_W, TILE_W)
    d, _ = scenes.lattice(6, 6, 3, 1.9, 0.6, seed=31 + rank, jitter=0.08, random_rot=True, origin_centered=False)
    d["pos"][:, 0] += origin[0] + 1.2
    d["pos"][:, 1] += origin[1] + 1.0
    d["shape_type"] = np.arange(len(d)) % 3
    d["shape"][:, :2] = np.where((np.arange(len(d)) % 3 == 0)[:, None], 0.5, np.float32([0.3, 0.45]))
    d["shape"][np.arange(len(d)) % 3 == 0, 2] = 0.5
    if rank == 0:      # a ball rolling into the other tile: ownership migrates
        b = scenes.dynamic_bodies(1)
        b["shape_type"] = abi.SHAPE_SPHERE
        b["shape"][0] = (0.5, 0, 0, 0)
        b["pos"][0] = (TILE_W - 2.5, 6.0, 3.6)
        b["lin_vel"][0] = (5.0, 0.3, 0.0)
        d = np.concatenate([d, b])
    return np.concatenate([scenes.ground(), d]), lo, hi


def exchange(worlds, boxes, margin, log):
    """What GhostExchange does across ranks, in-process for two worlds."""
    sent = []
    for r, w in enumerate(worlds):
        recs = w.export_boundary(boxes[r, :3], boxes[r, 3:], margin)
        send, counts, emig = tiles.route(recs, r, boxes, margin + 1.5)
        for i in emig:
            w.remove(int(i))
        sent.append(send)
        log.append((r, len(recs), len(send), len(emig)))
    for r, w in enumerate(worlds):
        ghosts, immigrants = tiles.split(sent[1 - r], boxes[r, :3], boxes[r, 3:])
        w.import_ghosts(ghosts)
        if len(immigrants):
            w.add_batch(tiles.records_to_descs(immigrants))
        log.append((r, len(ghosts), len(immigrants)))


def test_two_tiles_hip_against_oracle(oracle):
    from substrata_amd.lib import World
    scenes_, boxes = [], []
    for r in range(2):
        d, lo, hi = tile_scene(r)
        scenes_.append(d); boxes.append(np.concatenate([lo, hi]))
    boxes = np.array(boxes, np.float32)
    gpu = [World(max_bodies=512) for _ in range(2)]
    cpu = [oracle.OracleWorld(max_bodies=512) for _ in range(2)]
    for r in range(2):
        gpu[r].add_batch(scenes_[r]); cpu[r].add_batch(scenes_[r])
    migrated = 0
    for s in range(1, 241):
        lg, lc = [], []
        exchange(gpu, boxes, 1.5, lg)
        exchange(cpu, boxes, 1.5, lc)
        assert lg == lc, (s, lg, lc)
        migrated += sum(e[3] for e in lg if len(e) == 4)
        for r in range(2):
            gpu[r].step(DT); cpu[r].step(DT)
        if s % 30 == 0:
            for r in range(2):
                sg, sc = gpu[r].read_states(0, 512), cpu[r].read_states(0, 512)
                d = parity.state_diff(sg, sc)
                assert d["bit_exact"] and d["active_mismatch"] == 0, (s, r, d)
    assert migrated >= 1                                  # the ball changed owner
    assert any(e[1] > 0 for e in lg if len(e) == 3)        # ghosts are still being exchanged at the end
    for w in gpu + cpu:
        w.close()


def test_two_tiles_native_exchange_against_oracle(oracle):
    """The same scene with the HIP worlds exchanged by the native path (sgp_tiles_exchange_group: routing kernels, device-to-device
    copies, device-side ghost refresh while the set is unchanged) and the oracle worlds by the Python statement of the rules: identical
    counts every step, identical bits every 30 steps, ownership migrations reported with the body's user data."""
    from substrata_amd.lib import World
    scenes_, boxes = [], []
    for r in range(2):
        d, lo, hi = tile_scene(r)
        d["userdata"] = 1000 * (r + 1) + np.arange(len(d))
        scenes_.append(d); boxes.append(np.concatenate([lo, hi]))
    boxes = np.array(boxes, np.float32)
    gpu = [World(max_bodies=512) for _ in range(2)]
    cpu = [oracle.OracleWorld(max_bodies=512) for _ in range(2)]
    for r in range(2):
        gpu[r].add_batch(scenes_[r]); cpu[r].add_batch(scenes_[r])
    nt = [tiles.NativeTiles(gpu[r], r, 2, boxes, 1.5) for r in range(2)]
    migrations = []
    for s in range(1, 241):
        lc = []
        tiles.NativeTiles.exchange_group(nt)
        exchange(cpu, boxes, 1.5, lc)
        for r in range(2):
            st = nt[r].stats()
            exp = [e for e in lc if len(e) == 4 and e[0] == r][0]; imp = [e for e in lc if len(e) == 3 and e[0] == r][0]
            assert (st.exported, st.emigrated, st.ghosts, st.immigrated) == (exp[2], exp[3], imp[1], imp[2]), (s, r)
            migrations += list(nt[r].drain_migrations())
        for r in range(2):
            gpu[r].step(DT); cpu[r].step(DT)
        if s % 30 == 0:
            for r in range(2):
                d = parity.state_diff(gpu[r].read_states(0, 512), cpu[r].read_states(0, 512))
                assert d["bit_exact"] and d["active_mismatch"] == 0, (s, r, d)
    out = [m for m in migrations if m["direction"] == 0]; inn = [m for m in migrations if m["direction"] == 1]
    assert len(out) >= 1 and len(out) == len(inn)
    assert sorted(int(m["userdata"]) for m in out) == sorted(int(m["userdata"]) for m in inn)       # the same objects left and arrived
    assert all(int(m["new_id"]) != abi.INVALID_ID and int(m["peer"]) in (0, 1) for m in inn)
    # the steady state ran on the device: most imports never touched the host
    assert nt[0].stats().fast_imports > nt[0].stats().slow_imports
    # a ray in the new owner's world finds the migrated ball under its original user data
    ball_ud = int(scenes_[0]["userdata"][-1])
    assert ball_ud in [int(m["userdata"]) for m in inn]
    new_id = [int(m["new_id"]) for m in inn if int(m["userdata"]) == ball_ud][-1]
    stt = gpu[1].get_state([new_id])[0]
    ray = np.zeros(1, dtype=abi.ray_dtype); ray["origin"] = stt["pos"] + np.float32([0, 0, 5]); ray["dir"] = (0, 0, -1); ray["max_t"] = 10; ray["ignore_id"] = abi.INVALID_ID
    h = gpu[1].raycast(ray)[0]
    assert int(h["id"]) == new_id and int(h["userdata"]) == ball_ud
    for t in nt:
        t.close()
    for w in gpu + cpu:
        w.close()


def test_rccl_binding_self_test():
    """The RCCL entry points libsgp.so binds at run time (ncclGetUniqueId, CommInitRank, AllGather, grouped Send / Recv), as far as one GPU
    allows: a one-rank communicator, a counts all-gather and a grouped send/recv of records to itself, checked byte for byte."""
    from substrata_amd.lib import World
    w = World(max_bodies=64)
    w.add_batch(scenes.ground())
    rc = w._lib.sgp_tiles_selftest_rccl(w._h, 3000)
    assert rc == 0, w._lib.sgp_last_error().decode()
    w.close()
