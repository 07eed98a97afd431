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
