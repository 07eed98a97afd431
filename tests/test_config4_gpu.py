"""BASELINE config 4 (1M boxes, 100^3 lattice, spacing 1.25 m, 2x2x2 spatial tiles).

(i)  the full 1M-box world on ONE GPU: size-independent properties (nothing dropped, valid colouring, finite state, no energy
     created, run-to-run bit equality);
(ii) the 2x2x2 split on a scaled-down lattice, eight tile worlds in one process, HIP against oracle tiles bit for bit, with bodies
     falling through the z faces of the tiling (ownership migrates downwards) -- the tile path of the named config measured against
     the oracle like every other part of the step;
(iii) the two-process z-split over gloo lives in tests/test_tiles_gloo.py (CPU).
"""
import numpy as np
import pytest

from substrata_amd import abi, scenes, tiles
from helpers import DT
import parity
import ghost_exchange

pytestmark = pytest.mark.gpu


def colouring_valid(cons, movable):
    """Vectorised parity.check_colouring_valid: no two constraints of one (non-overflow) colour share a movable body."""
    c = cons[cons["colour"] != 63]
    keys = []
    for side in ("a", "b"):
        m = movable[c[side]]
        keys.append(c["colour"][m].astype(np.int64) << np.int64(32) | c[side][m].astype(np.int64))
    k = np.concatenate(keys)
    return len(np.unique(k)) == len(k)


def potential_energy(descs, st):
    return float(np.sum(descs["mass"][1:].astype(np.float64) * 9.81 * st["pos"][1:, 2].astype(np.float64)))


def kinetic_energy(descs, st):
    return float(np.sum(0.5 * descs["mass"][1:].astype(np.float64) * np.sum(st["lin_vel"][1:].astype(np.float64) ** 2, axis=1)))


@pytest.mark.timeout(1500)
def test_config4_1m_boxes_single_gpu_properties():
    from substrata_amd.lib import World
    descs = scenes.config4_1m_boxes()
    n = len(descs)
    assert n == 1_000_001
    runs = []
    for rep in range(2):
        w = World(max_bodies=n + 64)
        w.add_batch(descs)
        e0 = potential_energy(descs, w.read_states(0, n))
        for _ in range(48):
            w.step(DT)
        st = w.read_states(0, n)
        stats = w.stats()
        runs.append(st)
        if rep == 0:
            cons = w.dump_constraints(cap=stats.num_manifolds + 1024)
            assert stats.num_bodies == n and stats.pairs_dropped == 0 and stats.manifolds_dropped == 0
            # the lowest layers have landed on the ground and on each other
            assert stats.num_manifolds > 10000 and stats.num_overflow_constraints == 0 and stats.num_colours <= 40
            for f in ("pos", "rot", "lin_vel", "ang_vel"):
                assert np.all(np.isfinite(st[f])), f
            assert np.max(np.abs(np.linalg.norm(st["rot"], axis=1) - 1.0)) < 1e-5
            assert st["pos"][1:, 2].min() > -0.5
            # the top layer, 120 m above the contact front, still follows the closed-form semi-implicit Euler free fall -- up to the small
            # kicks randomly rotated unit cubes at 1.25 m spacing give each other (corners reach 0.87 m: neighbours touch from step 1)
            top = st["pos"][-10000:, 2]
            v, z = 0.0, float(descs["pos"][-1, 2])
            for _ in range(48):
                v = np.float32(np.float32(v + np.float32(-9.81) * np.float32(DT)) * np.float32(1.0 - 0.05 * DT))
                z = np.float32(z + v * np.float32(DT))
            assert abs(float(np.mean(top)) - float(z)) < 0.1 and np.max(np.abs(top - z)) < 1.0
            assert potential_energy(descs, st) + kinetic_energy(descs, st) <= e0 * (1.0 + 1e-4)
            movable = st["active"].astype(bool)
            movable[0] = False
            assert len(cons) == stats.num_manifolds
            assert colouring_valid(cons, movable)
            key = cons["a"].astype(np.uint64) << np.uint64(32) | cons["b"].astype(np.uint64)
            assert np.all(np.diff(key.astype(np.int64)) > 0)
        w.close()
    for f in ("pos", "rot", "lin_vel", "ang_vel"):
        assert np.array_equal(runs[0][f].view(np.uint32), runs[1][f].view(np.uint32)), f


@pytest.mark.timeout(900)
def test_config4_scaled_down_2x2x2_tiles_against_oracle(oracle):
    """n = 8 lattice (512 boxes) split 2x2x2: the four upper tiles own the upper half of the tower, which falls through the z = 5 m
    faces into the lower tiles.  Eight HIP worlds and eight oracle worlds go through the same exchange; states must agree bit for bit,
    no body may be lost or duplicated, and bodies must have migrated downwards through the z faces."""
    from substrata_amd.lib import World
    n, n_tiles = 8, 8
    assert tiles.tile_grid(n_tiles) == (2, 2, 2)
    tile_descs, boxes = [], []
    for r in range(n_tiles):
        d, lo, hi = scenes.config4_tile_descs(r, n_tiles, n=n)
        tile_descs.append(d); boxes.append(np.concatenate([lo, hi]))
    boxes = np.array(boxes, np.float32)
    total = sum(len(d) - 1 for d in tile_descs)
    assert total == n ** 3 and all(len(d) - 1 == n ** 3 // 8 for d in tile_descs)
    cap = 2048
    gpu = [World(max_bodies=cap) for _ in range(n_tiles)]
    cpu = [oracle.OracleWorld(max_bodies=cap) for _ in range(n_tiles)]
    for r in range(n_tiles):
        assert np.array_equal(gpu[r].add_batch(tile_descs[r]), cpu[r].add_batch(tile_descs[r]))
    margin = 2.0                       # ghost margin = 1 cell (2 m)
    migrated_down = 0
    # the HIP tiles go through the native exchange (sgp_tiles_*: routing on the device, device-to-device copies between the tiles of this
    # process), the oracle tiles through the Python statement of the same rules
    nt = [tiles.NativeTiles(gpu[r], r, n_tiles, boxes, margin) for r in range(n_tiles)]
    for s in range(1, 181):
        lc = []
        tiles.NativeTiles.exchange_group(nt)
        ghost_exchange.exchange_in_process(cpu, boxes, margin, lc)
        lg = []
        for r in range(n_tiles):
            st = nt[r].stats()
            exp = [e for e in lc if e[0] == "export" and e[1] == r][0]; imp = [e for e in lc if e[0] == "import" and e[1] == r][0]
            assert (st.exported, st.emigrated, st.ghosts, st.immigrated) == (sum(exp[3]), exp[4], imp[2], imp[3]), (s, r)
        lg = lc
        migrated_down += sum(e[4] for e in lg if e[0] == "export" and e[1] >= 4)
        for r in range(n_tiles):
            gpu[r].step(DT); cpu[r].step(DT)
        if s % 30 == 0:
            for r in range(n_tiles):
                dd = parity.state_diff(gpu[r].read_states(0, cap), cpu[r].read_states(0, cap))
                assert dd["bit_exact"] and dd["active_mismatch"] == 0, (s, r, dd)
            owned = sum(gpu[r].num_bodies() - 1 - [e for e in lg if e[0] == "import" and e[1] == r][0][2] for r in range(n_tiles))
            assert owned == total, (s, owned, total)
    # the tower compacts: well over a layer of the upper tiles' bodies (4 x 64) fell through the z = 5 m faces and changed owner,
    # the rest now rests on the pile above the face -- still owned by the upper tiles, with ghosts crossing the face both ways
    assert migrated_down >= 100
    upper_owned = sum(gpu[r].num_bodies() - 1 - [e for e in lg if e[0] == "import" and e[1] == r][0][2] for r in range(4, 8))
    assert 0 < upper_owned < n ** 3 // 2
    assert all([e for e in lg if e[0] == "import" and e[1] == r][0][2] > 0 for r in range(8))      # every tile holds ghosts at the end
    for t in nt:
        t.close()
    for w in gpu + cpu:
        w.close()


def test_config4_scaled_down_tiles_with_rebalancing_against_oracle(oracle):
    """The same falling tower with the tile regions re-balanced every 12 steps (sgp_tiles_rebalance_group: split planes at the quantiles of the
    owned bodies, by body count and by bodies + contacts in turn).  The oracle tiles exchange with the regions the device computed; states stay
    bit-identical, no body is lost, and -- the point of it -- no tile runs empty while the static split's upper tiles lose most of theirs."""
    from substrata_amd.lib import World
    n, n_tiles = 8, 8
    grid = tiles.tile_grid(n_tiles)
    tile_descs, boxes = [], []
    for r in range(n_tiles):
        d, lo, hi = scenes.config4_tile_descs(r, n_tiles, n=n)
        tile_descs.append(d); boxes.append(np.concatenate([lo, hi]))
    boxes = np.array(boxes, np.float32)
    total = n ** 3
    cap = 2048
    gpu = [World(max_bodies=cap) for _ in range(n_tiles)]
    cpu = [oracle.OracleWorld(max_bodies=cap) for _ in range(n_tiles)]
    for r in range(n_tiles):
        assert np.array_equal(gpu[r].add_batch(tile_descs[r]), cpu[r].add_batch(tile_descs[r]))
    margin = 2.0
    nt = [tiles.NativeTiles(gpu[r], r, n_tiles, boxes, margin) for r in range(n_tiles)]
    min_share, rebalances = 1.0, 0
    for s in range(0, 240):
        if s % 12 == 0:
            tiles.NativeTiles.rebalance_group(nt, grid, by_contacts=(s // 12) % 2 == 1)
            rebalances += 1
            boxes = nt[0].boxes()
            assert all(np.array_equal(boxes, t.boxes()) for t in nt)
            # a partition of space: the tiles of a column share their z plane, of a slab their y plane, all of them the x plane
            b = boxes.reshape(2, 2, 2, 6)                      # [iz][iy][ix]
            assert np.all(b[:, :, 0, 3] == b[:, :, 1, 0]) and np.all(b[:, 0, :, 4] == b[:, 1, :, 1]) and np.all(b[0, :, :, 5] == b[1, :, :, 2])
        lc = []
        tiles.NativeTiles.exchange_group(nt)
        ghost_exchange.exchange_in_process(cpu, boxes, margin, lc)
        for r in range(n_tiles):
            st = nt[r].stats()
            exp = [e for e in lc if e[0] == "export" and e[1] == r][0]; imp = [e for e in lc if e[0] == "import" and e[1] == r][0]
            assert (st.exported, st.emigrated, st.ghosts, st.immigrated) == (sum(exp[3]), exp[4], imp[2], imp[3]), (s, r)
        owned = [gpu[r].num_bodies() - 1 - nt[r].stats().ghosts for r in range(n_tiles)]
        assert sum(owned) == total, (s, owned)
        if s >= 13:
            min_share = min(min_share, min(owned) / total)
        for r in range(n_tiles):
            gpu[r].step(DT); cpu[r].step(DT)
        if s % 30 == 29:
            for r in range(n_tiles):
                dd = parity.state_diff(gpu[r].read_states(0, cap), cpu[r].read_states(0, cap))
                assert dd["bit_exact"] and dd["active_mismatch"] == 0, (s, r, dd)
    assert nt[0].stats().rebalances == rebalances
    assert min_share >= 0.06, min_share                        # (a static split leaves the upper tiles far fewer: see the test above)
    for t in nt:
        t.close()
    for w in gpu + cpu:
        w.close()
