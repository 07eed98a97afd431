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


@pytest.mark.parametrize("host_records", [0, 1])
def test_two_tiles_native_exchange_against_oracle(oracle, host_records, monkeypatch):
    """host_records = 0 (round 6, the default): newcomers to the ghost set and bodies that change owner are created ON THE DEVICE from the received
    records, the host hands out the slots from 16 + 16-byte keys; 1 (SGP_TILES_HOST_RECORDS=1): the round-5 road, records to the host and create
    commands back.  Both must equal the oracle bit for bit.
    The same scene with the HIP worlds exchanged by the native path (sgp_tiles_exchange_group: routing kernels, device-to-device
    copies, device-side ghost refresh while the set is unchanged) and the oracle worlds by the Python statement of the rules: identical
    counts every step, identical bits every 30 steps, ownership migrations reported with the body's user data."""
    from substrata_amd.lib import World
    monkeypatch.setenv("SGP_TILES_HOST_RECORDS", str(host_records))
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
    # ... and a changed set (primitives only here) did not bring the records to the host either: the device created the newcomers from them
    for t_ in nt:
        st_ = t_.stats()
        if host_records:
            assert st_.device_creates == 0 and st_.slow_imports > 0
        else:
            assert st_.slow_imports == 0 and st_.device_creates > 0
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


def test_a_car_crossing_the_border_keeps_its_vehicle(oracle):
    """A vehicle's chassis never changes owner (SGP_GHOST_FLAG_CHASSIS: the vehicle record -- engine, gearbox, wheel state -- lives with the tile
    that created it): the car drives out of tile 0 into tile 1, stays tile 0's body, keeps driving, and tile 1 meets it as a ghost that pushes
    its boxes.  Native exchange on the HIP worlds, the Python statement of the rules on the oracle worlds: same counts, same bits."""
    from substrata_amd.lib import World
    from helpers import add_car
    boxes = []
    for r in range(2):
        lo, hi, _ = tiles.tile_bounds(r, 2, TILE_W, TILE_W)
        boxes.append(np.concatenate([lo, hi]))
    boxes = np.array(boxes, np.float32)
    gpu = [World(max_bodies=256) for _ in range(2)]
    cpu = [oracle.OracleWorld(max_bodies=256) for _ in range(2)]
    cars = []
    for ws in (gpu, cpu):
        for r in range(2):
            ws[r].add_batch(scenes.ground())
        # tile 1: a few boxes in the car's way, just behind the border
        d = scenes.dynamic_bodies(4, mass=20.0)
        d["shape"][:, :3] = 0.4
        d["pos"] = [(TILE_W + 2.0 + 0.9 * k, 6.0 + 0.5 * (k % 2), 0.41) for k in range(4)]
        ws[1].add_batch(d)
        q = (0.0, 0.0, -np.sin(np.pi / 4), np.cos(np.pi / 4))                      # the car's forward (+y) turned to +x
        cars.append(add_car(ws[0], pos=(TILE_W - 5.0, 6.0, 0.8), rot=q))
    assert cars[0] == cars[1]
    body, veh = cars[0]
    nt = [tiles.NativeTiles(gpu[r], r, 2, boxes, 1.5) for r in range(2)]
    for s in range(1, 301):
        for ws in (gpu, cpu):
            ws[0].vehicle_set_input(veh, forward=1.0)
        lc = []
        tiles.NativeTiles.exchange_group(nt)
        exchange(cpu, boxes, 1.5, lc)
        for r in range(2):
            st = nt[r].stats()
            exp = [e for e in lc if len(e) == 4 and e[0] == r][0]; imp = [e for e in lc if len(e) == 3 and e[0] == r][0]
            assert (st.exported, st.emigrated, st.ghosts, st.immigrated) == (exp[2], exp[3], imp[1], imp[2]), (s, r)
            assert st.emigrated == 0 or r == 1                  # the car never emigrates (tile 1's boxes may, once pushed around)
        for r in range(2):
            gpu[r].step(DT); cpu[r].step(DT)
        if s % 50 == 0:
            for r in range(2):
                d = parity.state_diff(gpu[r].read_states(0, 256), cpu[r].read_states(0, 256))
                assert d["bit_exact"] and d["active_mismatch"] == 0, (s, r, d)
    car = gpu[0].get_state([body])[0]
    assert car["pos"][0] > TILE_W + 1.0, car["pos"]                         # it is well inside tile 1's region ...
    vs_g, vs_c = gpu[0].vehicle_get_state(veh), cpu[0].vehicle_get_state(veh)
    assert float(vs_g["engine_rpm"]) == float(vs_c["engine_rpm"]) and float(vs_g["engine_rpm"]) > 1000.0     # ... and still a car, in tile 0's world
    moved = gpu[1].read_states(1, 4)
    assert np.max(np.abs(moved["pos"][:, 0] - [TILE_W + 2.0 + 0.9 * k for k in range(4)])) > 0.3      # its ghost pushed tile 1's boxes
    for t in nt:
        t.close()
    for w in gpu + cpu:
        w.close()


def sensor_border_scene(rank):
    """Tile 1 holds a box at rest next to the border; tile 0 holds a SENSOR box and a box on the non-collidable moving layer, both overlapping it
    across the border.  If their ghosts were solid (the round-2 behaviour) they would shove the resting box away."""
    lo, hi, origin = tiles.tile_bounds(rank, 2, TILE_W, TILE_W)
    if rank == 1:
        d = scenes.dynamic_bodies(1)
        d["shape_type"] = abi.SHAPE_BOX; d["shape"][0] = (0.5, 0.5, 0.5, 0)
        d["pos"][0] = (TILE_W + 0.7, 6.0, 0.5)
    else:
        d = scenes.dynamic_bodies(2)
        d["shape_type"] = abi.SHAPE_BOX; d["shape"][:, :3] = 0.5
        d["pos"][0] = (TILE_W - 0.1, 6.0, 0.5); d["is_sensor"][0] = 1
        d["pos"][1] = (TILE_W - 0.2, 6.0, 1.6); d["layer"][1] = abi.LAYER_MOVING_NON_COLLIDABLE
        d["gravity_factor"][:] = 0.0            # they hover where they are
    return np.concatenate([scenes.ground(), d]), lo, hi


def test_sensor_and_non_collidable_ghosts_are_not_solid(oracle):
    from substrata_amd.lib import World
    scenes_, boxes = [], []
    for r in range(2):
        d, lo, hi = sensor_border_scene(r)
        scenes_.append(d); boxes.append(np.concatenate([lo, hi]))
    boxes = np.array(boxes, np.float32)
    gpu = [World(max_bodies=64) for _ in range(2)]
    cpu = [oracle.OracleWorld(max_bodies=64) for _ in range(2)]
    for r in range(2):
        gpu[r].add_batch(scenes_[r]); cpu[r].add_batch(scenes_[r])
    nt = [tiles.NativeTiles(gpu[r], r, 2, boxes, 1.5) for r in range(2)]
    for s in range(1, 61):
        lc = []
        tiles.NativeTiles.exchange_group(nt)
        exchange(cpu, boxes, 1.5, lc)
        for r in range(2):
            gpu[r].step(DT); cpu[r].step(DT)
    assert nt[1].stats().ghosts == 2                       # both hovering boxes are ghosts in tile 1 ...
    for r in range(2):
        d = parity.state_diff(gpu[r].read_states(0, 64), cpu[r].read_states(0, 64))
        assert d["bit_exact"] and d["active_mismatch"] == 0, (r, d)
    rest = gpu[1].get_state([1])[0]                         # ... and the resting box has not been pushed by either
    assert abs(float(rest["pos"][0]) - (TILE_W + 0.7)) < 1e-3 and abs(float(rest["pos"][1]) - 6.0) < 1e-3
    for t in nt:
        t.close()
    for w in gpu + cpu:
        w.close()


def test_boxes_with_a_gap_keep_their_bodies():
    """sgp_tiles_create takes arbitrary boxes.  A dynamic body whose centre leaves its tile into a region NO tile covers must stay with its
    owner (it used to be removed by the owner and accepted by nobody)."""
    from substrata_amd.lib import World
    boxes = np.float32([[-1e9, -1e9, -1e9, 5.0, 1e9, 1e9], [8.0, -1e9, -1e9, 1e9, 1e9, 1e9]])      # a 3 m gap between x = 5 and x = 8
    gpu = [World(max_bodies=64) for _ in range(2)]
    d = scenes.dynamic_bodies(1)
    d["shape_type"] = abi.SHAPE_SPHERE; d["shape"][0] = (0.5, 0, 0, 0); d["pos"][0] = (4.0, 0.0, 0.5); d["lin_vel"][0] = (4.0, 0, 0); d["friction"] = 0.0
    gpu[0].add_batch(np.concatenate([scenes.ground(), d])); gpu[1].add_batch(scenes.ground())
    nt = [tiles.NativeTiles(gpu[r], r, 2, boxes, 1.0) for r in range(2)]
    owners = []
    for s in range(90):
        tiles.NativeTiles.exchange_group(nt)
        owned = [w.num_bodies() - 1 - t.stats().ghosts for w, t in zip(gpu, nt)]
        assert sum(owned) == 1, (s, owned)               # never lost, never duplicated
        owners.append(owned.index(1))
        for w in gpu:
            w.step(DT)
    assert owners[0] == 0 and owners[-1] == 1            # it crossed the gap with tile 0 and was handed over once tile 1's region held it
    for t in nt:
        t.close()
    for w in gpu:
        w.close()


def test_route_retry_with_a_small_send_buffer(oracle):
    """More boundary bodies than the first send buffer holds (16384 records): the exchange grows the buffer and routes again on its own --
    in the RCCL form without a second all-gather and without returning to the caller -- and the result is what the oracle tiles get."""
    from substrata_amd.lib import World
    n = 18000
    boxes = np.float32([[-1e9, -1e9, -1e9, 0.0, 1e9, 1e9], [0.0, -1e9, -1e9, 1e9, 1e9, 1e9]])
    d = scenes.dynamic_bodies(n)
    d["shape_type"] = abi.SHAPE_SPHERE; d["shape"][:, 0] = 0.2
    k = np.arange(n)
    d["pos"][:, 0] = -0.5; d["pos"][:, 1] = 0.5 * (k % 150); d["pos"][:, 2] = 0.3 + 0.5 * (k // 150)
    d["gravity_factor"] = 0.0
    gpu = [World(max_bodies=2 * n + 64) for _ in range(2)]
    cpu = [oracle.OracleWorld(max_bodies=2 * n + 64) for _ in range(2)]
    for ws in (gpu, cpu):
        ws[0].add_batch(np.concatenate([scenes.ground(), d])); ws[1].add_batch(scenes.ground())
    nt = [tiles.NativeTiles(gpu[r], r, 2, boxes, 1.0) for r in range(2)]
    for s in range(3):
        lc = []
        tiles.NativeTiles.exchange_group(nt)
        exchange(cpu, boxes, 1.0, lc)
        for r in range(2):
            gpu[r].step(DT); cpu[r].step(DT)
    assert nt[1].stats().ghosts == n
    dd = parity.state_diff(gpu[1].read_states(0, n + 8), cpu[1].read_states(0, n + 8))
    assert dd["bit_exact"], dd
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
