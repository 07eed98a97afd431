"""The N > 1 path on CPU: two processes (gloo, world_size 2), one tile each, exchanging ghost bodies through
substrata_amd/tiles.py exactly as bench.py does over RCCL.  The worlds here are oracle worlds (the product has no CPU
path); the exchange code, the export/import ABI semantics and the tiling are the ones the GPU run uses."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from substrata_amd import abi, scenes, tiles  # noqa: E402
import ghost_exchange  # noqa: E402

DT = 1.0 / 60.0
TILE_W = 12.0


def scene_for_tile(rank, n_tiles):
    """A 6x6x2 box lattice per tile (spacing 2 m, so columns straddle the tile border closely) + the ground quad."""
    lo, hi, origin = tiles.tile_bounds(rank, n_tiles, TILE_W, TILE_W)
    d, _ = scenes.lattice(6, 6, 2, 2.0, 0.6, seed=11 + rank, jitter=0.05, random_rot=False, origin_centered=False)
    d["pos"][:, 0] += origin[0] + 1.0
    d["pos"][:, 1] += origin[1] + 1.0
    return np.concatenate([scenes.ground(), d]), lo, hi


def migration_worker(rank, world_size, port, steps, out_dir):
    """Tile 0 owns one sphere rolling in +x across the border at x = TILE_W; tile 1 must take it over."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from oracle import oracle
    lo, hi, origin = tiles.tile_bounds(rank, world_size, TILE_W, TILE_W)
    descs = scenes.ground()
    if rank == 0:
        b = scenes.dynamic_bodies(1)
        b["shape_type"] = abi.SHAPE_SPHERE
        b["shape"][0] = (0.5, 0, 0, 0)
        b["pos"][0] = (TILE_W - 3.0, 5.0, 0.5)
        b["lin_vel"][0] = (4.0, 0.0, 0.0)
        descs = np.concatenate([descs, b])
    w = oracle.OracleWorld(max_bodies=64)
    w.add_batch(descs)
    ex = ghost_exchange.GhostExchange(w, rank, world_size, lo, hi, margin=1.5, dist=dist, device=torch.device("cpu"), cap=64)
    emig = immig = 0
    for _ in range(steps):
        ex.exchange()
        emig += ex.last_emigrated
        immig += ex.last_immigrated
        w.step(DT)
    st = w.read_states(0, 8)
    np.save(os.path.join(out_dir, f"mig{rank}.npy"), st)
    np.save(os.path.join(out_dir, f"migcount{rank}.npy"), np.array([emig, immig, w.num_bodies()]))
    dist.barrier()
    dist.destroy_process_group()


def worker(rank, world_size, port, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from oracle import oracle
    descs, lo, hi = scene_for_tile(rank, world_size)
    w = oracle.OracleWorld(max_bodies=1024)
    w.add_batch(descs)
    ex = ghost_exchange.GhostExchange(w, rank, world_size, lo, hi, margin=1.5, dist=dist, device=torch.device("cpu"), cap=512)
    log = []
    for _ in range(steps):
        ex.exchange()
        w.step(DT)
        log.append((ex.last_exported, ex.last_imported))
    st = w.read_states(0, 256)
    np.save(os.path.join(out_dir, f"tile{rank}.npy"), st)
    np.save(os.path.join(out_dir, f"log{rank}.npy"), np.array(log))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_grid_and_bounds():
    # SURVEY 8(e) / north_star: 2x1x1, 2x2x1, 2x2x2
    assert tiles.tile_grid(1) == (1, 1, 1) and tiles.tile_grid(2) == (2, 1, 1) and tiles.tile_grid(4) == (2, 2, 1) and tiles.tile_grid(8) == (2, 2, 2)
    lo, hi, org = tiles.tile_bounds(1, 2, 150.0, 150.0)
    assert lo[0] == 150.0 and hi[0] > 1e8 and org[0] == 150.0 and lo[2] < -1e8 and hi[2] > 1e8
    lo, hi, org = tiles.tile_bounds(5, 8, 150.0, 150.0, grid=(4, 2, 1))     # flat side-by-side layout: ix=1, iy=1
    assert (lo[0], hi[0], lo[1]) == (150.0, 300.0, 150.0)
    # the 2x2x2 split of a box that starts at (-10, -10, 0): rank 7 = upper (+x, +y) octant, rank 1 = lower (+x, -y)
    lo, hi, org = tiles.tile_bounds(7, 8, 10.0, 10.0, 20.0, origin=(-10.0, -10.0, 0.0))
    assert tuple(lo) == (0.0, 0.0, 20.0) and all(hi > 1e8) and tuple(org) == (0.0, 0.0, 20.0)
    lo, hi, org = tiles.tile_bounds(1, 8, 10.0, 10.0, 20.0, origin=(-10.0, -10.0, 0.0))
    assert lo[0] == 0.0 and lo[1] < -1e8 and lo[2] < -1e8 and hi[0] > 1e8 and hi[1] == 0.0 and hi[2] == 20.0
    # every point belongs to exactly one tile
    rng = np.random.default_rng(0)
    pts = rng.uniform(-30, 50, size=(2000, 3)).astype(np.float32)
    owners = np.zeros(len(pts), int)
    for r in range(8):
        lo, hi, _ = tiles.tile_bounds(r, 8, 10.0, 10.0, 20.0, origin=(-10.0, -10.0, 0.0))
        owners += (np.all(pts >= lo, axis=1) & np.all(pts < hi, axis=1)).astype(int)
    assert np.all(owners == 1)


def test_select_ghosts_filters_by_region():
    recs = np.zeros(4, dtype=abi.ghost_dtype)
    recs["pos"] = [[149.5, 10, 1], [100, 10, 1], [151, 10, 1], [149.5, 400, 1]]
    lo, hi, _ = tiles.tile_bounds(1, 2, 150.0, 150.0)
    sel = tiles.select_ghosts(recs, lo, hi, margin=2.0)
    assert len(sel) == 3 and 100.0 not in sel["pos"][:, 0]


@pytest.mark.timeout(300)
def test_two_tiles_gloo_ghost_exchange(tmp_path, oracle):
    steps = 150
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(worker, args=(2, port, steps, str(tmp_path)), nprocs=2, join=True)
    t0, t1 = np.load(tmp_path / "tile0.npy"), np.load(tmp_path / "tile1.npy")
    l0, l1 = np.load(tmp_path / "log0.npy"), np.load(tmp_path / "log1.npy")
    # every step each tile exported its border bodies and imported the other tile's
    assert l0[:, 0].min() > 0 and l1[:, 0].min() > 0
    assert np.array_equal(l0[:, 1] > 0, l1[:, 0] > 0)
    # bodies 1..72 are the tile's own; ghosts (kinematic copies) sit above them in the id space
    own0, own1 = t0[1:73], t1[1:73]
    assert np.all(own0["id"] != abi.INVALID_ID) and np.all(own1["id"] != abi.INVALID_ID)
    # the piles settled on the ground on both sides of the border: nothing fell through or exploded
    for own in (own0, own1):
        assert own["pos"][:, 2].min() > 0.4 and own["pos"][:, 2].max() < 4.0
        assert np.abs(own["lin_vel"]).max() < 1.0
    # ghosts present in tile 0 mirror bodies tile 1 owns (same pose within one step of motion)
    ghosts0 = t0[73:]
    ghosts0 = ghosts0[ghosts0["id"] != abi.INVALID_ID]
    assert len(ghosts0) > 0
    dmin = np.min(np.linalg.norm(ghosts0["pos"][:, None, :] - own1["pos"][None, :, :], axis=2), axis=1)
    assert dmin.max() < 0.05
    # no deep interpenetration across the border: closest centre distance between the two tiles' own boxes
    dd = np.linalg.norm(own0["pos"][:, None, :] - own1["pos"][None, :, :], axis=2)
    assert dd.min() > 0.9


@pytest.mark.timeout(300)
def test_ownership_migrates_across_the_tile_border(tmp_path, oracle):
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(migration_worker, args=(2, port, 120, str(tmp_path)), nprocs=2, join=True)
    c0, c1 = np.load(tmp_path / "migcount0.npy"), np.load(tmp_path / "migcount1.npy")
    assert c0[0] == 1 and c1[1] == 1                    # emigrated once from tile 0, immigrated once into tile 1
    s1 = np.load(tmp_path / "mig1.npy")
    own = s1[(s1["id"] != abi.INVALID_ID)]
    ball = own[np.argmax(own["pos"][:, 0])]
    # it kept rolling: well past the border, still moving in +x, still on the ground
    assert ball["pos"][0] > TILE_W + 1.0 and ball["lin_vel"][0] > 1.0 and abs(ball["pos"][2] - 0.5) < 0.05
    # tile 0 now only holds its ground quad plus (at most) a ghost copy while the ball is still near the border
    assert c0[2] <= 2


def test_route_and_split_match_a_numpy_reference():
    """sgp_tiles_route / sgp_tiles_split (the C helpers behind GhostExchange) against the plain numpy statement of the same rules."""
    rng = np.random.default_rng(5)
    n_tiles, w = 8, 30.0
    boxes = np.array([np.concatenate(tiles.tile_bounds(r, n_tiles, w, w, grid=(4, 2, 1))[:2]) for r in range(n_tiles)], np.float32)
    for rank in (0, 5):
        lo, hi = boxes[rank, :3], boxes[rank, 3:]
        n = 3000
        recs = np.zeros(n, dtype=abi.ghost_dtype)
        recs["pos"] = rng.uniform(-10, 130, (n, 3)).astype(np.float32) * np.float32([1, 0.55, 0.05])
        recs["motion_type"] = rng.choice([abi.MOTION_DYNAMIC, abi.MOTION_KINEMATIC], n, p=[0.9, 0.1])
        recs["global_id"] = np.arange(n, dtype=np.uint64) * 3 + 1
        send, counts, emig = tiles.route(recs, rank, boxes, pad=3.5)
        inside_own = tiles.inside(recs, lo, hi)
        want_emig = ~inside_own & (recs["motion_type"] == abi.MOTION_DYNAMIC)
        assert np.array_equal(np.sort(emig), np.sort((recs["global_id"][want_emig] & np.uint64(0xFFFFFFFF)).astype(np.uint32)))
        off = 0
        for r in range(n_tiles):
            if r == rank:
                assert counts[r] == 0
                continue
            m = np.all(recs["pos"] >= boxes[r, :3] - np.float32(3.5), axis=1) & np.all(recs["pos"] < boxes[r, 3:] + np.float32(3.5), axis=1)
            part = send[off:off + counts[r]]
            off += counts[r]
            assert counts[r] == m.sum()
            assert np.array_equal(part["global_id"], recs["global_id"][m] | (np.uint64(rank) << np.uint64(40)))
            assert np.array_equal((part["motion_type"] & tiles.GHOST_TAKE_OWNERSHIP) != 0, want_emig[m])
            assert np.array_equal(part["pos"], recs["pos"][m])
        assert off == len(send)
        # receiving side, as tile 1 would see what `rank` sent it
        part = send[:counts[0]] if rank != 0 else send[:counts[1]]
        dst = 0 if rank != 0 else 1
        ghosts, immigrants = tiles.split(part, boxes[dst, :3], boxes[dst, 3:])
        take = (part["motion_type"] & tiles.GHOST_TAKE_OWNERSHIP) != 0
        assert np.array_equal(ghosts["global_id"], part["global_id"][~take])
        assert np.array_equal(immigrants["global_id"], part["global_id"][take & tiles.inside(part, boxes[dst, :3], boxes[dst, 3:])])
    # empty input
    send, counts, emig = tiles.route(np.zeros(0, dtype=abi.ghost_dtype), 0, boxes, pad=3.5)
    assert len(send) == 0 and counts.sum() == 0 and len(emig) == 0


def corner_worker(rank, world_size, port, steps, out_dir):
    """2 x 2 tiles; tile 0 owns a small pile sitting right at the common corner, so its bodies are ghosts in all three other tiles, and
    tile 3 owns a box that slides across the corner region into tile 0 (migration with four ranks)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from oracle import oracle
    lo, hi, origin = tiles.tile_bounds(rank, world_size, TILE_W, TILE_W)
    descs = scenes.ground()
    if rank == 0:
        b = scenes.dynamic_bodies(4)
        b["pos"] = np.float32([[TILE_W - 0.7, TILE_W - 0.7, 0.5], [TILE_W - 0.7, TILE_W - 1.9, 0.5], [TILE_W - 1.9, TILE_W - 0.7, 0.5], [TILE_W - 0.7, TILE_W - 0.7, 1.6]])
        descs = np.concatenate([descs, b])
    if rank == 3:
        b = scenes.dynamic_bodies(1)
        b["pos"][0] = (TILE_W + 3.0, TILE_W + 3.0, 0.5)
        b["lin_vel"][0] = (-3.0, -3.0, 0.0)
        b["friction"] = 0.0
        descs = np.concatenate([descs, b])
    w = oracle.OracleWorld(max_bodies=64)
    w.add_batch(descs)
    ex = ghost_exchange.GhostExchange(w, rank, world_size, lo, hi, margin=1.5, dist=dist, device=torch.device("cpu"), cap=64)
    log = []
    for _ in range(steps):
        ex.exchange()
        log.append((ex.last_exported, ex.last_sent, ex.last_imported, ex.last_emigrated, ex.last_immigrated))
        w.step(DT)
    np.save(os.path.join(out_dir, f"corner{rank}.npy"), np.array(log))
    np.save(os.path.join(out_dir, f"cornerstate{rank}.npy"), w.read_states(0, 32))
    np.save(os.path.join(out_dir, f"cornercount{rank}.npy"), np.array([w.num_bodies()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_four_tiles_corner_bodies_reach_all_neighbours(tmp_path, oracle):
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(corner_worker, args=(4, port, 90, str(tmp_path)), nprocs=4, join=True)
    logs = [np.load(tmp_path / f"corner{r}.npy") for r in range(4)]
    # step 0: tile 0 exports its 4 corner bodies; each goes to all three other tiles (12 records sent), which import 4 ghosts each
    # (+ tile 3's slider is not yet near anybody)
    assert logs[0][0, 0] == 4 and logs[0][0, 1] == 12
    for r in (1, 2, 3):
        assert logs[r][0, 2] == 4
    # the slider left tile 3 (it runs into the ghosts of the corner pile there and is deflected); every emigration was matched by
    # exactly one immigration somewhere, and no body was lost or duplicated: 5 dynamic bodies are owned in total at the end
    assert logs[3][:, 3].sum() >= 1
    assert sum(int(l[:, 3].sum()) for l in logs) == sum(int(l[:, 4].sum()) for l in logs)
    owned = sum(int(np.load(tmp_path / f"cornercount{r}.npy")[0]) - 1 - int(logs[r][-1, 2]) for r in range(4))
    assert owned == 5
    for r in range(4):
        st = np.load(tmp_path / f"cornerstate{r}.npy")
        own = st[st["id"] != abi.INVALID_ID]
        assert own["pos"][:, 2].min() > -0.6 and np.abs(own["lin_vel"]).max() < 8.0      # nothing fell through or exploded


def zsplit_worker(rank, world_size, port, steps, out_dir):
    """A z split (grid 1 x 1 x 2, the third axis of the 2x2x2 tiling of config 4): the upper tile owns a short column of boxes that
    falls through the z face onto a 2 x 2 base the lower tile owns."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from oracle import oracle
    lo, hi, origin = tiles.tile_bounds(rank, world_size, TILE_W, TILE_W, 3.0, grid=(1, 1, 2))
    descs = scenes.ground()
    if rank == 0:
        b = scenes.dynamic_bodies(4)
        b["pos"] = np.float32([[5.0, 5.0, 0.5], [6.02, 5.0, 0.5], [5.0, 6.02, 0.5], [6.02, 6.02, 0.5]])
    else:
        b = scenes.dynamic_bodies(3)
        b["pos"] = np.float32([[5.5, 5.5, 4.0], [5.52, 5.5, 5.5], [5.5, 5.52, 7.0]])
    descs = np.concatenate([descs, b])
    w = oracle.OracleWorld(max_bodies=64)
    w.add_batch(descs)
    ex = ghost_exchange.GhostExchange(w, rank, world_size, lo, hi, margin=2.0, dist=dist, device=torch.device("cpu"), cap=64)
    log = []
    for _ in range(steps):
        ex.exchange()
        log.append((ex.last_exported, ex.last_sent, ex.last_imported, ex.last_emigrated, ex.last_immigrated))
        w.step(DT)
    st = w.read_states(0, 32)
    np.save(os.path.join(out_dir, f"z{rank}.npy"), st)
    np.save(os.path.join(out_dir, f"zlog{rank}.npy"), np.array(log))
    np.save(os.path.join(out_dir, f"zcount{rank}.npy"), np.array([w.num_bodies(), ex.last_imported]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_tiles_split_in_z_bodies_fall_through_the_face(tmp_path, oracle):
    lo, hi, org = tiles.tile_bounds(1, 2, TILE_W, TILE_W, 3.0, grid=(1, 1, 2))
    assert lo[2] == 3.0 and hi[2] > 1e8 and lo[0] < -1e8 and org[2] == 3.0
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(zsplit_worker, args=(2, port, 200, str(tmp_path)), nprocs=2, join=True)
    l0, l1 = np.load(tmp_path / "zlog0.npy"), np.load(tmp_path / "zlog1.npy")
    c0, c1 = np.load(tmp_path / "zcount0.npy"), np.load(tmp_path / "zcount1.npy")
    # the boxes of the upper tile that came to rest below the z = 3 face emigrated through it, and the lower tile took every one of them.
    # (With the body-pair contact cache the column stands: its top box rests at z = 3.5 and stays with the upper tile; without it the
    # column used to sway apart and all three ended up below the face.)
    k = int(l1[:, 3].sum())
    assert k in (2, 3) and l0[:, 4].sum() == k and l0[:, 3].sum() == 0
    # before it crossed, the lowest falling box was a ghost in the lower tile (imported across the z face)
    assert l0[:, 2].max() >= 1
    s0, s1 = np.load(tmp_path / "z0.npy"), np.load(tmp_path / "z1.npy")
    assert c0[0] - c0[1] == 1 + 4 + k                    # ground + 4 base boxes + the immigrants (ghost copies excluded)
    assert c1[0] - c1[1] == 1 + (3 - k)                  # the upper tile keeps its ground quad and what still rests above the face
    own0 = s0[s0["id"] != abi.INVALID_ID][1:]
    dyn = own0[np.argsort(own0["pos"][:, 2])][:4 + k]
    # the column came to rest on the base: nothing fell through, nothing is still moving fast
    assert dyn["pos"][:, 2].min() > 0.45 and dyn["pos"][:, 2].max() < 4.2
    assert np.abs(dyn["lin_vel"]).max() < 4.0 and np.all(np.isfinite(dyn["pos"]))     # (the boxes that tumbled off the column -- from 3 m up -- may still be rolling away: 1 .. 2.6 m/s, whichever way the contact cache treats sleeping pairs)


def test_sensor_ghost_is_a_sensor_next_door():
    """Oracle tiles, one process: a sensor box and a non-collidable box of tile 0 overlap a resting box of tile 1 across the border.  Their ghosts
    carry the layer and the sensor flag of the originals (sgp_ghost_record.flags), so the resting box is not pushed."""
    from oracle import oracle
    TILE_W = 12.0
    boxes = np.array([np.concatenate(tiles.tile_bounds(r, 2, TILE_W, TILE_W)[:2]) for r in range(2)], np.float32)
    a = scenes.dynamic_bodies(2)
    a["shape_type"] = abi.SHAPE_BOX; a["shape"][:, :3] = 0.5
    a["pos"][0] = (TILE_W - 0.1, 6.0, 0.5); a["is_sensor"][0] = 1
    a["pos"][1] = (TILE_W - 0.2, 6.0, 1.6); a["layer"][1] = abi.LAYER_MOVING_NON_COLLIDABLE
    a["gravity_factor"][:] = 0.0
    b = scenes.dynamic_bodies(1)
    b["shape_type"] = abi.SHAPE_BOX; b["shape"][0] = (0.5, 0.5, 0.5, 0); b["pos"][0] = (TILE_W + 0.7, 6.0, 0.5)
    ws = [oracle.OracleWorld(max_bodies=64) for _ in range(2)]
    ws[0].add_batch(np.concatenate([scenes.ground(), a])); ws[1].add_batch(np.concatenate([scenes.ground(), b]))
    for _ in range(60):
        ghost_exchange.exchange_in_process(ws, boxes, 1.5)
        for w in ws:
            w.step(1.0 / 60.0)
    assert ws[1].num_bodies() == 4                       # ground + box + two ghosts
    rest = ws[1].get_state([1])[0]
    assert abs(float(rest["pos"][0]) - (TILE_W + 0.7)) < 1e-3 and abs(float(rest["pos"][1]) - 6.0) < 1e-3
    for w in ws:
        w.close()
