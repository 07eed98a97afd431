"""sgp_tiles_exchange ITSELF with more than one rank: 2 and 4 OS processes, one tile each, all on cuda:0, the product's own
routing kernels -> all-gather of the rows -> grouped send / receive -> import, carried between the processes by the test-only collective library
tests/rccl_standin (bound through SGP_RCCL_LIBRARY; real RCCL wants one GPU per rank, a test box has one).  Every rank also steps the ORACLE world
of its tile, exchanged by the host statement of the rules over gloo (tests/ghost_exchange.py) with the regions the device computed; the device
tile must report the same counts every step and equal its oracle twin bit for bit, through migration and re-tiling.

And the failure path: a rank whose buffer growth fails after the gather must not leave its peers waiting -- every rank returns an error from
that exchange (the failing one its own, the others SGP_ERR_PEER), and the next exchange works again.

Nothing timed through the stand-in is a scaling number ("transport": "test stand-in")."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

DT = 1.0 / 60.0
STANDIN_DIR = os.path.join(ROOT, "tests", "rccl_standin")
STANDIN_LIB = os.path.join(STANDIN_DIR, "librccl_standin.so")


def build_standin():
    src = os.path.join(STANDIN_DIR, "rccl_standin.cpp")
    if os.path.exists(STANDIN_LIB) and os.path.getmtime(STANDIN_LIB) >= os.path.getmtime(src):
        return STANDIN_LIB
    subprocess.run(["g++", "-shared", "-fPIC", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src,
                    "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", STANDIN_LIB], check=True)
    return STANDIN_LIB


def _setup(rank, world_size, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SGP_RCCL_LIBRARY"] = STANDIN_LIB
    os.environ.setdefault("SGP_RCCL_STANDIN_TIMEOUT_S", "60")
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    return dist


def _unique_id(dist, rank):
    import torch
    from substrata_amd import tiles
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        uid.copy_(torch.frombuffer(bytearray(tiles.NativeTiles.unique_id()), dtype=torch.uint8))
    dist.broadcast(uid, src=0)
    return bytes(uid.numpy().tobytes())


def parity_worker(rank, world_size, port, steps, out_dir):
    import torch
    dist = _setup(rank, world_size, port)
    from substrata_amd import scenes, tiles
    from substrata_amd.lib import World, init
    from oracle import oracle
    import parity
    import ghost_exchange
    init()
    n = 8                                                   # the 8^3 scaled-down tower of BASELINE config 4
    grid = tiles.tile_grid(world_size)
    boxes = []
    for r in range(world_size):
        d_r, lo, hi = scenes.config4_tile_descs(r, world_size, n=n)
        boxes.append(np.concatenate([lo, hi]))
        if r == rank:
            descs = d_r
    boxes = np.array(boxes, np.float32)
    cap = 2048
    g = World(max_bodies=cap, device=0); c = oracle.OracleWorld(max_bodies=cap)
    assert np.array_equal(g.add_batch(descs), c.add_batch(descs))
    margin = 2.0
    nt = tiles.NativeTiles(g, rank, world_size, boxes, margin, unique_id=_unique_id(dist, rank))
    exc = ghost_exchange.GhostExchange(c, rank, world_size, boxes[rank, :3], boxes[rank, 3:], margin=margin, dist=dist, device=torch.device("cpu"), cap=4096)
    ok, why = True, ""
    migrated_in = migrated_out = 0
    max_ghosts = 0
    rebalances = 0
    for s in range(steps):
        if s % 12 == 0:                                     # re-tiling: the device decides (collective over the stand-in), the oracle tiles follow
            nt.rebalance(grid, by_contacts=(s // 12) % 2 == 1)
            rebalances += 1
            b = nt.boxes()
            exc.boxes = b.copy(); exc.lo = b[rank, :3].copy(); exc.hi = b[rank, 3:].copy()
        nt.exchange(); exc.exchange()
        st = nt.stats()
        if (st.exported, st.emigrated, st.ghosts, st.immigrated) != (exc.last_sent, exc.last_emigrated, exc.last_imported, exc.last_immigrated):
            ok = False; why = why or f"step {s}: counts device {(st.exported, st.emigrated, st.ghosts, st.immigrated)} oracle {(exc.last_sent, exc.last_emigrated, exc.last_imported, exc.last_immigrated)}"
        migrated_in += st.immigrated; migrated_out += st.emigrated; max_ghosts = max(max_ghosts, st.ghosts)
        g.step(DT); c.step(DT)
        if s % 30 == 29:
            dd = parity.state_diff(g.read_states(0, cap), c.read_states(0, cap))
            if not (dd["bit_exact"] and dd["active_mismatch"] == 0):
                ok = False; why = why or f"step {s}: {dd}"
    st = nt.stats()
    owned = g.num_bodies() - 1 - st.ghosts
    np.save(os.path.join(out_dir, f"mr{rank}.npy"), np.array([int(ok), migrated_in, migrated_out, max_ghosts, owned, st.comm_ranks, st.rebalances, rebalances, st.fast_imports, st.slow_imports, st.device_creates]))
    if why:
        open(os.path.join(out_dir, f"mr{rank}.txt"), "w").write(why)
    dist.barrier()
    nt.close(); g.close(); c.close()
    dist.destroy_process_group()


def _run(worker, world_size, args, tmp_path, salt):
    import torch.multiprocessing as mp
    build_standin()
    port = 36100 + salt + (os.getpid() % 1500)
    mp.spawn(worker, args=(world_size, port) + args + (str(tmp_path),), nprocs=world_size, join=True)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world_size", [2, 4])
def test_native_exchange_across_processes_against_oracle_tiles(tmp_path, oracle, world_size):
    _run(parity_worker, world_size, (180,), tmp_path, 10 * world_size)
    res = [np.load(tmp_path / f"mr{r}.npy") for r in range(world_size)]
    for r, v in enumerate(res):
        note = (tmp_path / f"mr{r}.txt").read_text() if (tmp_path / f"mr{r}.txt").exists() else ""
        assert v[0] == 1, f"rank {r}: {note}"
        assert v[5] == world_size                          # the communicator really had one rank per tile
        assert v[6] == v[7] > 0                            # every re-tiling went through the collective
    assert sum(int(v[4]) for v in res) == 8 ** 3           # no body lost or duplicated across the processes
    assert all(int(v[9]) == 0 and int(v[10]) > 0 for v in res)      # round 6: no import brought records to the host; newcomers were created on the device
    assert all(v[3] > 0 for v in res)                      # ghosts flowed to every tile
    assert sum(int(v[1]) for v in res) == sum(int(v[2]) for v in res) >= 1       # bodies changed owner, and every emigrant arrived somewhere


def failure_worker(rank, world_size, port, out_dir):
    import torch
    if rank == 1:
        os.environ["SGP_TILES_TEST_FAIL_RANK"] = "1"      # this rank's first buffer growth fails (and it asks for one)
    dist = _setup(rank, world_size, port)
    from substrata_amd import scenes, tiles
    from substrata_amd.lib import World, init
    init()
    boxes = []
    for r in range(world_size):
        d_r, lo, hi = scenes.config4_tile_descs(r, world_size, n=6)
        boxes.append(np.concatenate([lo, hi]))
        if r == rank:
            descs = d_r
    g = World(max_bodies=1024, device=0)
    g.add_batch(descs)
    nt = tiles.NativeTiles(g, rank, world_size, np.array(boxes, np.float32), 2.0, unique_id=_unique_id(dist, rank))
    import time
    t0 = time.perf_counter()
    first = ""
    try:
        nt.exchange()
        first = "no error"
    except RuntimeError as e:
        first = str(e)
    took = time.perf_counter() - t0
    # the communicator is still usable: the next exchange (no forced failure any more) goes through on every rank
    second = "ok"
    try:
        nt.exchange(); g.step(DT); nt.exchange()
    except RuntimeError as e:
        second = str(e)
    open(os.path.join(out_dir, f"fail{rank}.txt"), "w").write(f"{took:.3f}\n{first}\n{second}\n{nt.stats().ghosts}\n")
    dist.barrier()
    nt.close(); g.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_a_rank_that_fails_after_the_gather_releases_its_peers(tmp_path):
    _run(failure_worker, 2, (), tmp_path, 77)
    out = [(tmp_path / f"fail{r}.txt").read_text().splitlines() for r in range(2)]
    for r in range(2):
        assert float(out[r][0]) < 20.0, out[r]                      # nobody waited for a message that never came (the stand-in's timeout is 60 s)
    assert "(-4)" in out[1][1] and "forced by SGP_TILES_TEST_FAIL_RANK" in out[1][1], out[1]      # the failing rank: its own error
    assert "(-7)" in out[0][1] and "another rank" in out[0][1], out[0]                             # its peer: SGP_ERR_PEER, not a hang
    assert out[0][2] == "ok" and out[1][2] == "ok", out                                            # and the exchange works again afterwards
    assert int(out[0][3]) > 0 and int(out[1][3]) > 0
