"""REAL device worlds in two processes driven through the host statement of the exchange rules (tests/ghost_exchange.py GhostExchange over
torch.distributed / gloo -- test infrastructure; the product's exchange is sgp_tiles_*, covered by tests/test_tiles_parity_gpu.py and
test_config4_gpu.py): both ranks put their tile on cuda:0 (one GPU is all a test box has), and every rank also runs the oracle on the same
tile through a second GhostExchange; after every 20 steps the device tile must equal the oracle tile bit for bit.  What this adds to the
in-process tests: the worlds of a tile world really live in different processes (export / import / migration through the C ABI only)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu

TILE_W = 12.0
DT = 1.0 / 60.0


def worker(rank, world_size, port, steps, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    from substrata_amd import abi, scenes, tiles
    from substrata_amd.lib import World, init
    from oracle import oracle
    import parity
    import ghost_exchange
    init()
    lo, hi, origin = tiles.tile_bounds(rank, world_size, TILE_W, TILE_W)
    d, _ = scenes.lattice(6, 6, 3, 1.9, 0.6, seed=41 + rank, jitter=0.08, random_rot=True, origin_centered=False)
    d["pos"][:, 0] += origin[0] + 1.2; d["pos"][:, 1] += origin[1] + 1.0
    d["shape_type"] = np.arange(len(d)) % 3
    d["shape"][:, :3] = 0.45
    d["shape"][np.arange(len(d)) % 3 == 2, 1] = 0.55
    d["lin_vel"][:, 0] = 3.0 if rank == 0 else -3.0            # the two piles run into each other across the border
    descs = np.concatenate([scenes.ground(), d])
    g = World(max_bodies=1024, device=0); c = oracle.OracleWorld(max_bodies=1024)
    g.add_batch(descs); c.add_batch(descs)
    exg = ghost_exchange.GhostExchange(g, rank, world_size, lo, hi, margin=1.5, dist=dist, device=torch.device("cpu"), cap=256)
    exc = ghost_exchange.GhostExchange(c, rank, world_size, lo, hi, margin=1.5, dist=dist, device=torch.device("cpu"), cap=256)
    ok = True; migrated = 0; imported = 0
    for s in range(1, steps + 1):
        exg.exchange(); exc.exchange()
        ok = ok and (exg.last_exported, exg.last_sent, exg.last_imported, exg.last_emigrated, exg.last_immigrated) == \
                    (exc.last_exported, exc.last_sent, exc.last_imported, exc.last_emigrated, exc.last_immigrated)
        migrated += exg.last_immigrated; imported = max(imported, exg.last_imported)
        g.step(DT); c.step(DT)
        if s % 20 == 0:
            dd = parity.state_diff(g.read_states(0, 1024), c.read_states(0, 1024))
            ok = ok and dd["bit_exact"] and dd["active_mismatch"] == 0
    np.save(os.path.join(out_dir, f"two{rank}.npy"), np.array([int(ok), migrated, imported, g.num_bodies()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_processes_device_tiles_against_oracle_tiles(tmp_path, oracle):
    import torch.multiprocessing as mp
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(worker, args=(2, port, 180, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "two0.npy"), np.load(tmp_path / "two1.npy")
    assert r0[0] == 1 and r1[0] == 1, "a device tile left its oracle twin, or the exchanges disagreed"
    assert r0[2] > 0 and r1[2] > 0                       # ghosts flowed both ways
    assert r0[1] + r1[1] >= 1                            # some body changed owner
