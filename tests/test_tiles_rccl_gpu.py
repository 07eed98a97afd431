"""The tile exchange on the RCCL path (torch.distributed backend "nccl"), as far as one GPU allows: a one-rank process group
runs the same two all-gathers (counts, padded records) on device tensors that the N-rank bench runs over xGMI.  The N > 1 logic
(ghost selection, migration) is covered on CPU by tests/test_tiles_gloo.py."""
import os

import numpy as np
import pytest

from substrata_amd import scenes, tiles
from helpers import DT

pytestmark = pytest.mark.gpu


def test_exchange_collectives_on_rccl_single_rank():
    import torch
    import torch.distributed as dist
    from substrata_amd.lib import World
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        descs = scenes.config3_100k_mixed(20, 20, 4, seed=9)
        w = World(max_bodies=len(descs) + 64)
        w.add_batch(descs)
        # a finite tile inside the scene so that boundary records exist
        lo = np.array([-8.0, -8.0, -1e9], np.float32); hi = np.array([8.0, 8.0, 1e9], np.float32)
        ex = tiles.GhostExchange(w, 0, 1, lo, hi, margin=2.0, dist=dist, device=torch.device("cuda", 0))
        for _ in range(5):
            ex.exchange()
            w.step(DT)
            dist.barrier()
        torch.cuda.synchronize()
        assert ex.last_exported > 50 and ex.last_imported == 0          # nobody else to import from
        assert int(ex.cnt_recv[0].item()) == ex.last_exported
        # the padded records came back through the collective unchanged
        rec = w.export_boundary(lo, hi, 2.0)
        assert len(rec) > 50
        w.close()
    finally:
        dist.destroy_process_group()
