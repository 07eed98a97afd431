"""The tile exchange on the RCCL path (torch.distributed backend "nccl"), as far as one GPU allows: a one-rank process group
runs the same collectives (all-gather of the count matrix, all-to-all-v of the records) on device tensors that the N-rank bench runs
over xGMI.  The N > 1 logic (routing, ghost selection, migration) is covered on CPU by tests/test_tiles_gloo.py (2 and 4 ranks)."""
import os

import numpy as np
import pytest

from substrata_amd import scenes, tiles
from helpers import DT
import ghost_exchange

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
        ex = ghost_exchange.GhostExchange(w, 0, 1, lo, hi, margin=2.0, dist=dist, device=torch.device("cuda", 0))
        for _ in range(5):
            ex.exchange()
            w.step(DT)
            dist.barrier()
        torch.cuda.synchronize()
        assert ex.last_exported > 50 and ex.last_imported == 0 and ex.last_sent == 0      # nobody else to send to or import from
        assert int(ex.cnt_recv.sum().item()) == 0
        rec = w.export_boundary(lo, hi, 2.0)
        assert len(rec) > 50
        # the all-to-all-v itself, with a non-empty payload to self, on the device buffers the exchange uses
        n = 64 * tiles.REC
        ex.send_dev[:n].copy_(torch.arange(n, dtype=torch.int64, device=ex.send_dev.device).to(torch.uint8))
        dist.all_to_all_single(ex.recv_dev[:n], ex.send_dev[:n], output_split_sizes=[n], input_split_sizes=[n])
        torch.cuda.synchronize()
        assert torch.equal(ex.recv_dev[:n], ex.send_dev[:n])
        w.close()
    finally:
        dist.destroy_process_group()
