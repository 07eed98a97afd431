"""Host statement of the tile exchange rules (test infrastructure, NOT part of the product: the product's exchange is sgp_tiles_* inside
libsgp.so, substrata_amd/tiles.py NativeTiles).

GhostExchange: export -> route -> all-gather of counts -> all-to-all-v of records over torch.distributed -> split -> import / immigrate,
for any world object with the export_boundary / import_ghosts / add_batch / remove methods -- i.e. for the ORACLE worlds that the parity
tests step next to the device tiles, and for the world_size-2 / -4 gloo tests of the N > 1 path on CPU.
exchange_in_process: the same for N tile worlds living in one process."""
import numpy as np

from substrata_amd import abi
from substrata_amd.tiles import REC, route, split, records_to_descs  # noqa: F401


def exchange_in_process(worlds, boxes, margin, log=None, radius_pad=1.5):
    """What GhostExchange does across ranks, for N tile worlds living in ONE process (tests, tools/fuzz_tiles.py, and a single-GPU
    dry run of a multi-tile world): export -> route -> [hand over] -> split -> import / immigrate, in rank order."""
    n = len(worlds)
    sent = []
    for r, w in enumerate(worlds):
        recs = w.export_boundary(boxes[r, :3], boxes[r, 3:], margin, cap=max(1 << 14, 4 * w.num_bodies()))
        send, counts, emig = route(recs, r, boxes, margin + radius_pad)
        for i in emig:
            w.remove(int(i))
        off = [0] + [int(x) for x in np.cumsum(counts)]
        sent.append([send[off[d]:off[d + 1]] for d in range(n)])
        if log is not None:
            log.append(("export", r, len(recs), [int(c) for c in counts], len(emig)))
    for r, w in enumerate(worlds):
        arrived = np.concatenate([sent[src][r] for src in range(n)]) if n > 1 else sent[0][0][:0]
        ghosts, immigrants = split(arrived, boxes[r, :3], boxes[r, 3:])
        w.import_ghosts(ghosts)
        if len(immigrants):
            w.add_batch(records_to_descs(immigrants))
        if log is not None:
            log.append(("import", r, len(ghosts), len(immigrants)))


class GhostExchange:
    """Per step: one small all-gather (every rank's per-destination record counts) and one all-to-all-v of the ghost records, each
    record travelling only to the tiles whose region (grown by margin + radius_pad) contains it.  With RCCL the all-to-all-v is the
    grouped ncclSend/ncclRecv exchange over xGMI; the CPU tests run the same code over gloo."""

    def __init__(self, world, rank, n_tiles, lo, hi, margin, dist=None, device=None, cap=1 << 16, radius_pad=1.5):
        self.world, self.rank, self.n = world, rank, n_tiles
        self.lo, self.hi, self.margin = np.asarray(lo, np.float32), np.asarray(hi, np.float32), float(margin)
        self.pad = float(margin) + float(radius_pad)
        self.dist, self.device, self.cap = dist, device, cap
        self.last_exported = 0
        self.last_sent = 0
        self.last_imported = 0
        self.last_emigrated = 0
        self.last_immigrated = 0
        self.boxes = np.concatenate([self.lo, self.hi])[None, :].astype(np.float32)
        if dist is not None:
            import torch
            self.torch = torch
            on_gpu = device is not None and torch.device(device).type == "cuda"
            self.on_gpu = on_gpu
            # every tile's region, once
            mine = torch.from_numpy(self.boxes[0].copy()).to(device)
            allb = torch.zeros(n_tiles * 6, dtype=torch.float32, device=device)
            dist.all_gather_into_tensor(allb, mine)
            self.boxes = allb.cpu().numpy().reshape(n_tiles, 6)
            self.cnt_send = torch.zeros(n_tiles, dtype=torch.int64, device=device)
            self.cnt_recv = torch.zeros(n_tiles * n_tiles, dtype=torch.int64, device=device)
            self._grow(cap)

    def _grow(self, cap):
        torch = self.torch
        self.cap = cap
        self.send_host = torch.zeros(cap * REC, dtype=torch.uint8, pin_memory=self.on_gpu)
        self.recv_host = torch.zeros(cap * REC, dtype=torch.uint8, pin_memory=self.on_gpu)
        self.send_dev = torch.zeros(cap * REC, dtype=torch.uint8, device=self.device) if self.on_gpu else self.send_host
        self.recv_dev = torch.zeros(cap * REC, dtype=torch.uint8, device=self.device) if self.on_gpu else self.recv_host

    def exchange(self):
        cap = max(1 << 14, 2 * self.last_exported)
        recs = self.world.export_boundary(self.lo, self.hi, self.margin, cap=cap)
        if len(recs) == cap:      # more boundary bodies than expected: ask again with room for every body
            recs = self.world.export_boundary(self.lo, self.hi, self.margin, cap=1 << 22)
        self.last_exported = len(recs)
        if self.dist is None or self.n == 1:
            # a single tile: nobody to talk to (the collectives still run when a process group is given, so that the one-rank RCCL
            # test exercises them)
            if self.dist is None:
                self.world.import_ghosts(recs[:0])
                self.last_imported = self.last_emigrated = self.last_immigrated = self.last_sent = 0
                return
        send, counts, emigrants = route(recs, self.rank, self.boxes, self.pad)
        # owned DYNAMIC bodies whose centre has left the tile emigrate: removed here, re-created by the tile that contains them
        for i in emigrants:
            self.world.remove(int(i))
        self.last_emigrated = len(emigrants)
        self.last_sent = len(send)
        torch = self.torch
        self.cnt_send.copy_(torch.from_numpy(counts.astype(np.int64)))
        pending = self.dist.all_gather_into_tensor(self.cnt_recv, self.cnt_send, async_op=True)
        # while the counts travel: stage this tile's records for the all-to-all-v
        n_send = int(counts.sum())
        if n_send > self.cap:
            self._grow(2 * n_send)
        if n_send:
            self.send_host[:n_send * REC].copy_(torch.from_numpy(send.view(np.uint8).reshape(-1)))
            if self.on_gpu:
                self.send_dev[:n_send * REC].copy_(self.send_host[:n_send * REC], non_blocking=True)
        pending.wait()
        matrix = self.cnt_recv.cpu().numpy().reshape(self.n, self.n)          # [source][destination]
        recv_counts = matrix[:, self.rank]
        n_recv = int(recv_counts.sum())
        if n_recv > self.cap:
            keep = self.send_dev[:n_send * REC].clone() if n_send else None
            self._grow(2 * max(n_send, n_recv))
            if n_send:
                self.send_dev[:n_send * REC].copy_(keep)
        if int(matrix.sum()) == 0:
            self.last_imported = self.last_immigrated = 0
            self.world.import_ghosts(recs[:0])
            return
        self.dist.all_to_all_single(self.recv_dev[:n_recv * REC], self.send_dev[:n_send * REC],
                                    output_split_sizes=[int(c) * REC for c in recv_counts],
                                    input_split_sizes=[int(c) * REC for c in counts])
        if self.on_gpu:
            self.recv_host[:n_recv * REC].copy_(self.recv_dev[:n_recv * REC])
        arrived = np.frombuffer(self.recv_host[:n_recv * REC].numpy(), dtype=abi.ghost_dtype) if n_recv else recs[:0]
        ghosts, immigrants = split(arrived, self.lo, self.hi)
        self.last_imported = len(ghosts)
        self.last_immigrated = len(immigrants)
        self.world.import_ghosts(ghosts)
        if len(immigrants):
            self.world.add_batch(records_to_descs(immigrants))
