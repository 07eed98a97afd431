"""Multi-GPU spatial tiles: one process per GPU, one sgp world per tile.

The path shards by space.  Each rank OWNS the bodies created in its tile and simulates them dynamically; bodies whose
AABB (inflated by the ghost margin) pokes out of the owner's tile are exported once per sub-step and imported by every
rank whose tile (inflated by the margin) they touch, where they are simulated as velocity-driven infinite-mass ghosts
(kinematic bodies) for that step.  The only collectives are, per step, one all-gather of the per-destination record counts (8 B x
tiles per rank) and one all-to-all-v of the ghost records -- each record goes only to the tiles it can touch -- (torch.distributed:
backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  Ghost traffic is ~1e3-1e4 records x 96 B per rank, so the
exchange is latency- not bandwidth-bound; the per-record routing is done by two C helpers (sgp_tiles_route / sgp_tiles_split).  Ownership migrates: when the centre of an owned body has left the tile, the owner removes it and the tile
that now contains the centre re-creates it as a dynamic body from the same record (its contact-cache entries restart).
"""
import ctypes as C

import numpy as np

from . import abi

REC = abi.ghost_dtype.itemsize


def tile_grid(n_tiles):
    """3-D tiling of the world box (north_star): 1 -> 1x1x1, 2 -> 2x1x1, 4 -> 2x2x1, 8 -> 2x2x2; beyond that the
    axes keep doubling in turn (x, y, z)."""
    g = [1, 1, 1]
    axis = 0
    n = 1
    while n < n_tiles:
        g[axis] *= 2
        n *= 2
        axis = (axis + 1) % 3
    assert n == n_tiles, "tile count must be a power of two"
    return tuple(g)


def tile_coords(rank, grid):
    """(ix, iy, iz) of tile `rank`: x fastest."""
    tx, ty, tz = grid
    return rank % tx, (rank // tx) % ty, rank // (tx * ty)


def tile_bounds(rank, n_tiles, tile_w, tile_d, tile_h=None, grid=None, origin=(0.0, 0.0, 0.0)):
    """Axis-aligned region [lo, hi) of tile `rank` in a grid of tile_w x tile_d x tile_h tiles starting at `origin`; the outer
    faces of the grid are unbounded (a body that leaves the world box stays with the nearest tile).  tile_h = None: the grid must
    be flat (one tile in z, unbounded in z) -- pass grid=(tx, ty, 1) for a side-by-side layout of more than 4 tiles.
    Returns (lo, hi, corner of the tile)."""
    tx, ty, tz = grid if grid is not None else tile_grid(n_tiles)
    assert tx * ty * tz == n_tiles
    assert tile_h is not None or tz == 1, "a z split needs a tile height"
    ix, iy, iz = tile_coords(rank, (tx, ty, tz))
    big = 1.0e9
    ox, oy, oz = (float(v) for v in origin)
    th = float(tile_h) if tile_h is not None else 0.0
    lo = np.array([ox + ix * tile_w if ix > 0 else -big, oy + iy * tile_d if iy > 0 else -big, oz + iz * th if iz > 0 else -big], dtype=np.float32)
    hi = np.array([ox + (ix + 1) * tile_w if ix < tx - 1 else big, oy + (iy + 1) * tile_d if iy < ty - 1 else big,
                   oz + (iz + 1) * th if iz < tz - 1 else big], dtype=np.float32)
    corner = np.array([ox + ix * tile_w, oy + iy * tile_d, oz + iz * th], dtype=np.float32)
    return lo, hi, corner


def inside(recs, lo, hi):
    """Mask of records whose centre lies in [lo, hi)."""
    if len(recs) == 0:
        return np.zeros(0, dtype=bool)
    p = recs["pos"]
    return np.all(p >= lo, axis=1) & np.all(p < hi, axis=1)


def records_to_descs(recs):
    """Body descs for immigrants: the body exactly as its previous owner described it (user data, layer, sensor / sleeping / drag flags,
    damping, gravity factor travel in the record), dynamic and awake."""
    d = np.zeros(len(recs), dtype=abi.body_desc_dtype)
    for f in ("pos", "rot", "lin_vel", "ang_vel", "shape_type", "shape", "mass", "friction", "restitution", "userdata", "gravity_factor"):
        d[f] = recs[f]
    d["linear_damping"] = recs["linear_damping"]
    d["angular_damping"] = recs["angular_damping"]
    fl = recs["flags"]
    d["motion_type"] = abi.MOTION_DYNAMIC
    d["layer"] = fl & abi.GHOST_FLAG_LAYER_MASK
    d["is_sensor"] = (fl & abi.GHOST_FLAG_SENSOR) != 0
    d["allow_sleeping"] = (fl & abi.GHOST_FLAG_ALLOW_SLEEP) != 0
    d["use_zero_linear_drag"] = (fl & abi.GHOST_FLAG_ZERO_DRAG) != 0
    d["activate"] = 1
    return d


def select_ghosts(recs, lo, hi, margin, radius_pad=1.5):
    """Records (from other ranks) that can touch the region [lo - margin, hi + margin)."""
    if len(recs) == 0:
        return recs
    p = recs["pos"]
    pad = margin + radius_pad
    m = np.all(p >= (lo - pad), axis=1) & np.all(p < (hi + pad), axis=1)
    return recs[m]


GHOST_TAKE_OWNERSHIP = 0x100       # sgp.h SGP_GHOST_TAKE_OWNERSHIP


def _routing_lib():
    """The host-side routing helpers (sgp_tiles_route / sgp_tiles_split) live in libsgp.so; they touch neither a world nor the device."""
    from .lib import load
    return load()


def route(recs, rank, boxes, pad, cap=None):
    """Group one tile's exported records by destination tile (sgp_tiles_route).  Returns (send records grouped in rank order,
    per-destination counts, local ids of the emigrants)."""
    n_tiles = len(boxes)
    cap = max(64, 3 * len(recs)) if cap is None else cap
    while True:
        send = np.empty(cap, dtype=abi.ghost_dtype)
        counts = np.zeros(n_tiles, dtype=np.uint32)
        emig = np.empty(max(16, len(recs)), dtype=np.uint32)
        n_emig = C.c_uint32(0)
        recs = np.ascontiguousarray(recs)
        boxes32 = np.ascontiguousarray(boxes, dtype=np.float32)
        rc = _routing_lib().sgp_tiles_route(recs.ctypes.data, len(recs), int(rank), boxes32.ctypes.data, n_tiles, float(pad),
                                            send.ctypes.data, cap, counts.ctypes.data, emig.ctypes.data, len(emig), C.byref(n_emig))
        if rc == abi.ERR_CAPACITY and cap < 64 * max(64, len(recs)):
            cap *= 4
            continue
        if rc != 0:
            raise RuntimeError(f"sgp_tiles_route failed ({rc})")
        return send[:int(counts.sum())], counts, emig[:n_emig.value].copy()


def split(recs, lo, hi):
    """What arrived at a tile -> (ghosts, immigrants) (sgp_tiles_split)."""
    n = len(recs)
    ghosts = np.empty(max(n, 1), dtype=abi.ghost_dtype)
    immigrants = np.empty(max(n, 1), dtype=abi.ghost_dtype)
    ng, ni = C.c_uint32(0), C.c_uint32(0)
    recs = np.ascontiguousarray(recs)
    lo32, hi32 = np.ascontiguousarray(lo, dtype=np.float32), np.ascontiguousarray(hi, dtype=np.float32)
    rc = _routing_lib().sgp_tiles_split(recs.ctypes.data, n, lo32.ctypes.data, hi32.ctypes.data, ghosts.ctypes.data, C.byref(ng),
                                        immigrants.ctypes.data, C.byref(ni))
    if rc != 0:
        raise RuntimeError(f"sgp_tiles_split failed ({rc})")
    return ghosts[:ng.value], immigrants[:ni.value]


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


class NativeTiles:
    """The exchange below the C ABI (sgp_tiles_*): routing on the device, counts all-gathered and records sent device to device over RCCL
    from inside libsgp.so, import on the device while the ghost set is unchanged.  Python only hands over the communicator's unique id.

    One tile per process:   t = NativeTiles(world, rank, n, boxes, margin, unique_id=<128 bytes from rank 0>);  t.exchange() each step
    All tiles in a process: ts = [NativeTiles(w_r, r, n, boxes, margin) ...];  NativeTiles.exchange_group(ts)"""

    def __init__(self, world, rank, n_tiles, boxes, margin, radius_pad=1.5, unique_id=None):
        self.world, self.rank, self.n = world, rank, n_tiles
        self._lib = world._lib
        self._h = C.c_void_p()
        boxes32 = np.ascontiguousarray(boxes, dtype=np.float32).reshape(n_tiles, 6)
        uid = None
        if unique_id is not None:
            uid = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        rc = self._lib.sgp_tiles_create(world._h, int(rank), int(n_tiles), boxes32.ctypes.data, float(margin), float(radius_pad), uid, C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"sgp_tiles_create failed ({rc}): {self._lib.sgp_last_error().decode()}")

    @staticmethod
    def unique_id():
        """rank 0: the communicator id every rank passes to the constructor (ncclGetUniqueId)."""
        from .lib import load
        lib = load()
        buf = (C.c_uint8 * 128)()
        rc = lib.sgp_tiles_unique_id(buf)
        if rc != 0:
            raise RuntimeError(f"sgp_tiles_unique_id failed ({rc}): {lib.sgp_last_error().decode()}")
        return bytes(buf)

    def exchange(self):
        rc = self._lib.sgp_tiles_exchange(self._h)
        if rc != 0:
            raise RuntimeError(f"sgp_tiles_exchange failed ({rc}): {self._lib.sgp_last_error().decode()}")

    @staticmethod
    def exchange_group(tiles_list):
        arr = (C.c_void_p * len(tiles_list))(*[t._h for t in tiles_list])
        lib = tiles_list[0]._lib
        rc = lib.sgp_tiles_exchange_group(arr, len(tiles_list))
        if rc != 0:
            raise RuntimeError(f"sgp_tiles_exchange_group failed ({rc}): {lib.sgp_last_error().decode()}")

    def stats(self):
        s = abi.TilesStats()
        self._lib.sgp_tiles_get_stats(self._h, C.byref(s))
        return s

    def drain_migrations(self, cap=4096):
        out = np.zeros(cap, dtype=abi.migration_dtype)
        n = C.c_uint32(0)
        self._lib.sgp_tiles_drain_migrations(self._h, out.ctypes.data, cap, C.byref(n))
        return out[:min(n.value, cap)]

    # the counters bench.py and the tests read from either exchange class
    @property
    def last_exported(self):
        return self.stats().exported

    @property
    def last_imported(self):
        return self.stats().ghosts

    @property
    def last_emigrated(self):
        return self.stats().emigrated

    @property
    def last_immigrated(self):
        return self.stats().immigrated

    def close(self):
        if self._h:
            self._lib.sgp_tiles_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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
