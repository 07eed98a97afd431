"""Multi-GPU spatial tiles: one process per GPU, one sgp world per tile.

The path shards by space.  Each rank OWNS the bodies created in its tile and simulates them dynamically; bodies whose
AABB (inflated by the ghost margin) pokes out of the owner's tile are exported once per sub-step and imported by every
rank whose tile (inflated by the margin) they touch, where they are simulated as velocity-driven infinite-mass ghosts
(kinematic bodies) for that step.  The only collectives are, per step, one all-gather of the record counts (8 B per rank) and
one all-gather of the ghost records padded to the largest count (torch.distributed: backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Ghost traffic is ~1e3-1e4 records x 104 B per rank, so the exchange is latency- not
bandwidth-bound.  Ownership migrates: when the centre of an owned body has left the tile, the owner removes it and the tile
that now contains the centre re-creates it as a dynamic body from the same record (its contact-cache entries restart).
"""
import numpy as np

from . import abi

REC = abi.ghost_dtype.itemsize


def tile_grid(n_tiles):
    """2-D tiling (the settled pile is shallow in z): 1 -> 1x1, 2 -> 2x1, 4 -> 2x2, 8 -> 4x2."""
    tx = 1
    while tx * tx < n_tiles:
        tx *= 2
    ty = max(1, n_tiles // tx)
    assert tx * ty == n_tiles, "tile count must be a power of two"
    return tx, ty


def tile_bounds(rank, n_tiles, tile_w, tile_d):
    """Axis-aligned region [lo, hi) of tile `rank`; z is unbounded."""
    tx, ty = tile_grid(n_tiles)
    ix, iy = rank % tx, rank // tx
    big = 1.0e9
    lo = np.array([ix * tile_w if ix > 0 else -big, iy * tile_d if iy > 0 else -big, -big], dtype=np.float32)
    hi = np.array([(ix + 1) * tile_w if ix < tx - 1 else big, (iy + 1) * tile_d if iy < ty - 1 else big, big], dtype=np.float32)
    origin = np.array([ix * tile_w, iy * tile_d, 0.0], dtype=np.float32)
    return lo, hi, origin


def inside(recs, lo, hi):
    """Mask of records whose centre lies in [lo, hi)."""
    if len(recs) == 0:
        return np.zeros(0, dtype=bool)
    p = recs["pos"]
    return np.all(p >= lo, axis=1) & np.all(p < hi, axis=1)


def records_to_descs(recs):
    """Body descs for immigrants: dynamic, awake, layer MOVING, Jolt default damping / gravity factor."""
    d = np.zeros(len(recs), dtype=abi.body_desc_dtype)
    for f in ("pos", "rot", "lin_vel", "ang_vel", "shape_type", "shape", "mass", "friction", "restitution"):
        d[f] = recs[f]
    d["motion_type"] = abi.MOTION_DYNAMIC
    d["layer"] = abi.LAYER_MOVING
    d["gravity_factor"] = 1.0
    d["linear_damping"] = 0.05
    d["angular_damping"] = 0.05
    d["allow_sleeping"] = 1
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


class GhostExchange:
    """Per step: one tiny all-gather of the record counts, then one all-gather of the records padded to the largest count."""

    def __init__(self, world, rank, n_tiles, lo, hi, margin, dist=None, device=None, cap=1 << 16):
        self.world, self.rank, self.n = world, rank, n_tiles
        self.lo, self.hi, self.margin = lo, hi, float(margin)
        self.dist, self.device, self.cap = dist, device, cap
        self.last_exported = 0
        self.last_imported = 0
        self.last_emigrated = 0
        self.last_immigrated = 0
        if dist is not None:
            import torch
            self.torch = torch
            self.cnt_send = torch.zeros(1, dtype=torch.int64, device=device)
            self.cnt_recv = torch.zeros(n_tiles, dtype=torch.int64, device=device)
            self.send = torch.zeros(cap * REC, dtype=torch.uint8, device=device)
            self.recv = torch.zeros(n_tiles * cap * REC, dtype=torch.uint8, device=device)

    def exchange(self):
        recs = self.world.export_boundary(self.lo, self.hi, self.margin, cap=self.cap)
        # owned DYNAMIC bodies whose centre has left the tile emigrate: removed here, re-created by the tile that contains them
        local_ids = (recs["global_id"] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        emigrant = ~inside(recs, self.lo, self.hi) & (recs["motion_type"] == abi.MOTION_DYNAMIC) if self.n > 1 else np.zeros(len(recs), bool)
        for i in local_ids[emigrant]:
            self.world.remove(int(i))
        self.last_emigrated = int(emigrant.sum())
        recs["global_id"] = recs["global_id"] | (np.uint64(self.rank) << np.uint64(40))
        recs["motion_type"] = np.where(emigrant, np.uint32(abi.MOTION_DYNAMIC | 0x100), recs["motion_type"])   # bit 8 = "take ownership"
        self.last_exported = len(recs)
        if self.dist is None:
            self.world.import_ghosts(recs[:0])
            return
        torch = self.torch
        self.cnt_send[0] = len(recs)
        self.dist.all_gather_into_tensor(self.cnt_recv, self.cnt_send)
        counts = self.cnt_recv.cpu().numpy()
        maxc = int(counts.max())
        if maxc == 0:
            self.last_imported = 0
            self.world.import_ghosts(recs[:0])
            return
        nbytes = maxc * REC
        if len(recs):
            self.send[:len(recs) * REC].copy_(torch.from_numpy(recs.view(np.uint8).reshape(-1).copy()))
        self.dist.all_gather_into_tensor(self.recv[:self.n * nbytes], self.send[:nbytes])
        allb = self.recv[:self.n * nbytes].cpu().numpy().reshape(self.n, nbytes)
        parts = []
        for r in range(self.n):
            c = int(counts[r])
            if r != self.rank and c:
                parts.append(np.frombuffer(allb[r, :c * REC].tobytes(), dtype=abi.ghost_dtype))
        others = np.concatenate(parts) if parts else np.zeros(0, dtype=abi.ghost_dtype)
        take = (others["motion_type"] & 0x100) != 0
        immigrants = others[take & inside(others, self.lo, self.hi)]
        ghosts = others[~take].copy()
        mine = select_ghosts(ghosts, self.lo, self.hi, self.margin)
        self.last_imported = len(mine)
        self.last_immigrated = len(immigrants)
        self.world.import_ghosts(mine)
        if len(immigrants):
            self.world.add_batch(records_to_descs(immigrants))
