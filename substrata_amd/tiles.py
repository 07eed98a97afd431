"""Multi-GPU spatial tiles: one process per GPU, one sgp world per tile.

The path shards by space.  Each rank OWNS the bodies created in its tile and simulates them dynamically; bodies whose
AABB (inflated by the ghost margin) pokes out of the owner's tile are exported once per sub-step and imported by every
rank whose tile (inflated by the margin) they touch, where they are simulated as velocity-driven infinite-mass ghosts
(kinematic bodies) for that step.  The only collectives are, per step, one all-gather of the per-destination record counts (4 B x
tiles per rank) and one grouped send / recv of the ghost records -- each record goes only to the tiles it can touch -- both issued on
RCCL from inside libsgp.so (NativeTiles = sgp_tiles_*; there is no second transport in the product).  Ghost traffic is ~1e3-1e4 records
x 128 B per rank, so the exchange is latency- not bandwidth-bound.  The host statement of the same rules over torch.distributed, which
the oracle worlds and the gloo tests use, lives in tests/ghost_exchange.py; it shares the two C routing helpers below
(sgp_tiles_route / sgp_tiles_split).  Ownership migrates: when the centre of an owned body has left the tile, the owner removes it and the tile
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

    def rebalance(self, grid, by_contacts=False):
        """Collective: move the grid's split planes to the body-count (or body + contact count) quantiles (sgp_tiles_rebalance)."""
        rc = self._lib.sgp_tiles_rebalance(self._h, int(grid[0]), int(grid[1]), int(grid[2]), int(by_contacts))
        if rc != 0:
            raise RuntimeError(f"sgp_tiles_rebalance failed ({rc}): {self._lib.sgp_last_error().decode()}")

    @staticmethod
    def rebalance_group(tiles_list, grid, by_contacts=False):
        arr = (C.c_void_p * len(tiles_list))(*[t._h for t in tiles_list])
        lib = tiles_list[0]._lib
        rc = lib.sgp_tiles_rebalance_group(arr, len(tiles_list), int(grid[0]), int(grid[1]), int(grid[2]), int(by_contacts))
        if rc != 0:
            raise RuntimeError(f"sgp_tiles_rebalance_group failed ({rc}): {lib.sgp_last_error().decode()}")

    def boxes(self):
        out = np.zeros((self.n, 6), dtype=np.float32)
        self._lib.sgp_tiles_get_boxes(self._h, out.ctypes.data)
        return out

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
