"""Synthetic inputs for the BASELINE.json configs.  Pure numpy, deterministic (numpy PCG64 streams
seeded per config).  Every generator returns a structured array of abi.body_desc_dtype ready for add_batch();
body 0 is always the ground quad (createGroundQuadShape(2000): static box half (1000,1000,0.5) centred at z=-0.5,
/root/reference/gui_client/PhysicsWorld.cpp:1123-1135, GUIClient.cpp:573-574).

Body parameters are the WorldObject defaults the reference gives scripted objects: mass 50, friction 0.5,
restitution 0.2 (shared/WorldObject.cpp:119,1268-1270); dynamic, layer MOVING, activated on add.
"""
import numpy as np
from . import abi


def _blank(n):
    d = np.zeros(n, dtype=abi.body_desc_dtype)
    d["rot"][:, 3] = 1.0
    d["shape_type"] = abi.SHAPE_BOX
    d["shape"][:, :3] = 0.5
    d["motion_type"] = abi.MOTION_STATIC
    d["layer"] = abi.LAYER_NON_MOVING
    d["mass"] = 100.0
    d["friction"] = 0.5
    d["restitution"] = 0.3
    d["gravity_factor"] = 1.0
    d["linear_damping"] = 0.05
    d["angular_damping"] = 0.05
    d["allow_sleeping"] = 1
    return d


def ground(width=2000.0, friction=0.5, restitution=0.3):
    g = _blank(1)
    g["shape"][0, :3] = (width / 2, width / 2, 0.5)
    g["pos"][0] = (0.0, 0.0, -0.5)
    g["friction"] = friction
    g["restitution"] = restitution
    return g


def dynamic_bodies(n, mass=50.0, friction=0.5, restitution=0.2):
    d = _blank(n)
    d["motion_type"] = abi.MOTION_DYNAMIC
    d["layer"] = abi.LAYER_MOVING
    d["mass"] = mass
    d["friction"] = friction
    d["restitution"] = restitution
    d["activate"] = 1
    return d


def _random_unit_quats(rng, n):
    q = rng.standard_normal((n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def lattice(nx, ny, nz, spacing, z0, seed, jitter=0.05, random_rot=True, origin_centered=True):
    """nx*ny*nz unit cubes on a lattice, x fastest; xy jitter +-jitter; optional uniformly random orientation."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = nx * ny * nz
    ix, iy, iz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    # x fastest ordering
    idx = np.arange(n)
    x = (idx % nx).astype(np.float32)
    y = ((idx // nx) % ny).astype(np.float32)
    z = (idx // (nx * ny)).astype(np.float32)
    d = dynamic_bodies(n)
    ox = (nx - 1) * spacing / 2 if origin_centered else 0.0
    oy = (ny - 1) * spacing / 2 if origin_centered else 0.0
    jit = (rng.random((n, 2)).astype(np.float32) * 2 - 1) * np.float32(jitter)
    d["pos"][:, 0] = x * spacing - ox + jit[:, 0]
    d["pos"][:, 1] = y * spacing - oy + jit[:, 1]
    d["pos"][:, 2] = z0 + z * spacing
    if random_rot:
        d["rot"] = _random_unit_quats(rng, n)
    return d, rng


def config1_256_boxes():
    """Config 1: 256 unit cubes, 8x8x4 lattice, spacing 1.5 m, lowest layer centre z=1.0, seed 1."""
    d, _ = lattice(8, 8, 4, 1.5, 1.0, seed=1)
    return np.concatenate([ground(), d])


def config2_10k_boxes():
    """Config 2: 10k unit cubes, 25x25x16 lattice, spacing 1.25 m, seed 2."""
    d, _ = lattice(25, 25, 16, 1.25, 1.0, seed=2)
    return np.concatenate([ground(), d])


def config3_100k_mixed(nx=100, ny=100, nz=10, seed=3):
    """Config 3: 100k mixed bodies, 100x100x10 lattice, spacing 1.5 m; type = index mod 3 -> box (half 0.5) /
    sphere (r 0.5) / capsule (half-height 0.65, r 0.3, axis z); uniform scale in [0.5,1.5]; mass = 50*scale^3."""
    d, rng = lattice(nx, ny, nz, 1.5, 1.0, seed=seed)
    n = len(d)
    s = (0.5 + rng.random(n)).astype(np.float32)
    t = np.arange(n) % 3
    d["mass"] = 50.0 * s ** 3
    box = t == 0
    sph = t == 1
    cap = t == 2
    d["shape_type"][box] = abi.SHAPE_BOX
    d["shape"][box, :3] = (0.5 * s[box])[:, None]
    d["shape_type"][sph] = abi.SHAPE_SPHERE
    d["shape"][sph, 0] = 0.5 * s[sph]
    d["shape"][sph, 1:] = 0
    d["shape_type"][cap] = abi.SHAPE_CAPSULE
    d["shape"][cap, 0] = 0.3 * s[cap]
    d["shape"][cap, 1] = 0.65 * s[cap]
    d["shape"][cap, 2:] = 0
    return np.concatenate([ground(), d])


CONFIG4_SPACING = 1.25


def config4_1m_boxes(n=100, seed=4, spacing=CONFIG4_SPACING):
    """Config 4: n^3 unit cubes (1M at n = 100) on an n x n x n lattice, spacing 1.25 m, lowest layer centre z = 1.0, seed 4;
    body 0 is the ground quad.  The lattice is centred on the origin in x and y."""
    d, _ = lattice(n, n, n, spacing, 1.0, seed=seed)
    return np.concatenate([ground(), d])


def config4_world_box(n=100, spacing=CONFIG4_SPACING):
    """The box the 3-D tiles of config 4 split: the lattice's extent (lo corner, size), z from the ground plane."""
    w = n * spacing
    return np.array([-w / 2, -w / 2, 0.0], np.float32), np.array([w, w, w], np.float32)


def config4_tile_descs(rank, n_tiles, n=100, seed=4, spacing=CONFIG4_SPACING):
    """What tile `rank` of the n_tiles-way 3-D split (tiles.tile_grid: 2x1x1 / 2x2x1 / 2x2x2) owns of config 4: its own ground quad
    (body 0 of every tile world) + the lattice bodies whose centre lies in the tile.  Returns (descs, lo, hi) with the tile's region."""
    from . import tiles
    full = config4_1m_boxes(n, seed, spacing)
    origin, size = config4_world_box(n, spacing)
    tx, ty, tz = tiles.tile_grid(n_tiles)
    lo, hi, _ = tiles.tile_bounds(rank, n_tiles, size[0] / tx, size[1] / ty, size[2] / tz, origin=origin)
    p = full["pos"][1:]
    mine = np.all(p >= lo, axis=1) & np.all(p < hi, axis=1)
    return np.concatenate([full[:1], full[1:][mine]]), lo, hi


def config4_tile(nx, ny, nz, spacing=1.25, seed=4, offset=(0.0, 0.0, 0.0)):
    """Config 4 building block: one spatial tile of the 1M-box lattice (100^3, spacing 1.25 m)."""
    d, _ = lattice(nx, ny, nz, spacing, 1.0, seed=seed, origin_centered=False)
    d["pos"] += np.asarray(offset, dtype=np.float32)
    return d


def small_mixed(n_side=6, layers=3, seed=7):
    """A small config-3-style scene for parity tests (oracle finishes in seconds)."""
    return config3_100k_mixed(n_side, n_side, layers, seed)


CAR_HALF_EXTENTS = (0.9, 2.0, 0.25)      # default car hull of the reference (Scripting.cpp:369-386: x +-0.9, up +-0.25 (roof 0.7), forward +-2) as a box;
CAR_MASS = 1200.0                        # x right, y forward, z up.  The reference has no default car mass (it uses ob->mass).


def config5_cars_debris(cars_side=32, n_debris=50000, spacing=8.0, seed=5):
    """BASELINE config 5: cars_side^2 cars on a grid (spacing 8 m) + n_debris unit boxes scattered at z in [0.5, 3].
    Returns (descs, car_body_ids): descs[0] is the ground, descs[1 : 1 + cars] the chassis bodies (box stand-in for the 12-point
    hull), the rest debris.  Vehicles are created per chassis with World.default_vehicle_desc(body) (CarPhysics' 4-wheel FWD
    layout with the Scripting.cpp:315-346 defaults); drive them with config5_inputs()."""
    rng = np.random.default_rng(seed)
    n_cars = cars_side * cars_side
    cars = dynamic_bodies(n_cars, mass=CAR_MASS, friction=0.5, restitution=0.0)
    ix, iy = np.meshgrid(np.arange(cars_side), np.arange(cars_side), indexing="ij")
    cars["pos"][:, 0] = (ix.ravel() - 0.5 * (cars_side - 1)) * spacing
    cars["pos"][:, 1] = (iy.ravel() - 0.5 * (cars_side - 1)) * spacing
    cars["pos"][:, 2] = 0.8
    cars["shape"][:, :3] = CAR_HALF_EXTENTS
    half = 0.5 * cars_side * spacing
    deb = dynamic_bodies(n_debris)
    pts = np.empty((0, 3), np.float32)
    while len(pts) < n_debris:                      # scatter, keeping the cars' footprints (+ margin) clear
        p = rng.uniform([-half, -half, 0.5], [half, half, 3.0], size=(n_debris, 3)).astype(np.float32)
        fx = np.abs(((p[:, 0] + half) % spacing) - 0.5 * spacing)
        fy = np.abs(((p[:, 1] + half) % spacing) - 0.5 * spacing)
        pts = np.concatenate([pts, p[(fx > 1.9) | (fy > 3.0)]])
    deb["pos"] = pts[:n_debris]
    deb["rot"] = _random_unit_quats(rng, n_debris)
    descs = np.concatenate([ground(), cars, deb])
    return descs, np.arange(1, 1 + n_cars, dtype=np.uint32)


# the default car hull of the reference (Scripting.cpp:369-386; model space: x right, y up, z forward) turned into x right, y forward, z up
CAR_HULL_POINTS = np.array([(sx * 0.9, sf * 2.0, su * 0.25) for sx in (-1, 1) for su in (-1, 1) for sf in (-1, 1)] +
                           [(0.9, 0.6, 0.7), (-0.9, 0.6, 0.7), (0.9, -1.2, 0.7), (-0.9, -1.2, 0.7)], dtype=np.float32)
CAR_COM_OFFSET = (0.0, 0.0, -0.2)        # OffsetCenterOfMassShape (CarPhysics.cpp:76-78 uses object->centre_of_mass_offset_os; no default in the reference)


def use_car_hull(world, descs, car_ids):
    """Turn the box chassis of config5_cars_debris() into the reference's 12-point convex hull (created in `world`, with the lowered
    centre of mass) before the descs are added.  Returns the hull info; the body frame sits at info.com of the hull's point frame, so
    the chassis positions are shifted accordingly (the principal axes of this symmetric hull are the point frame's axes up to a
    small pitch, which the wheel layout of default_vehicle_desc() ignores)."""
    info = world.hull_create(CAR_HULL_POINTS, com_offset=CAR_COM_OFFSET)
    idx = np.asarray(car_ids, dtype=np.int64)
    descs["shape_type"][idx] = abi.SHAPE_HULL
    descs["shape"][idx] = 0
    descs["shape"][idx, 0] = float(info.hull_id)
    descs["pos"][idx] += np.array(info.com[:], dtype=np.float32)
    descs["rot"][idx] = np.array(info.rot[:], dtype=np.float32)
    return info


def config5_inputs(n_cars, t):
    """Driver input of config 5 at time t: forward = 1, steer = sin(0.5 t + car id)."""
    inp = np.zeros(n_cars, dtype=abi.vehicle_input_dtype)
    inp["forward"] = 1.0
    inp["right"] = np.sin(0.5 * t + np.arange(n_cars)).astype(np.float32)
    return inp
