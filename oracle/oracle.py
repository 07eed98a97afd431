"""ORACLE loader (test infrastructure).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; the product package substrata_amd never does.

Binds oracle/libsgo_oracle.so (plain-C restatement of PhysicsWorld::think, see sgo_oracle.c) to the same thin
ctypes driver the product uses, so parity tests issue identical calls to both.
"""
import ctypes as C
import os
import subprocess
import numpy as np

from substrata_amd import abi
from substrata_amd.world import CWorld

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsgo_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("sgo_oracle.c", "sgo_collide.h", "sgo_hull.h", "sgo_hull_build.h", "sgo_mesh.h", "sgo_vehicle.h", "sgo_math.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "sgp.h"))
    if not force and os.path.exists(_LIB_PATH) and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B", "libsgo_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        abi.bind(_lib, "sgo_")
        _lib.sgo_collide_pair.restype = C.c_int
        _lib.sgo_collide_pair.argtypes = [C.POINTER(abi.BodyDesc), C.POINTER(abi.BodyDesc), C.c_float, C.c_void_p,
                                          C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
        _lib.sgo_layers_collide.restype = C.c_int
        _lib.sgo_layers_collide.argtypes = [C.c_int, C.c_int]
        _lib.sgo_set_threads.restype = C.c_int
        _lib.sgo_set_threads.argtypes = [C.c_int]
    return _lib


class OracleWorld(CWorld):
    def __init__(self, **kw):
        super().__init__(lib(), "sgo_", **kw)


def collide_pair(a, b, max_sep=0.02):
    """Narrow phase of the oracle on two body descs. Returns (normal, p1[np,3], p2[np,3]) or None."""
    n = np.zeros(3, np.float32)
    p1 = np.zeros((8, 3), np.float32)
    p2 = np.zeros((8, 3), np.float32)
    npts = C.c_int(0)
    hit = lib().sgo_collide_pair(C.byref(a), C.byref(b), float(max_sep), n.ctypes.data, C.byref(npts), p1.ctypes.data,
                                 p2.ctypes.data)
    if not hit:
        return None
    return n, p1[:npts.value].copy(), p2[:npts.value].copy()


def world_collide_pair(world, a, b, max_sep=0.02):
    """Narrow phase on two body descs of `world` (hull ids resolve against its hull table). Returns (normal, p1, p2) or None."""
    n = np.zeros(3, np.float32)
    p1 = np.zeros((8, 3), np.float32)
    p2 = np.zeros((8, 3), np.float32)
    npts = C.c_int(0)
    f = lib().sgo_world_collide_pair
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.POINTER(abi.BodyDesc), C.POINTER(abi.BodyDesc), C.c_float, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    hit = f(world._h, C.byref(a), C.byref(b), float(max_sep), n.ctypes.data, C.byref(npts), p1.ctypes.data, p2.ctypes.data)
    if hit < 0:
        raise ValueError("bad hull id")
    if not hit:
        return None
    return n, p1[:npts.value].copy(), p2[:npts.value].copy()


def hull_dump(world, hull_id):
    """The hull as stored (body frame): (verts[nv,3], planes[nf,4])."""
    v = np.zeros((256, 3), np.float32)
    pl = np.zeros((512, 4), np.float32)
    f = lib().sgo_hull_dump
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    r = f(world._h, int(hull_id), v.ctypes.data, pl.ctypes.data)
    if r < 0:
        raise ValueError("bad hull id")
    return v[:r & 0xFFFF].copy(), pl[:r >> 16].copy()


def cast_sphere(body, origin, direction, max_t, radius):
    """Sphere cast of the oracle against one body desc. Returns (t, normal, point) or None."""
    o = np.asarray(origin, np.float32)
    d = np.asarray(direction, np.float32)
    n = np.zeros(3, np.float32)
    p = np.zeros(3, np.float32)
    f = lib().sgo_cast_sphere
    f.restype = C.c_float
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    t = f(C.addressof(body), o.ctypes.data, d.ctypes.data, float(max_t), float(radius), n.ctypes.data, p.ctypes.data)
    return None if t < 0 else (float(t), n, p)


def cast_disc(body, origin, direction, across, leading, disc_r, rho, max_t):
    """The wheel cast of the oracle (VehicleCollisionTesterCastCylinder: a disc of radius disc_r in the plane spanned by `across` and `leading`, rounded by rho)
    against one body desc.  Returns (t, normal, point) or None."""
    a = [np.ascontiguousarray(x, np.float32) for x in (origin, direction, across, leading)]
    n = np.zeros(3, np.float32)
    p = np.zeros(3, np.float32)
    f = lib().sgo_cast_disc_hook
    f.restype = C.c_float
    f.argtypes = [C.c_void_p] * 5 + [C.c_float] * 3 + [C.c_void_p] * 2
    t = f(C.addressof(body), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, float(disc_r), float(rho), float(max_t), n.ctypes.data, p.ctypes.data)
    return None if t < 0 else (float(t), n, p)


def set_threads(n):
    """OpenMP threads for the order-independent loops of the oracle (cpu_baseline timing only; tests use 1)."""
    return int(lib().sgo_set_threads(int(n)))


def layers_collide(l1, l2):
    return bool(lib().sgo_layers_collide(int(l1), int(l2)))
