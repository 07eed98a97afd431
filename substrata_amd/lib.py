"""Loads substrata_amd/libsgp.so (the HIP product).  Fails loudly: there is no CPU or Python fallback."""
import ctypes as C
import os

import numpy as np

from . import abi
from .world import CWorld, SgpError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SGP_LIB_PATH") or os.path.join(_HERE, "libsgp.so")      # SGP_LIB_PATH: A/B runs against another build of the same ABI
_lib = None
_devices = None


def load():
    """dlopen libsgp.so and bind the prototypes of include/sgp.h. Raises if the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SgpError(f"{LIB_PATH} is missing: build it with `python -m substrata_amd.build` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        abi.bind(_lib, "sgp_")
    return _lib


def init():
    """PhysicsWorld::init() (PhysicsWorld.cpp:250-273). Returns the HIP device count; raises without a GPU."""
    global _devices
    lib = load()
    n = lib.sgp_init()
    if n <= 0:
        raise SgpError(f"sgp_init failed ({n}): {lib.sgp_last_error().decode()}")
    _devices = n
    return n


class World(CWorld):
    """One device-resident physics world (C ABI handle)."""

    def __init__(self, **kw):
        if _devices is None:
            init()
        super().__init__(load(), "sgp_", **kw)

    def kernel_class_names(self):
        names = []
        for k in range(abi.NUM_KERNEL_CLASSES):
            s = self._lib.sgp_kernel_class_name(k)
            if not s:
                break
            names.append(s.decode())
        return names


class SnapshotQueue:
    """De-jitter rings + insertion schedule of the network physics snapshots (sgp_snapshot_queue_*, include/sgp.h): host-side state, one
    queue for any number of objects keyed by uid.  poll() returns what is due now as (uids, pose_vel records) for ONE
    World.set_pose_vel_batch call."""

    PADDING_DELAY = 0.1       # GUIClient.cpp:7465

    def __init__(self):
        self._lib = load()
        self._h = C.c_void_p()
        rc = self._lib.sgp_snapshot_queue_create(C.byref(self._h))
        self._check(rc)

    def _check(self, rc):
        if rc != 0:
            raise SgpError(f"sgp_snapshot_queue call failed ({rc}): {self._lib.sgp_last_error().decode()}")

    def push_wire(self, msg, local_time):
        buf = (C.c_uint8 * abi.PHYSICS_UPDATE_BYTES).from_buffer_copy(bytes(msg))
        self._check(self._lib.sgp_snapshot_queue_push_wire(self._h, buf, float(local_time)))

    def push(self, uid, rec, client_time, local_time):
        r = abi.PoseVel.from_buffer_copy(np.asarray(rec, dtype=abi.pose_vel_dtype).tobytes())
        self._check(self._lib.sgp_snapshot_queue_push(self._h, int(uid), C.byref(r), float(client_time), float(local_time)))

    def ownership(self, uid, global_time_now, ownership_change_global_time, renewal=False):
        self._check(self._lib.sgp_snapshot_queue_ownership(self._h, int(uid), float(global_time_now), float(ownership_change_global_time), 1 if renewal else 0))

    def poll(self, global_time, padding_delay=PADDING_DELAY, cap=4096):
        uids = np.zeros(cap, dtype=np.uint64)
        recs = np.zeros(cap, dtype=abi.pose_vel_dtype)
        n = C.c_uint32(0)
        self._check(self._lib.sgp_snapshot_queue_poll(self._h, float(global_time), float(padding_delay), uids.ctypes.data, recs.ctypes.data, cap, C.byref(n)))
        m = min(n.value, cap)
        return uids[:m], recs[:m], n.value

    def expire(self, local_time_now, max_age=1.0):
        n = C.c_uint32(0)
        self._check(self._lib.sgp_snapshot_queue_expire(self._h, float(local_time_now), float(max_age), C.byref(n)))
        return n.value

    def peek(self, uid):
        a, b, off = C.c_uint32(0), C.c_uint32(0), C.c_double(0)
        self._check(self._lib.sgp_snapshot_queue_peek(self._h, int(uid), C.byref(a), C.byref(b), C.byref(off)))
        return a.value, b.value, off.value

    def close(self):
        if self._h:
            self._lib.sgp_snapshot_queue_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
