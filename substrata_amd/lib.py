"""Loads substrata_amd/libsgp.so (the HIP product).  Fails loudly: there is no CPU or Python fallback."""
import ctypes as C
import os

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
