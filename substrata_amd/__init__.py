"""substrata_amd -- MI355X-native rigid-body stepper behind Substrata's PhysicsWorld / PhysicsObject facade.

The compute path is hand-written HIP for gfx950 in substrata_amd/csrc (built to substrata_amd/libsgp.so) behind the
C ABI of include/sgp.h.  There is no CPU fallback: loading fails loudly when the library is missing.
"""
from . import abi  # noqa: F401

__all__ = ["abi"]
