import os
import sys

import pytest

# torch first: its wheel bundles its own HIP runtime, and a process that loads /opt/rocm's copy first (through libsgp.so) and torch's
# second ends up with two runtimes, the second of which sees no GPU ("No HIP GPUs are available").  Loaded in this order both resolve to
# one runtime.  Only the tile-exchange tests use torch, but every test process may load libsgp.so before they run.
try:
    import torch  # noqa: F401
except ImportError:      # the CPU-only parts of the suite do not need it
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o
