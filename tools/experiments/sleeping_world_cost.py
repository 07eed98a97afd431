"""What a step costs in a world that sleeps but for one body: 100 k boxes lying on the ground (asleep after a second), a kinematic body drifting far away.
The contact cache keeps the contacts of the sleepers: they are copied from buffer to buffer every step (docs/GAPS.md).  SGP_LIB_PATH selects the build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401
from substrata_amd import abi, scenes
from substrata_amd.lib import World
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
side = int(np.ceil(np.sqrt(n)))
d = scenes.dynamic_bodies(n)
ix, iy = np.meshgrid(np.arange(side), np.arange(side), indexing="ij")
d["pos"][:, 0] = (ix.ravel()[:n] - side / 2) * 1.5; d["pos"][:, 1] = (iy.ravel()[:n] - side / 2) * 1.5; d["pos"][:, 2] = 0.5
d["shape"][:, :3] = 0.5
k = scenes.dynamic_bodies(1); k["motion_type"] = abi.MOTION_KINEMATIC; k["pos"][0] = (0, 0, 50.0); k["lin_vel"][0] = (0.01, 0, 0)
w = World(max_bodies=n + 64)
w.add_batch(np.concatenate([scenes.ground(), d, k]))
for s in range(150):
    w.step(1.0 / 60.0)
st = w.stats()
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in range(200):
    w.step(1.0 / 60.0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
print(f"{os.environ.get('SGP_LIB_PATH', 'tree')}: {n} boxes, awake {st.num_active}, constraints {st.num_manifolds}: {dt * 1e6:.0f} us per step")
