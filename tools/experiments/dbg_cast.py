import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as fz, parity
from substrata_amd import abi
from oracle import oracle
oracle.build()
seed = int(sys.argv[1]); steps = int(sys.argv[2])
real_make = parity.make_twin
holder = {}
def mk(o, **kw):
    tw = real_make(o, **kw); holder["tw"] = tw
    descs = holder.setdefault("descs", {})
    g_add = tw.gpu.add_batch
    def add_logged(d):
        ids = g_add(d)
        for k, i in enumerate(ids):
            descs[int(i)] = {"shape_type": int(d["shape_type"][k]), "shape": d["shape"][k].tolist(), "motion": int(d["motion_type"][k]), "layer": int(d["layer"][k]), "sensor": int(d["is_sensor"][k])}
        return ids
    tw.gpu.add_batch = add_logged
    return tw
parity.make_twin = mk
try:
    fz.run_seed(oracle, seed, steps, verbose=True)
except AssertionError as e:
    print(str(e)[:700])
    tw = holder["tw"]
    info = e.args[0][3]
    for (k, ids, tg, tc, rs, _, o, d) in info:
        for i in ids:
            sg = tw.gpu.get_state([i])[0]; sc = tw.cpu.get_state([i])[0]
            print("body", i, holder["descs"].get(i), "pos", sg["pos"], "rot", sg["rot"], "active g/c", sg["active"], sc["active"])
        # single casts against each world, one at a time with ignore of the other
        ray = np.zeros(1, dtype=abi.ray_dtype); ray["origin"][0] = o; ray["dir"][0] = d; ray["max_t"] = 30.0; ray["ignore_id"] = abi.INVALID_ID
        for name, w in (("gpu", tw.gpu), ("cpu", tw.cpu)):
            h = w.spherecast(ray, np.float32([rs]))
            print(name, "cast again:", int(h["id"][0]), float(h["t"][0]))
            r2 = ray.copy(); r2["ignore_id"] = int(h["id"][0])
            h2 = w.spherecast(r2, np.float32([rs]))
            print(name, "ignoring that:", int(h2["id"][0]), float(h2["t"][0]))
