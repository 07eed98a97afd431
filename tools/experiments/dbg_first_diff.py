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
class Stop(Exception): pass
st = {"n": 0, "descs": {}}
def mk(o, **kw):
    tw = real_make(o, **kw)
    g_step, c_step = tw.gpu.step, tw.cpu.step
    g_add = tw.gpu.add_batch
    def add_logged(d):
        ids = g_add(d)
        for k, i in enumerate(ids):
            st["descs"][int(i)] = (int(d["shape_type"][k]), [round(float(x), 3) for x in d["shape"][k]], int(d["motion_type"][k]))
        return ids
    tw.gpu.add_batch = add_logged
    def both(dt):
        g_step(dt); c_step(dt); st["n"] += 1
        a, b = tw.gpu.read_states(0, 2048), tw.cpu.read_states(0, 2048)
        live = a["id"] != abi.INVALID_ID
        bad = np.flatnonzero(live & ((a["pos"].view(np.uint32) != b["pos"].view(np.uint32)).any(axis=1) | (a["lin_vel"].view(np.uint32) != b["lin_vel"].view(np.uint32)).any(axis=1) |
                                     (a["ang_vel"].view(np.uint32) != b["ang_vel"].view(np.uint32)).any(axis=1) | (a["rot"].view(np.uint32) != b["rot"].view(np.uint32)).any(axis=1)))
        if len(bad):
            print("first state difference at step", st["n"], "bodies", bad[:10].tolist())
            for i in bad[:4]:
                print("  ", i, st["descs"].get(int(i)), "gpu", a["pos"][i], a["lin_vel"][i], a["ang_vel"][i], "| cpu", b["pos"][i], b["lin_vel"][i], b["ang_vel"][i], "active", a["active"][i], b["active"][i])
            cg, cc = tw.gpu.dump_constraints(), tw.cpu.dump_constraints()
            for i in bad[:3]:
                mg = cg[(cg["a"] == i) | (cg["b"] == i)]; mc = cc[(cc["a"] == i) | (cc["b"] == i)]
                print("   constraints of", i, "gpu:", [(int(x["a"]), int(x["b"]), int(x["colour"]), int(x["np"]), [round(float(v), 6) for v in x["lam_n"]]) for x in mg][:6])
                print("   constraints of", i, "cpu:", [(int(x["a"]), int(x["b"]), int(x["colour"]), int(x["np"]), [round(float(v), 6) for v in x["lam_n"]]) for x in mc][:6])
            sg, sc = tw.gpu.stats(), tw.cpu.stats()
            print("   stats gpu", sg.num_pairs, sg.num_manifolds, sg.num_contact_points, sg.num_colours, "cpu", sc.num_pairs, sc.num_manifolds, sc.num_contact_points, sc.num_colours)
            raise Stop()
        return None, None
    tw.step = both
    return tw
parity.make_twin = mk
try:
    fz.run_seed(oracle, seed, steps, verbose=True)
except Stop:
    pass
