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
        if len(sys.argv) > 3 and st["n"] == int(sys.argv[3]):
            rng = np.random.default_rng(0)
            for body in [int(x) for x in sys.argv[4:]]:
                c0 = b["pos"][body]
                rays = np.zeros(6000, dtype=abi.ray_dtype)
                rays["origin"] = c0 + rng.uniform(-1.5, 1.5, (6000, 3)); dv = rng.normal(size=(6000, 3)); rays["dir"] = dv / np.linalg.norm(dv, axis=1, keepdims=True)
                rays["max_t"] = 0.8; rays["ignore_id"] = abi.INVALID_ID
                rad = np.full(6000, 0.3, np.float32)
                hg, hc = tw.gpu.spherecast(rays, rad), tw.cpu.spherecast(rays, rad)
                dif = np.flatnonzero((hg["id"] != hc["id"]) | (hg["t"].view(np.uint32) != hc["t"].view(np.uint32)))
                print("probe at step", st["n"], "around body", body, ": sphere casts that differ:", len(dif), "of 6000")
                for k in dif[:6]:
                    print("     ", rays["origin"][k].tolist(), rays["dir"][k].tolist(), "gpu", int(hg["id"][k]), float(hg["t"][k]), hg["normal"][k].tolist(), "cpu", int(hc["id"][k]), float(hc["t"][k]), hc["normal"][k].tolist())
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
            for x in cg[(cg["a"] == bad[0]) | (cg["b"] == bad[0])]: print("   gpu full:", {k: (x[k].tolist() if hasattr(x[k], "tolist") else x[k]) for k in x.dtype.names})
            for x in cc[(cc["a"] == bad[0]) | (cc["b"] == bad[0])]: print("   cpu full:", {k: (x[k].tolist() if hasattr(x[k], "tolist") else x[k]) for k in x.dtype.names})
            kg = {(int(x["a"]), int(x["b"])): x for x in cg}; kc = {(int(x["a"]), int(x["b"])): x for x in cc}
            difc = [k for k in kg if k in kc and (kg[k]["lam_n"].tobytes() != kc[k]["lam_n"].tobytes() or kg[k]["bias"].tobytes() != kc[k]["bias"].tobytes())]
            print("   constraints whose lambdas / bias differ:", difc[:12])
            for v in range(3):
                try:
                    vg, vc = tw.gpu.vehicle_get_state(v), tw.cpu.vehicle_get_state(v)
                except Exception:
                    break
                if vg.tobytes() != vc.tobytes():
                    for wi in range(4):
                        for wn in vg["wheels"].dtype.names:
                            a_, b_ = vg["wheels"][wi][wn], vc["wheels"][wi][wn]
                            if np.asarray(a_).tobytes() != np.asarray(b_).tobytes(): print("   vehicle", v, "wheel", wi, wn, np.asarray(a_).tolist(), np.asarray(b_).tolist())
                else:
                    print("   vehicle", v, "state identical; wheel contact bodies", [int(vg["wheels"][wi]["contact_body"]) for wi in range(4)], "has_contact", [int(vg["wheels"][wi]["has_contact"]) for wi in range(4)])
            rng = np.random.default_rng(0)
            c0 = b["pos"][bad[0]]
            rays = np.zeros(4000, dtype=abi.ray_dtype)
            rays["origin"] = c0 + rng.uniform(-1.2, 1.2, (4000, 3)); dv = rng.normal(size=(4000, 3)); rays["dir"] = dv / np.linalg.norm(dv, axis=1, keepdims=True)
            rays["max_t"] = 0.8; rays["ignore_id"] = abi.INVALID_ID
            rad = np.full(4000, 0.3, np.float32)
            hg, hc = tw.gpu.spherecast(rays, rad), tw.cpu.spherecast(rays, rad)
            dif = np.flatnonzero((hg["id"] != hc["id"]) | (hg["t"].view(np.uint32) != hc["t"].view(np.uint32)))
            print("   sphere casts around the body that differ:", len(dif))
            for k in dif[:5]:
                print("     ", rays["origin"][k].tolist(), rays["dir"][k].tolist(), "gpu", int(hg["id"][k]), float(hg["t"][k]), "cpu", int(hc["id"][k]), float(hc["t"][k]))
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
