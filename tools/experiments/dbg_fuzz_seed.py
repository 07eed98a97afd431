import os, sys
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as fz
import parity
from substrata_amd import abi
from oracle import oracle
oracle.build()
# monkeypatch: check stats each step
orig_step = parity.Twin.__getattr__
state = {"n": 0}
class Stop(Exception): pass
def run():
    import types
    real_make = parity.make_twin
    def mk(o, **kw):
        tw = real_make(o, **kw)
        g_step, c_step = tw.gpu.step, tw.cpu.step
        state["descs"] = {}
        g_add = tw.gpu.add_batch
        def add_batch_logged(d):
            ids = g_add(d)
            for k, i in enumerate(ids):
                state["descs"][int(i)] = {"shape_type": int(d["shape_type"][k]), "shape": d["shape"][k].tolist(), "mass": float(d["mass"][k]), "motion": int(d["motion_type"][k])}
            return ids
        tw.gpu.add_batch = add_batch_logged
        def both_step(dt):
            g_step(dt); c_step(dt)
            state["n"] += 1
            stg = tw.gpu.read_states(0, 2048)
            bad = np.flatnonzero(~np.isfinite(stg["pos"]).all(axis=1) & (stg["id"] != abi.INVALID_ID))
            if len(bad) and not state.get("nan_reported"):
                state["nan_reported"] = True
                print("first non-finite pose at step", state["n"], "bodies", bad[:5])
                prev = state.get("prev")
                for i in bad[:3]:
                    print("  body", i, "previous step state:", {k: prev[k][i].tolist() for k in ("pos", "rot", "lin_vel", "ang_vel")} if prev is not None else None)
                    print("  desc:", state["descs"].get(int(i)))
            state["prev"] = stg
            sg, sc = tw.gpu.stats(), tw.cpu.stats()
            tg = (sg.num_manifolds, sg.num_contact_points, sg.num_colours)
            tc = (sc.num_manifolds, sc.num_contact_points, sc.num_colours)
            if sg.num_pairs != sc.num_pairs and not state.get("pairs_reported"):
                state["pairs_reported"] = True
                print("pair counts differ first at step", state["n"], sg.num_pairs, sc.num_pairs)
            if tg != tc:
                print("first stats mismatch at step", state["n"], tg, tc)
                cg, cc = tw.gpu.dump_constraints(), tw.cpu.dump_constraints()
                pg = set(zip(cg["a"].tolist(), cg["b"].tolist())); pc = set(zip(cc["a"].tolist(), cc["b"].tolist()))
                print("only gpu:", sorted(pg - pc)[:20]); print("only cpu:", sorted(pc - pg)[:20])
                ids = sorted({x for p in (pc ^ pg) for x in p})[:12]
                st = tw.gpu.read_states(0, 2048); sc2 = tw.cpu.read_states(0, 2048)
                for i in ids:
                    print(i, state["descs"].get(int(i)), "gpu pos", st["pos"][i], "active", st["active"][i], "| cpu pos", sc2["pos"][i], "active", sc2["active"][i])
                raise Stop()
            return None, None
        tw.step = both_step
        return tw
    parity.make_twin = mk
    try:
        fz.run_seed(oracle, int(sys.argv[1]), int(sys.argv[2]), verbose=True)
    except Stop:
        pass
run()
