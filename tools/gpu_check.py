"""Developer probe (run on the GPU box): prints GPU-vs-oracle differences for a few scenes."""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from substrata_amd import abi, scenes
from substrata_amd.lib import World
from oracle import oracle
import parity

DT = 1 / 60


def run(name, descs, steps, every):
    tw = parity.Twin(World(max_bodies=len(descs) + 8), oracle.OracleWorld(max_bodies=len(descs) + 8))
    tw.add_batch(descs)
    for s in range(1, steps + 1):
        tw.step(DT)
        if s % every == 0 or s == 1:
            d = parity.compare(tw, len(descs))
            sg, sc = tw.stats()
            print(f"{name} step {s}: {d} gpu(pairs {sg.num_pairs} man {sg.num_manifolds} pts {sg.num_contact_points} col {sg.num_colours} rounds {sg.num_colour_rounds} act {sg.num_active}) "
                  f"cpu(pairs {sc.num_pairs} man {sc.num_manifolds} pts {sc.num_contact_points} col {sc.num_colours} rounds {sc.num_colour_rounds} act {sc.num_active})", flush=True)
    tw.close()


if __name__ == "__main__":
    run("config1", scenes.config1_256_boxes(), 240, 30)
    run("mixed", scenes.small_mixed(6, 3, 7), 180, 30)
    # perf probe
    for name, descs in (("config2_10k", scenes.config2_10k_boxes()), ("config3_100k", scenes.config3_100k_mixed())):
        w = World(max_bodies=len(descs) + 8)
        w.add_batch(descs)
        w.step(DT)
        t = time.time()
        n = 120
        for _ in range(n):
            w.step(DT)
        el = time.time() - t
        st = w.stats()
        print(f"{name}: {n / el:.1f} steps/s  pairs {st.num_pairs} man {st.num_manifolds} pts {st.num_contact_points} col {st.num_colours} rounds {st.num_colour_rounds} active {st.num_active} dropped {st.pairs_dropped}/{st.manifolds_dropped}", flush=True)
        p = w.step_profiled(DT)
        names = w.kernel_class_names()
        print("  stages ms:", [round(x, 3) for x in p.stage_ms], "total", round(p.total_ms, 3))
        print("  kernels:", {names[k]: (round(p.kernel_ms[k], 3), p.kernel_launches[k]) for k in range(len(names)) if p.kernel_launches[k]})
        cons = w.dump_constraints()
        print("  colour histogram:", np.bincount(cons["colour"], minlength=1).tolist())
        print("  points histogram:", np.bincount(cons["np"], minlength=5).tolist())
        w.close()
