"""Developer probe: steps/s for the small BASELINE configs (1: 256 boxes, 2: 10k boxes), each with the CPU port (oracle, NOT Jolt) timed beside it
on the same descs and the same number of steps, and the two end states compared bit for bit (column B2).
The oracle is the checker and the reported baseline here, never the thing measured as the product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from substrata_amd import scenes
from substrata_amd.lib import World
WARM, N = 60, 300
for name, descs, cpu_n in (("config1_256", scenes.config1_256_boxes(), 300), ("config2_10k", scenes.config2_10k_boxes(), 40)):
    w = World(max_bodies=len(descs) + 64)
    w.add_batch(descs)
    for _ in range(WARM):
        w.step(1 / 60)
    S_warm = w.read_states(0, len(descs))
    t = time.perf_counter()
    for _ in range(N):
        w.step(1 / 60)
    el = time.perf_counter() - t
    st = w.stats()
    p = w.step_profiled(1 / 60)
    names = w.kernel_class_names()
    print(f"{name}: {N / el:.1f} steps/s ({1000 * el / N:.3f} ms/step) active {st.num_active} manifolds {st.num_manifolds} colours {st.num_colours} rounds {st.num_colour_rounds}; launches {sum(p.kernel_launches[k] for k in range(len(names)))}")
    print('   total_ms', round(p.total_ms,3), {names[k]: (round(p.kernel_ms[k],3), p.kernel_launches[k]) for k in range(len(names)) if p.kernel_launches[k]})
    w.close()
    if "--no-cpu" in sys.argv:
        continue
    # the CPU port on the same scene: WARM untimed steps, then cpu_n timed ones (1 thread and the best of a small sweep), end state against a GPU world
    from oracle import oracle
    best = None
    for threads in (1, 8, 32):
        if threads > (os.cpu_count() or 1):
            continue
        oracle.set_threads(threads)
        c = oracle.OracleWorld(max_bodies=len(descs) + 64)
        c.add_batch(descs)
        for _ in range(WARM):
            c.step(1 / 60)
        if threads == 1:
            S_c = c.read_states(0, len(descs))
            same_warm = all(np.array_equal(S_c[f], S_warm[f]) for f in ("pos", "rot", "lin_vel", "ang_vel", "active"))
        t = time.perf_counter()
        for _ in range(cpu_n):
            c.step(1 / 60)
        rate = cpu_n / (time.perf_counter() - t)
        cst = c.stats()
        c.close()
        print(f"   cpu port (oracle, not Jolt), {threads} thread(s): {rate:.1f} steps/s over {cpu_n} steps after the same {WARM} ({cst.num_manifolds} constraints, {cst.num_active} active)")
        if best is None or rate > best[1]:
            best = (threads, rate)
    oracle.set_threads(1)
    print(f"   {name}: GPU {N / el:.1f} steps/s; B2 cpu port best {best[1]:.1f} steps/s ({best[0]} threads, host cpus {os.cpu_count()}); state after {WARM} steps bit-exact vs oracle: {same_warm}")
