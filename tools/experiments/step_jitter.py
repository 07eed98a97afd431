"""Step-time series while a scene changes (config 3 forming its pile): how much do launch-plan changes (graph capture / instantiate) cost?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from substrata_amd import scenes
from substrata_amd.lib import World, init
init()
for graphs in ("0", "1"):
    os.environ["SGP_NO_GRAPH"] = graphs
    descs = scenes.config3_100k_mixed()
    w = World(max_bodies=len(descs) + 64); w.add_batch(descs)
    ts = []
    for s in range(400):
        t = time.perf_counter(); w.step(1 / 60); ts.append(time.perf_counter() - t)
    ts = 1e3 * np.array(ts)
    for lo, hi in ((0, 100), (100, 200), (200, 400)):
        seg = ts[lo:hi]
        print(f"SGP_NO_GRAPH={graphs} steps {lo}-{hi}: median {np.median(seg):.2f} ms, mean {seg.mean():.2f}, p99 {np.percentile(seg, 99):.2f}, max {seg.max():.2f}, steps > 1.5 x median: {(seg > 1.5 * np.median(seg)).sum()}")
    w.close()
