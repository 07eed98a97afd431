#!/usr/bin/env python3
"""Where does a velocity-iteration launch spend its time?  Warms up BASELINE config 3, then times back-to-back launches of single
colours with parts of the kernel removed (sgp_debug_time_solve / k_solve_probe).  Run on the GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from substrata_amd import scenes                 # noqa: E402
from substrata_amd.lib import World, init, load as lib   # noqa: E402


def main():
    init()
    descs = scenes.config3_100k_mixed()
    w = World(max_bodies=len(descs) + 64)
    w.add_batch(descs)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
        w.step(1.0 / 60.0)
    st = w.stats()
    print(f"constraints {st.num_manifolds}, colours {st.num_colours}")
    fn = lib().sgp_debug_time_solve
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
    names = ["full", "loads + stores, no row arithmetic", "header + 2 body records", "colour range only"]
    print("| colour | constraints | " + " | ".join(names) + " |")
    print("|---|---|" + "---|" * len(names))
    for colour in (0, 4, 8, 12, 15):
        us = C.c_float(0); cnt = C.c_uint32(0)
        cells = []
        for v in (3, 2, 1, 0):
            rc = fn(w._h, v, colour, 200, C.byref(us), C.byref(cnt))
            assert rc == 0
            cells.append(f"{us.value:.2f}")
        print(f"| {colour} | {cnt.value} | " + " | ".join(reversed(cells)) + " |")


if __name__ == "__main__":
    main()
