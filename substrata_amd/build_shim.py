"""Builds the C++ PhysicsWorld / PhysicsObject facade (host code, g++) into substrata_amd/libsgp_shim.so, linked against
libsgp.so.  This is the drop-in layer a Substrata build compiles instead of gui_client/PhysicsWorld.cpp + Jolt."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "shim")
LIB = os.path.join(HERE, "libsgp_shim.so")
SOURCES = ["PhysicsWorld.cpp", "PhysicsObject.cpp"]


def flags():
    return ["-O2", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-parameter", "-I", SHIM]


def build(force=False, verbose=False):
    srcs = [os.path.join(SHIM, s) for s in SOURCES]
    deps = srcs + [os.path.join(SHIM, h) for h in ("PhysicsWorld.h", "PhysicsObject.h", "Jolt/JoltLite.h", "Jolt/JoltVehicleLite.h", "Jolt/JoltCharacterLite.h")]
    deps += [os.path.join(HERE, "..", "include", "sgp.h"), os.path.join(HERE, "libsgp.so")]      # the C ABI's structs are compiled into the facade
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    cmd = ["g++"] + flags() + ["-shared"] + srcs + ["-o", LIB, "-L", HERE, "-lsgp", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("g++ failed building libsgp_shim.so")
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
