"""File formats of oracle/jolt_ref/oracle_jolt.cpp (the real-Jolt oracle, built only where a JoltPhysics v5.3.0 checkout exists):
scene file = 'SGPJ', n, n x sgp_body_desc; dump file = 'SGPD', n, n_checkpoints, then per checkpoint (step, n x sgp_body_state)."""
import json
import os
import subprocess

import numpy as np

from substrata_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BINARY = os.path.join(ROOT, "oracle", "_ref", "oracle_jolt")


def available():
    return os.path.exists(BINARY) and os.access(BINARY, os.X_OK)


def write_scene(path, descs):
    descs = np.ascontiguousarray(descs, dtype=abi.body_desc_dtype)
    with open(path, "wb") as f:
        f.write(np.array([0x4A504753, len(descs)], dtype=np.uint32).tobytes())
        f.write(descs.tobytes())


def read_dump(path):
    raw = open(path, "rb").read()
    magic, n, ncp = np.frombuffer(raw[:12], dtype=np.uint32)
    assert magic == 0x44504753, "not an oracle_jolt dump"
    out, off, rec = {}, 12, abi.body_state_dtype.itemsize
    for _ in range(int(ncp)):
        step = int(np.frombuffer(raw[off:off + 4], dtype=np.uint32)[0]); off += 4
        out[step] = np.frombuffer(raw[off:off + rec * int(n)], dtype=abi.body_state_dtype).copy(); off += rec * int(n)
    return out


def run(descs, steps, checkpoints, tmp_dir, dt=1.0 / 60.0, threads=0, extra=()):
    """Steps `descs` through real Jolt.  Returns ({step: states}, timing dict)."""
    scene, dump = os.path.join(tmp_dir, "scene.bin"), os.path.join(tmp_dir, "dump.bin")
    write_scene(scene, descs)
    cmd = [BINARY, scene, dump, "--steps", str(steps), "--dt", repr(float(dt)), "--checkpoints", ",".join(str(c) for c in checkpoints)]
    if threads:
        cmd += ["--threads", str(threads)]
    r = subprocess.run(cmd + list(extra), capture_output=True, text=True, check=True)
    return read_dump(dump), json.loads(r.stdout.strip().splitlines()[-1])
