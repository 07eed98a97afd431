"""Drive the HIP product and the CPU oracle through identical calls and compare their states."""
import numpy as np
from substrata_amd import abi

FIELDS = ("pos", "rot", "lin_vel", "ang_vel")


class Twin:
    """Applies every call to both worlds."""

    def __init__(self, gpu, cpu):
        self.gpu, self.cpu = gpu, cpu

    def __getattr__(self, name):
        fg, fc = getattr(self.gpu, name), getattr(self.cpu, name)

        def both(*a, **k):
            rg = fg(*a, **k)
            rc = fc(*a, **k)
            return rg, rc
        return both

    def close(self):
        self.gpu.close()
        self.cpu.close()


def make_twin(oracle, **kw):
    from substrata_amd.lib import World
    return Twin(World(**kw), oracle.OracleWorld(**kw))


def state_diff(sg, sc):
    """Max abs difference per field over bodies alive in both; also whether everything is bit-identical."""
    assert np.array_equal(sg["id"], sc["id"]), "live id sets differ"
    live = sg["id"] != abi.INVALID_ID
    out = {}
    exact = True
    for f in FIELDS:
        a, b = sg[f][live], sc[f][live]
        if f == "rot":   # q and -q are the same rotation
            sign = np.sign(np.sum(a * b, axis=1, keepdims=True))
            sign[sign == 0] = 1
            b = b * sign
        out[f] = float(np.max(np.abs(a - b))) if a.size else 0.0
        exact = exact and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    out["active_mismatch"] = int(np.sum(sg["active"][live] != sc["active"][live]))
    out["bit_exact"] = bool(exact)
    return out


def compare(twin, n):
    sg = twin.gpu.read_states(0, n)
    sc = twin.cpu.read_states(0, n)
    return state_diff(sg, sc)


def constraint_sets(twin):
    cg = twin.gpu.dump_constraints()
    cc = twin.cpu.dump_constraints()
    return cg, cc


def check_colouring_valid(cons, movable):
    """No two constraints of one (non-overflow) colour share a movable body."""
    seen = set()
    for c in cons:
        col = int(c["colour"])
        if col == 63:
            continue
        for b in (int(c["a"]), int(c["b"])):
            if movable[b]:
                key = (col, b)
                if key in seen:
                    return False
                seen.add(key)
    return True
