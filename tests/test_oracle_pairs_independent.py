"""WHICH pairs get a contact constraint, checked without the oracle's broad phase and without its collision formulas: after every step of a heap of boxes, spheres
and capsules coming to rest, the set of constrained pairs must be exactly the pairs of bodies whose shapes -- at the poses the step started from -- are closer
than the speculative contact distance (PhysicsSettings::mSpeculativeContactDistance = 0.02 m, /root/reference/gui_client/PhysicsWorld.cpp:1359 runs Jolt's defaults),
where "closer" is measured by brute force over support functions (tests/test_collide_independent.py: min over 20 000 directions of h_A(d) + h_B(-d)), every pair of
bodies considered (no grid, no tree).  Pairs inside a 3 mm band around the threshold may go either way (the reference's resolution).  This pins the pair set the
GPU suite then compares bit for bit: a pair the oracle's cell grid lost, or a hit / miss decision its collision routines share with their device twins, fails here
(checked by mutation: a neighbour cell left out of the grid walk, or a speculative distance of 12 mm, each fail it at the first checked step)."""
import numpy as np

from substrata_amd import abi, scenes
from helpers import DT
from test_collide_independent import Shape, overlap, min_overlap, _DIRS, MAX_SEP

BAND = 3.0e-3
# (+ the coordinate axes: against the 2 km ground quad only the exact vertical separates -- a direction a hair off it sees the quad's far corners)
DIRS = np.vstack([_DIRS, np.eye(3), -np.eye(3)])


def _shapes(states, descs):
    out = []
    for s, d in zip(states, descs):
        out.append(Shape(int(d["shape_type"]), tuple(float(x) for x in d["shape"][:3]), s["pos"].astype(float), tuple(float(x) for x in s["rot"])))
    return out


def _bound_radius(sh):
    if sh.kind == abi.SHAPE_SPHERE:
        return float(sh.p[0])
    if sh.kind == abi.SHAPE_BOX:
        return float(np.linalg.norm(sh.p[:3]))
    return float(sh.p[0] + sh.p[1])


def test_constrained_pairs_are_the_pairs_within_the_speculative_distance(oracle):
    rng = np.random.default_rng(17)
    n = 140
    d = scenes.dynamic_bodies(n)
    d["pos"][:, 0] = rng.uniform(-2.2, 2.2, n); d["pos"][:, 1] = rng.uniform(-2.2, 2.2, n); d["pos"][:, 2] = rng.uniform(0.6, 5.0, n)
    q = rng.normal(size=(n, 4)); d["rot"] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    kind = rng.integers(0, 3, n); sc = rng.uniform(0.25, 0.6, n)
    for i in range(n):
        if kind[i] == 0: d["shape_type"][i] = abi.SHAPE_BOX; d["shape"][i, :3] = sc[i] * rng.uniform(0.5, 1.0, 3)
        elif kind[i] == 1: d["shape_type"][i] = abi.SHAPE_SPHERE; d["shape"][i, :3] = (sc[i] * 0.7, 0, 0)
        else: d["shape_type"][i] = abi.SHAPE_CAPSULE; d["shape"][i, :3] = (sc[i] * 0.4, sc[i] * 0.7, 0)
    d["allow_sleeping"] = 0                                     # (everybody awake: every close pair has an active member)
    descs = np.concatenate([scenes.ground(), d])
    w = oracle.OracleWorld(max_bodies=512)
    ids = w.add_batch(descs)
    assert list(ids) == list(range(n + 1))
    checked = constrained = ambiguous = 0
    for step in range(150):
        check = step in (20, 45, 70, 100, 149)
        if check:
            st = w.read_states(0, n + 1)
        w.step(DT)
        if not check:
            continue
        got = {(int(c["a"]), int(c["b"])) for c in w.dump_constraints()}
        shp = _shapes(st, descs)
        rad = np.array([_bound_radius(s) for s in shp]); pos = np.array([s.pos for s in shp])
        must, may = set(), set()
        for a in range(n + 1):
            for b in range(a + 1, n + 1):
                if a != 0 and np.linalg.norm(pos[a] - pos[b]) > rad[a] + rad[b] + 0.1:
                    continue                                    # (spheres around the shapes apart: certainly no contact; the ground is tested against everybody)
                if a == 0 and pos[b][2] - rad[b] > 0.1:
                    continue
                sep = -float(overlap(shp[a], shp[b], DIRS).min())              # distance between the shapes (negative: penetration); sampled, so never ABOVE the true one
                if sep >= MAX_SEP + BAND:
                    checked += 1
                    continue
                if sep > MAX_SEP - 0.02:
                    # near the threshold the sampling is too coarse for polytopes (the gap has kinks: an error of the first order in the angle, ~5 mm): refine
                    sep = -float(min_overlap(shp[a], shp[b])[0])
                if sep < MAX_SEP - BAND:
                    must.add((a, b))
                elif sep < MAX_SEP + BAND:
                    may.add((a, b))
                checked += 1
        missing, extra = must - got, got - must - may
        assert not missing, (step, "pairs within the speculative distance without a constraint", sorted(missing)[:5])
        assert not extra, (step, "constraints between shapes that are further apart", sorted(extra)[:5])
        constrained += len(got); ambiguous += len(may)
    assert constrained > 600 and checked > 2000 and ambiguous < constrained // 10
    w.close()
