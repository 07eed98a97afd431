"""Differential check of capsule - mesh contacts, device against oracle: small box meshes at random orientations, capsules around them, one step,
the constraint lists compared.   PYTHONPATH=. python tools/experiments/tri_capsule_diff.py [seed]"""
import sys; sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np
from substrata_amd import abi, scenes
import parity
import oracle as oracle_mod
import compound_scene as cs_
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
tw = parity.make_twin(oracle_mod, max_bodies=8192)
NM, PER = 200, 5
bv, bt = cs_.box_mesh((-0.6, -0.4, 0.0), (0.6, 0.4, 0.9))
ig, ic = tw.mesh_create(bv, bt)
d = scenes._blank(NM); d["shape_type"] = abi.SHAPE_MESH; d["shape"][:] = 0; d["shape"][:, 0] = float(ig.mesh_id)
cen = np.column_stack([(np.arange(NM) % 15) * 6.0, (np.arange(NM) // 15) * 6.0, np.full(NM, 3.0)])
d["pos"] = cen
q = rng.normal(size=(NM, 4)); d["rot"] = q / np.linalg.norm(q, axis=1, keepdims=True)
mg, mc = tw.add_batch(d)
b = scenes.dynamic_bodies(NM * PER)
b["shape_type"] = abi.SHAPE_CAPSULE; b["shape"][:, 0] = rng.uniform(0.15, 0.4, NM * PER); b["shape"][:, 1] = rng.uniform(0.2, 0.8, NM * PER)
b["pos"] = np.repeat(cen, PER, axis=0) + rng.normal(size=(NM * PER, 3)) * 0.55
q = rng.normal(size=(NM * PER, 4)); b["rot"] = q / np.linalg.norm(q, axis=1, keepdims=True)
b["gravity_factor"] = 0.0
tw.add_batch(b)
tw.step(1 / 60)
cg = tw.gpu.dump_constraints(); cc = tw.cpu.dump_constraints()
kg = {(int(c["a"]), int(c["b"])): c for c in cg}; kc = {(int(c["a"]), int(c["b"])): c for c in cc}
print(len(cg), len(cc), "only gpu", sorted(set(kg) - set(kc))[:10], "only cpu", sorted(set(kc) - set(kg))[:10])
bad = 0
for k in sorted(set(kg) & set(kc)):
    g, c = kg[k], kc[k]
    diff = [f for f in cg.dtype.names if not np.array_equal(np.atleast_1d(g[f]).view(np.uint8), np.atleast_1d(c[f]).view(np.uint8))]
    if diff:
        bad += 1
        if bad <= 5: print("DIFF", k, {f: (g[f], c[f]) for f in diff})
print("pairs compared", len(set(kg) & set(kc)), "different", bad)
