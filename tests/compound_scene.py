"""A Substrata portal as MeshBuilding::makePortalMeshes builds it (MeshBuilding.cpp:377-414): a static compound of the arch mesh (child 0)
and a thin box across the opening (child 1, half extents (0.5, 0.06, 1) at (0, 0, 1)).  The arch here is a synthetic mesh -- two posts
and a lintel, outward-facing triangles -- since portal.bmesh is not part of the tree."""
import numpy as np

from substrata_amd import abi, scenes


def box_mesh(lo, hi):
    lo, hi = np.asarray(lo, np.float32), np.asarray(hi, np.float32)
    V = np.array([(x, y, z) for z in (lo[2], hi[2]) for y in (lo[1], hi[1]) for x in (lo[0], hi[0])], np.float32)
    quads = [(0, 2, 3, 1), (4, 5, 7, 6), (0, 1, 5, 4), (2, 6, 7, 3), (0, 4, 6, 2), (1, 3, 7, 5)]          # outward-facing
    T = []
    for a, b, c, d in quads:
        T += [(a, b, c), (a, c, d)]
    return V, np.array(T, np.uint32)


def arch_mesh():
    parts = [box_mesh((-0.9, -0.15, 0.0), (-0.55, 0.15, 2.2)), box_mesh((0.55, -0.15, 0.0), (0.9, 0.15, 2.2)), box_mesh((-0.9, -0.15, 2.2), (0.9, 0.15, 2.6))]
    V = np.concatenate([p[0] for p in parts])
    T = np.concatenate([p[1] + 8 * i for i, p in enumerate(parts)])
    mats = np.repeat(np.uint32([0, 1, 2]), 12)
    return V, T.astype(np.uint32), mats


def portal_children(mesh_id):
    ch = np.zeros(2, dtype=abi.compound_child_dtype)
    ch["rot"][:, 3] = 1.0
    ch["shape_type"][0] = abi.SHAPE_MESH; ch["shape"][0, 0] = float(mesh_id)
    ch["shape_type"][1] = abi.SHAPE_BOX; ch["shape"][1, :3] = (0.5, 0.06, 1.0); ch["pos"][1] = (0.0, 0.0, 1.0)
    return ch


def add_portal(w, pos, rot=(0, 0, 0, 1), userdata=77):
    V, T, mats = arch_mesh()
    info = w.mesh_create(V, T, materials=mats)
    base = scenes._blank(1)
    base["pos"][0] = pos; base["rot"][0] = rot; base["userdata"] = userdata; base["friction"] = 0.5; base["restitution"] = 0.0
    return w.add_compound(base, portal_children(info.mesh_id)), info
