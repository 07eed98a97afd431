"""An INDEPENDENT restatement of the step's dynamics, written from docs/CONTRACT.md in numpy / float64 and in a different form than the oracle and the device
code (which are near-textual twins of each other: VERDICT r05, what's weak 1): bodies carry 6-vectors and a 6 x 6 inverse mass matrix, a constraint row is a
1 x 12 Jacobian, impulses go through W J^T.  It takes the contact manifolds from the narrow phase (pinned on its own by tests/test_collide_independent.py,
and re-derived here in closed form for sphere pairs), and does by itself: forces and damping, constraint set-up (lever arms, speculative bias, restitution
with Jolt 5's gravity compensation, effective masses, friction basis), the contact-cache match of the cached impulses, the warm start, 10 velocity
iterations in (colour, priority) order with the friction cone, the exact axis-angle pose integration and 2 Baumgarte position iterations.

The oracle -- float32, plain C -- must agree with it after every step to float32 accuracy.  A formula error shared by the oracle and the device (one text
compiled twice) shows here.  What it takes from the oracle: each constraint's COLOUR (a discrete choice; checked to be a proper colouring)."""
import numpy as np
import pytest

from substrata_amd import abi, scenes

DT = 1.0 / 60.0
G = np.array([0.0, 0.0, -9.81])


def mix64(z):
    m = (1 << 64) - 1
    z = (z + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def rotate_by(q, w):
    """Body::AddRotationStep: the exact rotation by the vector w (angle |w| about w / |w|) applied on the left."""
    a = np.linalg.norm(w)
    if a <= 1.0e-6:
        return q
    dq = np.concatenate([w / a * np.sin(0.5 * a), [np.cos(0.5 * a)]])
    r = quat_mul(dq, q)
    return r / np.linalg.norm(r)


def perpendicular(n):
    """Vec3::GetNormalizedPerpendicular"""
    if abs(n[0]) > abs(n[1]):
        l = np.hypot(n[0], n[2]); return np.array([n[2] / l, 0.0, -n[0] / l])
    l = np.hypot(n[1], n[2]); return np.array([0.0, n[2] / l, -n[1] / l])


class Body:
    def __init__(self, d):
        self.pos = np.array(d["pos"], float); self.q = np.array(d["rot"], float)
        self.u = np.concatenate([np.array(d["lin_vel"], float), np.array(d["ang_vel"], float)])
        self.type = int(d["shape_type"]); self.shape = np.array(d["shape"], float)
        self.dynamic = int(d["motion_type"]) == abi.MOTION_DYNAMIC
        self.mass = max(0.001, float(d["mass"])); self.friction = min(1.0, max(0.0, float(d["friction"]))); self.restitution = min(1.0, max(0.0, float(d["restitution"])))
        self.gf = float(d["gravity_factor"]); self.ld = float(d["linear_damping"]); self.ad = float(d["angular_damping"])
        m = self.mass
        if self.type == abi.SHAPE_SPHERE:
            I = np.full(3, 0.4 * m * self.shape[0] ** 2)
        elif self.type == abi.SHAPE_BOX:
            sx, sy, sz = 2 * self.shape[:3]
            I = m / 12.0 * np.array([sy * sy + sz * sz, sx * sx + sz * sz, sx * sx + sy * sy])
        else:
            raise NotImplementedError
        self.inv_I_local = 1.0 / I

    def W(self):
        """6 x 6 inverse mass matrix in the world frame (zero for a body that cannot move)"""
        W = np.zeros((6, 6))
        if self.dynamic:
            R = quat_to_R(self.q)
            W[:3, :3] = np.eye(3) / self.mass
            W[3:, 3:] = R @ np.diag(self.inv_I_local) @ R.T
        return W


def row(a, r1, r2):
    """Jacobian of one axis: J u = (a . v1 + (r1 x a) . w1) - (a . v2 + (r2 x a) . w2)"""
    return np.concatenate([a, np.cross(r1, a), -a, -np.cross(r2, a)])


class Reference:
    def __init__(self, descs, settings):
        self.bodies = [Body(d) for d in descs]
        self.descs = descs
        self.st = settings
        self.prev = {}          # pair key -> list of (local1, local2, lam_n, lam_t1, lam_t2)
        self.bounces = 0        # contact points that took the restitution branch

    def desc_now(self, i):
        d = abi.BodyDesc()
        for f, _ in abi.BodyDesc._fields_:
            v = self.descs[i][f]
            if hasattr(getattr(d, f), "__len__"):
                getattr(d, f)[:] = [float(x) for x in v]
            else:
                setattr(d, f, v.item() if hasattr(v, "item") else v)
        b = self.bodies[i]
        d.pos[:] = [float(np.float32(x)) for x in b.pos]; d.rot[:] = [float(np.float32(x)) for x in b.q]
        return d

    def step(self, oracle_mod, colours):
        st, B = self.st, self.bodies
        # 1. forces, damping
        for b in B:
            if not b.dynamic:
                continue
            b.u[:3] += G * b.gf * DT
            b.u[:3] *= max(0.0, 1.0 - b.ld * DT); b.u[3:] *= max(0.0, 1.0 - b.ad * DT)
        # 2. contacts of every pair with a movable body (the narrow phase is the oracle's own pure function; sphere pairs are cross-checked in closed form)
        cons = []
        for i in range(len(B)):
            for j in range(i + 1, len(B)):
                if not (B[i].dynamic or B[j].dynamic):
                    continue
                m = oracle_mod.collide_pair(self.desc_now(i), self.desc_now(j), st["speculative_contact_distance"])
                if m is None:
                    continue
                n, p1, p2 = [np.array(x, float) for x in m]
                if B[i].type == abi.SHAPE_SPHERE and B[j].type == abi.SHAPE_SPHERE:
                    dd = B[j].pos - B[i].pos; nn = dd / np.linalg.norm(dd)
                    assert np.allclose(n, nn, atol=1e-5) and np.allclose(p1[0], B[i].pos + nn * B[i].shape[0], atol=1e-5) and np.allclose(p2[0], B[j].pos - nn * B[j].shape[0], atol=1e-5)
                cons.append(dict(a=i, b=j, key=(i << 32) | j, n=n, p1=p1, p2=p2))
        # 3. set-up
        for c in cons:
            A, Bb = B[c["a"]], B[c["b"]]
            RA, RB = quat_to_R(A.q), quat_to_R(Bb.q)
            c["mu"] = np.sqrt(A.friction * Bb.friction); e = max(A.restitution, Bb.restitution)
            c["t1"] = perpendicular(c["n"]); c["t2"] = np.cross(c["n"], c["t1"])
            c["W"] = np.zeros((12, 12)); c["W"][:6, :6] = A.W(); c["W"][6:, 6:] = Bb.W()
            c["pts"] = []
            old = self.prev.get(c["key"], []) if st["warm_start"] else []
            for p1, p2 in zip(c["p1"], c["p2"]):
                l1 = RA.T @ (p1 - A.pos); l2 = RB.T @ (p2 - Bb.pos)
                lam = np.zeros(3)
                for (o1, o2, ln, lt1, lt2) in old:
                    if np.sum((l1 - o1) ** 2) < st["contact_point_preserve_lambda_max_dist_sq"] and np.sum((l2 - o2) ** 2) < st["contact_point_preserve_lambda_max_dist_sq"]:
                        lam = np.array([ln, lt1, lt2]); break
                mid = 0.5 * (p1 + p2); r1 = mid - A.pos; r2 = mid - Bb.pos
                Jn, J1, J2 = row(c["n"], r1, r2), row(c["t1"], r1, r2), row(c["t2"], r1, r2)
                u = np.concatenate([A.u, Bb.u])
                vn = -(Jn @ u)                                            # velocity of the point on body 2 relative to the one on body 1, along n
                pen = (p1 - p2) @ c["n"]
                spec = max(0.0, -pen / DT)
                bias = spec
                if e > 0.0 and vn < -st["min_velocity_for_restitution"] and vn < -spec:
                    acc = (G * Bb.gf if Bb.dynamic else 0.0) - (G * A.gf if A.dynamic else 0.0)
                    bias = e * (vn - min(0.0, float(np.dot(acc, c["n"]))) * DT)
                    self.bounces += 1
                def eff(J):
                    k = J @ c["W"] @ J
                    return 1.0 / k if k > 0.0 else 0.0
                c["pts"].append(dict(l1=l1, l2=l2, lam=lam, Jn=Jn, J1=J1, J2=J2, bias=bias, en=eff(Jn), e1=eff(J1), e2=eff(J2)))
        # a proper colouring: no two constraints of a colour share a body that can move
        for c in cons:
            c["colour"] = colours[c["key"]]
        seen = set()
        for c in cons:
            for body in (c["a"], c["b"]):
                if B[body].dynamic:
                    assert (c["colour"], body) not in seen, "the oracle's colouring is not proper"
                    seen.add((c["colour"], body))
        cons.sort(key=lambda c: (c["colour"], mix64(c["key"])))

        def apply(c, J, dlam):
            du = -(c["W"] @ J) * dlam
            B[c["a"]].u += du[:6]; B[c["b"]].u += du[6:]

        # 4. warm start (the summed form and the part-by-part form are the same impulse)
        if st["warm_start"]:
            for c in cons:
                for p in c["pts"]:
                    if c["mu"] > 0.0:
                        apply(c, p["J1"], p["lam"][1]); apply(c, p["J2"], p["lam"][2])
                    apply(c, p["Jn"], p["lam"][0])
        # 5. velocity iterations: friction rows of every point (cone from the normal impulse so far), then the non-penetration rows
        for _ in range(st["num_velocity_steps"]):
            for c in cons:
                if c["mu"] > 0.0:
                    for p in c["pts"]:
                        if p["e1"] <= 0.0 and p["e2"] <= 0.0:
                            continue
                        u = np.concatenate([B[c["a"]].u, B[c["b"]].u])
                        l1 = p["lam"][1] + p["e1"] * (p["J1"] @ u); l2 = p["lam"][2] + p["e2"] * (p["J2"] @ u)
                        lim = c["mu"] * p["lam"][0]
                        if l1 * l1 + l2 * l2 > lim * lim:
                            s = lim / np.sqrt(l1 * l1 + l2 * l2); l1 *= s; l2 *= s
                        apply(c, p["J1"], l1 - p["lam"][1]); p["lam"][1] = l1
                        apply(c, p["J2"], l2 - p["lam"][2]); p["lam"][2] = l2
                for p in c["pts"]:
                    if p["en"] <= 0.0:
                        continue
                    u = np.concatenate([B[c["a"]].u, B[c["b"]].u])
                    nl = max(0.0, p["lam"][0] + p["en"] * (p["Jn"] @ u - p["bias"]))
                    apply(c, p["Jn"], nl - p["lam"][0]); p["lam"][0] = nl
        # 6. integrate
        for b in B:
            if b.dynamic:
                b.pos = b.pos + b.u[:3] * DT
                b.q = rotate_by(b.q, b.u[3:] * DT)
        # 7. position iterations
        for _ in range(st["num_position_steps"]):
            for c in cons:
                A, Bb = B[c["a"]], B[c["b"]]
                for p in c["pts"]:
                    RA, RB = quat_to_R(A.q), quat_to_R(Bb.q)
                    w1 = A.pos + RA @ p["l1"]; w2 = Bb.pos + RB @ p["l2"]
                    sep = (w2 - w1) @ c["n"] + st["penetration_slop"]
                    if sep >= 0.0:
                        continue
                    sep = max(sep, -st["max_penetration_distance"])
                    mid = 0.5 * (w1 + w2)
                    J = row(c["n"], mid - A.pos, mid - Bb.pos)
                    W = np.zeros((12, 12)); W[:6, :6] = A.W(); W[6:, 6:] = Bb.W()
                    k = J @ W @ J
                    if k <= 0.0:
                        continue
                    lam = -(1.0 / k) * st["baumgarte"] * sep
                    dx = -(W @ J) * lam
                    if A.dynamic:
                        A.pos = A.pos + dx[:3]; A.q = rotate_by(A.q, dx[3:6])
                    if Bb.dynamic:
                        Bb.pos = Bb.pos + dx[6:9]; Bb.q = rotate_by(Bb.q, dx[9:])
        self.prev = {c["key"]: [(p["l1"], p["l2"], p["lam"][0], p["lam"][1], p["lam"][2]) for p in c["pts"]] for c in cons}
        return cons


def scene(seed):
    """a ground box, a few boxes resting / tilted on it and on each other, spheres dropped among them with some spin and speed (restitution and friction at work)"""
    rng = np.random.default_rng(seed)
    g = scenes.ground(width=40.0, friction=0.6, restitution=0.2)
    n = 10
    d = scenes.dynamic_bodies(n, mass=10.0, friction=0.5, restitution=0.3)
    d["allow_sleeping"] = 0
    for k in range(n):
        if k < 4:
            d["shape_type"][k] = abi.SHAPE_BOX; d["shape"][k, :3] = rng.uniform(0.3, 0.6, 3)
            d["pos"][k] = (rng.uniform(-1.0, 1.0), rng.uniform(-1.0, 1.0), d["shape"][k, 2] - 0.005 + 1.25 * (k // 2) * 0.0)
            ang = rng.uniform(-0.4, 0.4); d["rot"][k] = (0, 0, np.sin(ang / 2), np.cos(ang / 2))
            d["pos"][k, 0] += 2.0 * (k - 1.5)
        else:
            r = rng.uniform(0.25, 0.45)
            d["shape_type"][k] = abi.SHAPE_SPHERE; d["shape"][k, 0] = r
            d["pos"][k] = (2.0 * (k - 5.5) * 0.6 + rng.uniform(-0.1, 0.1), rng.uniform(-0.3, 0.3), r + rng.uniform(-0.01, 0.3))
            d["lin_vel"][k] = rng.uniform(-2.0, 2.0, 3) * (1, 1, 0.5); d["ang_vel"][k] = rng.uniform(-3, 3, 3)
        d["mass"][k] = rng.uniform(5.0, 40.0)
    d["lin_vel"][4] = (0.5, 0.0, -3.0); d["pos"][4, 2] = d["shape"][4, 0] + 0.03      # one sphere hits the ground at 3 m/s: restitution
    d["lin_vel"][6] = (3.0, 0.0, 0.0); d["pos"][6] = d["pos"][1] + np.float32([-(d["shape"][1, 0] + d["shape"][6, 0] + 0.3), 0.0, 0.0]); d["pos"][6, 2] = d["shape"][6, 0] + 0.002      # one rolls into a box
    # two spheres that touch each other, one box resting on another
    d["pos"][9] = d["pos"][8] + np.float32([d["shape"][8, 0] + d["shape"][9, 0] - 0.004, 0, 0.0]); d["pos"][9, 2] = d["pos"][8, 2]
    d["pos"][3] = d["pos"][2] + np.float32([0.1, 0.05, d["shape"][2, 2] + d["shape"][3, 2] - 0.003]); d["rot"][3] = d["rot"][2]
    return np.concatenate([g, d])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_step_agrees_with_the_independent_restatement(oracle, seed):
    descs = scene(seed)
    # (the body-pair contact cache -- a resting box pair keeps last step's manifold instead of colliding again -- is switched off: the restatement collides
    #  every pair afresh; the cache has its own tests, test_body_pair_contact_cache_on_and_off)
    w = oracle.OracleWorld(max_bodies=64, settings=dict(use_body_pair_contact_cache=0))
    w.add_batch(descs)
    st = {f: getattr(w.desc.settings, f) for f, _ in abi.Settings._fields_}
    ref = Reference(descs, st)
    total_cons = 0; warm = 0
    for step in range(12):
        w.step(DT)
        dump = w.dump_constraints()
        colours = {(int(c["a"]) << 32) | int(c["b"]): int(c["colour"]) for c in dump}
        cons = ref.step(oracle, colours)
        assert len(cons) == len(dump)                                            # the same contact pairs
        total_cons += len(cons)
        S = w.read_states(0, len(descs))
        for i, b in enumerate(ref.bodies):
            # float32 against float64 over a step of ~ 100 dependent operations; velocities of a few m/s
            assert np.allclose(S["lin_vel"][i], b.u[:3], atol=3e-4), (step, i, S["lin_vel"][i], b.u[:3])
            assert np.allclose(S["ang_vel"][i], b.u[3:], atol=1.5e-3), (step, i, S["ang_vel"][i], b.u[3:])
            assert np.allclose(S["pos"][i], b.pos, atol=2e-5 * (step + 1)), (step, i)
            assert min(np.abs(S["rot"][i] - b.q).max(), np.abs(S["rot"][i] + b.q).max()) < 2e-5 * (step + 1), (step, i)
        # the accumulated impulses themselves
        by_key = {(int(c["a"]) << 32) | int(c["b"]): c for c in dump}
        for c in cons:
            dc = by_key[c["key"]]
            assert int(dc["np"]) == len(c["pts"])
            # (how a resting box's load is shared among its four points is statically indeterminate: rounding shows there a hundred times enlarged, while the
            #  manifold's total and every velocity agree -- so the single points of a multi-point manifold get the wider bound)
            tol = 2e-3 if len(c["pts"]) == 1 else 2e-2
            tot = max(1.0, sum(abs(p["lam"][0]) for p in c["pts"]))
            assert abs(sum(float(dc["lam_n"][k]) for k in range(len(c["pts"]))) - sum(p["lam"][0] for p in c["pts"])) < 2e-3 * tot, (step, c["a"], c["b"])
            for k, p in enumerate(c["pts"]):
                assert abs(float(dc["lam_n"][k]) - p["lam"][0]) < tol * tot and abs(float(dc["bias"][k]) - p["bias"]) < 2e-3 * max(1.0, abs(p["bias"])), (step, c["a"], c["b"], k)
                assert abs(float(dc["lam_t1"][k]) - p["lam"][1]) < tol * tot and abs(float(dc["lam_t2"][k]) - p["lam"][2]) < tol * tot
            warm += sum(1 for p in c["pts"] if step > 0 and abs(p["lam"][0]) > 0)
            # the next step's cached impulses are the oracle's (so that an indeterminate split does not drift apart over the steps)
            ref.prev[c["key"]] = [(p["l1"], p["l2"], float(dc["lam_n"][k]), float(dc["lam_t1"][k]), float(dc["lam_t2"][k])) for k, p in enumerate(c["pts"])]
        # keep the two from drifting apart through chaos: the reference continues from the oracle's float32 state
        for i, b in enumerate(ref.bodies):
            b.pos = np.array(S["pos"][i], float); b.q = np.array(S["rot"][i], float); b.u = np.concatenate([S["lin_vel"][i], S["ang_vel"][i]]).astype(float)
    assert total_cons >= 60 and warm >= 30 and ref.bounces >= 1                  # contacts, warm-started ones and a bounce were really exercised
    w.close()
