"""Thin numpy/ctypes driver over the C ABI (include/sgp.h).

`CWorld` is ABI plumbing only: it owns no physics.  The product binds it to substrata_amd/libsgp.so (HIP, gfx950);
the test suite binds the same class to its CPU checker library (a different symbol prefix) so that parity tests
drive both sides through identical calls.
"""
import ctypes as C
import numpy as np
from . import abi


class SgpError(RuntimeError):
    pass


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f3(v):
    return np.ascontiguousarray(v, dtype=np.float32).reshape(3)


def _f4(v):
    return np.ascontiguousarray(v, dtype=np.float32).reshape(4)


class CWorld:
    def __init__(self, lib, prefix, max_bodies=65536, gravity=(0.0, 0.0, -9.81), device=0, settings=None,
                 max_body_pairs=0, max_manifolds=0, large_body_radius=0.0):
        self._lib, self._p = lib, prefix
        self._h = C.c_void_p()
        d = abi.WorldDesc()
        self._fn("default_world_desc")(C.byref(d))
        d.max_bodies = int(max_bodies)
        d.max_body_pairs = int(max_body_pairs)
        d.max_manifolds = int(max_manifolds)
        d.device = int(device)
        d.gravity[:] = gravity
        if large_body_radius:
            d.large_body_radius = float(large_body_radius)
        if settings:
            for k, v in settings.items():
                setattr(d.settings, k, v)
        self.desc = d
        self._check(self._fn("world_create")(C.byref(d), C.byref(self._h)), "world_create")
        self.max_bodies = int(max_bodies)

    # -- plumbing -------------------------------------------------------------------------------------------
    def _fn(self, name):
        return getattr(self._lib, self._p + name)

    def _check(self, rc, what):
        if rc != abi.OK:
            msg = ""
            fn = getattr(self._lib, self._p + "last_error", None)
            if fn is not None:
                m = fn()
                msg = m.decode() if m else ""
            raise SgpError(f"{self._p}{what} failed: rc={rc} {msg}")

    def close(self):
        if self._h:
            self._fn("world_destroy")(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- bodies ---------------------------------------------------------------------------------------------
    def default_body_desc(self):
        d = abi.BodyDesc()
        self._fn("default_body_desc")(C.byref(d))
        return d

    def add(self, desc):
        out = C.c_uint32(abi.INVALID_ID)
        rc = self._fn("body_add")(self._h, C.byref(desc), C.byref(out))
        if rc == abi.ERR_REJECTED:
            return abi.INVALID_ID
        self._check(rc, "body_add")
        return out.value

    def add_batch(self, descs):
        """descs: numpy structured array of abi.body_desc_dtype. Returns ids (uint32; INVALID_ID where rejected)."""
        descs = np.ascontiguousarray(descs, dtype=abi.body_desc_dtype)
        ids = np.empty(len(descs), dtype=np.uint32)
        self._check(self._fn("body_add_batch")(self._h, descs.ctypes.data, len(descs), ids.ctypes.data), "body_add_batch")
        return ids

    def add_compound(self, base_desc, children):
        """StaticCompoundShapeSettings::AddShape x n + Create on a static body (MeshBuilding.cpp:396-407).  base_desc: one record of
        abi.body_desc_dtype (pose, layer, material, userdata); children: array of abi.compound_child_dtype.  Returns the compound's id."""
        base = np.ascontiguousarray(base_desc, dtype=abi.body_desc_dtype).reshape(-1)[:1].copy()
        ch = np.ascontiguousarray(children, dtype=abi.compound_child_dtype)
        out = C.c_uint32(abi.INVALID_ID)
        self._check(self._fn("body_add_compound")(self._h, C.cast(base.ctypes.data, C.POINTER(abi.BodyDesc)), ch.ctypes.data, len(ch), C.byref(out)),
                    "body_add_compound")
        return out.value

    def compound_size(self, i):
        n = C.c_uint32(0)
        self._check(self._fn("body_compound_size")(self._h, int(i), C.byref(n)), "body_compound_size")
        return n.value

    def remove(self, i):
        self._check(self._fn("body_remove")(self._h, int(i)), "body_remove")

    def activate(self, i):
        self._check(self._fn("body_activate")(self._h, int(i)), "body_activate")

    def get_volume(self, i):
        v = C.c_float(0.0)
        self._check(self._fn("body_get_volume")(self._h, int(i), C.byref(v)), "body_get_volume")
        return v.value

    def set_layer(self, i, layer):
        self._check(self._fn("body_set_layer")(self._h, int(i), int(layer)), "body_set_layer")

    def set_pose_vel(self, i, pos, rot, lin_vel=(0, 0, 0), ang_vel=(0, 0, 0)):
        self._check(self._fn("body_set_pose_vel")(self._h, int(i), _fp(_f3(pos)), _fp(_f4(rot)), _fp(_f3(lin_vel)),
                                                  _fp(_f3(ang_vel))), "body_set_pose_vel")

    def set_pose_vel_batch(self, ids, recs):
        """Insert many physics snapshots at once (recs: structured array of abi.pose_vel_dtype)."""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        recs = np.ascontiguousarray(recs, dtype=abi.pose_vel_dtype)
        fn = getattr(self._lib, self._p + "body_set_pose_vel_batch", None)
        if fn is None:      # the CPU checker has no batched entry point
            for i, r in zip(ids, recs):
                self.set_pose_vel(int(i), r["pos"], r["rot"], r["lin_vel"], r["ang_vel"])
            return
        self._check(fn(self._h, ids.ctypes.data, recs.ctypes.data, len(ids)), "body_set_pose_vel_batch")

    def set_pose_shape(self, i, pos, rot, shape):
        self._check(self._fn("body_set_pose_shape")(self._h, int(i), _fp(_f3(pos)), _fp(_f4(rot)), _fp(_f4(shape))),
                    "body_set_pose_shape")

    def set_pos(self, i, pos):
        self._check(self._fn("body_set_pos")(self._h, int(i), _fp(_f3(pos))), "body_set_pos")

    def set_vel(self, i, lin_vel, ang_vel):
        self._check(self._fn("body_set_vel")(self._h, int(i), _fp(_f3(lin_vel)), _fp(_f3(ang_vel))), "body_set_vel")

    def move_kinematic(self, i, pos, rot, dt):
        self._check(self._fn("body_move_kinematic")(self._h, int(i), _fp(_f3(pos)), _fp(_f4(rot)), float(dt)),
                    "body_move_kinematic")

    def add_force(self, i, f):
        self._check(self._fn("body_add_force")(self._h, int(i), _fp(_f3(f))), "body_add_force")

    def add_force_at(self, i, f, p):
        self._check(self._fn("body_add_force_at")(self._h, int(i), _fp(_f3(f)), _fp(_f3(p))), "body_add_force_at")

    def add_torque(self, i, t):
        self._check(self._fn("body_add_torque")(self._h, int(i), _fp(_f3(t))), "body_add_torque")

    def get_state(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.zeros(len(ids), dtype=abi.body_state_dtype)
        self._check(self._fn("body_get_state")(self._h, ids.ctypes.data, len(ids), out.ctypes.data), "body_get_state")
        return out

    def read_states(self, first=0, n=None):
        n = self.max_bodies - first if n is None else n
        out = np.zeros(n, dtype=abi.body_state_dtype)
        self._check(self._fn("world_read_states")(self._h, int(first), int(n), out.ctypes.data), "world_read_states")
        return out

    def read_active(self, cap=None, out=None):
        """States of the active bodies (the application's per-frame read-back).  `out`: a reusable caller-owned record array."""
        if out is not None:
            cap = len(out)
        else:
            cap = self.max_bodies if cap is None else cap
            out = np.zeros(cap, dtype=abi.body_state_dtype)
        n = C.c_uint32(0)
        self._check(self._fn("world_read_active")(self._h, out.ctypes.data, int(cap), C.byref(n)), "world_read_active")
        return out[:min(n.value, cap)]

    def read_active_view(self):
        """States of the active bodies as a read-only view of the library's pinned host buffer (no copy into a caller buffer);
        valid until the next read-back call on this world."""
        ptr = C.c_void_p(0)
        n = C.c_uint32(0)
        self._check(self._fn("world_read_active_view")(self._h, C.byref(ptr), C.byref(n)), "world_read_active_view")
        if n.value == 0:
            return np.zeros(0, dtype=abi.body_state_dtype)
        buf = (C.c_char * (n.value * abi.body_state_dtype.itemsize)).from_address(ptr.value)
        a = np.frombuffer(buf, dtype=abi.body_state_dtype, count=n.value)
        a.flags.writeable = False
        return a

    def read_active_poses_view(self):
        """Id, position and rotation of the active bodies (32-byte records) as a read-only view of the library's pinned host buffer;
        valid until the next read-back call on this world."""
        ptr = C.c_void_p(0)
        n = C.c_uint32(0)
        self._check(self._fn("world_read_active_poses_view")(self._h, C.byref(ptr), C.byref(n)), "world_read_active_poses_view")
        if n.value == 0:
            return np.zeros(0, dtype=abi.body_pose_dtype)
        buf = (C.c_char * (n.value * abi.body_pose_dtype.itemsize)).from_address(ptr.value)
        a = np.frombuffer(buf, dtype=abi.body_pose_dtype, count=n.value)
        a.flags.writeable = False
        return a

    # -- world ----------------------------------------------------------------------------------------------
    def set_water(self, enabled, z):
        self._check(self._fn("world_set_water")(self._h, int(bool(enabled)), float(z)), "world_set_water")

    def set_contact_events(self, enabled):
        self._check(self._fn("world_set_contact_events")(self._h, int(bool(enabled))), "world_set_contact_events")

    def step(self, dt=1.0 / 60.0):
        self._check(self._fn("world_step")(self._h, float(dt)), "world_step")

    def step_n(self, dt, n):
        self._check(self._fn("world_step_n")(self._h, float(dt), int(n)), "world_step_n")

    def step_profiled(self, dt=1.0 / 60.0):
        p = abi.StepProfile()
        self._check(self._fn("world_step_profiled")(self._h, float(dt), C.byref(p)), "world_step_profiled")
        return p

    def stats(self):
        s = abi.StepStats()
        self._check(self._fn("world_stats")(self._h, C.byref(s)), "world_stats")
        return s

    def launch_counts(self):
        """(graph replays, eager steps, idle steps) so far -- product library only."""
        g, e, i = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._check(self._fn("world_launch_counts")(self._h, C.byref(g), C.byref(e), C.byref(i)), "world_launch_counts")
        return g.value, e.value, i.value

    def num_bodies(self):
        n = C.c_uint32(0)
        self._check(self._fn("world_num_bodies")(self._h, C.byref(n)), "world_num_bodies")
        return n.value

    def event_counts(self):
        """Events waiting per kind (index = abi.EVENT_*), nothing drained -- product library only."""
        c = (C.c_uint32 * 5)()
        self._check(self._fn("world_event_counts")(self._h, c), "world_event_counts")
        return list(c)

    def drain_events(self, kind, cap=1 << 16):
        dt = abi.body_event_dtype if kind <= abi.EVENT_ENTERED_WATER else abi.contact_event_dtype
        n = C.c_uint32(0)
        while True:
            out = np.zeros(cap, dtype=dt)
            self._check(self._fn("world_drain_events")(self._h, int(kind), out.ctypes.data, int(cap), C.byref(n)),
                        "world_drain_events")
            return out[:min(n.value, cap)]

    def raycast(self, rays):
        rays = np.ascontiguousarray(rays, dtype=abi.ray_dtype)
        hits = np.zeros(len(rays), dtype=abi.hit_dtype)
        self._check(self._fn("raycast")(self._h, rays.ctypes.data, len(rays), hits.ctypes.data), "raycast")
        return hits

    # -- static triangle meshes -------------------------------------------------------------------------------------
    def mesh_create(self, vertices, triangles, materials=None):
        """MeshShapeSettings(vertices, triangles).Create(): returns abi.MeshInfo (mesh_id for static body descs).  `materials`: one
        user-data word per triangle (the material index ray hits report)."""
        v = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3)
        info = abi.MeshInfo()
        if materials is None:
            self._check(self._fn("mesh_create")(self._h, v.ctypes.data, len(v), t.ctypes.data, len(t), C.byref(info)), "mesh_create")
        else:
            m = np.ascontiguousarray(materials, dtype=np.uint32).reshape(len(t))
            self._check(self._fn("mesh_create_with_materials")(self._h, v.ctypes.data, len(v), t.ctypes.data, len(t), m.ctypes.data, C.byref(info)),
                        "mesh_create_with_materials")
        return info

    def mesh_destroy(self, mesh_id):
        self._check(self._fn("mesh_destroy")(self._h, int(mesh_id)), "mesh_destroy")

    def hull_destroy(self, hull_id):
        self._check(self._fn("hull_destroy")(self._h, int(hull_id)), "hull_destroy")

    # -- convex hulls ---------------------------------------------------------------------------------------------
    def hull_create(self, points, com_offset=None):
        """ConvexHullShapeSettings(points).Create() (wrapped in OffsetCenterOfMassShape when com_offset is given): returns
        abi.HullInfo (hull_id for body descs, com / rot = body frame in the frame of the points)."""
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        info = abi.HullInfo()
        if com_offset is None:
            self._check(self._fn("hull_create")(self._h, pts.ctypes.data, len(pts), C.byref(info)), "hull_create")
        else:
            self._check(self._fn("hull_create_com")(self._h, pts.ctypes.data, len(pts), _fp(_f3(com_offset)), C.byref(info)), "hull_create_com")
        return info

    # -- wheeled vehicles ---------------------------------------------------------------------------------------
    def default_vehicle_desc(self, body=None):
        d = abi.VehicleDesc()
        self._fn("default_vehicle_desc")(C.byref(d))
        if body is not None:
            d.body = int(body)
        return d

    def vehicle_create(self, desc):
        out = C.c_uint32(abi.INVALID_ID)
        self._check(self._fn("vehicle_create")(self._h, C.byref(desc), C.byref(out)), "vehicle_create")
        return out.value

    def vehicle_destroy(self, vid):
        self._check(self._fn("vehicle_destroy")(self._h, int(vid)), "vehicle_destroy")

    def vehicle_set_input(self, vid, forward=0.0, right=0.0, brake=0.0, hand_brake=0.0):
        i = abi.VehicleInput(float(forward), float(right), float(brake), float(hand_brake))
        self._check(self._fn("vehicle_set_input")(self._h, int(vid), C.byref(i)), "vehicle_set_input")

    def vehicle_set_inputs(self, first, inputs):
        inputs = np.ascontiguousarray(inputs, dtype=abi.vehicle_input_dtype)
        self._check(self._fn("vehicle_set_inputs")(self._h, int(first), len(inputs), inputs.ctypes.data), "vehicle_set_inputs")

    def vehicle_get_state(self, vid):
        return self.vehicle_get_states(vid, 1)[0]

    def vehicle_get_states(self, first, n):
        out = np.zeros(n, dtype=abi.vehicle_state_dtype)
        self._check(self._fn("vehicle_get_states")(self._h, int(first), int(n), out.ctypes.data), "vehicle_get_states")
        return out

    def vehicle_enable_lean_controller(self, vid, enabled=True):
        self._check(self._fn("vehicle_enable_lean_controller")(self._h, int(vid), int(bool(enabled))), "vehicle_enable_lean_controller")

    def vehicle_reset_drivetrain(self, vid, engine_rpm=0.0, wheel_angular_velocity=0.0):
        self._check(self._fn("vehicle_reset_drivetrain")(self._h, int(vid), float(engine_rpm), float(wheel_angular_velocity)),
                    "vehicle_reset_drivetrain")

    def collide_capsules(self, queries, cap=4096):
        queries = np.ascontiguousarray(queries, dtype=abi.capsule_query_dtype)
        out = np.zeros(cap, dtype=abi.query_contact_dtype)
        n = C.c_uint32(0)
        self._check(self._fn("collide_capsules")(self._h, queries.ctypes.data, len(queries), out.ctypes.data, int(cap), C.byref(n)), "collide_capsules")
        return out[:min(n.value, cap)]

    def spherecast(self, rays, radii):
        rays = np.ascontiguousarray(rays, dtype=abi.ray_dtype)
        radii = np.ascontiguousarray(np.broadcast_to(np.asarray(radii, dtype=np.float32), (len(rays),)), dtype=np.float32)
        hits = np.zeros(len(rays), dtype=abi.hit_dtype)
        self._check(self._fn("spherecast")(self._h, rays.ctypes.data, radii.ctypes.data, len(rays), hits.ctypes.data), "spherecast")
        return hits

    def dump_constraints(self, cap=None):
        cap = (8 * self.max_bodies + 1024) if cap is None else cap
        out = np.zeros(cap, dtype=abi.constraint_dump_dtype)
        n = C.c_uint32(0)
        self._check(self._fn("world_dump_constraints")(self._h, out.ctypes.data, int(cap), C.byref(n)),
                    "world_dump_constraints")
        return out[:min(n.value, cap)]

    def export_boundary(self, lo, hi, margin, cap=1 << 16):
        out = np.empty(cap, dtype=abi.ghost_dtype)          # only the records the call fills in are returned
        n = C.c_uint32(0)
        self._check(self._fn("world_export_boundary")(self._h, _fp(_f3(lo)), _fp(_f3(hi)), float(margin),
                                                      out.ctypes.data, int(cap), C.byref(n)), "world_export_boundary")
        return out[:min(n.value, cap)]

    def import_ghosts(self, recs):
        recs = np.ascontiguousarray(recs, dtype=abi.ghost_dtype)
        self._check(self._fn("world_import_ghosts")(self._h, recs.ctypes.data, len(recs)), "world_import_ghosts")
