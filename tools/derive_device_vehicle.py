"""Regenerates substrata_amd/csrc/sgp_device_vehicle.h, sgp_device_hull.h (device code) and sgp_hull_build.h (host code), all
committed, from the formulas of oracle/sgo_vehicle.h, sgo_hull.h and sgo_hull_build.h.

The per-vehicle arithmetic has to be the same expression tree on both sides for the bit-exact parity tests, so the device header
is produced by a mechanical rewrite (prefix sgo_ -> sgd_, C `static inline` -> __device__, C structs -> C++ structs) instead of
by hand.  The result is an independent file: the product never includes anything under oracle/.  Run after editing the oracle
header:  python tools/derive_device_vehicle.py"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = '''// sgp_device_vehicle.h -- gfx950 wheeled vehicle constraint: per-vehicle arithmetic (device code only).
//
// Role of JPH::VehicleConstraint + JPH::WheeledVehicleController / MotorcycleController + the sphere-cast collision tester behind
// CarPhysics and BikePhysics (/root/reference/gui_client/CarPhysics.cpp:62,94-231; BikePhysics.cpp:124-227; defaults
// /root/reference/gui_client/Scripting.cpp:315-346): per wheel one sphere cast along the suspension, tyre slip -> friction,
// engine / clutch / gearbox / differential, brakes, anti-roll bars, then 4 axis rows per wheel (soft suspension spring, hard
// max-up stop, longitudinal, lateral) and, for motorcycles, the lean spring.
// One wave owns one vehicle; vehicles never share a chassis and treat the body under a wheel as kinematic (its contact
// point velocity is sampled at cast time), so the vehicle phases need no colouring.
// The arithmetic (expression order included) is the contract checked by tests/test_vehicle_parity_gpu.py against the CPU
// oracle; no libm call sits on this path (polynomial sin/cos/acos).  Regenerate with tools/derive_device_vehicle.py.
#pragma once
#include "sgp_device_collide.h"     // sgd_hull (wheel casts against hull bodies)

'''


HULL_HDR = '''// sgp_device_hull.h -- gfx950 convex hull shapes: hull - hull / box / sphere / capsule manifolds, rays (device code only).
//
// Role of JPH::ConvexHullShape in CollideShape / CastRay for the dynamic meshes and vehicle bodies Substrata creates
// (/root/reference/gui_client/PhysicsWorld.cpp:735-1166 with is_dynamic, CarPhysics.cpp:66-92, BikePhysics.cpp:76-112): separating
// axis test over face normals and edge pairs + reference / incident face clipping (<= 4 points) instead of GJK/EPA.
// A hull is stored in its body frame (origin = centre of mass, axes = principal axes); a box is the +-1 cube template scaled.
// Included by sgp_device_collide.h after sgd_manifold / sgd_closest_on_segment.  Regenerate with tools/derive_device_vehicle.py.
#pragma once
#include "sgp_device_math.h"

'''

MESH_HDR = '''// sgp_device_mesh.h -- gfx950 static triangle meshes: per-triangle collision (a triangle is a thin 3-vertex hull), grouping of the
// triangle manifolds of one body pair by normal (<= 3 groups, <= 4 points each), ray - triangle (device code only).
//
// Role of JPH::MeshShape / HeightFieldShape in CollideShape / CastRay for Substrata's static meshes and terrain
// (/root/reference/gui_client/PhysicsWorld.cpp:735-1166 with is_dynamic = false, :1020-1120; TerrainSystem.cpp:1300).
// Included after sgp_device_collide.h.  Regenerate with tools/derive_device_vehicle.py.
#pragma once
#include "sgp_device_vehicle.h"     // sgd_ray_sphere / sgd_ray_capsule_z, sgd_hull through sgp_device_collide.h

'''

BUILD_HDR = '''// sgp_hull_build.h -- HOST side of the convex hull shapes: hull from a point cloud, volume / centre of mass / inertia, body frame.
//
// Role of JPH::ConvexHullShapeSettings::Create + MassProperties (+ OffsetCenterOfMassShape) (/root/reference/gui_client/
// CarPhysics.cpp:66-92, BikePhysics.cpp:76-112): brute-force supporting planes for <= 32 hull vertices, signed-tetrahedra mass
// properties, Jacobi principal axes; double precision, rounded to float once.  Regenerate with tools/derive_device_vehicle.py.
#pragma once
#include <math.h>
#include <string.h>
#include "sgp_device_collide.h"     // sgd_hull

static inline v3 sgh_v3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }

'''


def rewrite(body):
    body = body.replace('sgo_', 'sgd_').replace('SGO_', 'SGD_')
    body = body.replace('static inline ', 'SGP_DEV static ')
    body = re.sub(r'v3_set\(&(\w+),', r'v3_set(\1,', body)
    body = re.sub(r'/\*(.*?)\*/', lambda m: '//' + m.group(1).rstrip() if '\n' not in m.group(1) else m.group(0), body)
    body = re.sub(r'typedef struct \{(.*?)\} (\w+);', lambda m: 'struct %s {%s};' % (m.group(2), m.group(1)), body, flags=re.S)
    return body


def main():
    csrc = os.path.join(ROOT, "substrata_amd", "csrc")
    s = open(os.path.join(ROOT, "oracle", "sgo_vehicle.h")).read()
    body = rewrite(s[s.index('#define SGO_MAX_WHEELS'):s.rindex('#endif')])
    open(os.path.join(csrc, "sgp_device_vehicle.h"), "w").write(HDR + body)
    s = open(os.path.join(ROOT, "oracle", "sgo_hull.h")).read()
    body = rewrite(s[s.index('#define SGO_HULL_MAX_VERTS'):s.rindex('#endif')])
    open(os.path.join(csrc, "sgp_device_hull.h"), "w").write(HULL_HDR + body)
    s = open(os.path.join(ROOT, "oracle", "sgo_mesh.h")).read()
    body = rewrite(s[s.index('#define SGO_MESH_MAX_GROUPS'):s.rindex('#endif')])
    body = re.sub(r'->p\[(\d)\]', r'->p\1', body)          # the device shape record names its parameters p0, p1, p2
    open(os.path.join(csrc, "sgp_device_mesh.h"), "w").write(MESH_HDR + body)
    # host-side builder: plain host functions (no __device__), host constructor for v3
    s = open(os.path.join(ROOT, "oracle", "sgo_hull_build.h")).read()
    body = s[s.index('typedef struct { double x, y, z; } sgo_d3;'):s.rindex('#endif')]
    body = body.replace('sgo_hull', 'sgd_hull').replace('SGO_HULL', 'SGD_HULL').replace('sgo_', 'sgh_')
    body = body.replace('V3(', 'sgh_v3(')
    body = re.sub(r'/\*(.*?)\*/', lambda m: '//' + m.group(1).rstrip() if '\n' not in m.group(1) else m.group(0), body)
    body = re.sub(r'typedef struct \{(.*?)\} (\w+);', lambda m: 'struct %s {%s};' % (m.group(2), m.group(1)), body, flags=re.S)
    open(os.path.join(csrc, "sgp_hull_build.h"), "w").write(BUILD_HDR + body)


if __name__ == "__main__":
    main()
