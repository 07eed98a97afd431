"""Regenerates substrata_amd/csrc/sgp_device_vehicle.h (device code, committed) from the formulas of oracle/sgo_vehicle.h.

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
#include "sgp_device_math.h"

'''


def main():
    s = open(os.path.join(ROOT, "oracle", "sgo_vehicle.h")).read()
    body = s[s.index('#define SGO_MAX_WHEELS'):s.rindex('#endif')]
    body = body.replace('sgo_', 'sgd_').replace('SGO_', 'SGD_')
    body = body.replace('static inline ', 'SGP_DEV static ')
    body = re.sub(r'v3_set\(&(\w+),', r'v3_set(\1,', body)
    body = re.sub(r'/\*(.*?)\*/', lambda m: '//' + m.group(1).rstrip() if '\n' not in m.group(1) else m.group(0), body)
    body = re.sub(r'typedef struct \{(.*?)\} (\w+);', lambda m: 'struct %s {%s};' % (m.group(2), m.group(1)), body, flags=re.S)
    open(os.path.join(ROOT, "substrata_amd", "csrc", "sgp_device_vehicle.h"), "w").write(HDR + body)


if __name__ == "__main__":
    main()
