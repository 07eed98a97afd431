"""Tier 2 of the boundary, checked against the reference's own text: every `JPH::` name the six gui_client callers (and JoltUtils.h) use must
exist in the look-alike set under substrata_amd/shim/Jolt.  The names are extracted from /root/reference at test time, so this test only
runs where the reference tree is present (the authoring container); elsewhere it checks the committed list, which the first test keeps in
step with the tree."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "substrata_amd", "shim")
REF = "/root/reference/gui_client"
CALLER_FILES = ["PlayerPhysics.h", "PlayerPhysics.cpp", "CarPhysics.h", "CarPhysics.cpp", "BikePhysics.h", "BikePhysics.cpp", "HoverCarPhysics.h",
                "HoverCarPhysics.cpp", "BoatPhysics.h", "BoatPhysics.cpp", "ParticleManager.h", "ParticleManager.cpp", "VehiclePhysics.h", "JoltUtils.h"]
TOKEN = re.compile(r"JPH::[A-Za-z_][A-Za-z_0-9]*(?:::[A-Za-z_][A-Za-z_0-9]*)?")

# the committed list (sorted): what the extraction yields for the reference snapshot this repository was written against
COMMITTED = """JPH::Array JPH::Body JPH::BodyCreationSettings JPH::BodyFilter JPH::BodyID JPH::BodyInterface JPH::BodyLockRead JPH::BroadPhaseLayerFilter
JPH::CapsuleShape JPH::CharacterBase::EGroundState JPH::CharacterContactListener JPH::CharacterContactSettings JPH::CharacterVirtual
JPH::CharacterVirtual::EGroundState JPH::CharacterVirtual::ExtendedUpdateSettings JPH::CharacterVirtualSettings JPH::ConvexHullShapeSettings
JPH::DegreesToRadians JPH::EActivation::Activate JPH::EActivation::DontActivate JPH::EMotionType::Dynamic JPH::EOverrideMassProperties::CalculateInertia
JPH::ETransmissionMode::Manual JPH::Float4 JPH::IgnoreSingleBodyFilter JPH::JPH_PI JPH::Mat44 JPH::MotorcycleController JPH::MotorcycleControllerSettings
JPH::ObjectLayer JPH::ObjectLayerFilter JPH::OffsetCenterOfMassShapeSettings JPH::PhysicsMaterial JPH::PhysicsSystem JPH::Plane JPH::Quat
JPH::Quat::sIdentity JPH::Quat::sRotation JPH::RVec3 JPH::RVec3Arg JPH::RadiansToDegrees JPH::Ref JPH::RefConst JPH::RotatedTranslatedShapeSettings
JPH::Shape JPH::ShapeFilter JPH::Square JPH::SubShapeID JPH::TempAllocator JPH::Vec3 JPH::Vec3::sAxisX JPH::Vec3::sAxisY JPH::Vec3::sAxisZ JPH::Vec3::sZero
JPH::Vec3Arg JPH::Vec4 JPH::VehicleCollisionTester JPH::VehicleCollisionTesterCastCylinder JPH::VehicleCollisionTesterCastSphere JPH::VehicleConstraint
JPH::VehicleConstraintSettings JPH::Wheel JPH::WheelSettings JPH::WheelSettingsWV JPH::WheeledVehicleController JPH::WheeledVehicleControllerSettings""".split()

ALL_HEADERS = """#include <Jolt/Jolt.h>
#include <Jolt/Physics/PhysicsSystem.h>
#include <Jolt/Physics/Collision/ObjectLayer.h>
#include <Jolt/Physics/Collision/Shape/CapsuleShape.h>
#include <Jolt/Physics/Collision/Shape/RotatedTranslatedShape.h>
#include <Jolt/Physics/Collision/Shape/BoxShape.h>
#include <Jolt/Physics/Collision/Shape/OffsetCenterOfMassShape.h>
#include <Jolt/Physics/Collision/Shape/ConvexHullShape.h>
#include <Jolt/Physics/Body/BodyCreationSettings.h>
#include <Jolt/Physics/Vehicle/VehicleConstraint.h>
#include <Jolt/Physics/Vehicle/WheeledVehicleController.h>
#include <Jolt/Physics/Vehicle/MotorcycleController.h>
#include <Jolt/Physics/Character/Character.h>
#include <Jolt/Physics/Character/CharacterVirtual.h>
"""


def extract():
    names = set()
    for f in CALLER_FILES:
        with open(os.path.join(REF, f), errors="replace") as fh:
            names.update(TOKEN.findall(fh.read()))
    return sorted(names)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_committed_list_is_what_the_reference_uses():
    assert extract() == sorted(COMMITTED)


def test_every_jph_name_of_the_callers_exists_in_the_lookalikes():
    # one translation unit: each name must be usable as a type, as a template, or as a value / function (whichever it is in Jolt)
    probes = []
    for i, n in enumerate(sorted(COMMITTED)):
        probes.append(f"template <class T = void> struct P{i} {{"
                      f" template <class U = T> static auto a(int) -> decltype(sizeof(typename std::conditional<true, {n}, U>::type), 1);"      # a type
                      f" template <class U = T> static auto b(int) -> decltype((void)({n}), 1);"                                                 # a value / function / enumerator
                      f" }};")
    # SFINAE cannot see a non-dependent name fail, so each name is compiled on its own in the two roles and one of them has to pass;
    # templates (Ref, RefConst, Array, Square) are named with an argument
    templates = {"JPH::Ref": "JPH::Ref<JPH::Shape>", "JPH::RefConst": "JPH::RefConst<JPH::Shape>", "JPH::Array": "JPH::Array<int>", "JPH::Square": "JPH::Square<float>"}
    missing = []
    for n in sorted(COMMITTED):
        t = templates.get(n, n)
        ok = False
        for body in (f"typedef {t} probe_t;", f"void probe() {{ (void)({t}); }}", f"void probe() {{ (void)&{t}; }}"):
            r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", SHIM, "-x", "c++", "-"], input=ALL_HEADERS + body + "\n", text=True, capture_output=True)
            if r.returncode == 0:
                ok = True
                break
        if not ok:
            missing.append(n)
    assert not missing, missing
