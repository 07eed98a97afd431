"""Tier 2 of the boundary, checked by USE rather than by name: every member the six gui_client callers apply to an
object of a JPH:: look-alike type -- `obj.Member(`, `ptr->Member(`, `settings.mField` -- must exist in that look-alike class.

How: the callers' text (comments stripped) is scanned for declarations `JPH::Type [*&] name` / `JPH::Ref<JPH::Type> name`, which gives a
variable -> class table, and for `name.Member` / `name->Member` with a Jolt-style member (`UpperCamel`, `mField`, `sStatic`); members reached
through a call chain (`GetWheel(i)->GetSettings()->mRadius`) are attributed by CHAINED_OWNER below.  Each (class, member) pair is then
compiled as a probe against substrata_amd/shim/Jolt: a class has a member named M iff deriving from it AND from a struct that declares M
makes `&Derived::M` ambiguous (works for overloaded and for data members alike).  Nothing of the reference is copied: the committed list
holds (class, member) names only, and the first test keeps it in step with the tree where the tree exists."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "substrata_amd", "shim")
REF = "/root/reference/gui_client"
FILES = ["PlayerPhysics.h", "PlayerPhysics.cpp", "CarPhysics.h", "CarPhysics.cpp", "BikePhysics.h", "BikePhysics.cpp", "HoverCarPhysics.h",
         "HoverCarPhysics.cpp", "BoatPhysics.h", "BoatPhysics.cpp", "ParticleManager.h", "ParticleManager.cpp", "VehiclePhysics.h"]

DECL = re.compile(r"(?:const\s+)?(?:JPH::(?:Ref|RefConst)<\s*)?JPH::([A-Za-z_0-9]+)(?:::([A-Za-z_0-9]+))?\s*>?\s*(?:const\s*)?[*&]?\s*(?:const\s+)?"
                  r"([a-z_][A-Za-z_0-9]*)\s*(?:=|;|\(|\)|,|\{)")
USE = re.compile(r"\b([a-z_][A-Za-z_0-9]*)\s*(?:\.|->)\s*([A-Z][A-Za-z_0-9]*|m[A-Z][A-Za-z_0-9]*|s[A-Z][A-Za-z_0-9]*)\b")
CHAIN = re.compile(r"(?:\)|\])\s*(?:\.|->)\s*([A-Z][A-Za-z_0-9]*|m[A-Z][A-Za-z_0-9]*)\b")

# typedefs of the look-alikes (a variable declared with the alias uses the class)
ALIASES = {"Vec3Arg": "Vec3", "RVec3": "Vec3", "RVec3Arg": "Vec3", "QuatArg": "Quat", "Mat44Arg": "Mat44", "RMat44": "Mat44"}
# members reached through a call chain: the class Jolt's API returns at that point (GetWheel() -> Wheel*, GetEngine() -> VehicleEngine&, ...)
CHAINED_OWNER = {
    "Create": "ShapeSettings", "Get": "ShapeSettings::ShapeResult", "Dot": "Vec3", "GetZ": "Vec3", "Normalized": "Vec3",
    "EnableLeanController": "MotorcycleController", "GetCurrentRPM": "VehicleEngine", "SetCurrentRPM": "VehicleEngine",
    "GetAngularVelocity": "Wheel", "SetAngularVelocity": "Wheel", "GetContactLateral": "Wheel", "GetContactLongitudinal": "Wheel", "GetContactNormal": "Wheel",
    "GetContactPointVelocity": "Wheel", "GetContactPosition": "Wheel", "GetLateralLambda": "Wheel", "GetSuspensionLambda": "Wheel",
    "GetRotationAngle": "Wheel", "GetSteerAngle": "Wheel", "GetSuspensionLength": "Wheel", "HasContact": "Wheel", "GetSettings": "Wheel",
    "GetVolume": "Shape", "TryGetBody": "BodyLockInterface", "mLateralFriction": "WheelSettingsWV", "mLongitudinalFriction": "WheelSettingsWV",
    "mPosition": "WheelSettings", "mWidth": "WheelSettings", "mLeftWheel": "VehicleDifferentialSettings", "mRightWheel": "VehicleDifferentialSettings",
    "mLeftRightSplit": "VehicleDifferentialSettings", "mY": "LinearCurve::Point", "mPoints": "LinearCurve",
}
# not members of look-alike classes: the reference's own PhysicsInput fields matched by the pattern
NOT_JOLT = re.compile(r"^[A-Z]+_down$")

HEADERS = """#include <Jolt/Jolt.h>
#include <Jolt/Physics/PhysicsSystem.h>
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

# (classes a variable name may have | member): the extraction of the reference snapshot this repository was written against
COMMITTED = """
Body:GetID Body:GetShape Body:GetUserData BodyCreationSettings:mMassPropertiesOverride BodyCreationSettings:mOverrideMassProperties
BodyCreationSettings:mUserData BodyInterface:ActivateBody BodyInterface:AddBody BodyInterface:AddForce BodyInterface:AddTorque BodyInterface:CreateBody
BodyInterface:GetAngularVelocity BodyInterface:GetCenterOfMassPosition BodyInterface:GetLinearVelocity BodyInterface:GetPointVelocity BodyInterface:GetRotation
BodyInterface:GetWorldTransform BodyLockInterface:TryGetBody BodyLockRead:GetBody BodyLockRead:Succeeded
CharacterVirtual::ExtendedUpdateSettings:mStickToFloorStepDown CharacterVirtual::ExtendedUpdateSettings:mWalkStairsCosAngleForwardContact
CharacterVirtual::ExtendedUpdateSettings:mWalkStairsMinStepForward CharacterVirtual::ExtendedUpdateSettings:mWalkStairsStepDownExtra
CharacterVirtual::ExtendedUpdateSettings:mWalkStairsStepForwardTest CharacterVirtual::ExtendedUpdateSettings:mWalkStairsStepUp
CharacterVirtual::ExtendedUpdateSettings|CharacterVirtualSettings:mMaxStrength CharacterVirtual::ExtendedUpdateSettings|CharacterVirtualSettings:mShape
CharacterVirtual::ExtendedUpdateSettings|CharacterVirtualSettings:mStickToFloorStepDown
CharacterVirtual::ExtendedUpdateSettings|CharacterVirtualSettings:mSupportingVolume CharacterVirtual::ExtendedUpdateSettings|CharacterVirtualSettings:mUp
CharacterVirtual::ExtendedUpdateSettings|CharacterVirtualSettings:mWalkStairsStepUp CharacterVirtual:CanWalkStairs
CharacterVirtual:CancelVelocityTowardsSteepSlopes CharacterVirtual:ExtendedUpdate CharacterVirtual:GetGroundNormal CharacterVirtual:GetGroundVelocity
CharacterVirtual:GetLinearVelocity CharacterVirtual:GetPosition CharacterVirtual:GetShape CharacterVirtual:GetUp CharacterVirtual:IsSlopeTooSteep
CharacterVirtual:IsSupported CharacterVirtual:SetLinearVelocity CharacterVirtual:SetListener CharacterVirtual:SetPosition CharacterVirtual:SetShape
CharacterVirtual:StickToFloor CharacterVirtual:Update CharacterVirtual:UpdateGroundVelocity CharacterVirtual:WalkStairs ConvexHullShapeSettings:Create
LinearCurve::Point:mY LinearCurve:mPoints Mat44:Multiply3x3 Mat44:StoreFloat4x4 MotorcycleController:EnableLeanController
MotorcycleControllerSettings:mDifferentials MotorcycleControllerSettings:mEngine MotorcycleControllerSettings:mLeanSmoothingFactor
MotorcycleControllerSettings:mLeanSpringConstant MotorcycleControllerSettings:mLeanSpringDamping MotorcycleControllerSettings:mLeanSpringIntegrationCoefficient
MotorcycleControllerSettings:mMaxLeanAngle MotorcycleControllerSettings:mTransmission PhysicsSystem:AddConstraint PhysicsSystem:AddStepListener
PhysicsSystem:GetBodyInterface PhysicsSystem:GetBodyLockInterface PhysicsSystem:GetDefaultBroadPhaseLayerFilter PhysicsSystem:GetDefaultLayerFilter
PhysicsSystem:GetGravity PhysicsSystem:RemoveConstraint PhysicsSystem:RemoveStepListener Quat:Conjugated Quat:GetAxisAngle Shape:GetVolume
ShapeSettings::ShapeResult:Get ShapeSettings:Create Vec3:Cross Vec3:Dot Vec3:GetX Vec3:GetY Vec3:GetZ Vec3:IsNearZero Vec3:Length Vec3:Normalized
Vec3:NormalizedOr VehicleConstraint:GetController VehicleConstraint:GetLocalForward VehicleConstraint:GetLocalUp VehicleConstraint:GetWheel
VehicleConstraint:GetWheelLocalBasis VehicleConstraint:GetWheelLocalTransform VehicleConstraint:GetWheelWorldTransform
VehicleConstraint:SetVehicleCollisionTester VehicleConstraintSettings:mAntiRollBars VehicleConstraintSettings:mController VehicleConstraintSettings:mForward
VehicleConstraintSettings:mUp VehicleConstraintSettings:mWheels VehicleDifferentialSettings:mLeftRightSplit VehicleDifferentialSettings:mLeftWheel
VehicleDifferentialSettings:mRightWheel VehicleEngine:GetCurrentRPM VehicleEngine:SetCurrentRPM Wheel:GetAngularVelocity Wheel:GetContactLateral
Wheel:GetContactLongitudinal Wheel:GetContactNormal Wheel:GetContactPointVelocity Wheel:GetContactPosition Wheel:GetLateralLambda Wheel:GetRotationAngle
Wheel:GetSettings Wheel:GetSteerAngle Wheel:GetSuspensionLambda Wheel:GetSuspensionLength Wheel:HasContact Wheel:SetAngularVelocity WheelSettings:mPosition
WheelSettings:mRadius WheelSettings:mSuspensionMaxLength WheelSettings:mSuspensionMinLength WheelSettings:mWidth WheelSettingsWV:mInertia
WheelSettingsWV:mLateralFriction WheelSettingsWV:mLongitudinalFriction WheelSettingsWV:mMaxBrakeTorque WheelSettingsWV:mMaxHandBrakeTorque
WheelSettingsWV:mMaxSteerAngle WheelSettingsWV:mPosition WheelSettingsWV:mRadius WheelSettingsWV:mSteeringAxis WheelSettingsWV:mSuspensionDirection
WheelSettingsWV:mSuspensionMaxLength WheelSettingsWV:mSuspensionMinLength WheelSettingsWV:mSuspensionSpring WheelSettingsWV:mWheelForward
WheelSettingsWV:mWheelUp WheelSettingsWV:mWidth WheeledVehicleController|WheeledVehicleControllerSettings:GetEngine
WheeledVehicleController|WheeledVehicleControllerSettings:SetDriverInput WheeledVehicleController|WheeledVehicleControllerSettings:mDifferentials
WheeledVehicleController|WheeledVehicleControllerSettings:mEngine
""".split()


def extract():
    texts, var_types = {}, {}
    for f in FILES:
        with open(os.path.join(REF, f), errors="replace") as fh:
            t = fh.read()
        t = re.sub(r"//[^\n]*", "", t)
        t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
        texts[f] = t
        for m in DECL.finditer(t):
            cls = m.group(1) if not m.group(2) else m.group(1) + "::" + m.group(2)
            var_types.setdefault(m.group(3), set()).add(ALIASES.get(cls, cls))
    out = set()
    for t in texts.values():
        for m in USE.finditer(t):
            v, mem = m.group(1), m.group(2)
            if NOT_JOLT.match(mem):
                continue
            if v in var_types:
                out.add("|".join(sorted(var_types[v])) + ":" + mem)
            elif mem in CHAINED_OWNER:
                out.add(CHAINED_OWNER[mem] + ":" + mem)
            else:
                out.add("?:" + mem)
        for m in CHAIN.finditer(t):
            mem = m.group(1)
            out.add(CHAINED_OWNER.get(mem, "?") + ":" + mem)
    return sorted(out)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this machine")
def test_committed_member_list_is_what_the_reference_uses():
    got = extract()
    assert not [x for x in got if x.startswith("?:")], [x for x in got if x.startswith("?:")]      # every chained member has an owner
    assert got == sorted(COMMITTED)


def probe(cls, mem):
    return (f"namespace probe_{abs(hash((cls, mem)))} {{ struct Fallback {{ int {mem}; }}; struct Derived : JPH::{cls}, Fallback {{}};\n"
            f"template <class U, U> struct Check; template <class U> char (&f(Check<int Fallback::*, &U::{mem}>*))[1]; template <class U> char (&f(...))[2];\n"
            f"static_assert(sizeof(f<Derived>(0)) == 2, \"JPH::{cls} has no member {mem}\"); }}\n")


def compiles(src):
    return subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", SHIM, "-x", "c++", "-"], input=HEADERS + src, text=True, capture_output=True)


def test_every_member_the_callers_use_exists_in_its_lookalike_class():
    items = [([c for c in ":".join(x.split(":")[:-1]).split("|")], x.rsplit(":", 1)[1]) for x in COMMITTED]
    # one translation unit with the first candidate class of each item; what fails is then tried class by class
    r = compiles("".join(probe(cs[0], m) for cs, m in items))
    if r.returncode == 0:
        return
    missing = []
    for cs, m in items:
        if not any(compiles(probe(c, m)).returncode == 0 for c in cs):
            missing.append(("|".join(cs), m))
    assert not missing, missing
