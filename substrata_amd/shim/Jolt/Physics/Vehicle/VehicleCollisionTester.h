// <Jolt/Physics/Vehicle/VehicleCollisionTester.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: VehicleCollisionTesterRay / CastSphere / CastCylinder.  Implementation: Jolt/JoltVehicleLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../../JoltVehicleLite.h"
