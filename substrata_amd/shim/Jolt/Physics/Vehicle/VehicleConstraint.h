// <Jolt/Physics/Vehicle/VehicleConstraint.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: VehicleConstraint, VehicleConstraintSettings, Wheel, WheelSettings, VehicleAntiRollBar.  Implementation: Jolt/JoltVehicleLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../../JoltVehicleLite.h"
