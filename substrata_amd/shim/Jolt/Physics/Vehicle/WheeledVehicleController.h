// <Jolt/Physics/Vehicle/WheeledVehicleController.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: WheeledVehicleController(+Settings), WheelSettingsWV, VehicleEngine, differentials, transmission.  Implementation: Jolt/JoltVehicleLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../../JoltVehicleLite.h"
