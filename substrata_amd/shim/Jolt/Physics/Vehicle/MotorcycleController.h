// <Jolt/Physics/Vehicle/MotorcycleController.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: MotorcycleController(+Settings).  Implementation: Jolt/JoltVehicleLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../../JoltVehicleLite.h"
