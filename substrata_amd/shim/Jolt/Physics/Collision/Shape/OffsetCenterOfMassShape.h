// <Jolt/Physics/Collision/Shape/OffsetCenterOfMassShape.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: OffsetCenterOfMassShapeSettings.  Implementation: Jolt/JoltLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../../../JoltLite.h"
