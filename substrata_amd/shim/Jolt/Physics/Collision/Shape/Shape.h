// <Jolt/Physics/Collision/Shape/Shape.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: Shape, ShapeSettings, ShapeSettings::ShapeResult, SubShapeID.  Implementation: Jolt/JoltLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../../../JoltLite.h"
