// <Jolt/Physics/Collision/ObjectLayer.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: ObjectLayer filters (the layer table itself is sgp's 4x4 mask).  Implementation: Jolt/JoltLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../../JoltLite.h"
