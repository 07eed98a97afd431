// <Jolt/Jolt.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: core types: Vec3, Quat, Mat44, Float3/4, Ref / RefConst / RefTarget, Array, uint.  Implementation: Jolt/JoltLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "JoltLite.h"
