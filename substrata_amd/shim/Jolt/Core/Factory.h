// <Jolt/Core/Factory.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: no factory: shapes are plain descriptions bound to sgp_* ids.  Implementation: Jolt/JoltLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../JoltLite.h"
