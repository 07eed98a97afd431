// <Jolt/Core/TempAllocator.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: TempAllocator (an empty tag type: the device owns its arenas).  Implementation: Jolt/JoltCharacterLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../JoltCharacterLite.h"
