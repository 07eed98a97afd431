// <Jolt/Physics/Character/Character.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: only CharacterVirtual is used by PlayerPhysics; this header exists because PlayerPhysics.h includes it.  Implementation: Jolt/JoltCharacterLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../../JoltCharacterLite.h"
