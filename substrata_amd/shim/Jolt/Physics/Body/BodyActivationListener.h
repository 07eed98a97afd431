// <Jolt/Physics/Body/BodyActivationListener.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: activation events are replayed by PhysicsWorld::think from sgp_world_drain_events.  Implementation: Jolt/JoltLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../../JoltLite.h"
