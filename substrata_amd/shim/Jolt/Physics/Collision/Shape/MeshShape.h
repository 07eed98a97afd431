// <Jolt/Physics/Collision/Shape/MeshShape.h> of the JPH look-alike set (SURVEY 8b Tier 2): the include path the reference's callers use.
// Provides: meshes are built by PhysicsWorld::createJoltShapeFor...Mesh -> sgp_mesh_create.  Implementation: Jolt/JoltLite.h over the sgp C ABI; no Jolt code.
#pragma once
#include "../../../JoltLite.h"
