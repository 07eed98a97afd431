// Stand-in for glare-core maths/Vec4.h as far as the physics callers need it: the reference's JoltUtils.h includes this header and
// expects Vec4f, Vec3f / Vec3d, Quatf and Matrix4f to be visible afterwards.
#pragma once
#include "Vec4f.h"
#include "vec3.h"
#include "Matrix4f.h"
#include "Quat.h"
