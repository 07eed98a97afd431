#pragma once
#include "../maths/Vec4f.h"
namespace js { class AABBox { public: AABBox() {} AABBox(const Vec4f& mn, const Vec4f& mx) : min_(mn), max_(mx) {} Vec4f min_, max_; }; }
