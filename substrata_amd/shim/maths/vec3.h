// Minimal stand-in for glare-core maths/vec3.h (see Vec4f.h).
#pragma once
#include <cmath>
#include "Vec4f.h"
template <class T> class Vec3
{
public:
	Vec3() : x(0), y(0), z(0) {}
	explicit Vec3(T f) : x(f), y(f), z(f) {}
	Vec3(T a, T b, T c) : x(a), y(b), z(c) {}
	explicit Vec3(const Vec4f& v) : x((T)v[0]), y((T)v[1]), z((T)v[2]) {}
	T& operator[](int i) { return (&x)[i]; }
	T operator[](int i) const { return (&x)[i]; }
	bool operator==(const Vec3& o) const { return x == o.x && y == o.y && z == o.z; }
	bool operator!=(const Vec3& o) const { return !(*this == o); }
	bool isFinite() const { return std::isfinite(x) && std::isfinite(y) && std::isfinite(z); }
	Vec4f toVec4fVector() const { return Vec4f((float)x, (float)y, (float)z, 0.f); }
	Vec4f toVec4fPoint() const { return Vec4f((float)x, (float)y, (float)z, 1.f); }
	T x, y, z;
};
typedef Vec3<float> Vec3f;
typedef Vec3<double> Vec3d;
