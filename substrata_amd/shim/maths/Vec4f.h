// Minimal stand-in for glare-core maths/Vec4f.h (glare-core is un-vendored: docs/building.txt:17-19), holding exactly the
// members the physics facade and its callers use.  Written from scratch for this repo; not a copy of glare-core.
#pragma once
#include <cmath>

class Vec4f
{
public:
	Vec4f() { x[0] = x[1] = x[2] = x[3] = 0.f; }
	explicit Vec4f(float f) { x[0] = x[1] = x[2] = x[3] = f; }
	Vec4f(float a, float b, float c, float d) { x[0] = a; x[1] = b; x[2] = c; x[3] = d; }
	float& operator[](int i) { return x[i]; }
	float operator[](int i) const { return x[i]; }
	Vec4f operator+(const Vec4f& o) const { return Vec4f(x[0] + o.x[0], x[1] + o.x[1], x[2] + o.x[2], x[3] + o.x[3]); }
	Vec4f operator-(const Vec4f& o) const { return Vec4f(x[0] - o.x[0], x[1] - o.x[1], x[2] - o.x[2], x[3] - o.x[3]); }
	Vec4f operator*(float f) const { return Vec4f(x[0] * f, x[1] * f, x[2] * f, x[3] * f); }
	Vec4f operator-() const { return Vec4f(-x[0], -x[1], -x[2], -x[3]); }
	bool operator==(const Vec4f& o) const { return x[0] == o.x[0] && x[1] == o.x[1] && x[2] == o.x[2] && x[3] == o.x[3]; }
	bool operator!=(const Vec4f& o) const { return !(*this == o); }
	bool isFinite() const { return std::isfinite(x[0]) && std::isfinite(x[1]) && std::isfinite(x[2]) && std::isfinite(x[3]); }
	float length() const { return std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]); }
	float x[4];
};
inline Vec4f maskWToZero(const Vec4f& v) { return Vec4f(v[0], v[1], v[2], 0.f); }
inline Vec4f setWToOne(const Vec4f& v) { return Vec4f(v[0], v[1], v[2], 1.f); }
inline float dot(const Vec4f& a, const Vec4f& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3]; }
inline Vec4f div(const Vec4f& a, const Vec4f& b) { return Vec4f(a[0] / b[0], a[1] / b[1], a[2] / b[2], a[3] / b[3]); }
inline Vec4f normalise(const Vec4f& v) { return v * (1.f / v.length()); }
inline Vec4f crossProduct(const Vec4f& a, const Vec4f& b) { return Vec4f(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0], 0.f); }
