// Minimal stand-in for glare-core maths/Quat.h: (x,y,z,w) in a Vec4f, same memory order as JPH::Quat (JoltUtils.h:48-56).
#pragma once
#include "Vec4f.h"
#include "Matrix4f.h"
template <class T> class Quat
{
public:
	Quat() : v(0.f, 0.f, 0.f, 1.f) {}
	explicit Quat(const Vec4f& v_) : v(v_) {}
	Quat(T x, T y, T z, T w) : v(x, y, z, w) {}
	static Quat identity() { return Quat(0, 0, 0, 1); }
	static Quat fromAxisAndAngle(const Vec4f& unit_axis, T angle) { const T s = std::sin(angle / 2); return Quat(unit_axis[0] * s, unit_axis[1] * s, unit_axis[2] * s, std::cos(angle / 2)); }
	Matrix4f toMatrix() const
	{
		const T x = v[0], y = v[1], z = v[2], w = v[3];
		Matrix4f m = Matrix4f::identity();
		m.setColumn(0, Vec4f(1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y), 0));
		m.setColumn(1, Vec4f(2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x), 0));
		m.setColumn(2, Vec4f(2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y), 0));
		return m;
	}
	Quat conjugate() const { return Quat(-v[0], -v[1], -v[2], v[3]); }
	Quat operator*(const Quat& b) const            // Hamilton product: (this * b) rotates by b first, then by this
	{
		const T ax = v[0], ay = v[1], az = v[2], aw = v[3], bx = b.v[0], by = b.v[1], bz = b.v[2], bw = b.v[3];
		return Quat(aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz);
	}
	Vec4f rotateVector(const Vec4f& p) const
	{
		const T x = v[0], y = v[1], z = v[2], w = v[3];
		const T tx = 2 * (y * p[2] - z * p[1]), ty = 2 * (z * p[0] - x * p[2]), tz = 2 * (x * p[1] - y * p[0]);
		return Vec4f(p[0] + w * tx + (y * tz - z * ty), p[1] + w * ty + (z * tx - x * tz), p[2] + w * tz + (x * ty - y * tx), p[3]);
	}
	Vec4f v;
};
typedef Quat<float> Quatf;
