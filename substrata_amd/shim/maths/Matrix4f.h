// Minimal stand-in for glare-core maths/Matrix4f.h: column-major 4x4, only what the facade uses.
#pragma once
#include "Vec4f.h"
class Matrix4f
{
public:
	Matrix4f() { for (int i = 0; i < 16; ++i) e[i] = 0.f; }
	explicit Matrix4f(const float* d) { for (int i = 0; i < 16; ++i) e[i] = d[i]; }
	static Matrix4f identity() { Matrix4f m; m.e[0] = m.e[5] = m.e[10] = m.e[15] = 1.f; return m; }
	void setColumn(int c, const Vec4f& v) { for (int r = 0; r < 4; ++r) e[c * 4 + r] = v[r]; }
	Vec4f getColumn(int c) const { return Vec4f(e[c * 4], e[c * 4 + 1], e[c * 4 + 2], e[c * 4 + 3]); }
	Matrix4f getTranspose() const { Matrix4f t; for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) t.e[r * 4 + c] = e[c * 4 + r]; return t; }
	Vec4f operator*(const Vec4f& v) const
	{
		Vec4f r;
		for (int i = 0; i < 4; ++i) r[i] = e[i] * v[0] + e[4 + i] * v[1] + e[8 + i] * v[2] + e[12 + i] * v[3];
		return r;
	}
	Matrix4f operator*(const Matrix4f& b) const { Matrix4f m; for (int c = 0; c < 4; ++c) m.setColumn(c, (*this) * b.getColumn(c)); return m; }
	float e[16];
};
// M * translation(t)
inline Matrix4f rightTranslate(const Matrix4f& m, const Vec4f& t)
{
	Matrix4f r = m;
	r.setColumn(3, m * Vec4f(t[0], t[1], t[2], 1.f));
	return r;
}
