// JoltUtils.h of the shim: the conversion helpers the reference's callers use between glare-core maths types and JPH types, under
// the names of /root/reference/gui_client/JoltUtils.h:14-64 (toJoltVec3 x3, toVec3f, toVec4fVec, toVec4fPos, toJoltQuat, toQuat,
// toMatrix4f).  The reference bit-casts SSE registers; the look-alike types here are plain structs, so these go component by component.
#pragma once
#include <Jolt/Jolt.h>
#include <maths/Vec4.h>

inline JPH::Vec3 toJoltVec3(const Vec4f& v) { return JPH::Vec3(v[0], v[1], v[2]); }       // w is dropped (vector or point alike)
inline JPH::Vec3 toJoltVec3(const Vec3f& v) { return JPH::Vec3(v.x, v.y, v.z); }
inline JPH::Vec3 toJoltVec3(const Vec3d& v) { return JPH::Vec3((float)v.x, (float)v.y, (float)v.z); }

inline Vec3f toVec3f(const JPH::Vec3& v) { return Vec3f(v.GetX(), v.GetY(), v.GetZ()); }
inline Vec4f toVec4fVec(const JPH::Vec3& v) { return Vec4f(v.GetX(), v.GetY(), v.GetZ(), 0.f); }   // direction: w = 0
inline Vec4f toVec4fPos(const JPH::Vec3& v) { return Vec4f(v.GetX(), v.GetY(), v.GetZ(), 1.f); }   // point: w = 1

inline JPH::Quat toJoltQuat(const Quatf& q) { return JPH::Quat(q.v[0], q.v[1], q.v[2], q.v[3]); }   // both are (x, y, z, w)
inline Quatf toQuat(const JPH::Quat& q) { return Quatf(q.GetX(), q.GetY(), q.GetZ(), q.GetW()); }

inline Matrix4f toMatrix4f(const JPH::Mat44& mat)
{
	JPH::Float4 columns[4];
	mat.StoreFloat4x4(columns);               // column-major, 16 contiguous floats
	return Matrix4f(&columns[0].x);
}
