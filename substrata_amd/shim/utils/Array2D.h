// Minimal stand-in for glare-core utils/Array2D.h: what PhysicsWorld::createJoltHeightFieldShape(int, const Array2D<float>&, float) reads
// (PhysicsWorld.cpp:1086-1119: getWidth(), getData(); TerrainSystem.cpp:1300 fills it with elem(x, y)).  Written for this repo.
#pragma once
#include <vector>
#include <cstddef>
template <class T> class Array2D
{
public:
	Array2D() : w(0), h(0) {}
	Array2D(size_t width, size_t height) : w(width), h(height), data(width * height) {}
	void resize(size_t width, size_t height) { w = width; h = height; data.assign(width * height, T()); }
	size_t getWidth() const { return w; }
	size_t getHeight() const { return h; }
	T& elem(size_t x, size_t y) { return data[y * w + x]; }
	const T& elem(size_t x, size_t y) const { return data[y * w + x]; }
	T* getData() { return data.data(); }
	const T* getData() const { return data.data(); }
private:
	size_t w, h;
	std::vector<T> data;
};
