#pragma once
template <class T> class Vec2 { public: Vec2() : x(0), y(0) {} explicit Vec2(T f) : x(f), y(f) {} Vec2(T a, T b) : x(a), y(b) {} T x, y; };
typedef Vec2<float> Vec2f;
