#pragma once
#include <string>
namespace glare { class Exception { public: explicit Exception(const std::string& s_) : s(s_) {} const std::string& what() const { return s; } private: std::string s; }; }
