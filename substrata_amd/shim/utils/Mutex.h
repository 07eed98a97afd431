#pragma once
#include <mutex>
class Mutex { public: std::mutex m; };
class Lock { public: explicit Lock(Mutex& mu) : l(mu.m) {} private: std::lock_guard<std::mutex> l; };
#define GUARDED_BY(x)
