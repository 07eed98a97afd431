#pragma once
#include <cstddef>
template <class T> class Reference
{
public:
	Reference() : p(nullptr) {}
	Reference(T* p_) : p(p_) { if (p) p->incRefCount(); }
	Reference(std::nullptr_t) : p(nullptr) {}
	Reference(const Reference& o) : p(o.p) { if (p) p->incRefCount(); }
	~Reference() { release(); }
	Reference& operator=(const Reference& o) { if (o.p) o.p->incRefCount(); release(); p = o.p; return *this; }
	Reference& operator=(std::nullptr_t) { release(); return *this; }
	T* operator->() const { return p; }
	T& operator*() const { return *p; }
	T* ptr() const { return p; }
	T* getPointer() const { return p; }
	explicit operator bool() const { return p != nullptr; }
	bool isNull() const { return p == nullptr; }
private:
	void release() { if (p && p->decRefCount() == 0) delete p; p = nullptr; }
	T* p;
};
