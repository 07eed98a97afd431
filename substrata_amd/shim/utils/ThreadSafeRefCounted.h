#pragma once
#include <atomic>
class ThreadSafeRefCounted
{
public:
	ThreadSafeRefCounted() : refcount(0) {}
	virtual ~ThreadSafeRefCounted() {}
	void incRefCount() const { refcount.fetch_add(1); }
	long decRefCount() const { return refcount.fetch_sub(1) - 1; }
	long getRefCount() const { return refcount.load(); }
private:
	mutable std::atomic<long> refcount;
};
#define GLARE_ALIGNED_16_NEW_DELETE
