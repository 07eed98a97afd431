// Minimal stand-in for glare-core utils/HashSet.h (open-addressing set with an explicit empty key) over std::unordered_set.
#pragma once
#include <unordered_set>
template <class K> class HashSet
{
public:
	typedef typename std::unordered_set<K>::iterator iterator;
	typedef typename std::unordered_set<K>::const_iterator const_iterator;
	explicit HashSet(K /*empty_key*/ = K()) {}
	std::pair<iterator, bool> insert(const K& k) { return s.insert(k); }
	void erase(const K& k) { s.erase(k); }
	iterator begin() { return s.begin(); }
	iterator end() { return s.end(); }
	const_iterator begin() const { return s.begin(); }
	const_iterator end() const { return s.end(); }
	size_t size() const { return s.size(); }
	size_t count(const K& k) const { return s.count(k); }
	void clear() { s.clear(); }
private:
	std::unordered_set<K> s;
};
