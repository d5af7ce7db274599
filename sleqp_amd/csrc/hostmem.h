// Large host arrays of the symbolic analysis (graph of S, product lists: > 100 MB at config 4).
//
// A cold analysis touches every page of them for the first time; with 4 KB pages that is 30-40 k page faults per
// 100 MB, 14-21 ms on the bench host (scripts/probe/page_touch.cpp) - a fifth of the whole analysis.  The kernel
// backs 2 MB-aligned ranges with transparent huge pages when asked (THP mode "madvise"): 1-6 ms for the same
// touches.  huge_resize() reserves, advises, then resizes (glibc serves allocations of this size by mmap, so the
// pages are untouched until the resize).
#pragma once
#include <sys/mman.h>

#include <cstdint>
#include <cstdlib>
#include <new>
#include <utility>
#include <vector>

namespace hipfact {

// std::vector::resize() value-initialises: 150 MB of zeros written by one thread (15 ms) into arrays whose every
// entry is about to be written by the threaded passes anyway.  BigVec default-initialises instead (its resize leaves
// the new entries indeterminate - only for arrays that are filled completely before they are read).
template <class T>
struct NoInitAlloc {
  using value_type = T;
  NoInitAlloc() = default;
  template <class U>
  NoInitAlloc(const NoInitAlloc<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(::operator new(n * sizeof(T))); }
  void deallocate(T* p, size_t) { ::operator delete(p); }
  template <class U>
  void construct(U* p) noexcept {
    ::new ((void*)p) U;
  }
  template <class U, class A0, class... A>
  void construct(U* p, A0&& a0, A&&... a) {
    ::new ((void*)p) U(std::forward<A0>(a0), std::forward<A>(a)...);
  }
  template <class U>
  bool operator==(const NoInitAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <class T>
using BigVec = std::vector<T, NoInitAlloc<T>>;

template <class V>
inline void huge_advise(V& v) {
  using T = typename V::value_type;
  const size_t bytes = v.capacity() * sizeof(T);
  static const bool off = getenv("HIPFACT_NO_THP") != nullptr;
  if (off || bytes < ((size_t)4 << 20)) return;
  const uintptr_t lo = ((uintptr_t)v.data() + 4095) & ~(uintptr_t)4095;
  const uintptr_t hi = ((uintptr_t)v.data() + bytes) & ~(uintptr_t)4095;
  if (hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);  // advisory: failure leaves ordinary pages
}

template <class V>
inline void huge_resize(V& v, size_t n) {
  if (n > v.capacity()) {
    v.reserve(n);
    huge_advise(v);
  }
  v.resize(n);
}

template <class V>
inline void huge_assign(V& v, size_t n, const typename V::value_type& value) {
  v.clear();
  if (n > v.capacity()) {
    v.reserve(n);
    huge_advise(v);
  }
  v.assign(n, value);
}

}  // namespace hipfact
