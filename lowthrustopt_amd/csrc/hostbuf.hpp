// hostbuf.hpp -- host-side storage for code behind the C ABI.
//
// The library is built with -fno-exceptions (nothing may unwind across `extern "C"` into a Julia session), so a std::vector that
// cannot grow ends the caller's process.  Everything the ABI layer allocates on the host per call, and the few lists it keeps, go
// through these two types instead: allocation failure is a value (`ok()` / `push` returning false) that the entry point turns
// into LTO_ENOMEM.
#pragma once
#include <cstdlib>
#include <cstring>

namespace lto {

// fixed-size array of trivially copyable T, zero-filled or filled with `fill`
template <class T>
struct HostBuf {
  T* p = nullptr;
  size_t n = 0;
  HostBuf() = default;
  explicit HostBuf(size_t count) { alloc(count); }
  HostBuf(size_t count, const T& fill) {
    if (alloc(count)) for (size_t k = 0; k < n; ++k) p[k] = fill;
  }
  HostBuf(const HostBuf&) = delete;
  HostBuf& operator=(const HostBuf&) = delete;
  ~HostBuf() { std::free(p); }
  bool alloc(size_t count) {
    std::free(p);
    n = 0;
    p = (T*)std::calloc(count ? count : 1, sizeof(T));
    if (p) n = count;
    return p != nullptr;
  }
  bool ok() const { return p != nullptr; }
  T* data() { return p; }
  const T* data() const { return p; }
  size_t size() const { return n; }
  T& operator[](size_t k) { return p[k]; }
  const T& operator[](size_t k) const { return p[k]; }
};

// growable list of trivially copyable T (order not preserved by erase_at)
template <class T>
struct HostList {
  T* p = nullptr;
  size_t n = 0, cap = 0;
  HostList() = default;
  HostList(const HostList&) = delete;
  HostList& operator=(const HostList&) = delete;
  ~HostList() { std::free(p); }
  bool push(const T& v) {
    if (n == cap) {
      const size_t ncap = cap ? 2 * cap : 8;
      T* q = (T*)std::realloc(p, ncap * sizeof(T));
      if (!q) return false;
      p = q; cap = ncap;
    }
    p[n++] = v;
    return true;
  }
  void erase_at(size_t k) { p[k] = p[n - 1]; --n; }
  void clear() { n = 0; }
  bool empty() const { return n == 0; }
  size_t size() const { return n; }
  T& operator[](size_t k) { return p[k]; }
  const T& operator[](size_t k) const { return p[k]; }
  T* begin() { return p; }
  T* end() { return p + n; }
  const T* begin() const { return p; }
  const T* end() const { return p + n; }
};

}  // namespace lto
