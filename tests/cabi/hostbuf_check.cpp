// CPU check of lowthrustopt_amd/csrc/hostbuf.hpp (the containers behind the C ABI that report allocation failure as a value):
// built and run by tests/test_hostbuf.py with g++, no GPU.
#include <cstdio>
#include <cstdint>
#include "../../lowthrustopt_amd/csrc/hostbuf.hpp"

struct Pair { void* p; long tag; };

static int fail(const char* what) { std::printf("FAIL %s\n", what); return 1; }

int main() {
  {   // HostBuf: zero-filled, fill value, re-allocation, failure as a value
    lto::HostBuf<double> z(1000);
    if (!z.ok() || z.size() != 1000) return fail("alloc");
    for (size_t k = 0; k < z.size(); ++k) if (z[k] != 0.0) return fail("zero fill");
    lto::HostBuf<int> f(17, 7);
    for (size_t k = 0; k < f.size(); ++k) if (f[k] != 7) return fail("fill");
    lto::HostBuf<char> e(0);
    if (!e.ok() || e.size() != 0) return fail("empty");
    if (!z.alloc(10) || z.size() != 10 || z[9] != 0.0) return fail("realloc");
    lto::HostBuf<double> huge;
    if (huge.alloc(SIZE_MAX / sizeof(double)) || huge.ok() || huge.size() != 0) return fail("an impossible size must fail quietly");
  }
  {   // HostList: growth past the first capacity, unordered erase, iteration
    lto::HostList<Pair> l;
    if (!l.empty()) return fail("new list");
    for (long k = 0; k < 1000; ++k) if (!l.push({(void*)(uintptr_t)(k + 1), k})) return fail("push");
    if (l.size() != 1000) return fail("size");
    long sum = 0;
    for (const Pair& q : l) sum += q.tag;
    if (sum != 999L * 1000 / 2) return fail("iteration");
    l.erase_at(0);                       // the last entry takes its place
    if (l.size() != 999 || l[0].tag != 999) return fail("erase_at");
    for (size_t k = 0; k < l.size();) { if (l[k].tag % 2) l.erase_at(k); else ++k; }
    for (const Pair& q : l) if (q.tag % 2) return fail("erase while scanning");
    l.clear();
    if (!l.empty() || !l.push({nullptr, 5}) || l[0].tag != 5) return fail("reuse after clear");
  }
  std::printf("hostbuf ok\n");
  return 0;
}
