// pipe_hooks.hpp (probe version) -- same interface as lowthrustopt_amd/csrc/hooks/pipe_hooks.hpp, with the counters behind it.
// Used by `make probe` only (build/liblto_probe.so; tools/probe_pipe_roles.py, tools/probe_coop2.py): the probe scripts pass a
// defect buffer with spare rows, and the hooks write their figures there.
#pragma once
#include <hip/hip_runtime.h>

namespace lto {
namespace hook {

constexpr bool kProbeBuild = true;

struct BarrierWait {
  long long waited = 0;
  __device__ __forceinline__ void sync() { const long long t = clock64(); __syncthreads(); waited += clock64() - t; }
  __device__ __forceinline__ int sync_or(const int pred) { const long long t = clock64(); const int r = __syncthreads_or(pred); waited += clock64() - t; return r; }
  __device__ __forceinline__ void report(double* rows, long ld, int row, long col) const { if (rows) rows[row * ld + col] = (double)waited; }
};
struct RegionClock {
  long long c0 = 0, w0 = 0;
  __device__ __forceinline__ void start() { c0 = clock64(); w0 = wall_clock64(); }
  __device__ __forceinline__ void report(double* rows, long ld, int row_ticks, int row_wall, long col) const {
    if (rows) { rows[row_ticks * ld + col] = (double)(clock64() - c0); rows[row_wall * ld + col] = (double)(wall_clock64() - w0); }
  }
  __device__ __forceinline__ void report_ticks(double* rows, long ld, int row, long col) const { if (rows) rows[row * ld + col] = (double)(clock64() - c0); }
};
struct Counter {
  int n = 0;
  __device__ __forceinline__ void bump() { ++n; }
  __device__ __forceinline__ void report(double* rows, long ld, int row, long col) const { if (rows) rows[row * ld + col] = (double)n; }
};
// 100 MHz wall-clock stamps 0 .. 3 into rows row0 .. row0 + 3, and the hardware id of the compute unit (XCC, SE, CU) into row0 + 4
struct Stamps {
  long long t[4] = {0, 0, 0, 0};
  __device__ __forceinline__ void mark(int i) { t[i] = wall_clock64(); }
  __device__ __forceinline__ void report(double* rows, long ld, int row0, long col) const {
    if (!rows) return;
    for (int i = 0; i < 4; ++i) rows[(row0 + i) * ld + col] = (double)t[i];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // HW_ID: cu_id bits 11:8, sh_id bit 12, se_id bits 15:13; XCC_ID bits 3:0
    rows[(row0 + 4) * ld + col] = (double)(((xcc & 0xf) << 8) | (((hw >> 13) & 0x7) << 5) | (((hw >> 12) & 0x1) << 4) | ((hw >> 8) & 0xf));
  }
};
template <class Args>
__device__ __forceinline__ bool role_on(const Args& a, int bit) { return !(a.max_steps & bit); }

}  // namespace hook
}  // namespace lto
