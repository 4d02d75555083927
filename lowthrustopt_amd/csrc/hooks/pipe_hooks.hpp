// pipe_hooks.hpp (product version) -- the development probes of the pipeline / cooperative kernels, as no-ops.
//
// The kernels call these hooks where a probe build measures something (ticks waited at a barrier, ticks of a loop, trial
// counts) or switches a role off.  This header is what liblto_hip.so is built with: every hook is empty and folds away, so the
// benchmarked translation units contain no `#ifdef`.  `make probe` puts tools/probe_hooks/ first on the include path instead;
// its pipe_hooks.hpp has the same interface with the counters behind it (tools/probe_pipe_roles.py, tools/probe_coop2.py).
#pragma once
#include <hip/hip_runtime.h>

namespace lto {
namespace hook {

constexpr bool kProbeBuild = false;

// a workgroup barrier, and the ticks this wave waited at such barriers
struct BarrierWait {
  __device__ __forceinline__ void sync() { __syncthreads(); }
  __device__ __forceinline__ int sync_or(const int pred) { return __syncthreads_or(pred); }
  __device__ __forceinline__ void report(double*, long, int, long) const {}          // (rows, ld, row, column)
};
// ticks (s_memtime) and 100 MHz ticks of a region
struct RegionClock {
  __device__ __forceinline__ void start() {}
  __device__ __forceinline__ void report(double*, long, int, int, long) const {}     // (rows, ld, row of the ticks, row of the wall ticks, column)
  __device__ __forceinline__ void report_ticks(double*, long, int, long) const {}
};
struct Counter {
  __device__ __forceinline__ void bump() {}
  __device__ __forceinline__ void report(double*, long, int, long) const {}
};
// wall-clock stamps of a workgroup's life (entry, loop start, loop end, exit) and the compute unit it ran on (tools/probe_c5_turnover.py)
struct Stamps {
  __device__ __forceinline__ void mark(int) {}
  __device__ __forceinline__ void report(double*, long, int, long) const {}
};
// role switches of the pipeline kernels (probe build: bits of IndirectArgs::max_steps switch roles off)
template <class Args>
__device__ __forceinline__ constexpr bool role_on(const Args&, int) { return true; }

}  // namespace hook
}  // namespace lto
