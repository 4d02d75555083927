// kernels_indirect.hip -- ND = 12 instantiations of indirect_kernel.hpp (the reference's state+costate system).
#include "indirect_kernel.hpp"

namespace lto {

hipError_t launch_indirect_defect(int pm, int method, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  switch (method) {
    case M_RK4: return launch_pm<12, M_RK4, 0>(pm, a, st);
    case M_RKF78_FIXED: return launch_pm<12, M_RKF78_FIXED, 0>(pm, a, st);
    case M_RKF78_ADAPTIVE: return launch_pm<12, M_RKF78_ADAPTIVE, 0>(pm, a, st);
    case M_DOP853_ADAPTIVE: return launch_pm<12, M_DOP853_ADAPTIVE, 0>(pm, a, st);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_indirect_stm(int pm, int method, int cols, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (method == M_RK4) {
    if (cols == 0) {
      // Fill the chip first (1024 SIMDs): fewer columns per lane = more waves but more redundant base
      // work; more columns per lane amortise the base RHS once the machine is full.  (Round 6: the two-column form is gone -- it won
      // only around 8 192 segments, 18.5 against 21.5 us, and AUTO ran it up to 43 690 where three columns are 20 % faster:
      // tools/probe_cols.py, profiles/r06_probe_cols.txt.)
      const long waves1 = ((long)a.S + 63) / 64 * 12;
      cols = (waves1 <= 1536) ? 1 : 3;
    }
    switch (cols) {
      case 1: return launch_pm<12, M_RK4, 1>(pm, a, st);
      case 3: return launch_pm<12, M_RK4, 3>(pm, a, st);
    }
    return hipErrorInvalidValue;
  }
  // 13-stage methods: no per-lane STM form (round 6: the memory-resident one-column-per-lane kernels, never AUTO's choice, are gone --
  // lto_api.hip sends such sweeps to the cooperative kernels)
  return hipErrorInvalidValue;
}

hipError_t launch_indirect_dense(int ndim, int pm, int method, const IndirectArgs& a, const DenseArgs& d, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (ndim != 12) return hipErrorInvalidValue;          // (lto_api.hip refuses these shapes with LTO_EUNSUPPORTED before it gets here)
  switch (method) {
    case M_RK4: return launch_dense_pm<12, M_RK4>(pm, a, d, st);
    case M_DOP853_ADAPTIVE: return launch_dense_pm<12, M_DOP853_ADAPTIVE>(pm, a, d, st);
  }
  return hipErrorInvalidValue;
}

}  // namespace lto
