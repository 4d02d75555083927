// kernels_indirect14.hip -- ND = 14 instantiations of indirect_kernel.hpp: CRTBP state + mass + costates +
// mass costate (BASELINE configs[1]; an extension with no reference counterpart, see dynamics.hpp / DESIGN.md).
#include "indirect_kernel.hpp"

namespace lto {

hipError_t launch_indirect14_defect(int pm, int method, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  switch (method) {
    case M_RK4: return launch_pm<14, M_RK4, 0>(pm, a, st);
    case M_RKF78_FIXED: return launch_pm<14, M_RKF78_FIXED, 0>(pm, a, st);
    case M_RKF78_ADAPTIVE: return launch_pm<14, M_RKF78_ADAPTIVE, 0>(pm, a, st);
    case M_DOP853_ADAPTIVE: return launch_pm<14, M_DOP853_ADAPTIVE, 0>(pm, a, st);
  }
  return hipErrorInvalidValue;
}

hipError_t launch_indirect14_stm(int pm, int method, int cols, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (method == M_RK4) {
    if (cols == 0) cols = (((long)a.S + 63) / 64 * 14 <= 4096) ? 1 : 2;
    switch (cols) {
      case 1: return launch_pm<14, M_RK4, 1>(pm, a, st);
      case 2: return launch_pm<14, M_RK4, 2>(pm, a, st);
    }
    return hipErrorInvalidValue;
  }
  return hipErrorInvalidValue;      // 13-stage methods: cooperative kernels only (see kernels_indirect.hip)
}

}  // namespace lto
