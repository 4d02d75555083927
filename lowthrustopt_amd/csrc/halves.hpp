// halves.hpp -- two lanes per 12-component state: the cross-lane pieces shared by kernels_indirect_coop2.hip and
// kernels_indirect_defect2.hip (device code only; the per-lane arithmetic, rhs12_base_half, is in dynamics.hpp).
//
// Lane layout inside a 16-lane DPP row: 4-lane banks A B A B; lane A of a segment owns (r, v), lane B -- four lanes up -- owns
// (lambda_v, lambda_r).  Own rows in global numbering: A (0..5), B (9, 10, 11, 6, 7, 8).
#pragma once
#include <hip/hip_runtime.h>

namespace lto {

// src's value from the lane 4 below (CTRL = row_shr:4) or 4 above (row_shl:4) into the lanes of the banks in BANK; the other
// lanes keep old.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_bank_merge(const double old, const double src) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANK, false);
  return __hiloint2double(hi, lo);
}
// the A lane's x in both lanes of a pair / the B lane's x in both lanes
__device__ __forceinline__ double from_lane_a(const double x) { return dpp_bank_merge<0x114, 0xA>(x, x); }   // row_shr:4 into the B banks
__device__ __forceinline__ double from_lane_b(const double x) { return dpp_bank_merge<0x104, 0x5>(x, x); }   // row_shl:4 into the A banks
// x_A + x_B, the same bits in both lanes
__device__ __forceinline__ double pair_sum(const double x) { return from_lane_a(x) + from_lane_b(x); }

}  // namespace lto
