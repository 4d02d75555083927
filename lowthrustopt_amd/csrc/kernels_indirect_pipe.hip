// kernels_indirect_pipe.hip -- three-role software pipeline for the fixed-step RK4 STM sweep (BASELINE configs[1]).
//
// At 4 096 segments the chip offers exactly 16 lanes per segment (1 024 SIMDs x 64 lanes) and the sweep lasts as long
// as ONE lane's instruction stream: the per-lane kernel (indirect_kernel.hpp) spends ~60 % of that stream on work
// every column lane of a segment repeats (base trajectory, gravity, control law), and the cooperative kernel
// (kernels_indirect_coop.hip) pays one workgroup barrier per RK stage with the base role's full RHS + coefficient
// build on the critical path.  Here a workgroup owns 16 segments and runs three roles in different wavefronts,
// skewed by one RK4 STEP each, so that they execute concurrently and meet at ONE barrier per step:
//
//   base      integrates the ND-dim base state (rhs*_base: the RHS alone -- the shortest instruction stream) and
//             publishes, once per step, the part of the four stage arguments the coefficients depend on
//   coef      one step behind: lane = (segment, RK stage); builds G, H, U (+ mass couplings) at those arguments
//   columns   two steps behind: c' = F(t) c with the coefficients of the stage -- no gravity, no control law, no base
//             state in these lanes
//
// Both hand-overs are double-buffered per step in LDS; a sweep of n steps takes n + 2 phases; every wavefront executes
// exactly n + 2 barriers and nothing spins.  Two forms differ in the column role only:
//
//   k_indirect_pipe   4 waves, one per SIMD: base, coef, 2 column waves; column lane = (segment, PAIR of columns),
//                     17 / 25 coefficients read from LDS per stage
//   k_indirect_pipe6  6 waves: base and coef alone on a SIMD each, 4 column waves two per SIMD; column lane =
//                     (segment, ONE column) with a DPP row = one segment, coefficients broadcast inside the FMA
//
// Replaces the serial loop of jacobianCalc (src/multiShoot_CRTBP_indirect.jl:93-146) for the fixed-step setting.
#include "pipe_common.hpp"

namespace lto {

// ---------------------------------------------------------------------------------------------------------- base role
// lane = (segment, lane group): the wave holds FOUR identical copies of its 16 segments' base state, one per 16-lane
// row.  An LDS store costs the issuing wave ~6 FMA slots whatever it carries, so the stage arguments are not stored
// stage by stage (16 stores per step): row g keeps the argument of stage g in registers (8 moves under that row's EXEC
// mask) and ONE set of stores at the end of the step publishes all four stages, every lane carrying useful data.
// s_int: [8][NI][PIPE_SEG], slab = step parity * 4 + stage.
template <int ND, int PM>
__device__ __forceinline__ void pipe_role_base(const IndirectArgs& a, const PipeLane& L, const int seg, const int slot,
                                               double* s_int) {
  constexpr int NI = PipeArg<ND, PM>::N;
  const int steps = a.steps;
  const double h = L.h, h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);
  auto rhs = [&](const double (&y)[ND], double (&k)[ND]) {
    if constexpr (ND == 12) rhs12_base<PM>(y, L.tp, k);
    else rhs14_base<PM>(y, L.tp, k);
  };
  double y[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + L.node];
  for (int p = 0; p < steps + 2; ++p) {
    if (p < steps && PIPE_ROLE_ON(a, 1)) {
      double k[ND], yt[ND], acc[ND], keep[NI];
      auto remember = [&](int stage, const double (&arg)[ND]) {
        if (slot == stage) {
#pragma unroll
          for (int e = 0; e < NI; ++e) keep[e] = arg[PipeArg<ND, PM>::idx[e]];
        }
      };
      remember(0, y);
      rhs(y, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
      remember(1, yt);
      rhs(yt, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
      remember(2, yt);
      rhs(yt, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h, k[c], y[c]); }
      remember(3, yt);
      rhs(yt, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) y[c] = __builtin_fma(h6, k[c], acc[c]);
      double* dst = s_int + (((p & 1) * 4 + slot) * NI) * PIPE_SEG + seg;
#pragma unroll
      for (int e = 0; e < NI; ++e) dst[e * PIPE_SEG] = keep[e];
    }
    __syncthreads();
  }
  if (L.in_range && slot == 0) {
    if (a.defect) {
#pragma unroll
      for (int c = 0; c < ND; ++c) a.defect[c * a.ldd + L.s] = y[c] - a.X[c * a.ldx + L.node + 1];
    }
    if (a.errors) a.errors[L.s] = 0.0;
    if (a.nacc) a.nacc[L.s] = steps;
    if (a.nrej) a.nrej[L.s] = 0;
  }
}

// --------------------------------------------------------------------------------------------------- coefficient role
// lane = (segment, RK stage): the four stages of step p - 1 are built side by side in phase p.  The stage argument's
// components outside PipeArg<ND, PM>::idx are unused by the coefficients (the slopes rhs* also produces are dead code).
template <int ND, int PM, class Layout>
__device__ __forceinline__ void pipe_role_coef(const IndirectArgs& a, const PipeLane& L, const int seg, const int stage,
                                               const double* s_int, double* s_coef) {
  using Coef = typename PipeCoef<ND>::type;
  constexpr int NI = PipeArg<ND, PM>::N;
  constexpr int NC = sizeof(Coef) / sizeof(double);
  const int steps = a.steps;
  for (int p = 0; p < steps + 2; ++p) {
    if (p >= 1 && p <= steps && PIPE_ROLE_ON(a, 2)) {
      const int buf = (p - 1) & 1;
      double arg[ND], dead[ND];
#pragma unroll
      for (int c = 0; c < ND; ++c) arg[c] = 0.0;
      const double* src = s_int + ((buf * 4 + stage) * NI) * PIPE_SEG + seg;
#pragma unroll
      for (int e = 0; e < NI; ++e) arg[PipeArg<ND, PM>::idx[e]] = src[e * PIPE_SEG];
      Coef vc;
      if constexpr (ND == 12) rhs12<PM, true>(arg, L.tp, dead, vc);
      else rhs14<PM, true>(arg, L.tp, dead, vc);
      const double* o = reinterpret_cast<const double*>(&vc);
      double* dst = s_coef + (buf * 4 + stage) * Layout::template stage_doubles<NC>();
      // Layout::SCALED: every coefficient except the unit vector lhat (entries 14..16) is stored times the stage's
      // RK4 argument weight (h/2, h/2, h, h/6), so that the column lanes accumulate h a_s F c straight onto y
      // (the last stage carries h/2 = 3 h/6: the column lanes advance 3 y per step, col_dpp_step)
      const double as = Layout::SCALED ? ((stage == 2) ? L.h : 0.5 * L.h) : 1.0;
#pragma unroll
      for (int e = 0; e < NC; ++e) dst[Layout::template at<ND>(e, seg)] = (Layout::SCALED && (e < 14 || e > 16)) ? o[e] * as : o[e];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------- column role, two columns per lane
// lane = (segment, pair of columns); coefficients of the stage read from LDS (CoefByValue).
template <int ND>
__device__ __forceinline__ void pipe_role_columns2(const IndirectArgs& a, const PipeLane& L, const int seg, const int pair_raw,
                                                   const double* s_coef) {
  using Coef = typename PipeCoef<ND>::type;
  constexpr int NC = sizeof(Coef) / sizeof(double);
  const int steps = a.steps;
  const double h = L.h, h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0), w2 = L.w2;
  const bool col_lane = pair_raw < ND / 2;
  const int pair = col_lane ? pair_raw : 0;        // spare lanes shadow pair 0 and store nothing
  double y[2][ND];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < ND; ++r) y[j][r] = (r == 2 * pair + j) ? 1.0 : 0.0;
  for (int p = 0; p < steps + 2; ++p) {
    if (p >= 2 && PIPE_ROLE_ON(a, 4)) {
      const int buf = p & 1;
      double acc[2][ND], yt[2][ND];
#pragma unroll
      for (int stage = 0; stage < 4; ++stage) {
        Coef vc;
        double* v = reinterpret_cast<double*>(&vc);
        const double* src = s_coef + (buf * 4 + stage) * CoefByValue::stage_doubles<NC>();
#pragma unroll
        for (int e = 0; e < NC; ++e) v[e] = src[CoefByValue::at<ND>(e, seg)];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          double k[ND];
          if constexpr (ND == 12) { if (stage == 0) var_col12(vc, w2, y[j], k); else var_col12(vc, w2, yt[j], k); }
          else { if (stage == 0) var_col14(vc, w2, y[j], k); else var_col14(vc, w2, yt[j], k); }
#pragma unroll
          for (int c = 0; c < ND; ++c) {
            if (stage == 0) { acc[j][c] = __builtin_fma(h6, k[c], y[j][c]); yt[j][c] = __builtin_fma(h2, k[c], y[j][c]); }
            else if (stage == 1) { acc[j][c] = __builtin_fma(h3, k[c], acc[j][c]); yt[j][c] = __builtin_fma(h2, k[c], y[j][c]); }
            else if (stage == 2) { acc[j][c] = __builtin_fma(h3, k[c], acc[j][c]); yt[j][c] = __builtin_fma(h, k[c], y[j][c]); }
            else y[j][c] = __builtin_fma(h6, k[c], acc[j][c]);
          }
        }
      }
    }
    __syncthreads();
  }
  if (L.in_range && col_lane) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < ND; ++r) a.Phi[(long)((2 * pair + j) * ND + r) * a.ldp + L.s] = y[j][r];
  }
}


// lane = (row = segment, column); coefficients of the stage: CoefBySegment (scaled); one RK4 step = col_dpp_step.
template <int ND>
__device__ __forceinline__ void pipe_role_columns_dpp(const IndirectArgs& a, const PipeLane& L, const int seg, const int col,
                                                      const double* s_coef) {
  constexpr int NC = sizeof(typename PipeCoef<ND>::type) / sizeof(double);
  constexpr int SD = CoefBySegment::stage_doubles<NC>();
  const int steps = a.steps;
  const ColStepConst k(L.h, L.w2);
  double y[ND];
#pragma unroll
  for (int r = 0; r < ND; ++r) y[r] = (r == col) ? 1.0 : 0.0;
  for (int p = 0; p < steps + 2; ++p) {
    if (p >= 2 && PIPE_ROLE_ON(a, 4) && col < ND)     // the spare lanes of a row stay switched off: they are never DPP sources
      col_dpp_step<ND, SD>(s_coef + ((p & 1) * 4) * SD + CoefBySegment::lane_base(col, seg), k, p - 2, y);
    __syncthreads();
  }
  if (L.in_range && col < ND) {
#pragma unroll
    for (int r = 0; r < ND; ++r) a.Phi[(long)(col * ND + r) * a.ldp + L.s] = y[r] * a.stm_scale;
  }
}

// ------------------------------------------------------------------------------------------------------------ kernels
// Four waves, one per SIMD: wave 0 base, wave 1 coef, waves 2-3 columns (two per lane).  LDS 23 / 33 KB (ND = 12 / 14).
template <int ND, int PM>
__global__ __launch_bounds__(256, 2) void k_indirect_pipe(const IndirectArgs a) {
  constexpr int NI = PipeArg<ND, PM>::N;
  constexpr int NC = sizeof(typename PipeCoef<ND>::type) / sizeof(double);
  __shared__ double s_int[8 * NI * PIPE_SEG];
  __shared__ double s_coef[2 * 4 * CoefByValue::stage_doubles<NC>()];
  // Batches of several chip-fulls: two workgroups share a CU (<= 256 registers per lane), and the hardware gives
  // wave w of every workgroup on a CU the same SIMD.  Workgroups 256 apart (the ones that meet on a CU under the
  // round-robin dispatch over 8 XCDs x 32 CUs) therefore swap the role pairs, so that each SIMD carries
  // base + columns or coef + columns instead of 2 x base, 2 x coef, 2 x columns, 2 x columns.
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) ^ (((blockIdx.x >> 8) & 1) << 1);
  const int lane = threadIdx.x & 63;
  const int seg = lane & (PIPE_SEG - 1), slot = lane >> 4;
  const PipeLane L = pipe_lane<PM>(a, seg);
  if (!__syncthreads_or(L.mine)) return;         // workgroup-uniform
  if (wave == 0) pipe_role_base<ND, PM>(a, L, seg, slot, s_int);
  else if (wave == 1) pipe_role_coef<ND, PM, CoefByValue>(a, L, seg, slot, s_int, s_coef);
  else pipe_role_columns2<ND>(a, L, seg, (wave - 2) * 4 + slot, s_coef);
}

// Six waves.  The hardware places the waves of a workgroup on the four SIMDs round-robin (measured:
// tools/micro/dpp_probe.hip), so waves 0, 1, 4, 5 -- two per SIMD on two SIMDs -- take the columns (column wave cw owns
// segments 4 cw .. 4 cw + 3, one per DPP row) and waves 2 and 3, each alone on its SIMD, the base and coefficient roles.
// LDS 39 / 41 KB.
template <int ND, int PM>
__global__ __launch_bounds__(384) void k_indirect_pipe6(const IndirectArgs a) {
  constexpr int NI = PipeArg<ND, PM>::N;
  constexpr int NC = sizeof(typename PipeCoef<ND>::type) / sizeof(double);
  __shared__ double s_int[8 * NI * PIPE_SEG];
  __shared__ double s_coef[2 * 4 * CoefBySegment::stage_doubles<NC>()];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const bool col_wave = (wave != 2 && wave != 3);
  const int cw = wave < 2 ? wave : wave - 2;                  // waves 0, 1, 4, 5 -> column waves 0, 1, 2, 3
  const int seg = col_wave ? cw * 4 + (lane >> 4) : (lane & (PIPE_SEG - 1));
  const PipeLane L = pipe_lane<PM>(a, seg);
  if (!__syncthreads_or(L.mine)) return;         // workgroup-uniform
  if (wave == 2) pipe_role_base<ND, PM>(a, L, seg, lane >> 4, s_int);
  else if (wave == 3) pipe_role_coef<ND, PM, CoefBySegment>(a, L, seg, lane >> 4, s_int, s_coef);
  else pipe_role_columns_dpp<ND>(a, L, seg, lane & 15, s_coef);
}

template <int ND, int PM, bool SIX>
static hipError_t launch_pipe_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + PIPE_SEG - 1) / PIPE_SEG);
  if (SIX) hipLaunchKernelGGL((k_indirect_pipe6<ND, PM>), grid, dim3(384), 0, st, a);
  else hipLaunchKernelGGL((k_indirect_pipe<ND, PM>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

template <int ND, bool SIX>
static hipError_t launch_pipe_pm(int pm, const IndirectArgs& a0, hipStream_t st) {
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_pipe_one<ND, PM_P0, SIX>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_pipe_one<ND, PM_P1, SIX>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_pipe_one<ND, PM_P2, SIX>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_pipe_one<ND, PM_PGEN, SIX>(a, st);
  return e;
}

// RK4 only (the 13-stage methods use the cooperative kernel); steps >= 1.
hipError_t launch_indirect_stm_pipe(int ndim, int pm, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (a.steps < 1) return hipErrorInvalidValue;
  if (ndim == 12) return launch_pipe_pm<12, false>(pm, a, st);
  if (ndim == 14) return launch_pipe_pm<14, false>(pm, a, st);
  return hipErrorInvalidValue;
}

hipError_t launch_indirect_stm_pipe6(int ndim, int pm, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (a.steps < 1) return hipErrorInvalidValue;
  if (ndim == 12) return launch_pipe_pm<12, true>(pm, a, st);
  if (ndim == 14) return launch_pipe_pm<14, true>(pm, a, st);
  return hipErrorInvalidValue;
}

}  // namespace lto
