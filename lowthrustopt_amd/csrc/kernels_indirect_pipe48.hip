// kernels_indirect_pipe48.hip -- the three-role RK4 pipeline for LARGE batches: 48 (14-dim) or 44 (12-dim) segments and 16
// wavefronts per workgroup.
//
// The pipeline kernels of kernels_indirect_pipe.hip / _pipe8.hip are built for the latency regime (4 096 segments = one
// workgroup of 16 segments per CU): their base wave integrates 16 segments in 64 lanes.  Once the chip is full many times over
// (BASELINE configs[3]: 256 x 1 024 segments) only the number of instructions issued per segment counts, and there the per-lane
// kernel (every lane re-integrates the base state with three columns; 384 live registers, a fifth of its instructions AGPR
// moves) was level with them.  This form keeps the three roles and the one-step skew but fills the wavefronts:
//   wave 0       base: lane = segment (48 of 64 lanes), publishes the stage arguments stage by stage (no lane groups to
//                remember them in -- 12 stores per step more on a stream that is no longer the critical one)
//   waves 1-3    coefficients: lane = (segment, RK stage), 16 segments per wave, one step behind
//   waves 4-15   columns: DPP row = segment (12 or 14 column lanes + spare lanes), four segments per wave, two steps behind;
//                coefficients broadcast inside v_fmac_f64_dpp (col_dpp_step, pipe_common.hpp)
// Wave w runs on SIMD w mod 4: every SIMD carries one base / coefficient wave and three column waves (~1 350-1 500
// instructions per step).  Hand-overs double-buffered per step in LDS (12-dim: 18 + 101 KB; one workgroup per CU), one
// __syncthreads() per step, steps + 2 phases, every wave executes the same barriers, nothing spins.
// Instructions issued per segment and RK4 step: 541 / 48 + 3 x 400 / 48 + 12 x 310 / 48 = 114 against 184 in the per-lane kernel.
#include "pipe_common.hpp"

namespace lto {

// Segments per workgroup, SEG.  48: twelve column waves, three on every SIMD next to the base / a coefficient wave.  44 (round 4,
// 12-dim only): eleven column waves, the base wave's SIMD carries two of them (wave 12 leaves at once) -- with three, that SIMD
// issues 565 + 3 x 310 instructions per step against 400 + 3 x 310 on the others and every step waits for it: one round of
// 12 288 segments 163.9 us = 13.3 ns per segment, one of 11 264 139.3 us = 12.4 ns; C4 (262 144 segments) 3.56 -> 3.30 ms.  Which
// form a 12-dim sweep takes is the caller's choice by round cost (lto_api.hip).  14-dim: 48 only -- there a step lasts as long as
// the base wave's own dependent chain whatever shares its SIMD (189 us for 11 264 segments and for 12 288).


template <int ND, int PM, int SEG_> struct Pipe48 {
  using Arg = PipeArg<ND, PM>;
  static constexpr int NI = Arg::N;
  static constexpr int NC = sizeof(typename PipeCoef<ND>::type) / sizeof(double);
  static constexpr int SEG = SEG_;
  static constexpr int SD = SEG * CoefBySegment::LD;           // doubles of one stage's coefficient records
  static constexpr int INT_DOUBLES = 2 * 4 * NI * SEG;         // [step parity][stage][value][segment]
  static constexpr int COEF_DOUBLES = 2 * 4 * SD;                  // [step parity][stage][segment record]
  static constexpr bool PARK = (ND == 14) || (PM == PM_PGEN);      // base role with the step's base point and RK4 sum in LDS
  static constexpr int PARK_DOUBLES = PARK ? 2 * ND * SEG : 1;
};

template <int ND, int PM, int SEG>
__device__ __forceinline__ void pipe48_role_base(const IndirectArgs& a, const PipeLane& L, const int seg, const bool live, double* s_int) {
  using P = Pipe48<ND, PM, SEG>;
  constexpr int NI = P::NI, P48_SEG = P::SEG;
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
    if (p < steps) {
      double k[ND], yt[ND], acc[ND];
      double* slab = s_int + ((p & 1) * 4 * NI) * P48_SEG + seg;
      auto publish = [&](const int stage, const double (&arg)[ND]) {
        if (live) {
#pragma unroll
          for (int e = 0; e < NI; ++e) slab[(stage * NI + e) * P48_SEG] = arg[P::Arg::idx[e]];
        }
      };
      publish(0, y);
      rhs(y, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
      publish(1, yt);
      rhs(yt, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
      publish(2, yt);
      rhs(yt, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h, k[c], y[c]); }
      publish(3, yt);
      rhs(yt, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) y[c] = __builtin_fma(h6, k[c], acc[c]);
    }
    __syncthreads();
  }
  // (the segment's index and node pass through an empty asm: the compiler otherwise forms the ND addresses of x_{i+1} next to those
  // of x_i before the loop and parks them across it -- 20 spilled dwords in a kernel that runs at 128 registers per lane)
  int s_e = L.s;
  long node_e = L.node;
  if constexpr (ND == 12) {
    // (14-dim: the same pin moves its spills INTO the step loop: 193 -> 233 us at 12 288 segments)
    // Round 5: the node index is not carried across the loop at all (it was the one value the 44-segment form still spilled: a
    // store before the loop, a load behind it) but formed again from the segment index, which is pinned so that the division
    // cannot be hoisted in front of the loop.
    asm volatile("" : "+v"(s_e));
    const int traj_e = s_e / a.seg_per_traj;
    node_e = (long)traj_e * a.n_nodes + (s_e - traj_e * a.seg_per_traj);
  }
  if (L.in_range && live) {
    if (a.defect) {
#pragma unroll
      for (int c = 0; c < ND; ++c) a.defect[c * a.ldd + s_e] = y[c] - a.X[c * a.ldx + node_e + 1];
    }
    if (a.errors) a.errors[s_e] = 0.0;
    if (a.nacc) a.nacc[s_e] = steps;
    if (a.nrej) a.nrej[s_e] = 0;
  }
}

// Base role for the instantiations whose four RK4 arrays do not fit: y, k, the stage argument and the RK4 sum are 4 x ND doubles --
// 112 of the 128 registers a lane has at four wavefronts per SIMD (14-dim) -- and the rest went to scratch inside the step loop
// (67 scratch operations per step; 12 288 segments 14-dim: 193 us per sweep, 233 us once the epilogue's addresses were pinned).
// Here the step's base point and the running sum live in LDS (s_park [2][ND][segment]): written by this lane, read back through an
// offset the compiler cannot see through (it would otherwise forward the stored registers and keep them alive) -- 126 LDS
// operations per step on the base wave, no scratch.  Same operations in the same order as pipe48_role_base: same bits.
template <int ND, int PM, int SEG>
__device__ __forceinline__ void pipe48_role_base_parked(const IndirectArgs& a, const PipeLane& L, const int seg, const bool live,
                                                        double* s_int, double* s_park) {
  using P = Pipe48<ND, PM, SEG>;
  constexpr int NI = P::NI, P48_SEG = P::SEG;
  const int steps = a.steps;
  const double h = L.h, h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);
  auto rhs = [&](const double (&y)[ND], double (&k)[ND]) {
    if constexpr (ND == 12) rhs12_base<PM>(y, L.tp, k);
    else rhs14_base<PM>(y, L.tp, k);
  };
  int off = seg;
  asm volatile("" : "+v"(off));
  double* yw = s_park + seg;
  double* sw = s_park + ND * P48_SEG + seg;
  const double* yr = s_park + off;
  const double* sr = s_park + ND * P48_SEG + off;
  double y[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + L.node];
  for (int p = 0; p < steps + 2; ++p) {
    if (p < steps) {
      double k[ND], yt[ND];
      double* slab = s_int + ((p & 1) * 4 * NI) * P48_SEG + seg;
      auto publish = [&](const int stage, const double (&arg)[ND]) {
        if (live) {
#pragma unroll
          for (int e = 0; e < NI; ++e) slab[(stage * NI + e) * P48_SEG] = arg[P::Arg::idx[e]];
        }
      };
#pragma unroll
      for (int c = 0; c < ND; ++c) yw[c * P48_SEG] = y[c];
      publish(0, y);
      rhs(y, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) { sw[c * P48_SEG] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
      publish(1, yt);
      rhs(yt, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) { sw[c * P48_SEG] = __builtin_fma(h3, k[c], sr[c * P48_SEG]); yt[c] = __builtin_fma(h2, k[c], yr[c * P48_SEG]); }
      publish(2, yt);
      rhs(yt, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) { sw[c * P48_SEG] = __builtin_fma(h3, k[c], sr[c * P48_SEG]); yt[c] = __builtin_fma(h, k[c], yr[c * P48_SEG]); }
      publish(3, yt);
      rhs(yt, k);
#pragma unroll
      for (int c = 0; c < ND; ++c) y[c] = __builtin_fma(h6, k[c], sr[c * P48_SEG]);
    }
    __syncthreads();
  }
  if (L.in_range && live) {
    if (a.defect) {
#pragma unroll
      for (int c = 0; c < ND; ++c) a.defect[c * a.ldd + L.s] = y[c] - a.X[c * a.ldx + L.node + 1];
    }
    if (a.errors) a.errors[L.s] = 0.0;
    if (a.nacc) a.nacc[L.s] = steps;
    if (a.nrej) a.nrej[L.s] = 0;
  }
}

// lane = (segment, RK stage): the four stages of step p - 1 are built side by side in phase p; every coefficient except the
// unit vector lhat (entries 14..16) is stored times the stage's RK4 argument weight (h/2, h/2, h, h/2: col_dpp_step).
template <int ND, int PM, int SEG>
__device__ __forceinline__ void pipe48_role_coef(const IndirectArgs& a, const PipeLane& L, const int seg, const int stage,
                                                 const double* s_int, double* s_coef) {
  using P = Pipe48<ND, PM, SEG>;
  using Coef = typename PipeCoef<ND>::type;
  constexpr int NI = P::NI, NC = P::NC, P48_SEG = P::SEG;
  const int steps = a.steps;
  const double as = (stage == 2) ? L.h : 0.5 * L.h;
  for (int p = 0; p < steps + 2; ++p) {
    if (p >= 1 && p <= steps) {
      const int buf = (p - 1) & 1;
      double arg[ND], dead[ND];
#pragma unroll
      for (int c = 0; c < ND; ++c) arg[c] = 0.0;
      const double* src = s_int + ((buf * 4 + stage) * NI) * P48_SEG + seg;
#pragma unroll
      for (int e = 0; e < NI; ++e) arg[P::Arg::idx[e]] = src[e * P48_SEG];
      Coef vc;
      if constexpr (ND == 12) rhs12<PM, true>(arg, L.tp, dead, vc);
      else rhs14<PM, true>(arg, L.tp, dead, vc);
      const double* o = reinterpret_cast<const double*>(&vc);
      double* dst = s_coef + (buf * 4 + stage) * P::SD;
#pragma unroll
      for (int e = 0; e < NC; ++e) dst[CoefBySegment::at<ND>(e, seg)] = (e < 14 || e > 16) ? o[e] * as : o[e];
    }
    __syncthreads();
  }
}

// lane = (row = segment, column); one RK4 step = col_dpp_step
template <int ND, int SEG>
__device__ __forceinline__ void pipe48_role_columns(const IndirectArgs& a, const PipeLane& L, const int seg, const int col,
                                                    const double* s_coef) {
  constexpr int SD = SEG * CoefBySegment::LD;
  const int steps = a.steps;
  const ColStepConst k(L.h, L.w2);
  double y[ND];
#pragma unroll
  for (int r = 0; r < ND; ++r) y[r] = (r == col) ? 1.0 : 0.0;
  for (int p = 0; p < steps + 2; ++p) {
    if (p >= 2 && col < ND)                    // the spare lanes of a row stay switched off: they are never DPP sources
      col_dpp_step<ND, SD>(s_coef + ((p & 1) * 4) * SD + CoefBySegment::lane_base(col, seg), k, p - 2, y);
    __syncthreads();
  }
  if (L.in_range && col < ND) {
#pragma unroll
    for (int r = 0; r < ND; ++r) a.Phi[(long)(col * ND + r) * a.ldp + L.s] = y[r] * a.stm_scale;
  }
}

template <int ND, int PM, int SEG>
__global__ __launch_bounds__(1024) void k_indirect_pipe48(const IndirectArgs a) {
  using P = Pipe48<ND, PM, SEG>;
  __shared__ double s_int[P::INT_DOUBLES];
  __shared__ double s_coef[P::COEF_DOUBLES];
  __shared__ double s_park[P::PARK_DOUBLES];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // wave 0: base (lane = segment; lanes past the last segment shadow it); waves 1-3: coefficients of segments 16 (w - 1) ..;
  // waves 4-15: column jobs, four segments each.  SEG = 44: wave 12 (the fourth wave of the base wave's SIMD) leaves at once and
  // waves 13..15 are column jobs 8..10.
  constexpr bool DROP12 = (SEG == 44);
  const int cj = (DROP12 && wave > 12) ? wave - 5 : wave - 4;
  const int seg_raw = (wave == 0) ? lane : (wave < 4) ? (wave - 1) * 16 + (lane & 15) : cj * 4 + (lane >> 4);
  const int seg = seg_raw < SEG ? seg_raw : SEG - 1;   // shadow lanes (base, last coefficient wave) repeat the last segment: same values to the same places
  const PipeLane L = pipe_lane<PM, SEG>(a, seg);
  if (!__syncthreads_or(L.mine)) return;         // workgroup-uniform
  if (DROP12 && wave == 12) return;              // before the first step barrier: the hardware barrier counts the waves still alive
  // Every wave shares its SIMD with three others and all meet at one barrier per step: the two roles with the long dependent
  // streams and the fewest instructions (base 565, coefficients ~400 per step) issue first, or the step would last four times
  // the base wave's stream.
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    if constexpr (P::PARK) pipe48_role_base_parked<ND, PM, SEG>(a, L, seg, lane < SEG, s_int, s_park);
    else pipe48_role_base<ND, PM, SEG>(a, L, seg, lane < SEG, s_int);
  }
  else if (wave < 4) { __builtin_amdgcn_s_setprio(2); pipe48_role_coef<ND, PM, SEG>(a, L, seg, lane >> 4, s_int, s_coef); }
  else pipe48_role_columns<ND, SEG>(a, L, seg, lane & 15, s_coef);
}

template <int ND, int PM, int SEG>
static hipError_t launch_pipe48_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + SEG - 1) / SEG);
  hipLaunchKernelGGL((k_indirect_pipe48<ND, PM, SEG>), grid, dim3(1024), 0, st, a);
  return hipGetLastError();
}

template <int ND, int SEG>
static hipError_t launch_pipe48_pm(int pm, const IndirectArgs& a0, hipStream_t st) {
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_pipe48_one<ND, PM_P0, SEG>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_pipe48_one<ND, PM_P1, SEG>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_pipe48_one<ND, PM_P2, SEG>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_pipe48_one<ND, PM_PGEN, SEG>(a, st);
  return e;
}

// RK4 only; steps >= 1.  seg44: the 44-segment form (12-dim only; ignored for 14).
hipError_t launch_indirect_stm_pipe48(int ndim, int pm, const IndirectArgs& a, bool seg44, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (a.steps < 1) return hipErrorInvalidValue;
  if (ndim == 12) return seg44 ? launch_pipe48_pm<12, 44>(pm, a, st) : launch_pipe48_pm<12, 48>(pm, a, st);
  if (ndim == 14) return launch_pipe48_pm<14, 48>(pm, a, st);
  return hipErrorInvalidValue;
}

}  // namespace lto
