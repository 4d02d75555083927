// kernels_indirect_pipe.hip -- three-role software pipeline for the fixed-step RK4 STM sweep (BASELINE configs[1]).
//
// At 4 096 segments the chip offers exactly 16 lanes per segment (1 024 SIMDs x 64 lanes) and the sweep lasts as long
// as ONE lane's instruction stream: the per-lane kernel (indirect_kernel.hpp) spends ~60 % of that stream on work
// every column lane of a segment repeats (base trajectory, gravity, control law), and the cooperative kernel
// (kernels_indirect_coop.hip) pays one workgroup barrier per RK stage with the base role's full RHS + coefficient
// build on the critical path.  Here a workgroup (4 wavefronts, one per SIMD) owns 16 segments and runs three roles
// that are skewed by one RK4 STEP each, so they execute concurrently and meet at ONE barrier per step:
//
//   wave 0  base      integrates the ND-dim base state (rhs*_base: the RHS alone -- the shortest instruction stream)
//                     and publishes, per stage, the part of the stage argument the coefficients depend on
//   wave 1  coef      one step behind: lane = (segment, RK stage); builds G, H, U (+ mass couplings) at those
//                     arguments and publishes them
//   waves 2,3 columns two steps behind: lane = (segment, PAIR of STM columns); c' = F(t) c with the coefficients
//                     read from LDS -- no gravity, no control law, no base state in these lanes
//
// Both hand-overs are double-buffered per step in LDS (46 KB for ND = 14); a sweep of n steps takes n + 2 phases.
// Two columns per lane (7 pairs x 16 segments = 112 lanes for ND = 14) is what lets the column role fit the two
// remaining SIMDs, so that every SIMD carries one wave of ~170-190 fp64 instructions per stage instead of the
// per-lane kernel's 281.  Every wavefront executes exactly steps + 2 barriers; nothing spins.
//
// Replaces the serial loop of jacobianCalc (src/multiShoot_CRTBP_indirect.jl:93-146) for the fixed-step setting.
#include "kernels.hpp"

namespace lto {

constexpr int PIPE_SEG = 16;   // segments per workgroup

// The coefficients of an RK stage are functions of the stage argument's position, lambda_v (and, for ND = 14, mass and
// lambda_m) only: that is all the base wave publishes (6 / 8 doubles per stage -- LDS stores of a 64-lane wave cost
// issue time on the critical path), in the order of PipeArg<ND>::idx.
template <int ND> struct PipeArg {
  static constexpr int N = 6;
  static constexpr int idx[6] = {0, 1, 2, 9, 10, 11};
  using Coef = VarCoef12;
};
template <> struct PipeArg<14> {
  static constexpr int N = 8;
  static constexpr int idx[8] = {0, 1, 2, 6, 10, 11, 12, 13};
  using Coef = VarCoef14;
};

template <int ND, int PM>
__device__ __forceinline__ void pipe_base_rhs(const double (&y)[ND], const TrajParams& tp, double (&k)[ND]) {
  if constexpr (ND == 12) rhs12_base<PM>(y, tp, k);
  else rhs14_base<PM>(y, tp, k);
}
// G, H, U (+ mass couplings) at the stage argument `arg` (components outside PipeArg<ND>::idx are unused: the slopes
// this call also produces are dead code)
template <int ND, int PM>
__device__ __forceinline__ void pipe_coef(const double (&arg)[ND], const TrajParams& tp, typename PipeArg<ND>::Coef& vc) {
  double dead[ND];
  if constexpr (ND == 12) rhs12<PM, true>(arg, tp, dead, vc);
  else rhs14<PM, true>(arg, tp, dead, vc);
}
template <int ND>
__device__ __forceinline__ void pipe_col(const typename PipeArg<ND>::Coef& vc, const double w2, const double (&c)[ND],
                                         double (&dc)[ND]) {
  if constexpr (ND == 12) var_col12(vc, w2, c, dc);
  else var_col14(vc, w2, c, dc);
}

template <int ND, int PM>
__global__ __launch_bounds__(256) void k_indirect_pipe(const IndirectArgs a) {
  using Coef = typename PipeArg<ND>::Coef;
  constexpr int NI = PipeArg<ND>::N;
  constexpr int NC = sizeof(Coef) / sizeof(double);
  constexpr int NPAIR = ND / 2;

  // base -> coef: slab = parity * 4 + stage, [value][segment]; slabs 8..10 take the stores of the base wave's three
  // spare lane groups, so that publishing needs no EXEC branch (a branch per stage would cut the base role's one
  // basic block per step into five and keep the scheduler from overlapping the stages' rsqrt / exp chains)
  __shared__ double s_int[8 + 3][NI][PIPE_SEG];
  __shared__ double s_coef[2][4][NC][PIPE_SEG];   // coef -> columns: [step parity][stage][value][segment]

  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int seg = lane & (PIPE_SEG - 1), slot = lane >> 4;
  const int s_raw = blockIdx.x * PIPE_SEG + seg;
  const int s_lin = s_raw < a.S ? s_raw : a.S - 1;             // shadow lanes repeat the last segment
  const int s = a.order ? a.order[s_lin] : s_lin;
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const double span = a.t[tg + 1] - a.t[tg];
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  const double w2 = 2.0 * tp.omega;
  // mixed-class batch: segments of another control-law class belong to that class's launch; here they run through the
  // barriers without storing
  const bool mine = !a.class_filter || p_class(tp.p) == PM;
  if (!__syncthreads_or(mine)) return;         // workgroup-uniform

  const int steps = a.steps;
  const double h = span / (double)steps;
  const double h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);

  const bool in_range = (s_raw < a.S) && mine;
  const int nphase = steps + 2;   // every role loop below executes exactly nphase barriers
#ifdef PIPE_PROBE
  const int probe = a.max_steps;  // development build only: bit 0 / 1 / 2 switches the base / coef / column work off
#else
  constexpr int probe = 0;
#endif

  if (wave == 0) {
    // -------------------------------------------------------------------- base: step p in phase p
    double y[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + node];
    for (int p = 0; p < nphase; ++p) {
      if (p < steps && !(probe & 1)) {
        const int buf = p & 1;
        double k[ND], yt[ND], acc[ND];
        auto publish = [&](int stage, const double (&arg)[ND]) {
          double* dst = &s_int[slot == 0 ? buf * 4 + stage : 7 + slot][0][seg];
#pragma unroll
          for (int e = 0; e < NI; ++e) dst[e * PIPE_SEG] = arg[PipeArg<ND>::idx[e]];
        };
        publish(0, y);
        pipe_base_rhs<ND, PM>(y, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
        publish(1, yt);
        pipe_base_rhs<ND, PM>(yt, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
        publish(2, yt);
        pipe_base_rhs<ND, PM>(yt, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h, k[c], y[c]); }
        publish(3, yt);
        pipe_base_rhs<ND, PM>(yt, tp, k);
#pragma unroll
        for (int c = 0; c < ND; ++c) y[c] = __builtin_fma(h6, k[c], acc[c]);
      }
      __syncthreads();
    }
    if (in_range && slot == 0) {
      if (a.defect) {
#pragma unroll
        for (int c = 0; c < ND; ++c) a.defect[c * a.ldd + s] = y[c] - a.X[c * a.ldx + node + 1];
      }
      if (a.errors) a.errors[s] = 0.0;
      if (a.nacc) a.nacc[s] = steps;
      if (a.nrej) a.nrej[s] = 0;
    }
  } else if (wave == 1) {
    // -------------------------------------------------------------------- coefficients of step p - 1 in phase p
    // lane = (segment, RK stage): the four stages of a step are built side by side
    for (int p = 0; p < nphase; ++p) {
      if (p >= 1 && p <= steps && !(probe & 2)) {
        const int buf = (p - 1) & 1;
        double arg[ND];
#pragma unroll
        for (int c = 0; c < ND; ++c) arg[c] = 0.0;
        const double* src = &s_int[buf * 4 + slot][0][seg];
#pragma unroll
        for (int e = 0; e < NI; ++e) arg[PipeArg<ND>::idx[e]] = src[e * PIPE_SEG];
        Coef vc;
        pipe_coef<ND, PM>(arg, tp, vc);
        const double* o = reinterpret_cast<const double*>(&vc);
        double* dst = &s_coef[buf][slot][0][seg];
#pragma unroll
        for (int e = 0; e < NC; ++e) dst[e * PIPE_SEG] = o[e];
      }
      __syncthreads();
    }
  } else {
    // -------------------------------------------------------------------- columns: step p - 2 in phase p
    const int pair_raw = (wave - 2) * 4 + slot;
    const bool col_lane = pair_raw < NPAIR;
    const int pair = col_lane ? pair_raw : 0;        // spare lanes shadow pair 0 and store nothing
    double y[2][ND];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < ND; ++r) y[j][r] = (r == 2 * pair + j) ? 1.0 : 0.0;
    for (int p = 0; p < nphase; ++p) {
      if (p >= 2 && !(probe & 4)) {
        const int buf = p & 1;
        double acc[2][ND], yt[2][ND];
#pragma unroll
        for (int stage = 0; stage < 4; ++stage) {
          Coef vc;
          double* v = reinterpret_cast<double*>(&vc);
#pragma unroll
          for (int e = 0; e < NC; ++e) v[e] = s_coef[buf][stage][e][seg];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            double k[ND];
            if (stage == 0) pipe_col<ND>(vc, w2, y[j], k);
            else pipe_col<ND>(vc, w2, yt[j], k);
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
    if (in_range && col_lane) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < ND; ++r) a.Phi[(long)((2 * pair + j) * ND + r) * a.ldp + s] = y[j][r];
    }
  }
}

template <int ND, int PM>
static hipError_t launch_pipe_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + PIPE_SEG - 1) / PIPE_SEG);
  hipLaunchKernelGGL((k_indirect_pipe<ND, PM>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

template <int ND>
static hipError_t launch_pipe_pm(int pm, const IndirectArgs& a0, hipStream_t st) {
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_pipe_one<ND, PM_P0>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_pipe_one<ND, PM_P1>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_pipe_one<ND, PM_P2>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_pipe_one<ND, PM_PGEN>(a, st);
  return e;
}

// RK4 only (the 13-stage methods use the cooperative kernel); steps >= 1.
hipError_t launch_indirect_stm_pipe(int ndim, int pm, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (a.steps < 1) return hipErrorInvalidValue;
  if (ndim == 12) return launch_pipe_pm<12>(pm, a, st);
  if (ndim == 14) return launch_pipe_pm<14>(pm, a, st);
  return hipErrorInvalidValue;
}

}  // namespace lto
