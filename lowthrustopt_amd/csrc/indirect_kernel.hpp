// indirect_kernel.hpp -- batched propagation of the 12-dim state+costate system and its STM.
//
// Replaces the serial `for i = 1:n_nodes-1` loops of defectCalc / jacobianCalc in
// src/multiShoot_CRTBP_indirect.jl:63-90 / :93-146.  One shooting segment (x one group of STM
// columns) per lane; state, RK stages and STM columns live in VGPRs for the whole integration; HBM is
// touched once on entry (node i, node i+1, t_i, t_{i+1}) and once on exit (defect, Phi), all
// struct-of-arrays so a wavefront moves 512 contiguous bytes per component.
//
// Why columns per lane instead of the whole 12x12 STM: 12 + 144 doubles x 3-4 RK work vectors
// exceeds the 512-VGPR file.  STM columns are independent given the base trajectory
// (Phi_dot = F(y(t)) Phi), so lane (segment s, column group g) re-integrates the cheap base state
// together with COLS columns.  The column group is blockIdx.y, i.e. wave-uniform.
#pragma once
#include <type_traits>
#include "kernels.hpp"
#include "rk.hpp"

namespace lto {

// ND = 12: the reference's state+costate system.  ND = 14: + mass and mass costate (extension, dynamics.hpp).
template <int ND, int PM, int COLS>
struct SysIndirect {
  static constexpr int DIM = ND + ND * COLS;
  double w2;
  TrajParams tp;
  __device__ __forceinline__ void rhs(const double (&y)[DIM], double (&k)[DIM]) const {
    if constexpr (ND == 12 && COLS == 1) {
      rhs12_fused1<PM>(y, tp, w2, k);
      return;
    }
    if constexpr (ND == 14 && COLS == 1) {
      rhs14_fused1<PM>(y, tp, w2, k);
      return;
    }
    if constexpr (COLS == 0) {     // defect-only sweeps: the lean base RHS of the pipeline kernels (no selects, short exp)
      if constexpr (ND == 12) rhs12_base<PM>(y, tp, k);
      else rhs14_base<PM, true>(y, tp, k);
      return;
    }
    double yb[ND], kb[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) yb[i] = y[i];
    typename std::conditional<ND == 12, VarCoef12, VarCoef14>::type vc;
    if constexpr (ND == 12) rhs12<PM, (COLS > 0)>(yb, tp, kb, vc);
    else rhs14<PM, (COLS > 0)>(yb, tp, kb, vc);
#pragma unroll
    for (int i = 0; i < ND; ++i) k[i] = kb[i];
#pragma unroll
    for (int j = 0; j < COLS; ++j) {
      double c[ND], dc[ND];
#pragma unroll
      for (int i = 0; i < ND; ++i) c[i] = y[ND + ND * j + i];
      if constexpr (ND == 12) var_col12(vc, w2, c, dc);
      else var_col14(vc, w2, c, dc);
#pragma unroll
      for (int i = 0; i < ND; ++i) k[ND + ND * j + i] = dc[i];
    }
  }
};

// x^(-1/8) and x^(1/9) without pow()
__device__ __forceinline__ double pow_m8th(double x) { return 1.0 / sqrt(sqrt(sqrt(x))); }
__device__ __forceinline__ double pow_9th(double x) { return cbrt(cbrt(x)); }

// ode78 (GeneralCode/ode.jl:479-534): h0 = span/50, hmax = span/2.5, hmin = span/1e7; accept when
// delta <= tau = tol max(|x|_inf, 1); h <- min(hmax, 0.8 h (tau/delta)^(1/8)).  Error and |x|_inf run
// over the first NERR components (the base state), so every column lane of a segment takes the same
// step sequence.
template <class Sys, int NERR>
__device__ __forceinline__ void run_rkf78_adaptive(const Sys& sys, const double span, const double tol, const int max_steps,
                                                   double (&y)[Sys::DIM], int& nacc, int& nrej) {
  constexpr int D = Sys::DIM;
  const double hmax = span / 2.5, hmin = span / 1e7;
  double h = span / 50.0, t = 0.0;
  nacc = 0; nrej = 0;
  while (t < span && h >= hmin && nacc + nrej < max_steps) {
    if (t + h > span) h = span - t;
    double yn[D];
    double delta;
    delta = rkf78_step<Sys, NERR, true>(sys, h, y, yn);
    double nx = 0.0;
#pragma unroll
    for (int i = 0; i < NERR; ++i) nx = fmax(nx, fabs(y[i]));
    const double tau = tol * fmax(nx, 1.0);
    if (delta <= tau) {
      t += h;
#pragma unroll
      for (int i = 0; i < D; ++i) y[i] = yn[i];
      ++nacc;
    } else {
      ++nrej;
      if (delta != delta) {                   // NaN step: poison the result instead of returning a partial one
#pragma unroll
        for (int i = 0; i < D; ++i) y[i] = delta;
        t = span;                             // (fmin below would drop the NaN and keep stepping)
      }
    }
    if (delta == 0.0) delta = 1e-16;
    h = fmin(hmax, 0.8 * h * sqrt(sqrt(sqrt(tau / delta))));
  }
  // not at t1 (max_steps used up, ode78's step-size floor ode.jl:479,524, or a decreasing grid: forward integration
  // only): no result -- NaN, never a state at some t < t1 that looks propagated
  if (t < span || !(span >= 0.0)) {           // a NaN span counts as a negative one
#pragma unroll
    for (int i = 0; i < D; ++i) y[i] = __builtin_nan("");
  }
}

// Adaptive DOP853 (DESIGN.md 'Integrators'): initial step by Hairer's d0/d1/d2 rule, accept if err < 1, factor = min(10, 0.9 err^(-1/8)) (<= 1 after a rejection),
// rejection factor max(0.2, 0.9 err^(-1/8)).
template <class Sys, int NERR>
__device__ __forceinline__ void run_dop853(const Sys& sys, const double span, const double rtol, const double atol,
                                           const int max_steps, double (&y)[Sys::DIM], int& nacc, int& nrej) {
  constexpr int D = Sys::DIM;
  double K[13][D];
  nacc = 0; nrej = 0;
  if (!(span > 0.0)) {
    if (span != 0.0) {     // decreasing grid (forward integration only) or NaN span: no result
#pragma unroll
      for (int i = 0; i < D; ++i) y[i] = __builtin_nan("");
    }
    return;
  }
  sys.rhs(y, K[0]);
  double h_abs;
  {
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int i = 0; i < NERR; ++i) {
      const double isc = rcp_nr(__builtin_fma(rtol, fabs(y[i]), atol));
      d0 = __builtin_fma(y[i] * isc, y[i] * isc, d0);
      d1 = __builtin_fma(K[0][i] * isc, K[0][i] * isc, d1);
    }
    d0 = sqrt(d0 / NERR); d1 = sqrt(d1 / NERR);
    const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    double yt[D];
#pragma unroll
    for (int i = 0; i < D; ++i) yt[i] = __builtin_fma(h0, K[0][i], y[i]);
    sys.rhs(yt, K[1]);
    double d2 = 0.0;
#pragma unroll
    for (int i = 0; i < NERR; ++i) {
      const double isc = rcp_nr(__builtin_fma(rtol, fabs(y[i]), atol));
      const double df = (K[1][i] - K[0][i]) * isc;
      d2 = __builtin_fma(df, df, d2);
    }
    d2 = sqrt(d2 / NERR) / h0;
    const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow_9th(0.01 / fmax(d1, d2));
    h_abs = fmin(fmin(100.0 * h0, h1), span);
  }
  double t = 0.0;
  // Flags are kept as per-lane doubles, not bools: hipcc 7.2 turns per-lane booleans that stay live across these
  // spill-heavy bodies into SGPR lane masks that are spilled/reloaded, and a 14-dim kernel returned garbage that
  // way (DESIGN.md "Compiler hazards").
  double rejected = 0.0;
  while (t < span && nacc + nrej < max_steps) {
    double h = h_abs;
    double last = 0.0;
    if (t + h >= span) { h = span - t; last = 1.0; }
    double yn[D];
    double E5, E3;
    (void)dop853_try<Sys, NERR>(sys, h, rtol, atol, y, K, yn, E5, E3);
    // the step decision every DOP853 kernel of this library takes (rk.hpp dp8_decide): the one- / two- / four-lane defect sweeps
    // AUTO switches between share one controller (advisor finding, round 4)
    double h_next, accept, bad;
    dp8_decide(E5, E3, h, rejected, (double)NERR, h_next, accept, bad);
    h_abs = h_next;
    if (accept != 0.0) {
      t = (last != 0.0) ? span : t + h;
#pragma unroll
      for (int i = 0; i < D; ++i) { y[i] = yn[i]; K[0][i] = K[12][i]; }
      ++nacc;
      rejected = 0.0;
    } else {
      rejected = 1.0;
      ++nrej;
      if (bad != 0.0) {                       // a NaN never recovers: poison and stop instead of max_steps retries
#pragma unroll
        for (int i = 0; i < D; ++i) y[i] = bad;
        t = span;
      }
    }
  }
  if (t < span) {                             // max_steps trial steps used up before t1: no result
#pragma unroll
    for (int i = 0; i < D; ++i) y[i] = __builtin_nan("");
  }
}

// Propagate y over `span` with the plan's integrator (shared by the sweep and the dense-output kernels).
template <class Sys, int ND, int METHOD>
__device__ __forceinline__ void advance(const Sys& sys, const double span, const IndirectArgs& a, double (&y)[Sys::DIM],
                                        int& nacc, int& nrej, double& maxErr) {
  constexpr int D = Sys::DIM;
  if (METHOD == M_RK4) {
    const double h = span / (double)a.steps;
    for (int k = 0; k < a.steps; ++k) rk4_step(sys, h, y);
    nacc += a.steps;
  } else if (METHOD == M_RKF78_FIXED) {
    const double h = span / (double)a.steps;
    for (int k = 0; k < a.steps; ++k) {
      double yn[D];
      double delta;
      delta = rkf78_step<Sys, ND>(sys, h, y, yn);
      maxErr = fmax(maxErr, delta);
#pragma unroll
      for (int c = 0; c < D; ++c) y[c] = yn[c];
    }
    nacc += a.steps;
  } else if (METHOD == M_RKF78_ADAPTIVE) {
    int na = 0, nr = 0;
    run_rkf78_adaptive<Sys, ND>(sys, span, a.rtol, a.max_steps, y, na, nr);
    nacc += na; nrej += nr;
  } else {
    int na = 0, nr = 0;
    run_dop853<Sys, D>(sys, span, a.rtol, a.atol, a.max_steps, y, na, nr);
    nacc += na; nrej += nr;
  }
}

// Dense output (replaces the per-segment re-propagation of densify, src/HelperFunctions.jl:51-101):
// lane = segment; the segment's sample times are td[first[s] .. first[s+1]) (sorted, inside [t_i, t_{i+1})); the
// lane integrates from sample to sample and stores the state at each one.  The reference evaluates Vern8's
// interpolant at those times; stepping exactly onto them is at least as accurate.  If `final_state` is set the
// last segment of each trajectory also stores x(t_n) (the `sol_forward[:,end]` column densify appends, :94-97).
template <int ND, int PM, int METHOD>
__global__ __launch_bounds__(64) void k_indirect_dense(const IndirectArgs a, const DenseArgs d) {
  const int s = blockIdx.x * 64 + threadIdx.x;
  if (s >= a.S) return;
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  using Sys = SysIndirect<ND, PM, 0>;
  Sys sys;
  sys.tp = a.tp[(long)traj * a.tp_stride];
  if (a.class_filter && p_class(sys.tp.p) != PM) return;
  sys.w2 = 2.0 * sys.tp.omega;
  double y[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + node];
  double tcur = a.t[tg];
  int nacc = 0, nrej = 0;
  double maxErr = 0.0;
  const int j0 = d.first[s], j1 = d.first[s + 1];
  for (int j = j0; j < j1; ++j) {
    const double ts = d.td[j];
    if (ts > tcur) { advance<Sys, ND, METHOD>(sys, ts - tcur, a, y, nacc, nrej, maxErr); tcur = ts; }
#pragma unroll
    for (int c = 0; c < ND; ++c) d.Y[c * d.ldy + j] = y[c];
  }
  if (d.final_state && i == a.seg_per_traj - 1) {
    const double te = a.t[tg + 1];
    if (te > tcur) advance<Sys, ND, METHOD>(sys, te - tcur, a, y, nacc, nrej, maxErr);
#pragma unroll
    for (int c = 0; c < ND; ++c) d.final_state[c * (a.S / a.seg_per_traj) + traj] = y[c];
  }
}

// COLS = 0: defect only (K1).  COLS >= 1: lane integrates base + COLS STM columns (K2); the
// column group is blockIdx.y and the g = 0 lanes also emit the defect.
template <int ND, int PM, int METHOD, int COLS>
__global__ __launch_bounds__(64) void k_indirect(const IndirectArgs a) {
  const int sl = xcd_unit(a, blockIdx.x, gridDim.x) * 64 + threadIdx.x;   // an XCD's wavefronts own a contiguous range of segments (kernels.hpp)
  if (sl >= a.S) return;
  const int s = a.order ? a.order[sl] : sl;    // balanced order: neighbouring lanes take similar numbers of steps
  const int g = blockIdx.y;
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const double span = a.t[tg + 1] - a.t[tg];

  using Sys = SysIndirect<ND, PM, COLS>;
  constexpr int D = Sys::DIM;
  Sys sys;
  sys.tp = a.tp[(long)traj * a.tp_stride];
  if (a.class_filter && p_class(sys.tp.p) != PM) return;   // mixed-class batch: another launch owns this trajectory
  sys.w2 = 2.0 * sys.tp.omega;

  double y[D];
#pragma unroll
  for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + node];
#pragma unroll
  for (int j = 0; j < COLS; ++j)
#pragma unroll
    for (int r = 0; r < ND; ++r) y[ND + ND * j + r] = (r == g * COLS + j) ? 1.0 : 0.0;

  double maxErr = 0.0;
  int nacc = 0, nrej = 0;
  advance<Sys, ND, METHOD>(sys, span, a, y, nacc, nrej, maxErr);

  if (COLS > 0) {
#pragma unroll
    for (int j = 0; j < COLS; ++j)
#pragma unroll
      for (int r = 0; r < ND; ++r) a.Phi[(long)((g * COLS + j) * ND + r) * a.ldp + s] = y[ND + ND * j + r];
  }
  if (g == 0) {
    if (a.defect) {
#pragma unroll
      for (int c = 0; c < ND; ++c) a.defect[c * a.ldd + s] = y[c] - a.X[c * a.ldx + node + 1];
    }
    if (a.errors) a.errors[s] = maxErr;
    if (a.nacc) a.nacc[s] = nacc;
    if (a.nrej) a.nrej[s] = nrej;
  }
}

template <int ND, int PM, int METHOD, int COLS>
static hipError_t launch_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + 63) / 64, COLS > 0 ? ND / COLS : 1);
  hipLaunchKernelGGL((k_indirect<ND, PM, METHOD, COLS>), grid, dim3(64), 0, st, a);
  return hipGetLastError();
}

// `pm` = bit mask of the control-law classes present in the batch.  One launch per class; with more than one class
// each launch filters its own trajectories (class_filter), so no kernel ever branches on p.

template <int ND, int METHOD>
static hipError_t launch_dense_pm(int pm, const IndirectArgs& a0, const DenseArgs& d, hipStream_t st) {
  dim3 grid((a0.S + 63) / 64);
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  if (pm & (1 << PM_P0)) hipLaunchKernelGGL((k_indirect_dense<ND, PM_P0, METHOD>), grid, dim3(64), 0, st, a, d);
  if (pm & (1 << PM_P1)) hipLaunchKernelGGL((k_indirect_dense<ND, PM_P1, METHOD>), grid, dim3(64), 0, st, a, d);
  if (pm & (1 << PM_P2)) hipLaunchKernelGGL((k_indirect_dense<ND, PM_P2, METHOD>), grid, dim3(64), 0, st, a, d);
  if (pm & (1 << PM_PGEN)) hipLaunchKernelGGL((k_indirect_dense<ND, PM_PGEN, METHOD>), grid, dim3(64), 0, st, a, d);
  return hipGetLastError();
}

template <int ND, int METHOD, int COLS>
static hipError_t launch_pm(int pm, const IndirectArgs& a0, hipStream_t st) {
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_one<ND, PM_P0, METHOD, COLS>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_one<ND, PM_P1, METHOD, COLS>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_one<ND, PM_P2, METHOD, COLS>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_one<ND, PM_PGEN, METHOD, COLS>(a, st);
  return e;
}

}  // namespace lto
