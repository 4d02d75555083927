// rk.hpp -- explicit Runge-Kutta steppers over fully unrolled register arrays (gfx950 device code).
//
// A "system" is any object with   static constexpr int DIM;   void rhs(const double (&y)[DIM], double (&k)[DIM]) const;
// All loops have compile-time bounds and are unrolled so that y / k live in VGPRs (runtime-indexed
// arrays would go to scratch).  Tableaus are constexpr: after unrolling, zero coefficients fold away.
//
// Tableaus restate mathematical constants:
//   RK4          GeneralCode/ode.jl:64-68
//   RKF7(8)      GeneralCode/ode.jl:875-892 (alpha_, beta_, chi_, psi_), used by ode7_8 (:773-953)
//   DOP853       Hairer/Norsett/Wanner 8(5,3) pair (dop853_tableau.h), standing in for Vern8
#pragma once
#include <hip/hip_runtime.h>
#include "dop853_tableau.h"

namespace lto {

// ------------------------------------------------------------------------------------ RK4
// One classical RK4 step, y <- y + h/6 (k1 + 2 k2 + 2 k3 + k4).
template <class Sys>
__device__ __forceinline__ void rk4_step(const Sys& sys, const double h, double (&y)[Sys::DIM]) {
  constexpr int D = Sys::DIM;
  double k[D], yt[D], acc[D];
  const double h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);
  sys.rhs(y, k);
#pragma unroll
  for (int i = 0; i < D; ++i) { acc[i] = __builtin_fma(h6, k[i], y[i]); yt[i] = __builtin_fma(h2, k[i], y[i]); }
  sys.rhs(yt, k);
#pragma unroll
  for (int i = 0; i < D; ++i) { acc[i] = __builtin_fma(h3, k[i], acc[i]); yt[i] = __builtin_fma(h2, k[i], y[i]); }
  sys.rhs(yt, k);
#pragma unroll
  for (int i = 0; i < D; ++i) { acc[i] = __builtin_fma(h3, k[i], acc[i]); yt[i] = __builtin_fma(h, k[i], y[i]); }
  sys.rhs(yt, k);
#pragma unroll
  for (int i = 0; i < D; ++i) y[i] = __builtin_fma(h6, k[i], acc[i]);
}

// ------------------------------------------------------------------------------------ RKF7(8)
struct TabRKF78 {
  static constexpr int NS = 13;
  // A[s][k]: weight of slope k in the argument of slope s  (= beta_[k+1, s] of ode.jl:877-889)
  static constexpr double A[13][13] = {
      {0},
      {2. / 27},
      {1. / 36, 1. / 12},
      {1. / 24, 0, 1. / 8},
      {5. / 12, 0, -25. / 16, 25. / 16},
      {0.05, 0, 0, 0.25, 0.2},
      {-25. / 108, 0, 0, 125. / 108, -65. / 27, 125. / 54},
      {31. / 300, 0, 0, 0, 61. / 225, -2. / 9, 13. / 900},
      {2, 0, 0, -53. / 6, 704. / 45, -107. / 9, 67. / 90, 3},
      {-91. / 108, 0, 0, 23. / 108, -976. / 135, 311. / 54, -19. / 60, 17. / 6, -1. / 12},
      {2383. / 4100, 0, 0, -341. / 164, 4496. / 1025, -301. / 82, 2133. / 4100, 45. / 82, 45. / 164, 18. / 41},
      {3. / 205, 0, 0, 0, 0, -6. / 41, -3. / 205, -3. / 41, 3. / 41, 6. / 41},
      {-1777. / 4100, 0, 0, -341. / 164, 4496. / 1025, -289. / 82, 2193. / 4100, 51. / 82, 33. / 164, 12. / 41, 0, 1}};
  // chi_: 8th-order weights (ode.jl:891); psi_ * 41/840: error term (ode.jl:892, :940)
  static constexpr double B[13] = {0, 0, 0, 0, 0, 34. / 105, 9. / 35, 9. / 35, 9. / 280, 9. / 280, 0, 41. / 840, 41. / 840};
  static constexpr double E[13] = {41. / 840, 0, 0, 0, 0, 0, 0, 0, 0, 0, 41. / 840, -41. / 840, -41. / 840};
};

// One RKF7(8) step.  ynew = y + h sum_k chi_k f_k (local extrapolation, ode.jl:937);
// returns delta = || h 41/840 sum_k psi_k f_k ||_inf over the first NERR components (ode.jl:940-943).
template <class Sys, int NERR>
__device__ __forceinline__ double rkf78_step(const Sys& sys, const double h, const double (&y)[Sys::DIM],
                                             double (&ynew)[Sys::DIM]) {
  constexpr int D = Sys::DIM;
  using T = TabRKF78;
  double K[T::NS][D];
  sys.rhs(y, K[0]);
#pragma unroll
  for (int s = 1; s < T::NS; ++s) {
    double yt[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < s; ++k)
        if (T::A[s][k] != 0.0) a = __builtin_fma(T::A[s][k], K[k][i], a);
      yt[i] = __builtin_fma(h, a, y[i]);
    }
    sys.rhs(yt, K[s]);
  }
  double delta = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < T::NS; ++k)
      if (T::B[k] != 0.0) a = __builtin_fma(T::B[k], K[k][i], a);
    ynew[i] = __builtin_fma(h, a, y[i]);
    if (i < NERR) {
      const double g = (K[0][i] + K[10][i] - K[11][i] - K[12][i]) * (h * (41.0 / 840.0));
      delta = fmax(delta, fabs(g));
    }
  }
  return delta;
}

// ------------------------------------------------------------------------------------ DOP853
// Stages 1..11 from K[0] = f(y); ynew; K[12] = f(ynew) (FSAL); returns the scaled error norm
// err = |h| e5^2 / sqrt((e5^2 + 0.01 e3^2) n) on scale_i = atol + rtol max(|y_i|, |ynew_i|),
// over the first NERR components.
template <class Sys, int NERR>
__device__ __forceinline__ double dop853_try(const Sys& sys, const double h, const double rtol, const double atol,
                                             const double (&y)[Sys::DIM], double (&K)[13][Sys::DIM],
                                             double (&ynew)[Sys::DIM]) {
  constexpr int D = Sys::DIM;
  constexpr int NS = DP8_NSTAGES;
#pragma unroll
  for (int s = 1; s < NS; ++s) {
    double yt[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < s; ++k)
        if (DP8_A[s][k] != 0.0) a = __builtin_fma(DP8_A[s][k], K[k][i], a);
      yt[i] = __builtin_fma(h, a, y[i]);
    }
    sys.rhs(yt, K[s]);
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double a = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (DP8_B[k] != 0.0) a = __builtin_fma(DP8_B[k], K[k][i], a);
    ynew[i] = __builtin_fma(h, a, y[i]);
  }
  sys.rhs(ynew, K[NS]);
  double e5 = 0.0, e3 = 0.0;
#pragma unroll
  for (int i = 0; i < NERR; ++i) {
    double a5 = 0.0, a3 = 0.0;
#pragma unroll
    for (int k = 0; k <= NS; ++k) {
      if (DP8_E5[k] != 0.0) a5 = __builtin_fma(DP8_E5[k], K[k][i], a5);
      if (DP8_E3[k] != 0.0) a3 = __builtin_fma(DP8_E3[k], K[k][i], a3);
    }
    const double inv_sc = 1.0 / __builtin_fma(rtol, fmax(fabs(y[i]), fabs(ynew[i])), atol);
    a5 *= inv_sc; a3 *= inv_sc;
    e5 = __builtin_fma(a5, a5, e5);
    e3 = __builtin_fma(a3, a3, e3);
  }
  if (e5 == 0.0 && e3 == 0.0) return 0.0;
  return fabs(h) * e5 / sqrt((e5 + 0.01 * e3) * (double)NERR);
}

}  // namespace lto
