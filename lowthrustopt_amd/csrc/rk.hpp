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
#include <type_traits>
#include "dop853_tableau.h"
#include "dynamics.hpp"  // rcp_nr

namespace lto {

// A tableau coefficient as a scalar-register operand materialised where it is used.  Left to itself the compiler hoists the
// 64-bit literals of an unrolled 13-stage loop out of it, runs out of scalar registers and spills them to VGPR lanes
// (v_readlane / v_writelane inside the loop); an asm statement cannot be hoisted.
__device__ __forceinline__ double coef_here(double c) {
  asm volatile("" : "+s"(c));
  return c;
}

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

// Error term of one component in the reference's operation order (ode.jl:940, `gamma1 = hi*41/840*f*psi_`): the slopes are scaled
// by (hi*41)/840 first, then summed over k in order against psi_ = (1, 0 ... 0, 1, -1, -1) -- every product rounded before it is added
// (no contraction).  The term is a difference of O(1) slopes at the 1e-16 level (errors ~ 1e-17 at the demo's step size), so the
// order of these seven operations is what its low bits are made of.
__device__ __forceinline__ double rkf78_err_term(const double h, const double k0, const double k10, const double k11, const double k12) {
  const double c = __ddiv_rn(__dmul_rn(h, 41.0), 840.0);
  double g = __dmul_rn(c, k0);
  g = __dadd_rn(g, __dmul_rn(c, k10));
  g = __dadd_rn(g, -__dmul_rn(c, k11));
  g = __dadd_rn(g, -__dmul_rn(c, k12));
  return g;
}

// One RKF7(8) step.  ynew = y + h sum_k chi_k f_k (local extrapolation, ode.jl:937);
// returns delta = || h 41/840 sum_k psi_k f_k ||_inf over the first NERR components (ode.jl:940-943).
// NANPROP (adaptive callers): a NaN component makes delta NaN, as the reference's maximum() does; fmax alone drops it.
template <class Sys, int NERR, bool NANPROP = false>
__device__ __forceinline__ double rkf78_step(const Sys& sys, const double h, const double (&y)[Sys::DIM],
                                             double (&ynew)[Sys::DIM]) {
  constexpr int D = Sys::DIM;
  // 12 and more components: arguments formed slope by slope with the coefficient materialised in place (coef_here) -- the
  // literals are otherwise hoisted and spilled to VGPR lanes (indirect RKF7(8) x 4 defect sweep: 28.7 -> 24.5 us).  The
  // 6/7-component systems of the direct path have the scalar registers to spare and are faster component by component
  // (direct defect sweep: 18.1 against 22.2 us).
  constexpr bool BY_SLOPE = (D >= 12);
  using T = TabRKF78;
  double K[T::NS][D];
  sys.rhs(y, K[0]);
#pragma unroll
  for (int s = 1; s < T::NS; ++s) {
    double yt[D], a[D];
#pragma unroll
    for (int i = 0; i < D; ++i) a[i] = 0.0;
    if constexpr (BY_SLOPE) {
#pragma unroll
      for (int k = 0; k < s; ++k)
        if (T::A[s][k] != 0.0) {
          const double w = coef_here(T::A[s][k]);
#pragma unroll
          for (int i = 0; i < D; ++i) a[i] = __builtin_fma(w, K[k][i], a[i]);
        }
    } else {
#pragma unroll
      for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int k = 0; k < s; ++k)
          if (T::A[s][k] != 0.0) a[i] = __builtin_fma(T::A[s][k], K[k][i], a[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) yt[i] = __builtin_fma(h, a[i], y[i]);
    sys.rhs(yt, K[s]);
  }
  double delta = 0.0, gsum = 0.0;              // fmax drops NaNs; gsum keeps them (the reference's maximum() propagates)
  double a[D];
#pragma unroll
  for (int i = 0; i < D; ++i) a[i] = 0.0;
  if constexpr (BY_SLOPE) {
#pragma unroll
    for (int k = 0; k < T::NS; ++k)
      if (T::B[k] != 0.0) {
        const double w = coef_here(T::B[k]);
#pragma unroll
        for (int i = 0; i < D; ++i) a[i] = __builtin_fma(w, K[k][i], a[i]);
      }
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
      for (int k = 0; k < T::NS; ++k)
        if (T::B[k] != 0.0) a[i] = __builtin_fma(T::B[k], K[k][i], a[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    ynew[i] = __builtin_fma(h, a[i], y[i]);
    if (i < NERR) {
      const double g = rkf78_err_term(h, K[0][i], K[10][i], K[11][i], K[12][i]);
      delta = fmax(delta, fabs(g));
      if (NANPROP) gsum += g;
    }
  }
  return (NANPROP && gsum != gsum) ? gsum : delta;
}

// ------------------------------------------------------------------------------------ DOP853
// Stages 1..11 from K[0] = f(y); ynew; K[12] = f(ynew) (FSAL); returns the scaled error norm
// err = |h| e5^2 / sqrt((e5^2 + 0.01 e3^2) n) on scale_i = atol + rtol max(|y_i|, |ynew_i|),
// over the first NERR components.
// E5 / E3: the two error sums themselves (what dp8_decide takes).
template <class Sys, int NERR>
__device__ __forceinline__ double dop853_try(const Sys& sys, const double h, const double rtol, const double atol,
                                             const double (&y)[Sys::DIM], double (&K)[13][Sys::DIM],
                                             double (&ynew)[Sys::DIM], double& E5, double& E3) {
  constexpr int D = Sys::DIM;
  constexpr int NS = DP8_NSTAGES;
#pragma unroll
  for (int s = 1; s < NS; ++s) {
    double yt[D], a[D];
#pragma unroll
    for (int i = 0; i < D; ++i) a[i] = 0.0;
#pragma unroll
    for (int k = 0; k < s; ++k)
      if (DP8_A[s][k] != 0.0) {
        const double w = coef_here(DP8_A[s][k]);
#pragma unroll
        for (int i = 0; i < D; ++i) a[i] = __builtin_fma(w, K[k][i], a[i]);
      }
#pragma unroll
    for (int i = 0; i < D; ++i) yt[i] = __builtin_fma(h, a[i], y[i]);
    sys.rhs(yt, K[s]);
  }
  {
    double a[D];
#pragma unroll
    for (int i = 0; i < D; ++i) a[i] = 0.0;
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (DP8_B[k] != 0.0) {
        const double w = coef_here(DP8_B[k]);
#pragma unroll
        for (int i = 0; i < D; ++i) a[i] = __builtin_fma(w, K[k][i], a[i]);
      }
#pragma unroll
    for (int i = 0; i < D; ++i) ynew[i] = __builtin_fma(h, a[i], y[i]);
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
    const double inv_sc = rcp_nr(__builtin_fma(rtol, fmax(fabs(y[i]), fabs(ynew[i])), atol));
    a5 *= inv_sc; a3 *= inv_sc;
    e5 = __builtin_fma(a5, a5, e5);
    e3 = __builtin_fma(a3, a3, e3);
  }
  E5 = e5; E3 = e3;
  if (e5 == 0.0 && e3 == 0.0) return 0.0;
  return fabs(h) * e5 / sqrt((e5 + 0.01 * e3) * (double)NERR);
}


// Step decision of the adaptive 8(5,3) controller from the two error sums (E5 = sum (e5_i / scale_i)^2, E3 likewise, over n
// components) -- Hairer's rule as in dop853_try's callers: err = |h| E5 / sqrt((E5 + 0.01 E3) n); accept if err < 1 with
// factor min(10, 0.9 err^(-1/8)) (at most 1 straight after a rejection), else factor max(0.2, 0.9 err^(-1/8)).  The roots are
// reciprocal-square-root chains (v_rsq_f64 + one third-order step each, ~1 ulp) instead of IEEE sqrt / divide sequences: the
// decision sits between a trial step's error sums and the next trial step of EVERY wavefront of a cooperative workgroup.
// Contraction is off so that every role (different template instantiations of the caller) forms the same bits from the same sums.
__device__ __forceinline__ void dp8_decide(const double E5, const double E3, const double h, const double rejected, const double ncomp,
                                           double& h_abs, double& accept, double& bad) {
#pragma clang fp contract(off)
  // selects only, no branch: the caller's scheduler interleaves this chain with the FSAL slopes of the same basic block
  const double den = (E5 + 0.01 * E3) * ncomp;
  const bool zero = (E5 == 0.0) & (E3 == 0.0);
  const double err_raw = (fabs(h) * E5) * rsqrt_nr(den);
  const double err = zero ? 0.0 : err_raw;
  const double r8 = rsqrt_nr(rsqrt_nr(rsqrt_nr(err)));          // err^(-1/8); NaN for err = 0 and for err = inf
  // The two non-finite r8 are selected away explicitly -- err = 0 (which also covers E5 = 0 with E3 != 0) grows by 10, err = inf
  // shrinks by 0.2 -- so the result does not rest on fmin / fmax returning their non-NaN operand (it would not survive
  // -ffinite-math-only on this unit; advisor finding, round 4).  A NaN err stays NaN in f and reaches `bad` below.
  const double inf = __builtin_huge_val();
  const double f = (err == 0.0) ? 10.0 : (err == inf ? 0.2 : 0.9 * r8);
  const bool acc = err < 1.0;
  double fa = fmin(10.0, f);
  fa = (rejected != 0.0) ? fmin(1.0, fa) : fa;
  const double fr = fmax(0.2, f);
  h_abs = h * (acc ? fa : fr);
  accept = acc ? 1.0 : 0.0;
  bad = (err != err) ? err : 0.0;                                // a NaN never recovers: the caller poisons the segment
}

// The DOP853 tableau for code that reads its coefficients with scalar loads where they are needed (the lone-wavefront kernels of the
// reference's integrator setting: as literals every coefficient costs two s_mov_b32 in an instruction stream that IS the sweep's
// duration -- 152 of ~2 000 instructions per trial step).  Packed: the non-zero weights of argument 1, 2, .. 11 (rows of A), of
// the new state (B) and of the two error estimators, each row contiguous and starting at an even index, so that one or two
// s_load_dwordx4..x16 fetch a row.
struct Dp8Packed {
  static constexpr int ROWS = 14;           // 0..10: A rows 1..11; 11: B; 12: E5; 13: E3
  double w[96];
};
constexpr double dp8_row_entry(int row, int k) {
  return row < 11 ? DP8_A[row + 1][k < 12 ? k : 0] * (k < 12 ? 1.0 : 0.0) : row == 11 ? (k < 12 ? DP8_B[k] : 0.0) : row == 12 ? DP8_E5[k] : DP8_E3[k];
}
constexpr int dp8_row_len(int row) { return row < 11 ? row + 1 : row == 11 ? 12 : 13; }      // entries that can be non-zero
constexpr int dp8_row_nnz(int row) {
  int n = 0;
  for (int k = 0; k < dp8_row_len(row); ++k) n += (dp8_row_entry(row, k) != 0.0) ? 1 : 0;
  return n;
}
constexpr int dp8_row_off(int row) {
  int off = 0;
  for (int r = 0; r < row; ++r) off += (dp8_row_nnz(r) + 1) & ~1;
  return off;
}
constexpr int dp8_slot(int row, int k) {     // position of entry k inside its packed row
  int n = 0;
  for (int i = 0; i < k; ++i) n += (dp8_row_entry(row, i) != 0.0) ? 1 : 0;
  return n;
}
constexpr Dp8Packed dp8_pack() {
  Dp8Packed P{};
  for (int r = 0; r < Dp8Packed::ROWS; ++r)
    for (int k = 0; k < dp8_row_len(r); ++k)
      if (dp8_row_entry(r, k) != 0.0) P.w[dp8_row_off(r) + dp8_slot(r, k)] = dp8_row_entry(r, k);
  return P;
}
static_assert(dp8_row_off(Dp8Packed::ROWS) <= 96, "packed DOP853 tableau");
static __device__ __constant__ Dp8Packed kDp8Packed = dp8_pack();

// The empty asm makes the address opaque, so the loads stay where the caller puts them -- one stage ahead of their use -- instead
// of being hoisted to the top of the kernel (74 doubles do not fit the scalar register file); constant address space: scalar loads.
typedef const Dp8Packed __attribute__((address_space(4))) * Dp8ConstPtr;
// dp8_tab_base(): once per kernel (an opaque value in a scalar register pair: the pc-relative address costs three instructions);
// dp8_tab_here(base): the pointer to load through, at the point of the call.
__device__ __forceinline__ unsigned long dp8_tab_base() {
  unsigned long p = (unsigned long)&kDp8Packed;
  asm volatile("" : "+s"(p));
  return p;
}
__device__ __forceinline__ Dp8ConstPtr dp8_tab_here(unsigned long base) {
  asm volatile("" : "+s"(base));
  return (Dp8ConstPtr)base;
}
// f(integral_constant<int, I>) for I = FIRST .. LAST - 1: a stage loop whose index is a constant expression inside the body
template <int FIRST, int LAST, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (FIRST < LAST) {
    f(std::integral_constant<int, FIRST>{});
    static_for<FIRST + 1, LAST>(f);
  }
}
// the non-zero weights of argument ST (ST = 1..11: row ST of A; ST = 12: B) into w[k], k < ST
template <int ST>
__device__ __forceinline__ void dp8_load_row(const unsigned long base, double (&w)[12]) {
  const Dp8ConstPtr T = dp8_tab_here(base);
  static_for<0, ST>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr bool used = dp8_row_entry(ST - 1, k) != 0.0;
    constexpr int at = dp8_row_off(ST - 1) + dp8_slot(ST - 1, k);
    if constexpr (used) w[k] = T->w[at];
  });
}
// The scalar loads share their counter with the LDS operations (lgkmcnt), and scalar loads return out of order: the first use of
// a loaded coefficient makes the compiler wait for EVERYTHING outstanding.  A kernel that wants its LDS traffic to overlap with
// arithmetic on these coefficients consumes them here, before it issues that traffic (an empty asm per entry: no instruction).
template <int ST>
__device__ __forceinline__ void dp8_pin_row(const double (&w)[12]) {
#pragma unroll
  for (int k = 0; k < ST; ++k) {
    if (dp8_row_entry(ST - 1, k) != 0.0) asm volatile("" :: "s"(w[k]));
  }
}
__device__ __forceinline__ void dp8_pin_err(const double (&e5)[13], const double (&e3)[13]) {
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    if (DP8_E5[k] != 0.0) asm volatile("" :: "s"(e5[k]));
    if (DP8_E3[k] != 0.0) asm volatile("" :: "s"(e3[k]));
  }
}
// the non-zero weights of the two error estimators
__device__ __forceinline__ void dp8_load_err(const unsigned long base, double (&e5)[13], double (&e3)[13]) {
  const Dp8ConstPtr T = dp8_tab_here(base);
  static_for<0, 13>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    constexpr int at5 = dp8_row_off(12) + dp8_slot(12, k), at3 = dp8_row_off(13) + dp8_slot(13, k);
    if constexpr (DP8_E5[k] != 0.0) e5[k] = T->w[at5];
    if constexpr (DP8_E3[k] != 0.0) e3[k] = T->w[at3];
  });
}

}  // namespace lto
