// kernels_indirect_coop.hip -- wave-specialised STM kernel for small/medium batches (latency regime).
//
// The per-lane STM kernels (indirect_kernel.hpp) make every column lane re-integrate the base trajectory and
// rebuild the variational coefficients (G, H, U: ~150 of ~230 fp64 instructions per RK stage), and for the
// 13-stage methods their 24-28 component state no longer fits the register file.  Here one workgroup owns 16
// segments and splits the roles across wavefronts:
//
//   base role     lane = segment.  Integrates the ND-dim base state, and at every RK stage publishes the
//                 variational coefficients of that stage (17 doubles for ND = 12, 25 for ND = 14) to LDS.
//   column waves  lane = (segment, STM column): 3 waves x (16 segments x 4 columns) for ND = 12.  Integrates ONLY
//                 its column  c' = F(t) c  with the coefficients read from LDS: no gravity, no control law, no
//                 redundant base work, and a 12/14-component state that stays in registers even with 13 slopes.
//
// One __syncthreads() per RK stage hands the coefficients over (double-buffered in LDS, so the base wave computes
// stage s+1 while the column waves consume stage s).  Adaptive methods take one common step sequence per
// segment: every lane contributes a partial error sum through LDS, the base lane decides (accept / next h / done)
// and broadcasts the decision through LDS.  The DOP853 error norm runs over the base state AND all ND columns
// (ND + ND^2 components) -- exactly the norm ForwardDiff duals see inside an adaptive solver
// (src/multiShoot_CRTBP_indirect.jl:107-110,121), so the step sequence is the one the oracle's dual-number run takes.
//
// Every loop is bounded (fixed step count, or max_steps trial steps), every wavefront executes the same number of
// barriers, and out-of-range lanes shadow a valid segment/column without storing: the grid always drains.
#include "kernels.hpp"
#include "rk.hpp"

namespace lto {

constexpr int COOP_SEG = 16;  // segments per workgroup

struct TabRK4c {
  static constexpr double A[4][4] = {{0}, {0.5}, {0, 0.5}, {0, 0, 1.0}};
  static constexpr double B[4] = {1. / 6, 1. / 3, 1. / 3, 1. / 6};
};

template <int ND> struct CoefOf { using type = VarCoef12; };
template <> struct CoefOf<14> { using type = VarCoef14; };

// Butcher coefficient access, uniform for the three tableaus
template <int METHOD> __device__ __forceinline__ constexpr double tabA(int s, int k) {
  return METHOD == M_RK4 ? TabRK4c::A[s][k] : (METHOD == M_DOP853_ADAPTIVE ? DP8_A[s][k] : TabRKF78::A[s][k]);
}
template <int METHOD> __device__ __forceinline__ constexpr double tabB(int k) {
  return METHOD == M_RK4 ? TabRK4c::B[k] : (METHOD == M_DOP853_ADAPTIVE ? DP8_B[k] : TabRKF78::B[k]);
}
template <int METHOD> struct TabN { static constexpr int NS = METHOD == M_RK4 ? 4 : (METHOD == M_DOP853_ADAPTIVE ? 12 : 13); };

struct CoopCtrl {   // ode78 decision of the base lane, broadcast through LDS
  double h;        // step the NEXT trial uses
  int accept;      // last trial accepted
};

template <int ND, int PM, int METHOD>
__global__ __launch_bounds__(256) void k_indirect_coop(const IndirectArgs a) {
  using Coef = typename CoefOf<ND>::type;
  constexpr int NC = sizeof(Coef) / sizeof(double);
  constexpr int NS = TabN<METHOD>::NS;
  constexpr bool ADAPT = (METHOD == M_RKF78_ADAPTIVE || METHOD == M_DOP853_ADAPTIVE);
  constexpr bool DOP = (METHOD == M_DOP853_ADAPTIVE);
  constexpr int NSL = DOP ? 13 : NS;         // slopes kept (DOP853: + FSAL slope)

  // ND = 12 (the reference's system): the base lane is the long pole of every RK stage when it also builds the coefficients
  // (~380 instructions against ~195 in a column lane), and the other way round when every column lane rebuilds them from
  // the bare argument (~245 / ~390: measured 0.308 -> 0.297 ms, DOP853 @ 1e-13, 4 096 segments).  So the work is split
  // where it balances: the base lane evaluates the lean RHS and publishes its argument (r, lambda_v) plus the by-products
  // the coefficients need (c_b, 1/d_b, ua, ub, 1/n: 13 doubles, rhs12_base_parts); the column lanes assemble G, H, U
  // from them without a reciprocal square root, an exponential or a division (coef12_from_parts): ~280 / ~245
  // instructions per stage.  (ND = 14 keeps the coefficient hand-over: there wave 3 runs base and column lanes one
  // after the other.)
  constexpr bool LEAN = (ND == 12);
  constexpr int NPUB = LEAN ? 13 : NC;
  __shared__ double s_coef[2][NPUB][COOP_SEG];
  __shared__ double s_part[3][ND + 1][COOP_SEG];   // partial norms: [which][role][segment]
  __shared__ double s_scale[ND][COOP_SEG];          // 1 / (atol + rtol |base value|): the error scale of row r
  __shared__ CoopCtrl s_ctrl[COOP_SEG];

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int seg = lane & (COOP_SEG - 1), slot = lane >> 4;
  // 16 roles per segment: rho < ND -> STM column rho; rho == ND -> base state; rho > ND -> shadow of the base
  // (repeats its work, stores nothing).  ND = 12: wave 3 is the base wave.  ND = 14: wave 3 holds columns 12, 13
  // and the base role in different lanes, so that wave runs both code paths (predicated) -- still 4 waves, one
  // per SIMD, which keeps the full 512-register budget.
  const int rho = wave * 4 + slot;
  const bool is_base = rho >= ND;
  const int col = is_base ? 0 : rho;
  const int role = is_base ? ND : col;                         // row of s_part
  const int s_raw = blockIdx.x * COOP_SEG + seg;
  const int s_lin = s_raw < a.S ? s_raw : a.S - 1;             // shadow lanes repeat the last segment
  const int s = a.order ? a.order[s_lin] : s_lin;              // balanced order (lto_indirect_plan_rebalance)
  const bool in_range = (s_raw < a.S) && (rho <= ND);

  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const double span = a.t[tg + 1] - a.t[tg];
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  const double w2 = 2.0 * tp.omega;
  // mixed-class batch: segments of another control-law class are owned by that class's launch; here they idle
  // through the barriers (never stored, never hold the adaptive loop open)
  const bool mine = !a.class_filter || p_class(tp.p) == PM;
  if (!__syncthreads_or(mine)) return;         // workgroup-uniform

  // state of this lane: base state (base wave) or one STM column (column waves)
  double y[ND], K[NSL][ND];
  if (is_base) {
#pragma unroll
    for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + node];
  } else {
#pragma unroll
    for (int r = 0; r < ND; ++r) y[r] = (r == col) ? 1.0 : 0.0;
  }

  // slope of the argument `arg` into `out`; base wave publishes the coefficients into buffer `buf`, column
  // waves consume them after the barrier.  Called by ALL threads (one barrier inside).
  auto slope = [&](const double (&arg)[ND], double (&out)[ND], int buf) {
    if (is_base) {
      if constexpr (LEAN) {
        BaseParts12 bp;
        rhs12_base_parts<PM>(arg, tp, out, bp);
        s_coef[buf][0][seg] = arg[0]; s_coef[buf][1][seg] = arg[1]; s_coef[buf][2][seg] = arg[2];
        s_coef[buf][3][seg] = arg[9]; s_coef[buf][4][seg] = arg[10]; s_coef[buf][5][seg] = arg[11];
        s_coef[buf][6][seg] = bp.c1; s_coef[buf][7][seg] = bp.c2; s_coef[buf][8][seg] = bp.i1s; s_coef[buf][9][seg] = bp.i2s;
        s_coef[buf][10][seg] = bp.ua; s_coef[buf][11][seg] = bp.ub; s_coef[buf][12][seg] = bp.inv_n;
      } else {
        Coef vc;
        if constexpr (ND == 12) rhs12<PM, true>(arg, tp, out, vc);
        else rhs14<PM, true>(arg, tp, out, vc);
        const double* v = reinterpret_cast<const double*>(&vc);
#pragma unroll
        for (int e = 0; e < NC; ++e) s_coef[buf][e][seg] = v[e];
      }
    }
    __syncthreads();
    if (!is_base) {
      Coef vc;
      if constexpr (LEAN) {
        BaseParts12 bp;
        bp.c1 = s_coef[buf][6][seg]; bp.c2 = s_coef[buf][7][seg]; bp.i1s = s_coef[buf][8][seg]; bp.i2s = s_coef[buf][9][seg];
        bp.ua = s_coef[buf][10][seg]; bp.ub = s_coef[buf][11][seg]; bp.inv_n = s_coef[buf][12][seg];
        coef12_from_parts(s_coef[buf][0][seg], s_coef[buf][1][seg], s_coef[buf][2][seg], s_coef[buf][3][seg], s_coef[buf][4][seg],
                          s_coef[buf][5][seg], bp, tp.MU, vc);
      } else {
        double* v = reinterpret_cast<double*>(&vc);
#pragma unroll
        for (int e = 0; e < NC; ++e) v[e] = s_coef[buf][e][seg];
      }
      if constexpr (ND == 12) var_col12(vc, w2, arg, out);
      else var_col14(vc, w2, arg, out);
    }
  };
  // sum over all roles of partial `which` for this lane's segment (fixed order => identical in every lane)
  auto total = [&](int which) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r <= ND; ++r) t += s_part[which][r][seg];
    return t;
  };

  double maxErr = 0.0;
  int nacc = 0, nrej = 0;
  int buf = 0;

  if constexpr (!ADAPT) {
    const double h = span / (double)a.steps;
    for (int step = 0; step < a.steps; ++step) {
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        double arg[ND], acc[ND];
#pragma unroll
        for (int c = 0; c < ND; ++c) acc[c] = 0.0;
#pragma unroll
        for (int k = 0; k < st; ++k)
          if (tabA<METHOD>(st, k) != 0.0) {
            const double w = coef_here(tabA<METHOD>(st, k));
#pragma unroll
            for (int c = 0; c < ND; ++c) acc[c] = __builtin_fma(w, K[k][c], acc[c]);
          }
#pragma unroll
        for (int c = 0; c < ND; ++c) arg[c] = (st == 0) ? y[c] : __builtin_fma(h, acc[c], y[c]);
        slope(arg, K[st], buf);
        buf ^= 1;
      }
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NS; ++k)
          if (tabB<METHOD>(k) != 0.0) acc = __builtin_fma(tabB<METHOD>(k), K[k][c], acc);
        y[c] = __builtin_fma(h, acc, y[c]);
      }
      if constexpr (METHOD == M_RKF78_FIXED) if (is_base) {
#pragma unroll
        for (int c = 0; c < ND; ++c)
          maxErr = fmax(maxErr, fabs(rkf78_err_term(h, K[0][c], K[10][c], K[11][c], K[12][c])));
      }
    }
    nacc = a.steps;
  } else {
    // ---------------------------------------------------------------- adaptive: common step sequence per segment
    const double rtol = a.rtol, atol = a.atol;
    double h_abs = 0.0, t = 0.0;
    double rejected = 0.0;
    int done = !(span > 0.0) || !mine;
    if (DOP) {
      // Hairer initial step over all ND + ND^2 components.  Row r of every column is scaled with the BASE value
      // of row r (a dual number's partials share the scale of its value), published by the base lane.
      if (is_base) {
#pragma unroll
        for (int c = 0; c < ND; ++c) s_scale[c][seg] = rcp_nr(__builtin_fma(rtol, fabs(y[c]), atol));
      }
      slope(y, K[0], buf); buf ^= 1;
      double isc0[ND];
      double p0 = 0.0, p1 = 0.0;
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        const double isc = s_scale[c][seg];
        isc0[c] = isc;
        p0 = __builtin_fma(y[c] * isc, y[c] * isc, p0);
        p1 = __builtin_fma(K[0][c] * isc, K[0][c] * isc, p1);
      }
      s_part[0][role][seg] = p0; s_part[1][role][seg] = p1;
      __syncthreads();
      constexpr double NCOMP = (double)(ND * (ND + 1));
      const double d0 = sqrt(total(0) / NCOMP), d1 = sqrt(total(1) / NCOMP);
      const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
      double arg[ND];
#pragma unroll
      for (int c = 0; c < ND; ++c) arg[c] = __builtin_fma(h0, K[0][c], y[c]);
      slope(arg, K[1], buf); buf ^= 1;       // barrier inside also separates the reads above from the writes below
      double p2 = 0.0;
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        const double df = (K[1][c] - K[0][c]) * isc0[c];
        p2 = __builtin_fma(df, df, p2);
      }
      s_part[2][role][seg] = p2;
      __syncthreads();
      const double d2 = sqrt(total(2) / NCOMP) / h0;
      const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : cbrt(cbrt(0.01 / fmax(d1, d2)));
      h_abs = fmin(fmin(100.0 * h0, h1), span);
    } else {
      h_abs = span / 50.0;                    // ode78: h = (tfinal - t)/50   (ode.jl:471)
    }
    const double hmax = span / 2.5, hmin = span / 1e7;   // ode78 only (ode.jl:464,470)

    for (int trial = 0; trial < a.max_steps; ++trial) {
      // every lane of a segment holds identical (t, h_abs, done): they are updated from identical data below
      double h = h_abs;
      double last = 0.0;
      if (DOP) { if (t + h >= span) { h = span - t; last = 1.0; } }
      else { if (t + h > span) h = span - t; }
      const int st0 = DOP ? 1 : 0;            // DOP853 enters with K[0] = f(y) (FSAL)
#pragma unroll
      for (int st = st0; st < NS; ++st) {
        double arg[ND], acc[ND];
#pragma unroll
        for (int c = 0; c < ND; ++c) acc[c] = 0.0;
#pragma unroll
        for (int k = 0; k < st; ++k)
          if (tabA<METHOD>(st, k) != 0.0) {
            const double w = coef_here(tabA<METHOD>(st, k));
#pragma unroll
            for (int c = 0; c < ND; ++c) acc[c] = __builtin_fma(w, K[k][c], acc[c]);
          }
#pragma unroll
        for (int c = 0; c < ND; ++c) arg[c] = (st == 0) ? y[c] : __builtin_fma(h, acc[c], y[c]);
        slope(arg, K[st], buf);
        buf ^= 1;
      }
      double yn[ND];
#pragma unroll
      for (int c = 0; c < ND; ++c) {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < NS; ++k)
          if (tabB<METHOD>(k) != 0.0) acc = __builtin_fma(tabB<METHOD>(k), K[k][c], acc);
        yn[c] = __builtin_fma(h, acc, y[c]);
      }
      double accept, bad = 0.0;
      if (DOP) {
        if (is_base) {
#pragma unroll
          for (int c = 0; c < ND; ++c) s_scale[c][seg] = rcp_nr(__builtin_fma(rtol, fmax(fabs(y[c]), fabs(yn[c])), atol));
        }
        slope(yn, K[12], buf); buf ^= 1;
        double e5 = 0.0, e3 = 0.0;
#pragma unroll
        for (int c = 0; c < ND; ++c) {
          double a5 = 0.0, a3 = 0.0;
#pragma unroll
          for (int k = 0; k <= 12; ++k) {
            if (DP8_E5[k] != 0.0) a5 = __builtin_fma(DP8_E5[k], K[k][c], a5);
            if (DP8_E3[k] != 0.0) a3 = __builtin_fma(DP8_E3[k], K[k][c], a3);
          }
          const double isc = s_scale[c][seg];
          a5 *= isc; a3 *= isc;
          e5 = __builtin_fma(a5, a5, e5);
          e3 = __builtin_fma(a3, a3, e3);
        }
        s_part[0][role][seg] = e5; s_part[1][role][seg] = e3;
        __syncthreads();
        const double E5 = total(0), E3 = total(1);
        constexpr double NCOMP = (double)(ND * (ND + 1));
        // identical arithmetic in every lane of the segment => identical decision, no broadcast needed (rk.hpp dp8_decide)
        dp8_decide(E5, E3, h, rejected, NCOMP, h_abs, accept, bad);
        __syncthreads();                       // s_part is rewritten by the next trial
      } else {
        // ode78: error and |x|_inf over the BASE state only (ode.jl:492-497): the base lane decides, LDS broadcasts
        if (is_base) {
          double delta = 0.0, nx = 0.0, gsum = 0.0;
#pragma unroll
          for (int c = 0; c < ND; ++c) {
            const double g = rkf78_err_term(h, K[0][c], K[10][c], K[11][c], K[12][c]);
            delta = fmax(delta, fabs(g));
            nx = fmax(nx, fabs(y[c]));
            gsum += g + y[c];
          }
          if (gsum != gsum) delta = gsum;     // fmax drops NaNs; the reference's maximum() propagates them
          const double tau = rtol * fmax(nx, 1.0);
          const int acc_i = delta <= tau;
          if (delta == 0.0) delta = 1e-16;
          const double hn = 0.8 * h * sqrt(sqrt(sqrt(tau / delta)));
          s_ctrl[seg].h = (hn != hn) ? hn : fmin(hmax, hn);   // keep a NaN visible (fmin would drop it)
          s_ctrl[seg].accept = acc_i;
        }
        __syncthreads();
        h_abs = s_ctrl[seg].h;
        accept = s_ctrl[seg].accept ? 1.0 : 0.0;
        if (h_abs != h_abs) bad = h_abs;
        __syncthreads();
      }
      if (!done) {
        if (bad != 0.0) {                      // NaN in the step: NaN results (status_flag 2 upstream), no max_steps stall
#pragma unroll
          for (int c = 0; c < ND; ++c) y[c] = bad;
          t = span;
        } else if (accept != 0.0) {
          t = (DOP && last != 0.0) ? span : t + h;
#pragma unroll
          for (int c = 0; c < ND; ++c) y[c] = yn[c];
          if (DOP) {
#pragma unroll
            for (int c = 0; c < ND; ++c) K[0][c] = K[12][c];
          }
          ++nacc;
          rejected = 0.0;
        } else {
          ++nrej;
          rejected = 1.0;
        }
        if (!(t < span)) done = 1;
        if (!DOP && !(h_abs >= hmin)) done = 1;      // ode78's "singularity" exit (ode.jl:479,524)
      }
      // workgroup-uniform exit: all 16 segments done
      if (!__syncthreads_or(!done)) break;
    }
    // A segment that did not reach t1 -- max_steps trial steps used up, ode78's step-size floor, or a decreasing time
    // grid (span < 0: the adaptive controllers here integrate forward only) -- has no result: NaN, which the driver
    // reports as status_flag 2 (indirect.jl:339-341), instead of a state at some t < t1 that looks propagated.
    if (mine && (t < span || !(span >= 0.0))) {   // unfinished, decreasing grid, or a NaN span (treated like a negative one)
#pragma unroll
      for (int c = 0; c < ND; ++c) y[c] = __builtin_nan("");
    }
  }

  const bool writer = in_range && mine;
  if (writer) {
    if (is_base) {
      if (a.defect) {
#pragma unroll
        for (int c = 0; c < ND; ++c) a.defect[c * a.ldd + s] = y[c] - a.X[c * a.ldx + node + 1];
      }
      if (a.errors) a.errors[s] = maxErr;
      if (a.nacc) a.nacc[s] = nacc;
      if (a.nrej) a.nrej[s] = nrej;
    } else {
#pragma unroll
      for (int r = 0; r < ND; ++r) a.Phi[(long)(col * ND + r) * a.ldp + s] = y[r];
    }
  }
}

template <int ND, int PM, int METHOD>
static hipError_t launch_coop_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + COOP_SEG - 1) / COOP_SEG);
  hipLaunchKernelGGL((k_indirect_coop<ND, PM, METHOD>), grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

template <int ND, int METHOD>
static hipError_t launch_coop_pm(int pm, const IndirectArgs& a0, hipStream_t st) {
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_coop_one<ND, PM_P0, METHOD>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_coop_one<ND, PM_P1, METHOD>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_coop_one<ND, PM_P2, METHOD>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_coop_one<ND, PM_PGEN, METHOD>(a, st);
  return e;
}

hipError_t launch_indirect_stm_coop(int ndim, int pm, int method, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (ndim == 12) {
    switch (method) {
      // (round 6: RK4 runs the pipelines at every size, 12-dim DOP853 the two-lanes-per-state form -- lto_api.hip never asks for them here)
      case M_RKF78_FIXED: return launch_coop_pm<12, M_RKF78_FIXED>(pm, a, st);
      case M_RKF78_ADAPTIVE: return launch_coop_pm<12, M_RKF78_ADAPTIVE>(pm, a, st);
    }
  } else if (ndim == 14) {
    switch (method) {
      case M_RKF78_FIXED: return launch_coop_pm<14, M_RKF78_FIXED>(pm, a, st);
      case M_RKF78_ADAPTIVE: return launch_coop_pm<14, M_RKF78_ADAPTIVE>(pm, a, st);
      case M_DOP853_ADAPTIVE: return launch_coop_pm<14, M_DOP853_ADAPTIVE>(pm, a, st);
    }
  }
  return hipErrorInvalidValue;
}

}  // namespace lto
