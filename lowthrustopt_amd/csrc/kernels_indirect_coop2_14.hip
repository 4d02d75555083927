// kernels_indirect_coop2_14.hip -- the reference's integrator setting (adaptive order 8, rtol = atol = 1e-13,
// src/multiShoot_CRTBP_indirect.jl:79,107-110,121) with the STM by variational equations on BASELINE configs[1]'s 14-dim system
// (state + mass + costates + mass costate; no reference counterpart: GeneralCode/twoBody_stateCostate_mass_deriv.jl:11-78 in CRTBP
// units, dynamics.hpp rhs14), always-thrust-limited control laws (p = 0, p = 1): the cooperative kernel of kernels_indirect_coop2.hip
// with every state split 7 + 7 over two lanes.  (Other laws, and RKF7(8): the one-piece kernel of kernels_indirect_coop.hip.)
//
// What changes against the 12-dim form.  A column is (a, b, mu | d, g, nu) = (delta r, delta v, delta m | delta lambda_v, delta
// lambda_r, delta lambda_m); for these laws nothing depends on lambda_m, so the lambda_m column of the STM is the unit vector and
// THIRTEEN columns are integrated.  12 of them fill three top and three bottom wavefronts as before (lane = segment x column, four
// columns per wavefront); the thirteenth lives in ONE wavefront whose lanes 0-15 are its top halves and lanes 16-31 its bottom halves
// (role chosen per lane inside the same barrier structure; the tableau arithmetic, two thirds of a half's stream, is common code).
// Eight wavefronts, two per SIMD (wave w runs on SIMD w mod 4):
//   SIMD 0  wave 0 top (columns 0-3)     wave 4 bottom (0-3)
//   SIMD 1  wave 1 top (4-7)             wave 5 bottom (4-7)
//   SIMD 2  wave 2 top (8-11)            wave 6 mixed (column 12)
//   SIMD 3  wave 3 base                  wave 7 bottom (8-11)
// so that the base wave (a DPP quad per segment: r | v, m | lambda_v | lambda_r, lambda_m; four components and 13 x 4 slopes per lane)
// shares its SIMD with the shorter of the two column streams and the mixed wavefront with the other.
// The stage record grows by two pairs -- (umn, mnn) for the top halves, (Lm, Lnn) for the bottom halves: the mass couplings of
// var_col14 with the unit vector folded in (dynamics.hpp var_col14_top_dy / _bottom_dy) -- and the top half hands mu over with the
// third component of delta r.  Step control as in the 12-dim kernel: one step sequence per segment, the error norm over the base
// state and all 14 x 14 partials (the unit column contributes zeros, as a dual number's would), rk.hpp dp8_decide.
// Every loop is bounded, every wavefront executes the same barriers, out-of-range lanes shadow a valid segment without storing.
#include "kernels.hpp"
#include "rk.hpp"
#include "halves.hpp"

namespace lto {

constexpr int C14_SEG = 16;      // segments per workgroup
constexpr int C14_WAVES = 8;     // wavefronts that contribute to a norm: three top, three bottom, the mixed one, the base wave
constexpr int C14_PLD = 9;       // entries per segment of the partial-sum table (eight used; 16-byte entries, pitch 36 dwords: conflict-free)
// Stage record of a segment as 16-byte pairs (pitch 52 dwords: the 16 segments of a 128-bit access start in 16 different 4-bank groups):
//   0 (A, y)  2 (z, e1)  4 (q1, es)   6 (z, e2)  8 (q2, omc)   10 (l0, l1)  12 (l2, -)  14 (ua, ubn)        as kernels_indirect_coop2.hip
//   16 (umn, mnn)  from base lane 0      18 (Lm, Lnn)  from base lane 1       20 .. 25  where lanes with nothing to add write
constexpr int C14_LD = 26;
constexpr int C14_SLD = 18;      // scale table [segment][base lane][4]
constexpr int C14_COLS = 13;

enum C14Role : int { C14_TOP = 0, C14_BOTTOM = 1, C14_BASE = 2, C14_MIXED = 3 };

typedef double c14_d2 __attribute__((ext_vector_type(2)));

struct C14Shared {
  alignas(16) double rec[2][C14_SEG][C14_LD];          // stage records, double-buffered by the parity of the stage index
  alignas(16) c14_d2 xa[2][2][C14_COLS][C14_SEG];      // [buffer][half that wrote][column][segment]: first triple of the stage argument, (0, 1) ...
  alignas(16) c14_d2 xb[2][2][C14_COLS][C14_SEG];      // ... and (2, mu): the top half's mass component travels with it
  alignas(16) double part[C14_SEG][C14_PLD][2];        // partial norms [segment][wavefront][which]
  alignas(16) double scale[C14_SEG][C14_SLD];          // 1 / (atol + rtol |base value|): base lane q's rows at 4 q .. 4 q + 3
};

template <int PM, int ROLE>
__device__ __forceinline__ void coop14_run(const IndirectArgs& a, C14Shared& sh, const int lane, const int cwave) {
  constexpr bool BASE = (ROLE == C14_BASE);
  constexpr bool MIXED = (ROLE == C14_MIXED);
  constexpr int NS = 12;
  constexpr int NC = BASE ? 4 : 7;       // components per lane
  // ---- who is this lane
  int seg, col = 0;
  const int q4 = lane & 3;               // base wave: the quad's lane (r | v, m | lambda_v | lambda_r, lambda_m)
  if (BASE) {
    seg = lane >> 2;
  } else {
    seg = lane & (C14_SEG - 1);
    col = MIXED ? 12 : cwave * 4 + (lane >> 4);
  }
  // column role of this lane: compile-time in the pure wavefronts, by lane in the mixed one (lanes 32-63 repeat lanes 0-31 and do not count)
  const bool top = MIXED ? (((lane >> 4) & 1) == 0) : (ROLE == C14_TOP);
  const bool counts = MIXED ? (lane < 32) : true;
  const int half = top ? 0 : 1;
  // own rows in the 14-dim numbering (r 0-2, v 3-5, m 6, lambda_r 7-9, lambda_v 10-12, lambda_m 13); -1: a base lane's unused fourth component
  int grow[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    if (BASE) grow[j] = (j < 3) ? ((q4 == 0) ? 0 : (q4 == 1) ? 3 : (q4 == 2) ? 10 : 7) + j : ((q4 == 1) ? 6 : (q4 == 3) ? 13 : -1);
    else grow[j] = top ? j : (j < 3 ? 10 + j : (j < 6 ? 4 + j : 13));
  }
  const int widx = BASE ? 7 : (MIXED ? 6 : (ROLE == C14_TOP ? cwave : 3 + cwave));

  const int s_raw = xcd_unit(a, blockIdx.x, gridDim.x) * C14_SEG + seg;
  const int s_lin = s_raw < a.S ? s_raw : a.S - 1;             // shadow lanes repeat the last segment
  const int s = a.order ? a.order[s_lin] : s_lin;
  const bool in_range = (s_raw < a.S);
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const double span = a.t[tg + 1] - a.t[tg];
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  const double w2 = 2.0 * tp.omega;
  const bool mine = !a.class_filter || p_class(tp.p) == PM;
  if (!__syncthreads_or(mine)) return;         // workgroup-uniform
  const QuadLane Q = quad_lane(q4, tp);
  // where this base lane's part of each record store goes
  const int o1 = (q4 == 0) ? 0 : (q4 == 2) ? 10 : 20;                       // (a0, P1): (A, y) / (l0, l1)
  const int o2 = (q4 == 0) ? 2 : (q4 == 1) ? 6 : (q4 == 2) ? 12 : 22;       // (P2, e): (z, e1) / (z, e2) / (l2, -)
  const int o3 = (q4 == 0) ? 4 : (q4 == 1) ? 8 : 22;                        // (q, es | omc)
  const int o4 = (q4 == 2) ? 14 : 24;                                       // (ua, ubn): the same in every lane of the quad
  const int o5 = (q4 == 0) ? 16 : (q4 == 1) ? 18 : 24;                      // lane 0: (umn, mnn); lane 1: (Lm, Lnn)
  const int sc0 = BASE ? 4 * q4 : (top ? 0 : 8);                            // this lane's rows in the scale table
  auto sc_at = [&](const int j) { return BASE ? sc0 + j : sc0 + (j < 3 ? j : (j < 6 ? j + 1 : 7)); };

  // ---- state of this lane
  double y[NC], K[13][NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) {
    if (BASE) y[j] = (grow[j] >= 0) ? a.X[(long)grow[j] * a.ldx + node] : 0.0;
    else y[j] = (grow[j] == col) ? 1.0 : 0.0;
  }
  // the base wave and the bottom wave that shares its SIMD: the base stream is the one everybody waits for
  if (BASE) __builtin_amdgcn_s_setprio(2);

  auto base_eval = [&](const double (&arg)[NC], double (&out)[NC], const int buf) {
   if constexpr (BASE) {
    QuadParts14 qp;
    double P[3];
    rhs14_base_quad<PM>(arg, Q, tp, out, qp, P);
    BaseParts12 bp;
    bp.ua = qp.q.ua; bp.ub = qp.q.ub;
    const bool nz = qp.q.n2 > 0.0;                 // lambda_v = 0 exactly: the reference's guard makes the control a constant (parts_guard_zero_norm)
    bp.ua = nz ? bp.ua : 0.0; bp.ub = nz ? bp.ub : 0.0;
    const double un = nz ? qp.un : 0.0;
    double* r = &sh.rec[buf][seg][0];
    const double ubn = (bp.ub * qp.q.inv_n) * qp.q.inv_n;
    const double x3 = Q.lane1 ? qp.q.omc : qp.q.es;
    // mass couplings (dynamics.hpp var_col14_top_dy / _bottom_dy)
    const double inv_m = qp.inv_m;
    const double umn = bp.ua * inv_m;                                            // (umag / mass) / n
    const double mnn = ((-tp.kappa_td * qp.mass) * un) * qp.q.inv_n;             // -kappa_td mass umag' / n
    const double Lm = nz ? 2.0 * ((qp.umag * qp.n) * inv_m) * inv_m : 0.0;
    const double Lnn = nz ? -(__builtin_fma(un, qp.n, qp.umag) * inv_m) * qp.q.inv_n : 0.0;
    const double v50 = Q.lane1 ? Lm : umn, v51 = Q.lane1 ? Lnn : mnn;
    *reinterpret_cast<c14_d2*>(r + o1) = c14_d2{qp.q.a0, P[1]};
    *reinterpret_cast<c14_d2*>(r + o2) = c14_d2{P[2], qp.q.e};
    *reinterpret_cast<c14_d2*>(r + o3) = c14_d2{qp.q.q, x3};
    *reinterpret_cast<c14_d2*>(r + o4) = c14_d2{bp.ua, ubn};
    *reinterpret_cast<c14_d2*>(r + o5) = c14_d2{v50, v51};
   }
  };
  auto col_hand = [&](const double a0, const double a1, const double a2, const double a3, const int buf) {
    if constexpr (!BASE) {
      sh.xa[buf][half][col][seg] = c14_d2{a0, a1};
      sh.xb[buf][half][col][seg] = c14_d2{a2, a3};
    }
  };
  // behind the stage's barrier: the slope of the own half (order of work as in the 12-dim kernel)
  auto col_finish = [&](const double (&arg)[NC], double (&out)[NC], const int buf, auto&& sums, auto&& early) {
   if constexpr (!BASE) {
    const double* r = &sh.rec[buf][seg][0];
    auto pair = [&](const int slot) { return *reinterpret_cast<const c14_d2*>(r + slot); };
    const c14_d2 ay = pair(0), ze1 = pair(2), l01 = pair(10);
    const double e2v = r[7], l2v = r[12];
    const c14_d2 o01 = sh.xa[buf][1 - half][col][seg];
    const c14_d2 o23 = sh.xb[buf][1 - half][col][seg];
    const c14_d2 qo = pair(8);                                    // (q2, omc)
    sums();                                                       // common code of both halves: the next argument's sum over the older slopes
    if (top) { out[0] = arg[3]; out[1] = arg[4]; out[2] = arg[5]; }
    else { out[0] = __builtin_fma(w2, arg[1], -arg[3]); out[1] = __builtin_fma(-w2, arg[0], -arg[4]); out[2] = -arg[5]; }
    early(out);
    DyadParts dp;
    dp.A = ay.x; dp.yy = ay.y; dp.z = ze1.x; dp.e1 = ze1.y; dp.e2 = e2v; dp.omc = qo.y;
    const double other[3] = {o01.x, o01.y, o23.x};
    auto dyn = [&](auto top_c) {
      constexpr bool T = decltype(top_c)::value;
      const c14_d2 rx = pair(T ? 14 : 4);                         // top: (ua, ubn); bottom: (q1, es)
      const c14_d2 mc = pair(T ? 16 : 18);                        // top: (umn, mnn); bottom: (Lm, Lnn)
      double dw[7];
      if constexpr (T) var_col14_top_dy(dp, l01.x, l01.y, l2v, rx.x, rx.y, mc.x, mc.y, w2, arg, other, dw);
      else var_col14_bottom_dy(dp, l01.x, l01.y, l2v, rx.x, qo.x, rx.y, mc.x, mc.y, w2, arg, other, o23.y, dw);
      out[3] = dw[3]; out[4] = dw[4]; out[5] = dw[5]; out[6] = dw[6];
    };
    if constexpr (ROLE == C14_TOP) dyn(std::true_type{});
    else if constexpr (ROLE == C14_BOTTOM) dyn(std::false_type{});
    else { if (top) dyn(std::true_type{}); else dyn(std::false_type{}); }
   }
  };
  auto nothing = [] {};
  auto nothing1 = [](const double (&)[NC]) {};
  auto slope = [&](const double (&arg)[NC], double (&out)[NC], const int buf) {
    if constexpr (BASE) base_eval(arg, out, buf);
    else col_hand(arg[0], arg[1], arg[2], arg[6], buf);
    __syncthreads();
    if constexpr (!BASE) col_finish(arg, out, buf, nothing, nothing1);
  };
  auto post2 = [&](double p0, double p1) {
    if (!counts) { p0 = 0.0; p1 = 0.0; }
    const double s0 = BASE ? quad_sum(p0) : rows_sum(p0), s1 = BASE ? quad_sum(p1) : rows_sum(p1);
    c14_d2 v; v.x = s0; v.y = s1;
    *reinterpret_cast<c14_d2*>(&sh.part[seg][widx][0]) = v;
  };
  auto totals2 = [&](double& T0, double& T1) {
    c14_d2 acc = *reinterpret_cast<const c14_d2*>(&sh.part[seg][0][0]);
#pragma unroll
    for (int w = 1; w < C14_WAVES; ++w) {
      const c14_d2 v = *reinterpret_cast<const c14_d2*>(&sh.part[seg][w][0]);
      acc.x += v.x; acc.y += v.y;
    }
    T0 = acc.x; T1 = acc.y;
  };

  const double rtol = a.rtol, atol = a.atol;
  const unsigned long tab = dp8_tab_base();
  double h_abs = 0.0, t = 0.0;
  double rejected = 0.0;
  int nacc = 0, nrej = 0;
  int done = !(span > 0.0) || !mine;
  constexpr double NCOMP = 210.0;              // 14 + 14 x 14 components (the unit column's are zeros)

  // ---- first step size: Hairer's rule over all components, row r of every column scaled with the base value of row r
  {
    if (BASE) {
#pragma unroll
      for (int j = 0; j < NC; ++j) sh.scale[seg][sc0 + j] = rcp_nr(__builtin_fma(rtol, fabs(y[j]), atol));
    }
    slope(y, K[0], 0);
    double isc0[NC];
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double isc = sh.scale[seg][sc_at(j)];
      isc0[j] = isc;
      p0 = __builtin_fma(y[j] * isc, y[j] * isc, p0);
      p1 = __builtin_fma(K[0][j] * isc, K[0][j] * isc, p1);
    }
    post2(p0, p1);
    __syncthreads();
    double t0, t1;
    totals2(t0, t1);
    const double d0 = sqrt(t0 / NCOMP), d1 = sqrt(t1 / NCOMP);
    const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    double arg[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) arg[j] = __builtin_fma(h0, K[0][j], y[j]);
    slope(arg, K[1], 1);   // the barrier inside also separates the reads above from the writes below
    double p2 = 0.0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double df = (K[1][j] - K[0][j]) * isc0[j];
      p2 = __builtin_fma(df, df, p2);
    }
    post2(p2, 0.0);
    __syncthreads();
    double t2, tu;
    totals2(t2, tu);
    const double d2 = sqrt(t2 / NCOMP) / h0;
    const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : cbrt(cbrt(0.01 / fmax(d1, d2)));
    h_abs = fmin(fmin(100.0 * h0, h1), span);
  }

  // ---- trial steps (structure of kernels_indirect_coop2.hip: error sums before the FSAL barrier, the decision behind it)
  for (int trial = 0; trial < a.max_steps; ++trial) {
    double h = h_abs;
    double last = 0.0;
    if (t + h >= span) { h = span - t; last = 1.0; }
    double arg[NC], yn[NC], a5[NC], a3[NC], iscb[NC];
    double wrow[2][12], we5[13], we3[13];
    dp8_load_row<2>(tab, wrow[0]);
    {
      const double w10 = coef_here(DP8_A[1][0]);
#pragma unroll
      for (int j = 0; j < NC; ++j) arg[j] = __builtin_fma(h, w10 * K[0][j], y[j]);
    }
    if constexpr (!BASE) col_hand(arg[0], arg[1], arg[2], arg[6], 1);
    int alive = 1;
    auto stage = [&](auto st_c) {                // enters with K[0] = f(y) (FSAL) and arg = argument st; leaves with K[st] and argument st + 1
      constexpr int st = decltype(st_c)::value;
      static_assert(st >= 1 && st < NS, "stage index");
      const double (&wn)[12] = wrow[(st + 1) & 1];
      if constexpr (st + 2 <= NS) dp8_load_row<st + 2>(tab, wrow[st & 1]);
      if constexpr (st == NS - 2) dp8_load_err(tab, we5, we3);
      double next[NC], argn[NC];
      auto sums = [&] {
#pragma unroll
        for (int j = 0; j < NC; ++j) next[j] = 0.0;
#pragma unroll
        for (int k = 0; k < st; ++k) {
          if (dp8_row_entry(st, k) != 0.0) {
#pragma unroll
            for (int j = 0; j < NC; ++j) next[j] = __builtin_fma(wn[k], K[k][j], next[j]);
          }
        }
        if constexpr (st == NS - 1) {
#pragma unroll
          for (int j = 0; j < NC; ++j) { a5[j] = 0.0; a3[j] = 0.0; }
#pragma unroll
          for (int k = 0; k < NS - 1; ++k) {
            if (DP8_E5[k] != 0.0) {
#pragma unroll
              for (int j = 0; j < NC; ++j) a5[j] = __builtin_fma(we5[k], K[k][j], a5[j]);
            }
            if (DP8_E3[k] != 0.0) {
#pragma unroll
              for (int j = 0; j < NC; ++j) a3[j] = __builtin_fma(we3[k], K[k][j], a3[j]);
            }
          }
        }
      };
      auto next_arg = [&](const int j) {
        const double acc = (dp8_row_entry(st, st) != 0.0) ? __builtin_fma(wn[st], K[st][j], next[j]) : next[j];
        argn[j] = __builtin_fma(h, acc, y[j]);
      };
      if constexpr (BASE) {
        base_eval(arg, K[st], st & 1);
        dp8_pin_row<st + 1>(wn);
        if constexpr (st == NS - 1) dp8_pin_err(we5, we3);
        sums();
#pragma unroll
        for (int j = 0; j < NC; ++j) next_arg(j);
        if constexpr (st == NS - 1) {
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            iscb[j] = rcp_nr(__builtin_fma(rtol, fmax(fabs(y[j]), fabs(argn[j])), atol));
            sh.scale[seg][sc0 + j] = iscb[j];
          }
        }
      } else {
        dp8_pin_row<st + 1>(wn);
        if constexpr (st == NS - 1) dp8_pin_err(we5, we3);
      }
      // the first stage's barrier carries the vote that ends the sweep of this workgroup (every segment done)
      if constexpr (st == 1) { alive = __syncthreads_or(!done); if (!alive) return; }
      else __syncthreads();
      if constexpr (!BASE) {
        col_finish(arg, K[st], st & 1, sums, [&](const double (&)[NC]) {
          next_arg(0); next_arg(1); next_arg(2);
        });
        next_arg(3); next_arg(4); next_arg(5); next_arg(6);
        col_hand(argn[0], argn[1], argn[2], argn[6], (st + 1) & 1);
      }
#pragma unroll
      for (int j = 0; j < NC; ++j) arg[j] = argn[j];
    };
    stage(std::integral_constant<int, 1>{});
    if (!alive) break;                       // workgroup-uniform: the vote is the barrier's
    static_for<2, NS>(stage);
    {
      double e5 = 0.0, e3 = 0.0;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        yn[j] = arg[j];
        double s5 = a5[j], s3 = a3[j];
        if (DP8_E5[NS - 1] != 0.0) s5 = __builtin_fma(we5[NS - 1], K[NS - 1][j], s5);
        if (DP8_E3[NS - 1] != 0.0) s3 = __builtin_fma(we3[NS - 1], K[NS - 1][j], s3);
        const double isc = BASE ? iscb[j] : sh.scale[seg][sc_at(j)];
        s5 *= isc; s3 *= isc;
        e5 = __builtin_fma(s5, s5, e5);
        e3 = __builtin_fma(s3, s3, e3);
      }
      post2(e5, e3);
    }
    static_assert(DP8_E5[NS] == 0.0 && DP8_E3[NS] == 0.0, "the error estimate must not involve the FSAL slope");
    if constexpr (BASE) base_eval(yn, K[NS], NS & 1);
    __syncthreads();
    if constexpr (!BASE) col_finish(yn, K[NS], NS & 1, nothing, nothing1);
#pragma unroll
    for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(K[NS][j]));
    double E5, E3;
    totals2(E5, E3);
    double accept, bad;
    dp8_decide(E5, E3, h, rejected, NCOMP, h_abs, accept, bad);
    asm volatile("" : "+v"(h_abs));
    if (!done) {
      if (bad != 0.0) {                    // NaN in the step: NaN results (status_flag 2 upstream), no max_steps stall
#pragma unroll
        for (int j = 0; j < NC; ++j) y[j] = bad;
        t = span;
      } else if (accept != 0.0) {
        t = (last != 0.0) ? span : t + h;
#pragma unroll
        for (int j = 0; j < NC; ++j) { y[j] = yn[j]; K[0][j] = K[NS][j]; }
        ++nacc;
        rejected = 0.0;
      } else {
        ++nrej;
        rejected = 1.0;
      }
      if (!(t < span)) done = 1;
    }
  }
  // A segment that did not reach t1 has no result: NaN (status_flag 2, indirect.jl:339-341)
  if (mine && (t < span || !(span >= 0.0))) {
#pragma unroll
    for (int j = 0; j < NC; ++j) y[j] = __builtin_nan("");
  }

  if (in_range && mine && counts) {
    if (BASE) {
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        if (grow[j] >= 0) {
          if (a.defect) a.defect[(long)grow[j] * a.ldd + s] = y[j] - a.X[(long)grow[j] * a.ldx + node + 1];
          // the lambda_m column of the STM: the unit vector (nothing feeds on lambda_m for these laws); a NaN segment poisons it too
          a.Phi[(long)(13 * 14 + grow[j]) * a.ldp + s] = (y[j] != y[j]) ? y[j] : ((grow[j] == 13) ? 1.0 : 0.0);
        }
      }
      if (q4 == 0) {
        if (a.errors) a.errors[s] = 0.0;
        if (a.nacc) a.nacc[s] = nacc;
        if (a.nrej) a.nrej[s] = nrej;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NC; ++j) a.Phi[(long)(col * 14 + grow[j]) * a.ldp + s] = y[j];
    }
  }
}

template <int PM>
__global__ __launch_bounds__(512) void k_indirect_coop2_14(const IndirectArgs a) {
  __shared__ C14Shared sh;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wave == 3) coop14_run<PM, C14_BASE>(a, sh, lane, 0);
  else if (wave < 3) coop14_run<PM, C14_TOP>(a, sh, lane, wave);
  else if (wave == 6) coop14_run<PM, C14_MIXED>(a, sh, lane, 0);
  else coop14_run<PM, C14_BOTTOM>(a, sh, lane, wave == 7 ? 2 : wave - 4);
}

template <int PM>
static hipError_t launch_coop14_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + C14_SEG - 1) / C14_SEG);
  hipLaunchKernelGGL((k_indirect_coop2_14<PM>), grid, dim3(512), 0, st, a);
  return hipGetLastError();
}

bool indirect_stm_coop2_14_available(int pm) { return (pm & ~((1 << PM_P0) | (1 << PM_P1))) == 0; }

// 14-dim system, DOP853 adaptive, batches of the always-thrust-limited laws only (p = 0, p = 1).
hipError_t launch_indirect_stm_coop2_14(int pm, const IndirectArgs& a0, hipStream_t st) {
  if (a0.S <= 0) return hipSuccess;
  if (!indirect_stm_coop2_14_available(pm) || !a0.Phi) return hipErrorInvalidValue;
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_coop14_one<PM_P0>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_coop14_one<PM_P1>(a, st);
  return e;
}

}  // namespace lto
