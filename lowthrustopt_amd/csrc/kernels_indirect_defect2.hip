// kernels_indirect_defect2.hip -- defect-only sweep with the reference's integrator setting (adaptive order 8, rtol = atol =
// 1e-13; src/multiShoot_CRTBP_indirect.jl:63-90, :79) on the 12-dim system, TWO LANES PER SEGMENT (k_indirect_defect2, round 2)
// and FOUR (k_indirect_defect4, round 3: below; what AUTO runs up to 131 072 segments on MI355X).
//
// The one-lane kernel (k_indirect<12, PM, M_DOP853_ADAPTIVE, 0>) keeps ten live slopes of 12 components: 240 of the 256
// registers a VALU instruction can address, so state, argument and temporaries overflow into AGPRs (620 v_accvgpr moves per
// trial step), and a sweep lasts as long as its slowest segment's instruction stream (C5: 69 trial steps against a mean of
// 8.5).  Here lane A owns (r, v) and lane B -- four lanes up, the next DPP bank -- owns (lambda_v, lambda_r): six components
// and 13 x 6 slopes per lane, the tableau arithmetic per lane halves, nothing spills.  Both lanes hold r and lambda_v (three
// doubles each cross over by v_mov_b32_dpp) and run one instruction stream (rhs12_base_half); error norms are pair sums
// formed in the same order in both lanes, so both take the same decisions and the pair's control flow never diverges.
// Twice the wavefronts of the one-lane kernel: AUTO runs it between 131 072 and 262 144 segments (lto_api.hip).
#include "kernels.hpp"
#include "rk.hpp"
#include "halves.hpp"

namespace lto {

template <int PM>
__global__ __launch_bounds__(64) void k_indirect_defect2(const IndirectArgs a) {
  const int lane = threadIdx.x;
  const bool is_a = ((lane >> 2) & 1) == 0;
  const int sl = xcd_unit(a, blockIdx.x, gridDim.x) * 32 + (lane >> 3) * 4 + (lane & 3);       // 32 segments per wavefront; an XCD's wavefronts own a contiguous range (kernels.hpp)
  if (sl >= a.S) return;                                               // both lanes of a pair leave together
  const int s = a.order ? a.order[sl] : sl;
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const double span = arg_span(a, node, tg);
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  if (a.class_filter && p_class(tp.p) != PM) return;                   // mixed-class batch: another launch owns this trajectory
  const double w2 = 2.0 * tp.omega;
  const double sg = is_a ? 1.0 : -1.0, kap = is_a ? 0.0 : w2;
  int grow[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) grow[j] = is_a ? j : (j < 3 ? 9 + j : 3 + j);

  double y[6], K[13][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) y[j] = arg_node(a, grow[j], node);

  auto rhs = [&](const double (&arg)[6], double (&out)[6]) {
    double R[3], L[3], q[3], kp[3], kq[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { R[j] = from_lane_a(arg[j]); L[j] = from_lane_b(arg[j]); q[j] = arg[3 + j]; }
    BaseParts12 bp;
    rhs12_base_half<PM>(R, L, q, is_a, sg, kap, tp, kp, kq, bp);
#pragma unroll
    for (int j = 0; j < 3; ++j) { out[j] = kp[j]; out[3 + j] = kq[j]; }
  };

  const double rtol = a.rtol, atol = a.atol;
  int nacc = 0, nrej = 0;
  double t = 0.0;
  if (span > 0.0) {
    rhs(y, K[0]);
    double h_abs;
    // warm start (lto_indirect_plan_set_warm_start): the first accepted step size of this segment in the previous defect-only sweep of
    // the plan -- consecutive Newton iterations and line-search trials sweep nearly the same trajectory -- instead of Hairer's rule,
    // which costs an extra RHS evaluation and starts a decade or two low.  Wave-uniform choice; any positive value is a valid start.
    if (a.warm) {
      const double hw = a.h_first[s];
      h_abs = (hw > 0.0) ? fmin(fmax(hw, 1e-6 * span), span) : 1e-3 * span;      // any positive value is a valid start; a tiny one would cost hundreds of trial steps
    } else {   // Hairer's initial step over the 12 components
      double isc[6], p0 = 0.0, p1 = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        isc[j] = rcp_nr(__builtin_fma(rtol, fabs(y[j]), atol));
        p0 = __builtin_fma(y[j] * isc[j], y[j] * isc[j], p0);
        p1 = __builtin_fma(K[0][j] * isc[j], K[0][j] * isc[j], p1);
      }
      const double d0 = sqrt(pair_sum(p0) / 12.0), d1 = sqrt(pair_sum(p1) / 12.0);
      const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
      double yt[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) yt[j] = __builtin_fma(h0, K[0][j], y[j]);
      rhs(yt, K[1]);
      double p2 = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double df = (K[1][j] - K[0][j]) * isc[j];
        p2 = __builtin_fma(df, df, p2);
      }
      const double d2 = sqrt(pair_sum(p2) / 12.0) / h0;
      const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : cbrt(cbrt(0.01 / fmax(d1, d2)));
      h_abs = fmin(fmin(100.0 * h0, h1), span);
    }
    double h_rec = 0.0;          // proposal that led to the first accepted step (what the next sweep starts from)
    double rejected = 0.0;       // per-lane flags as doubles (DESIGN.md "Compiler hazards")
    while (t < span && nacc + nrej < a.max_steps) {
      double h = h_abs;
      double last = 0.0;
      if (t + h >= span) { h = span - t; last = 1.0; }
#pragma unroll
      for (int st = 1; st < 12; ++st) {
        double arg[6], acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = 0.0;
#pragma unroll
        for (int k = 0; k < st; ++k)
          if (DP8_A[st][k] != 0.0) {
            const double w = coef_here(DP8_A[st][k]);
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[j] = __builtin_fma(w, K[k][j], acc[j]);
          }
#pragma unroll
        for (int j = 0; j < 6; ++j) arg[j] = __builtin_fma(h, acc[j], y[j]);
        rhs(arg, K[st]);
      }
      double yn[6], acc[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[j] = 0.0;
#pragma unroll
      for (int k = 0; k < 12; ++k)
        if (DP8_B[k] != 0.0) {
          const double w = coef_here(DP8_B[k]);
#pragma unroll
          for (int j = 0; j < 6; ++j) acc[j] = __builtin_fma(w, K[k][j], acc[j]);
        }
#pragma unroll
      for (int j = 0; j < 6; ++j) yn[j] = __builtin_fma(h, acc[j], y[j]);
      rhs(yn, K[12]);
      double a5[6], a3[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) { a5[j] = 0.0; a3[j] = 0.0; }
#pragma unroll
      for (int k = 0; k <= 12; ++k) {
        if (DP8_E5[k] != 0.0) {
          const double w = coef_here(DP8_E5[k]);
#pragma unroll
          for (int j = 0; j < 6; ++j) a5[j] = __builtin_fma(w, K[k][j], a5[j]);
        }
        if (DP8_E3[k] != 0.0) {
          const double w = coef_here(DP8_E3[k]);
#pragma unroll
          for (int j = 0; j < 6; ++j) a3[j] = __builtin_fma(w, K[k][j], a3[j]);
        }
      }
      double e5 = 0.0, e3 = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const double isc = rcp_nr(__builtin_fma(rtol, fmax(fabs(y[j]), fabs(yn[j])), atol));
        const double s5 = a5[j] * isc, s3 = a3[j] * isc;
        e5 = __builtin_fma(s5, s5, e5);
        e3 = __builtin_fma(s3, s3, e3);
      }
      const double E5 = pair_sum(e5), E3 = pair_sum(e3);
      double h_next, accept, bad;
      dp8_decide(E5, E3, h, rejected, 12.0, h_next, accept, bad);       // the controller of the four-lane and cooperative forms (rk.hpp)
      if (accept != 0.0) {
        if (nacc == 0) h_rec = h_abs;
        h_abs = h_next;
        t = (last != 0.0) ? span : t + h;
#pragma unroll
        for (int j = 0; j < 6; ++j) { y[j] = yn[j]; K[0][j] = K[12][j]; }
        ++nacc;
        rejected = 0.0;
      } else {
        h_abs = h_next;
        rejected = 1.0;
        ++nrej;
        if (bad != 0.0) {                     // a NaN never recovers: poison and stop instead of max_steps retries
#pragma unroll
          for (int j = 0; j < 6; ++j) y[j] = bad;
          t = span;
        }
      }
    }
    if (t < span) {                           // max_steps trial steps used up before t1: no result
#pragma unroll
      for (int j = 0; j < 6; ++j) y[j] = __builtin_nan("");
    }
    if (is_a && a.h_first) a.h_first[s] = h_rec;
  } else if (span != 0.0) {                   // decreasing grid (forward integration only) or NaN span: no result
#pragma unroll
    for (int j = 0; j < 6; ++j) y[j] = __builtin_nan("");
  }

  if (a.defect || a.Da) {
#pragma unroll
    for (int j = 0; j < 6; ++j) put_defect(a, grow[j], s, y[j] - arg_node(a, grow[j], node + 1));
  }
  if (is_a) {
    if (a.errors) a.errors[s] = 0.0;
    if (a.nacc) a.nacc[s] = nacc;
    if (a.nrej) a.nrej[s] = nrej;
  }
}

// ------------------------------------------------------------------------------------------- FOUR LANES PER SEGMENT
// A sweep of a few thousand segments lasts as long as the instruction stream of its slowest segment (4 096 segments: 16 trial
// steps against a mean of 5.3), so the stream is what to shorten: with a DPP quad per segment -- lane 0 owns r, lane 1 v, lane 2
// lambda_v, lane 3 lambda_r -- the tableau arithmetic per lane is half that of the two-lane kernel, and the three
// reciprocal-square-root chains of an evaluation run as one (rhs12_base_quad, halves.hpp).  The tableau rows come by scalar loads
// one stage ahead of their use (rk.hpp: dp8_load_row) instead of two s_mov_b32 per coefficient.  Same step control; the error
// norms are quad sums formed in the same order in all four lanes, so a quad's control flow never diverges.  Measured
// (tools/probe_defect2.py): 4 096 segments 90 -> 77 us per sweep, 29: 73 -> 63 us, 65 536 ordered: 0.31 -> 0.27 ms; AUTO up to
// eight wavefronts per SIMD (lto_api.hip).
// ND = 14 (round 6): BASELINE configs[1]'s system for the always-thrust-limited laws -- lane 1 owns (v, m), lane 3 (lambda_r, lambda_m),
// four components per lane with a zero in lanes 0 and 2 (halves.hpp rhs14_base_quad); norms over the 14 components.
template <int PM, int ND = 12>
__global__ __launch_bounds__(64) void k_indirect_defect4(const IndirectArgs a) {
  constexpr int NC = (ND == 14) ? 4 : 3;
  constexpr double NCOMP = (double)ND;
  const int lane = threadIdx.x;
  const int q4 = lane & 3;
  const int sl = xcd_unit(a, blockIdx.x, gridDim.x) * 16 + (lane >> 2);   // 16 segments per wavefront; an XCD's wavefronts own a contiguous range (kernels.hpp)
  if (sl >= a.S) return;                                               // the four lanes of a quad leave together
  const int s = a.order ? a.order[sl] : sl;
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const double span = (ND == 12) ? arg_span(a, node, tg) : a.t[tg + 1] - a.t[tg];
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  if (a.class_filter && p_class(tp.p) != PM) return;                   // mixed-class batch: another launch owns this trajectory
  const QuadLane Q = quad_lane(q4, tp);
  // own rows in global numbering; 14-dim: r 0-2 | v 3-5, m 6 | lambda_v 10-12 | lambda_r 7-9, lambda_m 13 (-1: the unused fourth component)
  const int row0 = (ND == 12) ? ((q4 == 0) ? 0 : (q4 == 1) ? 3 : (q4 == 2) ? 9 : 6) : ((q4 == 0) ? 0 : (q4 == 1) ? 3 : (q4 == 2) ? 10 : 7);
  const int row3 = (q4 == 1) ? 6 : (q4 == 3) ? 13 : -1;
  auto row_of = [&](const int j) { return j < 3 ? row0 + j : row3; };
  auto node_value = [&](const int j, const long nd) {
    if constexpr (ND == 12) return arg_node(a, row0 + j, nd);
    else return (row_of(j) >= 0) ? a.X[(long)row_of(j) * a.ldx + nd] : 0.0;
  };

  double y[NC], K[13][NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) y[j] = node_value(j, node);

  auto rhs = [&](const double (&arg)[NC], double (&out)[NC]) {
    double P[3];
    if constexpr (ND == 12) { QuadParts bp; rhs12_base_quad<PM>(arg, Q, tp, out, bp, P); }
    else { QuadParts14 bp; rhs14_base_quad<PM>(arg, Q, tp, out, bp, P); }
  };

  const double rtol = a.rtol, atol = a.atol;
  const unsigned long tab = dp8_tab_base();
  int nacc = 0, nrej = 0;
  double t = 0.0;
  if (span > 0.0) {
    rhs(y, K[0]);
    double h_abs;
    if (a.warm) {                // warm start: see k_indirect_defect2
      const double hw = a.h_first[s];
      h_abs = (hw > 0.0) ? fmin(fmax(hw, 1e-6 * span), span) : 1e-3 * span;      // any positive value is a valid start; a tiny one would cost hundreds of trial steps
    } else {                     // Hairer's initial step over the 12 components
      double isc[NC], p0 = 0.0, p1 = 0.0;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        isc[j] = rcp_nr(__builtin_fma(rtol, fabs(y[j]), atol));
        p0 = __builtin_fma(y[j] * isc[j], y[j] * isc[j], p0);
        p1 = __builtin_fma(K[0][j] * isc[j], K[0][j] * isc[j], p1);
      }
      const double d0 = sqrt(quad_sum(p0) / NCOMP), d1 = sqrt(quad_sum(p1) / NCOMP);
      const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
      double yt[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) yt[j] = __builtin_fma(h0, K[0][j], y[j]);
      rhs(yt, K[1]);
      double p2 = 0.0;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const double df = (K[1][j] - K[0][j]) * isc[j];
        p2 = __builtin_fma(df, df, p2);
      }
      const double d2 = sqrt(quad_sum(p2) / NCOMP) / h0;
      const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : cbrt(cbrt(0.01 / fmax(d1, d2)));
      h_abs = fmin(fmin(100.0 * h0, h1), span);
    }
    double h_rec = 0.0;
    double rejected = 0.0;
    while (t < span && nacc + nrej < a.max_steps) {
      double h = h_abs;
      double last = 0.0;
      if (t + h >= span) { h = span - t; last = 1.0; }
      // the weights of an argument are fetched (scalar loads, dp8_load_row) while the previous stage is evaluated
      double yn[NC];
      double wrow[2][12], we5[13], we3[13];
      dp8_load_row<1>(tab, wrow[1]);
      // stage arguments 1 .. 11 and the new state (12): y + h sum_k w_k K_k
      auto argument = [&](auto st_c, double (&arg)[NC]) {
        constexpr int st = decltype(st_c)::value;
        const double (&w)[12] = wrow[st & 1];
        double acc[NC];
#pragma unroll
        for (int j = 0; j < NC; ++j) acc[j] = 0.0;
#pragma unroll
        for (int k = 0; k < st; ++k) {
          const double c = (st < 12) ? DP8_A[st < 12 ? st : 0][k] : DP8_B[k];
          if (c != 0.0) {
#pragma unroll
            for (int j = 0; j < NC; ++j) acc[j] = __builtin_fma(w[k], K[k][j], acc[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < NC; ++j) arg[j] = __builtin_fma(h, acc[j], y[j]);
      };
      static_for<1, 12>([&](auto st_c) {
        constexpr int st = decltype(st_c)::value;
        double arg[NC];
        argument(st_c, arg);
        dp8_load_row<st + 1>(tab, wrow[(st + 1) & 1]);
        if constexpr (st == 11) dp8_load_err(tab, we5, we3);
        rhs(arg, K[st]);
      });
      argument(std::integral_constant<int, 12>{}, yn);
      // Round 4: the error estimate of the 8(5,3) pair does not involve the FSAL slope f(y_new) (E5[12] = E3[12] = 0): the sums, the
      // quad's norm and the step decision (rk.hpp: dp8_decide, reciprocal-square-root chains instead of IEEE square roots and a
      // division) are formed BEFORE that evaluation, in whose instruction stream their dependent chains then hide.
      static_assert(DP8_E5[12] == 0.0 && DP8_E3[12] == 0.0, "the error estimate must not involve the FSAL slope");
      double a5[NC], a3[NC];
#pragma unroll
      for (int j = 0; j < NC; ++j) { a5[j] = 0.0; a3[j] = 0.0; }
#pragma unroll
      for (int k = 0; k < 12; ++k) {
        if (DP8_E5[k] != 0.0) {
#pragma unroll
          for (int j = 0; j < NC; ++j) a5[j] = __builtin_fma(we5[k], K[k][j], a5[j]);
        }
        if (DP8_E3[k] != 0.0) {
#pragma unroll
          for (int j = 0; j < NC; ++j) a3[j] = __builtin_fma(we3[k], K[k][j], a3[j]);
        }
      }
      double e5 = 0.0, e3 = 0.0;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const double isc = rcp_nr(__builtin_fma(rtol, fmax(fabs(y[j]), fabs(yn[j])), atol));
        const double s5 = a5[j] * isc, s3 = a3[j] * isc;
        e5 = __builtin_fma(s5, s5, e5);
        e3 = __builtin_fma(s3, s3, e3);
      }
      const double E5 = quad_sum(e5), E3 = quad_sum(e3);
      double h_next, accept, bad;
      dp8_decide(E5, E3, h, rejected, NCOMP, h_next, accept, bad);
      asm volatile("" : "+v"(h_next));        // (unpinned, the compiler sinks the next proposal's chain to the loop's tail, behind the commit branch)
      rhs(yn, K[12]);
      if (accept != 0.0) {
        if (nacc == 0) h_rec = h_abs;
        t = (last != 0.0) ? span : t + h;
#pragma unroll
        for (int j = 0; j < NC; ++j) { y[j] = yn[j]; K[0][j] = K[12][j]; }
        ++nacc;
        rejected = 0.0;
      } else {
        rejected = 1.0;
        ++nrej;
        if (bad != 0.0) {                     // a NaN never recovers: poison and stop instead of max_steps retries
#pragma unroll
          for (int j = 0; j < NC; ++j) y[j] = bad;
          t = span;
        }
      }
      h_abs = h_next;
    }
    if (t < span) {                           // max_steps trial steps used up before t1: no result
#pragma unroll
      for (int j = 0; j < NC; ++j) y[j] = __builtin_nan("");
    }
    if (q4 == 0 && a.h_first) a.h_first[s] = h_rec;
  } else if (span != 0.0) {                   // decreasing grid (forward integration only) or NaN span: no result
#pragma unroll
    for (int j = 0; j < NC; ++j) y[j] = __builtin_nan("");
  }

  if (a.defect || a.Da) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if constexpr (ND == 12) put_defect(a, row0 + j, s, y[j] - arg_node(a, row0 + j, node + 1));
      else if (row_of(j) >= 0) a.defect[(long)row_of(j) * a.ldd + s] = y[j] - node_value(j, node + 1);
    }
  }
  if (q4 == 0) {
    if (a.errors) a.errors[s] = 0.0;
    if (a.nacc) a.nacc[s] = nacc;
    if (a.nrej) a.nrej[s] = nrej;
  }
}

template <int PM>
static hipError_t launch_defect2_one(const IndirectArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((k_indirect_defect2<PM>), dim3((a.S + 31) / 32), dim3(64), 0, st, a);
  return hipGetLastError();
}

template <int PM, int ND = 12>
static hipError_t launch_defect4_one(const IndirectArgs& a, hipStream_t st) {
  hipLaunchKernelGGL((k_indirect_defect4<PM, ND>), dim3((a.S + 15) / 16), dim3(64), 0, st, a);
  return hipGetLastError();
}

// 14-dim system, DOP853 adaptive, defect only, four lanes per segment: batches of the always-thrust-limited laws (p = 0, p = 1)
hipError_t launch_indirect14_defect4(int pm, const IndirectArgs& a0, hipStream_t st) {
  if (a0.S <= 0) return hipSuccess;
  if ((pm & ~((1 << PM_P0) | (1 << PM_P1))) != 0 || !a0.defect) return hipErrorInvalidValue;
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_defect4_one<PM_P0, 14>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_defect4_one<PM_P1, 14>(a, st);
  return e;
}

// 12-dim system, DOP853 adaptive, defect only, four lanes per segment.
hipError_t launch_indirect_defect4(int pm, const IndirectArgs& a0, hipStream_t st) {
  if (a0.S <= 0) return hipSuccess;
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_defect4_one<PM_P0>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_defect4_one<PM_P1>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_defect4_one<PM_P2>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_defect4_one<PM_PGEN>(a, st);
  return e;
}

// 12-dim system, DOP853 adaptive, defect only.
hipError_t launch_indirect_defect2(int pm, const IndirectArgs& a0, hipStream_t st) {
  if (a0.S <= 0) return hipSuccess;
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_defect2_one<PM_P0>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_defect2_one<PM_P1>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_defect2_one<PM_P2>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_defect2_one<PM_PGEN>(a, st);
  return e;
}

}  // namespace lto
