// kernels_indirect_coop2.hip -- the reference's integrator setting (adaptive order 8, rtol = atol = 1e-13) with the STM by
// variational equations, 12-dim system (src/multiShoot_CRTBP_indirect.jl:79,107-110,121): the cooperative kernel with every
// 12-component state SPLIT OVER TWO LANES.
//
// Why: a DOP853 lane of kernels_indirect_coop.hip keeps ten live slopes of 12 components (240 registers) next to its state,
// argument and coefficients; that does not fit the 256 registers a VALU instruction can address, and 18 % of the issue slots
// of BOTH roles were v_accvgpr_read / v_accvgpr_write moves (1 070 of 5 900 instructions per trial step).  With six components
// per lane the slopes take 120 registers, the tableau arithmetic per lane halves, and nothing spills.
//
// One workgroup = 16 segments = 8 wavefronts (wave i and wave i + 4 share a SIMD):
//   waves 0-2  TOP halves of the 12 STM columns: lane = (segment, column), owns (a, b) = (delta r, delta v); needs G and U.
//   waves 4-6  BOTTOM halves: owns (d, g) = (delta lambda_v, delta lambda_r); needs G and H.  A top and a bottom wave share a
//              SIMD (92 + 135 instructions per stage between them; round 2: 123 + 159).
//   wave 3     base wave, alone on its SIMD (wave 7 leaves at once).  Round 3: FOUR lanes per segment, a DPP quad -- lane 0 owns
//              r, lane 1 v, lane 2 lambda_v, lane 3 lambda_r (three components and 13 x 3 slopes each) -- and one instruction
//              stream in which the three reciprocal-square-root chains of an evaluation (two primaries, |lambda_v|) are ONE
//              (rhs12_base_quad, halves.hpp); the lanes trade their pieces by v_mov_b32_dpp quad_perm.  Lane 1, which then
//              holds both primaries' terms, also builds the gravity-gradient block G of the variational equations once per
//              segment and stage (the 24 column halves of a segment used to assemble it 24 times).
// Per RK stage one __syncthreads(): before it the base lanes evaluate the stage and publish their argument (r, lambda_v), the
// by-products and G (20 doubles, double-buffered; 14 store instructions: the lanes of a quad hold different quantities under the
// same name), and the column halves -- whose stage argument needs only their own earlier slopes -- publish the triple their
// partner needs; after it the column halves finish their coefficients (wave-uniform code per half: var_col12_top_g, and
// var_col12_bottom_g, which builds H from the parts) while the base lanes are already in the next stage.  The DOP853 tableau
// rows come by scalar loads one stage ahead of their use (rk.hpp: dp8_load_row) instead of two s_mov_b32 per coefficient, and the
// workgroup's exit vote rides on the first stage barrier of the next trial step instead of a barrier of its own.
// Measured at 4 096 segments (tools/probe_coop2.py, probe build): 22.0 k -> 18.8 k ticks per trial step; per sweep 236 -> 219 us
// cold start, 224 -> 206 us with the controller's warm start (tools/probe_warm.py).
//
// Step control as in kernels_indirect_coop.hip: one common step sequence per segment, the error norm over the base state AND
// all 144 column components (what ForwardDiff duals see inside the adaptive solver), partial sums of the 28 roles of a
// segment through LDS, identical arithmetic in every lane of the segment => identical decision, no broadcast.
// Every loop is bounded (max_steps trial steps), every wavefront executes the same barriers, out-of-range lanes shadow a valid
// segment without storing: the grid always drains.
#include "kernels.hpp"
#include "rk.hpp"
#include "halves.hpp"
#include <pipe_hooks.hpp>   // product: hooks/ (no-ops); `make probe`: tools/probe_hooks/

namespace lto {

constexpr int C2_SEG = 16;      // segments per workgroup
constexpr int C2_WAVES = 7;     // wavefronts that contribute to a norm: three top, three bottom, the base wave
constexpr int C2_PLD = 9;       // entries per segment of the partial-sum table (seven used): 16-byte entries at a pitch of 36 dwords are conflict-free
// The stage record of a segment, as 16-byte pairs (a pitch of 44 dwords keeps the 16 segments of a 128-bit access in different banks):
//   0 (A, y)  2 (z, e1)  4 (q1, es)            from base lane 0: A = x + MU, e_b = 3 kappa_b / d_b^{5/2}, q_b = 5 e_b (rho_b . lambda_v) / d_b
//   6 (z, e2) 8 (q2, omc)                      from base lane 1: omc = 1 - sum_b kappa_b / d_b^{3/2}
//   10 (l0, l1)  12 (l2, -)  14 (ua, ubn)      from base lane 2: lambda_v, umag / n, (umag / n - umag') / n^2
//   16 .. 21                                   where the lanes that have nothing to add to a store write
// -- everything the column halves apply G, H and U from (dynamics.hpp: var_col12_top_dy / var_col12_bottom_dy); no matrix is built.
constexpr int C2_LD = 22;
constexpr int C2_SLD = 18;      // pitch of the scale table: [segment][base lane][4]

enum C2Role : int { C2_TOP = 0, C2_BOTTOM = 1, C2_BASE = 2 };

typedef double c2_d2 __attribute__((ext_vector_type(2)));

struct C2Shared {
  alignas(16) double rec[2][C2_SEG][C2_LD];        // stage records, double-buffered by the parity of the stage index
  alignas(16) c2_d2 xa[2][2][12][C2_SEG];          // [buffer][half that wrote][column][segment]: first triple of the stage argument, (0, 1) ...
  double xb[2][2][12][C2_SEG];                     // ... and 2
  alignas(16) double part[C2_SEG][C2_PLD][2];      // partial norms [segment][wavefront][which]: every wavefront sums its own lanes of a segment first
  alignas(16) double scale[C2_SEG][C2_SLD];        // 1 / (atol + rtol |base value|): base lane q's rows at 4 q .. 4 q + 2
};

// Probe build (make probe): ticks every role waits at the stage barriers and the ticks of its trial loop, per workgroup,
// into rows 16-19 (base), 20-21 (top wave 0), 22-23 (bottom wave 0) of a 24-row defect buffer (tools/probe_coop2.py).
// The hooks are no-ops in the product build (hooks/pipe_hooks.hpp).
#define C2_SYNC() c2_wait.sync()

template <int PM, int ROLE>
__device__ __forceinline__ void coop2_run(const IndirectArgs& a, C2Shared& sh, const int lane, const int cwave) {
  constexpr bool BASE = (ROLE == C2_BASE);
  constexpr int NS = 12;
  constexpr int NC = BASE ? 3 : 6;       // components per lane: a base lane owns one triple, a column lane half a column
  hook::Stamps c2_life;                  // probe build: when this workgroup entered, looped and left, and where (no-ops in the product)
  c2_life.mark(0);
  // ---- who is this lane
  int seg, col = 0;
  const int q4 = lane & 3;               // base wave: the quad's lane (r, v, lambda_v, lambda_r)
  if (BASE) {
    seg = lane >> 2;
  } else {
    seg = lane & (C2_SEG - 1);
    col = cwave * 4 + (lane >> 4);
  }
  // own rows in global numbering.  Column halves: top 0..5, bottom (9, 10, 11, 6, 7, 8); base lanes: 0.., 3.., 9.., 6..
  int grow[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j)
    grow[j] = BASE ? ((q4 == 0) ? 0 : (q4 == 1) ? 3 : (q4 == 2) ? 9 : 6) + j : (ROLE == C2_TOP) ? j : (j < 3 ? 9 + j : 3 + j);
  const int widx = BASE ? 6 : (ROLE == C2_TOP ? cwave : 3 + cwave);     // this wavefront's entry in the partial-sum table

  const int s_raw = xcd_unit(a, blockIdx.x, gridDim.x) * C2_SEG + seg;      // an XCD's workgroups own a contiguous range of segments (kernels.hpp)
  const int s_lin = s_raw < a.S ? s_raw : a.S - 1;             // shadow lanes repeat the last segment
  const int s = a.order ? a.order[s_lin] : s_lin;              // balanced order (lto_indirect_plan_rebalance)
  const bool in_range = (s_raw < a.S);
  const int traj = s / a.seg_per_traj;
  const int i = s - traj * a.seg_per_traj;
  const long node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  const double span = arg_span(a, node, tg);
  const TrajParams tp = a.tp[(long)traj * a.tp_stride];
  const double w2 = 2.0 * tp.omega;
  const bool mine = !a.class_filter || p_class(tp.p) == PM;
  if (!__syncthreads_or(mine)) return;         // workgroup-uniform (wave 7 voted 0 and left)
  const QuadLane Q = quad_lane(q4, tp);        // base lanes: per-lane constants of the quad evaluation (halves.hpp)
  // where this base lane's part of each of the four record stores goes (slots of C2_LD; 16.. = nobody reads)
  const int o1 = (q4 == 0) ? 0 : (q4 == 2) ? 10 : 16;                       // (a0, P1): (A, y) / (l0, l1)
  const int o2 = (q4 == 0) ? 2 : (q4 == 1) ? 6 : (q4 == 2) ? 12 : 18;       // (P2, e): (z, e1) / (z, e2) / (l2, -)
  const int o3 = (q4 == 0) ? 4 : (q4 == 1) ? 8 : 18;                        // (q, es | omc)
  const int o4 = (q4 == 2) ? 14 : 20;                                       // (ua, ubn): the same in every lane of the quad
  const int sc0 = BASE ? 4 * q4 : (ROLE == C2_TOP ? 0 : 8);                 // this lane's rows in the scale table: sc0 .. +2 (and sc0 + 4 .. +6)

  // ---- state of this lane: six rows of the base state or of one STM column
  double y[NC], K[13][NC];
  if (BASE) {
#pragma unroll
    for (int j = 0; j < NC; ++j) y[j] = arg_node(a, grow[j], node);
  } else {
#pragma unroll
    for (int j = 0; j < NC; ++j) y[j] = (grow[j] == col) ? 1.0 : 0.0;
  }

  hook::BarrierWait c2_wait;
  // A top and a bottom wave share a SIMD and the top wave -- the older one, which the SIMD serves first -- has the shorter stream (92
  // against 135 instructions per stage): it used to finish early and wait while the bottom wave ran on alone.  With the bottom
  // wave at raised priority the two finish together: 217-222 -> 213.6 us per sweep on the same box (tools/probe_warm.py).
  if (ROLE == C2_BOTTOM) __builtin_amdgcn_s_setprio(2);
  // ---- the two sides of a stage.  Base lanes: evaluate the stage argument and publish the record the column halves apply G, H, U from.
  auto base_eval = [&](const double (&arg)[NC], double (&out)[NC], const int buf) {
   if constexpr (BASE) {
    QuadParts qp;
    double P[3];
    rhs12_base_quad<PM, true>(arg, Q, tp, out, qp, P);
    BaseParts12 bp;
    bp.ua = qp.ua; bp.ub = qp.ub;
    parts_guard_zero_norm<PM>(qp.n2, bp);
    // four 128-bit stores (round 3: fourteen 64-bit ones, a quarter of the base wave's issue time): the lanes of a quad hold
    // different quantities under the same name, and each writes its pair where the record wants it
    double* r = &sh.rec[buf][seg][0];
    const double ubn = (bp.ub * qp.inv_n) * qp.inv_n;
    const double x3 = Q.lane1 ? qp.omc : qp.es;
    *reinterpret_cast<c2_d2*>(r + o1) = c2_d2{qp.a0, P[1]};
    *reinterpret_cast<c2_d2*>(r + o2) = c2_d2{P[2], qp.e};
    *reinterpret_cast<c2_d2*>(r + o3) = c2_d2{qp.q, x3};
    *reinterpret_cast<c2_d2*>(r + o4) = c2_d2{bp.ua, ubn};
   }
  };
  // Column halves: hand the first triple of a stage argument to the partner half (the other wave of this SIMD) ...
  auto col_hand = [&](const double a0, const double a1, const double a2, const int buf) {
    if constexpr (!BASE) {
      sh.xa[buf][ROLE][col][seg] = c2_d2{a0, a1};
      sh.xb[buf][ROLE][col][seg] = a2;
    }
  };
  // ... and, behind the stage's barrier, the slope of the own half.  Order of work: the loads; `sums` (needs neither them nor
  // this slope: the next argument's sum over the older slopes) in the shadow of their latency; the slope of the FIRST triple,
  // which needs no loaded value (top: a' = b; bottom: d' = 2 w J d - g); `early(out)` -- the caller forms the next argument's
  // first triple from it and hands it over at once, so that the store's latency hides behind the second triple's arithmetic
  // instead of standing in front of the next barrier -- and then the second triple.
  auto col_finish = [&](const double (&arg)[NC], double (&out)[NC], const int buf, auto&& sums, auto&& early) {
   if constexpr (!BASE) {
    const double* r = &sh.rec[buf][seg][0];
    auto pair = [&](const int slot) { return *reinterpret_cast<const c2_d2*>(r + slot); };
    // (only whole pairs as 128-bit loads: a pair with a dead half lends that half's registers to the next load's destination, and
    // the compiler then waits for everything in between)
    const c2_d2 ay = pair(0), ze1 = pair(2), l01 = pair(10);
    const double e2v = r[7], l2v = r[12];
    const c2_d2 rx = pair(ROLE == C2_TOP ? 14 : 4);            // top: (ua, ubn); bottom: (q1, es)
    c2_d2 qo;                                                   // (q2, omc): the top half needs omc only
    if constexpr (ROLE == C2_TOP) { qo.x = 0.0; qo.y = r[9]; } else qo = pair(8);
    const c2_d2 o01 = sh.xa[buf][1 - ROLE][col][seg];
    const double o2v = sh.xb[buf][1 - ROLE][col][seg];
    // (no scheduling fences: two __builtin_amdgcn_sched_barrier(0) stood here and behind early() until the end of round 4; without
    // them the sweep takes 159.0 instead of 162.6 us at 4 096 segments -- profiles/r04_sched_strategy.txt)
    sums();
    if constexpr (ROLE == C2_TOP) { out[0] = arg[3]; out[1] = arg[4]; out[2] = arg[5]; }
    else { out[0] = __builtin_fma(w2, arg[1], -arg[3]); out[1] = __builtin_fma(-w2, arg[0], -arg[4]); out[2] = -arg[5]; }
    early(out);
    DyadParts dp;
    dp.A = ay.x; dp.yy = ay.y; dp.z = ze1.x; dp.e1 = ze1.y; dp.e2 = e2v; dp.omc = qo.y;
    const double other[3] = {o01.x, o01.y, o2v};
    double dw[6];
    if constexpr (ROLE == C2_TOP) var_col12_top_dy(dp, l01.x, l01.y, l2v, rx.x, rx.y, w2, arg, other, dw);
    else var_col12_bottom_dy(dp, l01.x, l01.y, l2v, rx.x, qo.x, rx.y, w2, arg, other, dw);
    out[3] = dw[3]; out[4] = dw[4]; out[5] = dw[5];
   }
  };
  auto nothing = [] {};
  auto nothing1 = [](const double (&)[NC]) {};
  // one whole evaluation with its barrier (the first-step rule below)
  auto slope = [&](const double (&arg)[NC], double (&out)[NC], const int buf) {
    if constexpr (BASE) base_eval(arg, out, buf);
    else col_hand(arg[0], arg[1], arg[2], buf);
    C2_SYNC();
    if constexpr (!BASE) col_finish(arg, out, buf, nothing, nothing1);
  };
  // Norms over all 156 components of a segment in two levels: every wavefront first sums its own lanes of the segment -- the four
  // columns of a column wave sit in the four 16-lane rows (rows_sum: row / half swaps), the base quad by DPP -- and publishes ONE
  // pair per segment; after the barrier every lane adds the seven pairs in the same order => identical bits, identical decision
  // in every lane, no broadcast.  (Round 3: 28 entries per segment, 56 LDS reads per lane and trial step.)
  auto post2 = [&](const double p0, const double p1) {
    const double s0 = BASE ? quad_sum(p0) : rows_sum(p0), s1 = BASE ? quad_sum(p1) : rows_sum(p1);
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 v; v.x = s0; v.y = s1;
    *reinterpret_cast<d2*>(&sh.part[seg][widx][0]) = v;
  };
  auto totals2 = [&](double& T0, double& T1) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    d2 acc = *reinterpret_cast<const d2*>(&sh.part[seg][0][0]);
#pragma unroll
    for (int w = 1; w < C2_WAVES; ++w) {
      const d2 v = *reinterpret_cast<const d2*>(&sh.part[seg][w][0]);
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
  constexpr double NCOMP = 156.0;              // 12 + 144 components

  // ---- first step size.  Warm start (lto_indirect_plan_set_warm_start): the first accepted step of this segment in the plan's
  // previous STM sweep -- consecutive Newton iterations sweep nearly the same trajectory -- which saves the two extra stage
  // evaluations of Hairer's rule and the trial steps it takes to grow from a start that is a decade or two low.  a.warm is
  // launch-uniform (every wave takes the same barriers); any positive value is a valid start.
  // (Stage records and exchanged triples are double-buffered by the parity of the stage index: the loop's first stage uses
  // buffer 1, and both evaluations here are consumed before a barrier that precedes it.)
  if (a.warm) {
    const double hw = a.h_first[s];
    h_abs = (hw > 0.0) ? fmin(fmax(hw, 1e-6 * span), span) : 1e-3 * span;      // any positive value is a valid start; a tiny one would cost hundreds of trial steps
    slope(y, K[0], 0);
  } else {
    // Hairer's initial step over all components.  Row r of every column is scaled with the BASE value of row r (a dual
    // number's partials share the scale of its value), published by the base lanes.
    if (BASE) {
#pragma unroll
      for (int j = 0; j < NC; ++j) sh.scale[seg][sc0 + j] = rcp_nr(__builtin_fma(rtol, fabs(y[j]), atol));
    }
    slope(y, K[0], 0);
    double isc0[NC];
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      const double isc = sh.scale[seg][sc0 + (j < 3 ? j : j + 1)];
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
  double h_rec = 0.0;            // proposal that led to the first accepted step (what the next sweep starts from)

  // ---- trial steps.  Round 4: the error estimate of the 8(5,3) pair does not involve the FSAL slope f(y_new) (E5[12] = E3[12] = 0), so
  // a trial step's error sums are formed and published as soon as the twelfth slope K[11] is there, BEFORE the barrier of the
  // FSAL evaluation; behind that barrier every lane reads the sums and takes the step decision in the same instruction stream
  // in which the column lanes form K[12] and the base lanes (one stage ahead, as in every stage) already evaluate the first
  // stage of the NEXT trial step.  The pipeline of base and column roles therefore never drains between trial steps: twelve
  // barriers per trial step instead of thirteen, no interval in which only one side works, and the decision's dependent chain
  // (table reads, sums, three reciprocal square roots) is hidden behind the FSAL slopes.  (Round 3: after the FSAL barrier the
  // columns formed K[12] with the base idle, then a barrier of its own for the sums, then every role evaluated the decision --
  // IEEE square roots and a division -- before the base lanes could start the next step alone: 18.7 k ticks per trial step.)
  hook::RegionClock c2_loop;
  hook::Counter c2_trials;
  c2_loop.start();
  c2_life.mark(1);
  for (int trial = 0; trial < a.max_steps; ++trial) {
    c2_trials.bump();
    // every lane of a segment holds identical (t, h_abs, done): they are updated from identical data below
    double h = h_abs;
    const double h_prop = h_abs;
    double last = 0.0;
    if (t + h >= span) { h = span - t; last = 1.0; }
    // Argument st + 1 (st + 1 = 2..11: weights DP8_A[st + 1][.]; 12: the new state, weights DP8_B) is y + h (next + w K[st]) with
    // next = the sum over the slopes before the newest one, formed in the shadow of the stage's LDS traffic (same summation order
    // as the one-piece loop: bit-identical arguments).  The tableau rows come by scalar loads (rk.hpp: dp8_load_row), row st + 2
    // while stage st runs, and are consumed (dp8_pin_row) BEFORE the stage's LDS traffic is issued: scalar loads and LDS
    // operations share one counter, and the first use of a row would otherwise wait for the record stores / loads as well.
    double arg[NC], yn[NC], a5[NC], a3[NC], iscb[NC];
    double wrow[2][12], we5[13], we3[13];
    dp8_load_row<2>(tab, wrow[0]);
    {
      const double w10 = coef_here(DP8_A[1][0]);
#pragma unroll
      for (int j = 0; j < NC; ++j) arg[j] = __builtin_fma(h, w10 * K[0][j], y[j]);
    }
    if constexpr (!BASE) col_hand(arg[0], arg[1], arg[2], 1);
    int alive = 1;
    auto stage = [&](auto st_c) {                // enters with K[0] = f(y) (FSAL) and arg = argument st; leaves with K[st] and argument st + 1
      constexpr int st = decltype(st_c)::value;
      static_assert(st >= 1 && st < NS, "stage index");
      const double (&wn)[12] = wrow[(st + 1) & 1];             // weights of argument st + 1
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
        if constexpr (st == NS - 1) {           // in the shadow of the last stage: the error sums over the slopes before K[11]
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
        dp8_pin_row<st + 1>(wn);              // (arrived long ago: the wait is for the scalar loads only, in front of nothing)
        if constexpr (st == NS - 1) dp8_pin_err(we5, we3);
        sums();
#pragma unroll
        for (int j = 0; j < NC; ++j) next_arg(j);
        if constexpr (st == NS - 1) {
          // last stage: K[11] is the base lanes' before the barrier, so the new state and with it the scale of the error norm
          // (which the column lanes need right behind this barrier) are formed and published here
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
      if constexpr (st == 1) { alive = c2_wait.sync_or(!done); if (!alive) return; }
      else C2_SYNC();
      if constexpr (!BASE) {
        col_finish(arg, K[st], st & 1, sums, [&](const double (&)[NC]) {
          next_arg(0); next_arg(1); next_arg(2);
          col_hand(argn[0], argn[1], argn[2], (st + 1) & 1);
        });
        next_arg(3); next_arg(4); next_arg(5);
      }
#pragma unroll
      for (int j = 0; j < NC; ++j) arg[j] = argn[j];
    };
    stage(std::integral_constant<int, 1>{});
    if (!alive) break;                       // workgroup-uniform: the vote is the barrier's
    static_for<2, NS>(stage);
    // arg is the new state now.  Error sums of this lane's rows: into the table before the FSAL barrier
    {
      double e5 = 0.0, e3 = 0.0;
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        yn[j] = arg[j];
        double s5 = a5[j], s3 = a3[j];
        if (DP8_E5[NS - 1] != 0.0) s5 = __builtin_fma(we5[NS - 1], K[NS - 1][j], s5);
        if (DP8_E3[NS - 1] != 0.0) s3 = __builtin_fma(we3[NS - 1], K[NS - 1][j], s3);
        const double isc = BASE ? iscb[j] : sh.scale[seg][sc0 + (j < 3 ? j : j + 1)];
        s5 *= isc; s3 *= isc;
        e5 = __builtin_fma(s5, s5, e5);
        e3 = __builtin_fma(s3, s3, e3);
      }
      post2(e5, e3);
    }
    static_assert(DP8_E5[NS] == 0.0 && DP8_E3[NS] == 0.0, "the error estimate must not involve the FSAL slope");
    // FSAL slope (the column halves handed the new state's first triple over inside stage 11); behind its barrier: the decision
    if constexpr (BASE) base_eval(yn, K[NS], NS & 1);
    C2_SYNC();
    if constexpr (!BASE) col_finish(yn, K[NS], NS & 1, nothing, nothing1);
    // (K[12] is used only by an accepted step: unpinned, the compiler sinks its whole evaluation into that branch, behind the decision)
#pragma unroll
    for (int j = 0; j < NC; ++j) asm volatile("" : "+v"(K[NS][j]));
    double E5, E3;
    totals2(E5, E3);
    double accept, bad;
    dp8_decide(E5, E3, h, rejected, NCOMP, h_abs, accept, bad);
    asm volatile("" : "+v"(h_abs));          // (likewise: the next proposal would sink to the loop's tail, behind the commit branch)
    if (!done) {
      if (bad != 0.0) {                    // NaN in the step: NaN results (status_flag 2 upstream), no max_steps stall
#pragma unroll
        for (int j = 0; j < NC; ++j) y[j] = bad;
        t = span;
      } else if (accept != 0.0) {
        if (nacc == 0) h_rec = h_prop;
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
    // (the exit vote rides on the next trial step's first stage barrier)
  }
  c2_life.mark(2);
  if ((lane & 15) == 0 && (BASE ? lane == 0 : (cwave == 0 && lane == 0))) {      // probe build only: the hooks write nothing otherwise
    const int r0 = BASE ? 16 : (ROLE == C2_TOP ? 20 : 22);
    const long at = (long)blockIdx.x * C2_SEG;
    c2_wait.report(a.defect, a.ldd, r0, at);
    c2_loop.report_ticks(a.defect, a.ldd, r0 + 1, at);
    if (BASE) c2_trials.report(a.defect, a.ldd, 18, at);
  }
  // A segment that did not reach t1 (max_steps trial steps used up, or a decreasing time grid: the controller integrates
  // forward only) has no result: NaN, which the driver reports as status_flag 2 (indirect.jl:339-341).
  if (mine && (t < span || !(span >= 0.0))) {   // unfinished, decreasing grid, or a NaN span (treated like a negative one)
#pragma unroll
    for (int j = 0; j < NC; ++j) y[j] = __builtin_nan("");
  }

  if (in_range && mine) {
    if (BASE) {
      if (a.defect || a.Da) {
#pragma unroll
        for (int j = 0; j < NC; ++j) put_defect(a, grow[j], s, y[j] - arg_node(a, grow[j], node + 1));
      }
      if (q4 == 0) {
        if (a.errors) a.errors[s] = 0.0;
        if (a.nacc) a.nacc[s] = nacc;
        if (a.nrej) a.nrej[s] = nrej;
        if (a.h_first) a.h_first[s] = h_rec;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NC; ++j) put_phi(a, col, grow[j], s, y[j]);
    }
  }
  c2_life.mark(3);
  if (BASE && lane == 0) c2_life.report(a.defect, a.ldd, 24, (long)blockIdx.x * C2_SEG);     // (rows 24 .. 28 of the probe's buffer)
}

template <int PM>
__global__ __launch_bounds__(512) void k_indirect_coop2(const IndirectArgs a) {
  __shared__ C2Shared sh;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (wave == 7) { (void)__syncthreads_or(0); return; }   // shares the base wave's SIMD: votes and leaves before the stage barriers
  if (wave == 3) coop2_run<PM, C2_BASE>(a, sh, lane, 0);
  else if (wave < 3) coop2_run<PM, C2_TOP>(a, sh, lane, wave);
  else coop2_run<PM, C2_BOTTOM>(a, sh, lane, wave - 4);
}

template <int PM>
static hipError_t launch_coop2_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + C2_SEG - 1) / C2_SEG);
  hipLaunchKernelGGL((k_indirect_coop2<PM>), grid, dim3(512), 0, st, a);
  return hipGetLastError();
}

// 12-dim system, DOP853 adaptive only.
hipError_t launch_indirect_stm_coop2(int pm, const IndirectArgs& a0, hipStream_t st) {
  if (a0.S <= 0) return hipSuccess;
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_coop2_one<PM_P0>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_coop2_one<PM_P1>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_coop2_one<PM_P2>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_coop2_one<PM_PGEN>(a, st);
  return e;
}

}  // namespace lto
