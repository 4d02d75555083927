// kernels_indirect_pipe8.hip -- the three-role software pipeline for the fixed-step RK4 STM sweep (BASELINE configs[1]); replaces
// the serial loop of jacobianCalc (src/multiShoot_CRTBP_indirect.jl:93-146).
//
// At 4 096 segments the chip offers 16 lanes per segment (1 024 SIMDs x 64 lanes) and a sweep lasts as long as the longest
// instruction stream of a workgroup.  A workgroup owns 16 segments and runs three roles in different wavefronts, skewed by one RK4
// STEP each, so that they execute concurrently:
//
//   base      integrates the ND-dim base state and publishes, per stage, the part of the stage argument the coefficients depend
//             on (r, lambda_v [, m]); round 3: two lanes of a segment's quad evaluate stages 1 | 2 and then 3 | 4 side by side
//             (pipe8_role_base_paired)
//   coef      one step behind: lane = (segment, RK stage); builds G, H, U (+ mass couplings) at those arguments
//   columns   two steps behind: c' = F(t) c with the coefficients of the stage -- no gravity, no control law, no base state in
//             these lanes; lane = (segment, ONE column), a DPP row = one segment, coefficient x entry = v_fmac_f64_dpp row_newbcast
//
// Four column waves do not fit three SIMDs evenly (a wavefront cannot be split and cannot move), so a phase is TWO RK4 steps (one
// workgroup barrier per phase) and the columns of segments 12..15 are advanced by two wavefronts in turn:
//
//   wave  SIMD  role                                         instructions per phase (two steps), ND = 14
//   w0    A     columns of segments 0..3,  both steps        680
//   w4    A     columns of segments 12..15, EVEN step        340   -> hands the 14 doubles per lane to w5 through LDS
//   w1    B     columns of segments 4..7,  both steps        680
//   w5    B     columns of segments 12..15, ODD step         340   <- waits for w4's flag
//   w2    C     base trajectory, both steps                  ~860 (paired stages; round 2: ~1100)
//   w6    C     exits at once
//   w3    D     coefficients of both steps                   ~360
//   w7    D     columns of segments 8..11, both steps        680
//
// (hardware places wave i and wave i + 4 of a workgroup on the same SIMD: tools/micro/sync_probe.hip).  w4 runs with
// raised priority so that its step finishes early in the phase and w5 can follow.  A SIMD works through its wavefronts' streams
// essentially one after the other at the issue rate of a lone wavefront (tools/micro/prio_probe.hip), so what bounds the sweep is
// (sum of the workgroup's streams) / 4: DESIGN.md section 6.
//
// Skew: ONE RK4 step per hand-over although the barrier comes every two steps.  In
// phase p the base wave integrates steps 2p, 2p + 1; the coefficient wave builds steps 2p - 1 (published before the last
// barrier) and 2p (published during this phase: it waits for the base wave's counter); the column waves advance steps
// 2p - 2 (coefficients complete since the last barrier) and 2p - 1 (built during this phase: they wait for the coefficient
// wave's counter, which by then is long set).  Both hand-overs are rings of four steps in LDS (91 KB in all: one
// workgroup per CU); steps / 2 + 1 phases per sweep.  Waiting: a wave only ever waits (bounded poll of one LDS word) for a
// wave that, in this phase, waits for nothing that depends on the waiter -- base: never; coefficients: base; w4: nothing;
// columns: coefficients' first pass; w5: w4 -- so every wave reaches every barrier.  A poll that runs out (it cannot
// while the producer runs) raises a workgroup flag that turns the workgroup's outputs into NaN instead of hanging.
//
// For the always-thrust-limited control laws (p = 0, p = 1) of the 14-dim system nothing depends on lambda_m: the base
// wave integrates 13 components, the coefficient wave -- which evaluates lambda_m_dot at every stage argument anyway --
// accumulates lambda_m off the critical stream, and the STM column d/d lambda_m(t0) is the unit vector, so its lane stays
// switched off (idle lanes let the device raise its clocks faster in short bursts; in steady state they cost nothing).
#include "pipe_common.hpp"

namespace lto {

constexpr int P8_SPIN_LIMIT = 1 << 22;   // polls before a waiting wave gives up (never hangs)

// Probe hooks (pipe_hooks.hpp: no-ops in the product build): ticks each wave waits at the phase barriers -> row 16 of the probe
// script's defect buffer, column 16 block + wave.
#define P8_SYNC() p8_wait.sync()
#define P8_WAIT_DECL hook::BarrierWait p8_wait
#define P8_WAIT_REPORT(a) do { if ((threadIdx.x & 63) == 0) p8_wait.report((a).defect, (a).ldd, 16, blockIdx.x * PIPE_SEG + (threadIdx.x >> 6)); } while (0)

template <int ND, int PM> struct Pipe8 {
  using Arg = PipeArg<ND, PM>;
  static constexpr int NI = Arg::N;
  static constexpr int NC = sizeof(typename PipeCoef<ND>::type) / sizeof(double);
  static constexpr int SD = CoefBySegment::stage_doubles<NC>();        // doubles per (step, stage) coefficient slab
  static constexpr bool LM_OFF = (ND == 14) && !Arg::LM;              // lambda_m integrated by the coefficient wave
  static constexpr int NB = LM_OFF ? ND - 1 : ND;                     // components the base wave integrates
  static constexpr int NA = LM_OFF ? ND - 1 : ND;                     // STM columns a row integrates (the rest: unit vectors)
  // Stage arguments travel as 16-byte pairs (round 4): ring [step & 3][stage][pair][segment][2], values in the order
  //   (r0 r1) (r2 l0) (l1 l2) [(m lambda_m)]      l = lambda_v
  // so that a stage's position and lambda_v are THREE 128-bit LDS stores on the base wave's stream instead of six 64-bit ones (an LDS
  // store costs the issuing wave 22.5 ticks, a 128-bit one 31: tools/micro/lds_probe.hip) and the coefficient wave reads them
  // back with three or four loads instead of six to eight.  Consecutive segments are 16 bytes apart: conflict-free both ways.
  static constexpr int NPAIR = (ND == 14) ? 4 : 3;
  static constexpr int SLABD = NPAIR * PIPE_SEG * 2;                  // doubles of one (step, stage) slab
  static constexpr int INT_DOUBLES = 4 * 4 * SLABD;
  // position in the slab's linear order of published value e (PipeArg::idx order: r0 r1 r2 [m] l0 l1 l2 [lambda_m])
  __host__ __device__ static constexpr int lin(int e) { return (ND == 12) ? e : (e < 3 ? e : e == 3 ? 6 : e < 7 ? e - 1 : 7); }
  __host__ __device__ static constexpr int at(int e, int seg) { return ((lin(e) >> 1) * PIPE_SEG + seg) * 2 + (lin(e) & 1); }
  static constexpr int COEF_DOUBLES = 4 * 4 * SD;                     // ring [step & 3][stage][record]
  static constexpr int HAND_DOUBLES = ND * 64;                        // [component][lane] of the alternating column job
};

typedef double p8_d2 __attribute__((ext_vector_type(2)));

struct Pipe8Flags {
  int base_steps;    // steps whose stage arguments the base wave has published
  int coef_steps;    // steps whose coefficients are complete
  int hand;          // phases in which w4 has handed its state to w5
  int fail;          // a poll ran out
};

// wait until *flag >= want (flags only grow); wave-uniform
__device__ __forceinline__ void p8_wait_for(int* flag, const int want, int* fail) {
  int spins = 0;
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
    if (++spins >= P8_SPIN_LIMIT) { __hip_atomic_store(fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); break; }
    __builtin_amdgcn_s_sleep(1);
  }
}
// The LDS operations of one wavefront execute in order, so a plain store issued after the data stores is a release for a
// reader that acquires; the empty asm keeps the compiler from moving the data stores below it.  (A release store proper
// makes the wave wait for its outstanding LDS stores first: ~90 cycles per phase on the base wave's chain.)
__device__ __forceinline__ void p8_signal(int* flag, const int value) {
  asm volatile("" ::: "memory");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}

// ------------------------------------------------- base role, one stage after the other (ND = 14 with lambda_m on the chain only)
// Row g of the wave keeps the argument of stage g in registers; ONE set of stores per step publishes all four stages; two steps per
// phase.  The unclamped p > 1 laws of the 14-dim system depend on the stage's lambda_m and mass, so their stages do not pair: every
// other instantiation uses pipe8_role_base_paired below.  Tried and dropped: rows 1..3 switched off and the stage arguments stored stage by
// stage from row 0 -- the three redundant rows cost ~5 us while the device is still raising its clocks (nothing in steady state), but twelve more
// LDS stores per step on the chain cost more (S = 29: 79 -> 84 us).
template <int ND, int PM>
__device__ __forceinline__ void pipe8_role_base(const IndirectArgs& a, const PipeLane& L, const int seg, const int slot,
                                                double* s_int, Pipe8Flags* fl) {
  using P = Pipe8<ND, PM>;
  constexpr int NI = P::NI, NB = P::NB;
  const int steps = a.steps, npairs = (steps + 1) >> 1;
  const double h = L.h, h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0);
  auto rhs = [&](const double (&y)[ND], double (&k)[ND]) {
    if constexpr (ND == 12) rhs12_base<PM>(y, L.tp, k);
    else rhs14_base<PM, !P::LM_OFF>(y, L.tp, k);
  };
  double y[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + L.node];
  hook::RegionClock loop_clock;
  loop_clock.start();
  P8_WAIT_DECL;
  for (int p = 0; p < npairs + 1; ++p) {
    if (p < npairs && PIPE_ROLE_ON(a, 1) && (PIPE_ROLE_ON(a, 16) || slot == 0)) {
      for (int j = 0; j < 2; ++j) {
        const int step = 2 * p + j;
        if (step >= steps) break;
        double k[ND], yt[ND], acc[ND], keep[NI];
#pragma unroll
        for (int c = 0; c < ND; ++c) { yt[c] = y[c]; acc[c] = y[c]; }
        auto remember = [&](int stage, const double (&arg)[ND]) {
          if (slot == stage) {
#pragma unroll
            for (int e = 0; e < NI; ++e) keep[e] = arg[P::Arg::idx[e]];
          }
        };
        remember(0, y);
        rhs(y, k);
#pragma unroll
        for (int c = 0; c < NB; ++c) { acc[c] = __builtin_fma(h6, k[c], y[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
        remember(1, yt);
        rhs(yt, k);
#pragma unroll
        for (int c = 0; c < NB; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h2, k[c], y[c]); }
        remember(2, yt);
        rhs(yt, k);
#pragma unroll
        for (int c = 0; c < NB; ++c) { acc[c] = __builtin_fma(h3, k[c], acc[c]); yt[c] = __builtin_fma(h, k[c], y[c]); }
        remember(3, yt);
        rhs(yt, k);
#pragma unroll
        for (int c = 0; c < NB; ++c) y[c] = __builtin_fma(h6, k[c], acc[c]);
        double* dst = s_int + ((step & 3) * 4 + slot) * P::SLABD;
#pragma unroll
        for (int e = 0; e < NI; ++e) dst[P::at(e, seg)] = keep[e];
        if (j == 0) p8_signal(&fl->base_steps, step + 1);   // the phase's first step: the coefficient wave is waiting for it
      }
    }
    P8_SYNC();
  }
  P8_WAIT_REPORT(a);
  if (L.in_range && slot == 0) {
    const bool fail = fl->fail != 0;
    if (a.defect) {
#pragma unroll
      for (int c = 0; c < NB; ++c) a.defect[c * a.ldd + L.s] = fail ? __builtin_nan("") : y[c] - a.X[c * a.ldx + L.node + 1];
    }
    if (a.errors) a.errors[L.s] = 0.0;
    if (a.nacc) a.nacc[L.s] = steps;
    if (a.nrej) a.nrej[L.s] = 0;
    if (seg == 0) loop_clock.report(a.defect, a.ldd, 17, 18, L.s);     // probe build: ticks of the phase loop, per workgroup
  }
}

// ------------------------------------------------------------------------------------- base role, paired stages (round 3)
// The chain of the role above is one RK4 stage after the other: ~135 instructions each, of which ~85 are the "expensive"
// part -- three reciprocal square roots, the logistic, the gravity coefficients -- and all four 16-lane rows of the wave
// compute the same thing.  But in RK4 on this system the position and lambda_v arguments of stage 2 do not depend on the
// expensive part of stage 1 (r_2 = r + h/2 v, lambda_v,2 = lambda_v + h/2 (2 w J lambda_v - lambda_r)), nor stage 4's on
// stage 3's (r_4 = r + h Y3_v, Y3_v from stage 2).  So a segment is four neighbouring lanes (a DPP quad), lanes 0 / 2 ("A")
// evaluate the expensive part of stages 1 and 3, lanes 1 / 3 ("B") of stages 2 and 4, at the same time with the same
// instructions: TWO expensive evaluations per step on the chain instead of four.  Each lane turns its evaluation into the
// pieces of its stage's slope that depend on it (base_stage_own: three components of v_dot without the Coriolis term, three
// of lambda_r_dot; for ND = 14 the thrust is kept apart as two scalars because it needs the stage's mass, which depends on
// the other lane's stage), the pair exchanges them by v_mov_b32_dpp quad_perm (two per double and direction; 64-bit DPP has
// row_newbcast only), and everything that is cheap -- Coriolis and identity rows, the stage arguments, the RK4 sums -- is
// computed by every lane of the quad redundantly, so after the exchange all four lanes hold the same bits again.
// Each lane publishes the argument of its own stage before evaluating it (the old role's `remember` moves are gone); the
// stage masses of ND = 14 follow once they exist.  Lanes 2 / 3 of a quad hold the same state as lanes 0 / 1 (they publish to the
// same addresses the same values) and take the second gravitating body off them: inside an evaluation a lane forms the
// inverse-distance powers of ONE primary and swaps c_b, t_b with the lane of the other (base_stage_own; one reciprocal square
// root and 13 instructions fewer per evaluation, four more DPP moves).  ND = 14 with lambda_m on the chain (unclamped p > 1 laws: the law itself depends on the stage's lambda_m and
// mass) keeps the one-stage-at-a-time role above.
template <int CTRL>
__device__ __forceinline__ double quad_from(const double x) {      // CTRL = quad_perm code
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// Lane q of a segment's quad: stage slot q & 1 (A: stages 1, 3; B: stages 2, 4), gravitating body q >> 1.  Everything that leaves
// an evaluation comes from lanes 0 (A) and 1 (B), the lanes of primary 1, whose operand order is that of rhs12_base / rhs14_base.
__device__ __forceinline__ double from_a(const double x) { return quad_from<0x00>(x); }       // quad_perm:[0,0,0,0]
__device__ __forceinline__ double from_b(const double x) { return quad_from<0x55>(x); }       // quad_perm:[1,1,1,1]
__device__ __forceinline__ double other_body(const double x) { return quad_from<0x4E>(x); }   // quad_perm:[2,3,0,1]

template <int ND, int PM>
__device__ __forceinline__ void pipe8_role_base_paired(const IndirectArgs& a, const PipeLane& L, const int seg, const int q,
                                                       double* s_int, Pipe8Flags* fl) {
  using P = Pipe8<ND, PM>;
  static_assert(ND == 12 || P::LM_OFF, "lambda_m on the chain: the stages do not pair");
  constexpr int NB = P::NB;
  constexpr bool M14 = (ND == 14);
  constexpr int V = 3, MI = 6, LR = M14 ? 7 : 6, LV = LR + 3;      // first row of v, mass row, first rows of lambda_r, lambda_v
  constexpr int SLAB = P::SLABD;
  const int steps = a.steps;
  const double h = L.h, h2 = 0.5 * h, h6 = h * (1.0 / 6.0), h3 = h * (1.0 / 3.0), w2 = L.w2;
  const bool is_b = (q & 1) != 0;
  const bool body2 = (q & 2) != 0;                   // this lane's gravitating body: primary 1 (x + MU) or primary 2 (x + MU - 1)
  const double body_off = body2 ? L.tp.MU - 1.0 : L.tp.MU, body_kap = body2 ? L.tp.MU : 1.0 - L.tp.MU, body_sgn = body2 ? 1.0 : -1.0;
  auto swap_body = [](const double v) { return other_body(v); };
  const double gA = is_b ? h2 : 0.0;                 // round 1, own stage argument (rows r, lambda_v): y + gA k1
  const double al = is_b ? 0.0 : h2, be = is_b ? h : 0.0;   // round 2: y + al k2 + be k3
  const int own = is_b ? SLAB : 0;                   // own stage's slab relative to the round's first
  const double kt = L.tp.kappa_td;
  double y[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + L.node];
  double inv_m = 0.0;
  if constexpr (M14) inv_m = rcp_nr(y[MI]);
  hook::RegionClock loop_clock;
  loop_clock.start();
  P8_WAIT_DECL;
  // one barrier per phase of two steps (after every odd step and after the last one), then the drain phase: npairs + 1 in all
  for (int step = 0; step < steps; ++step) {
    if (PIPE_ROLE_ON(a, 1)) {
      {
        double* slab = s_int + ((step & 3) * 4) * SLAB + 2 * seg;     // stage s of this step: slab + s * SLAB; pair q of it at + q * 2 PIPE_SEG
        auto publish = [&](double* d, const double (&pr)[3], const double (&pl)[3]) {
          *reinterpret_cast<p8_d2*>(d) = p8_d2{pr[0], pr[1]};
          *reinterpret_cast<p8_d2*>(d + 2 * PIPE_SEG) = p8_d2{pr[2], pl[0]};
          *reinterpret_cast<p8_d2*>(d + 4 * PIPE_SEG) = p8_d2{pl[1], pl[2]};
        };
        StageOwn o;
        double pr[3], pl[3];
        // ------------------------------------------------------------ round 1: stages 1 (A lanes) and 2 (B lanes)
        const double kl1[3] = {__builtin_fma(w2, y[LV + 1], -y[LR]), __builtin_fma(-w2, y[LV], -y[LR + 1]), -y[LR + 2]};
#pragma unroll
        for (int i = 0; i < 3; ++i) { pr[i] = __builtin_fma(gA, y[V + i], y[i]); pl[i] = __builtin_fma(gA, kl1[i], y[LV + i]); }
        publish(slab + own, pr, pl);
        base_stage_own<ND, PM>(pr[0], pr[1], pr[2], pl[0], pl[1], pl[2], L.tp, body_off, body_kap, body_sgn, swap_body, o);
        double a1[3], a2[3], g1[3], g2[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { a1[i] = from_a(o.av[i]); a2[i] = from_b(o.av[i]); g1[i] = from_a(o.gl[i]); g2[i] = from_b(o.gl[i]); }
        double kv1[3], kv2[3], km1 = 0.0, km2 = 0.0, y2m = 0.0, y3m = 0.0, im2 = 0.0, im3 = 0.0, gf2 = 0.0;
        if constexpr (M14) {
          const double u1 = from_a(o.gf) * inv_m;
          gf2 = from_b(o.gf);
          km1 = -kt * from_a(o.sc); km2 = -kt * from_b(o.sc);
          // the masses of stages 2 and 3 need only these two mass rates: both reciprocals from ONE (1/a = b/(ab), 1/b = a/(ab))
          y2m = __builtin_fma(h2, km1, y[MI]); y3m = __builtin_fma(h2, km2, y[MI]);
          const double r23 = rcp_nr(y2m * y3m);
          im2 = y3m * r23; im3 = y2m * r23;
          kv1[0] = __builtin_fma(w2, y[V + 1], __builtin_fma(-u1, y[LV], a1[0]));
          kv1[1] = __builtin_fma(-w2, y[V], __builtin_fma(-u1, y[LV + 1], a1[1]));
          kv1[2] = __builtin_fma(-u1, y[LV + 2], a1[2]);
        } else {
          kv1[0] = __builtin_fma(w2, y[V + 1], a1[0]); kv1[1] = __builtin_fma(-w2, y[V], a1[1]); kv1[2] = a1[2];
        }
        double y2v[3], y2g[3], y2l[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          y2v[i] = __builtin_fma(h2, kv1[i], y[V + i]); y2g[i] = __builtin_fma(h2, g1[i], y[LR + i]); y2l[i] = __builtin_fma(h2, kl1[i], y[LV + i]);
        }
        if constexpr (M14) {
          slab[6 * PIPE_SEG] = y[MI]; slab[SLAB + 6 * PIPE_SEG] = y2m;       // pair 3, first half
          const double u2 = gf2 * im2;
          kv2[0] = __builtin_fma(w2, y2v[1], __builtin_fma(-u2, y2l[0], a2[0]));
          kv2[1] = __builtin_fma(-w2, y2v[0], __builtin_fma(-u2, y2l[1], a2[1]));
          kv2[2] = __builtin_fma(-u2, y2l[2], a2[2]);
        } else {
          kv2[0] = __builtin_fma(w2, y2v[1], a2[0]); kv2[1] = __builtin_fma(-w2, y2v[0], a2[1]); kv2[2] = a2[2];
        }
        const double kl2[3] = {__builtin_fma(w2, y2l[1], -y2g[0]), __builtin_fma(-w2, y2l[0], -y2g[1]), -y2g[2]};
        double y3v[3], y3g[3], y3l[3], ar[3], avv[3], ag[3], alv[3], am = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          y3v[i] = __builtin_fma(h2, kv2[i], y[V + i]); y3g[i] = __builtin_fma(h2, g2[i], y[LR + i]); y3l[i] = __builtin_fma(h2, kl2[i], y[LV + i]);
          ar[i] = __builtin_fma(h3, y2v[i], __builtin_fma(h6, y[V + i], y[i]));
          avv[i] = __builtin_fma(h3, kv2[i], __builtin_fma(h6, kv1[i], y[V + i]));
          ag[i] = __builtin_fma(h3, g2[i], __builtin_fma(h6, g1[i], y[LR + i]));
          alv[i] = __builtin_fma(h3, kl2[i], __builtin_fma(h6, kl1[i], y[LV + i]));
        }
        if constexpr (M14) am = __builtin_fma(h3, km2, __builtin_fma(h6, km1, y[MI]));
        // ------------------------------------------------------------ round 2: stages 3 (A lanes) and 4 (B lanes)
        const double kl3[3] = {__builtin_fma(w2, y3l[1], -y3g[0]), __builtin_fma(-w2, y3l[0], -y3g[1]), -y3g[2]};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          pr[i] = __builtin_fma(be, y3v[i], __builtin_fma(al, y2v[i], y[i]));        // A: r + h/2 k2_r, B: r + h k3_r
          pl[i] = __builtin_fma(be, kl3[i], __builtin_fma(al, kl2[i], y[LV + i]));
        }
        publish(slab + 2 * SLAB + own, pr, pl);
        base_stage_own<ND, PM>(pr[0], pr[1], pr[2], pl[0], pl[1], pl[2], L.tp, body_off, body_kap, body_sgn, swap_body, o);
        double a3[3], a4[3], g3[3], g4[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { a3[i] = from_a(o.av[i]); a4[i] = from_b(o.av[i]); g3[i] = from_a(o.gl[i]); g4[i] = from_b(o.gl[i]); }
        double kv3[3], kv4[3], km3 = 0.0, km4 = 0.0, y4m = 0.0, mnew = 0.0, im4 = 0.0, gf4 = 0.0;
        if constexpr (M14) {
          const double u3 = from_a(o.gf) * im3;
          gf4 = from_b(o.gf);
          km3 = -kt * from_a(o.sc); km4 = -kt * from_b(o.sc);
          // stage 4's mass and the next step's: again both reciprocals from one
          y4m = __builtin_fma(h, km3, y[MI]);
          mnew = __builtin_fma(h6, km4, __builtin_fma(h3, km3, am));
          const double r4n = rcp_nr(y4m * mnew);
          im4 = mnew * r4n; inv_m = y4m * r4n;
          kv3[0] = __builtin_fma(w2, y3v[1], __builtin_fma(-u3, y3l[0], a3[0]));
          kv3[1] = __builtin_fma(-w2, y3v[0], __builtin_fma(-u3, y3l[1], a3[1]));
          kv3[2] = __builtin_fma(-u3, y3l[2], a3[2]);
        } else {
          kv3[0] = __builtin_fma(w2, y3v[1], a3[0]); kv3[1] = __builtin_fma(-w2, y3v[0], a3[1]); kv3[2] = a3[2];
        }
        double y4v[3], y4g[3], y4l[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          y4v[i] = __builtin_fma(h, kv3[i], y[V + i]); y4g[i] = __builtin_fma(h, g3[i], y[LR + i]); y4l[i] = __builtin_fma(h, kl3[i], y[LV + i]);
        }
        if constexpr (M14) {
          slab[2 * SLAB + 6 * PIPE_SEG] = y3m; slab[3 * SLAB + 6 * PIPE_SEG] = y4m;
          const double u4 = gf4 * im4;
          kv4[0] = __builtin_fma(w2, y4v[1], __builtin_fma(-u4, y4l[0], a4[0]));
          kv4[1] = __builtin_fma(-w2, y4v[0], __builtin_fma(-u4, y4l[1], a4[1]));
          kv4[2] = __builtin_fma(-u4, y4l[2], a4[2]);
        } else {
          kv4[0] = __builtin_fma(w2, y4v[1], a4[0]); kv4[1] = __builtin_fma(-w2, y4v[0], a4[1]); kv4[2] = a4[2];
        }
        const double kl4[3] = {__builtin_fma(w2, y4l[1], -y4g[0]), __builtin_fma(-w2, y4l[0], -y4g[1]), -y4g[2]};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          y[i] = __builtin_fma(h6, y4v[i], __builtin_fma(h3, y3v[i], ar[i]));
          y[V + i] = __builtin_fma(h6, kv4[i], __builtin_fma(h3, kv3[i], avv[i]));
          y[LR + i] = __builtin_fma(h6, g4[i], __builtin_fma(h3, g3[i], ag[i]));
          y[LV + i] = __builtin_fma(h6, kl4[i], __builtin_fma(h3, kl3[i], alv[i]));
        }
        if constexpr (M14) y[MI] = mnew;
        if (!(step & 1)) p8_signal(&fl->base_steps, step + 1);   // the phase's first step: the coefficient wave is waiting for it
      }
    }
    if ((step & 1) || step == steps - 1) P8_SYNC();
  }
  P8_SYNC();
  P8_WAIT_REPORT(a);
  if (L.in_range && q == 0) {
    const bool fail = fl->fail != 0;
    if (a.defect) {
#pragma unroll
      for (int c = 0; c < NB; ++c) a.defect[c * a.ldd + L.s] = fail ? __builtin_nan("") : y[c] - a.X[c * a.ldx + L.node + 1];
    }
    if (a.errors) a.errors[L.s] = 0.0;
    if (a.nacc) a.nacc[L.s] = steps;
    if (a.nrej) a.nrej[L.s] = 0;
    if (seg == 0) loop_clock.report(a.defect, a.ldd, 17, 18, L.s);     // probe build: ticks of the step loop, per workgroup
  }
}

// --------------------------------------------------------------------------------------------------- coefficient role
// lane = (segment, RK stage); phase p: step 2p - 1, then step 2p as soon as the base wave has published it.
template <int ND, int PM>
__device__ __forceinline__ void pipe8_role_coef(const IndirectArgs& a, const PipeLane& L, const int seg, const int stage,
                                                const double* s_int, double* s_coef, double* s_lm, Pipe8Flags* fl) {
  using P = Pipe8<ND, PM>;
  using Coef = typename PipeCoef<ND>::type;
  constexpr int NI = P::NI, NC = P::NC, SD = P::SD;
  const int steps = a.steps, npairs = (steps + 1) >> 1;
  // every coefficient except the unit vector lhat (entries 14..16) is stored times the stage's RK4 argument weight
  // (the last stage carries h/2 = 3 h/6: the column lanes advance 3 y per step, col_dpp_step)
  const double as = (stage == 2) ? L.h : 0.5 * L.h;
  const double bw = (stage == 0 || stage == 3) ? L.h * (1.0 / 6.0) : L.h * (1.0 / 3.0);   // RK4 weight of the stage's slope
  double lm_acc = 0.0;
  auto build = [&](const int step) {
    const int slab = (step & 3) * 4 + stage;
    double arg[ND], dead[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) arg[c] = 0.0;
    const p8_d2* src = reinterpret_cast<const p8_d2*>(s_int + slab * P::SLABD) + seg;
    double lv[2 * P::NPAIR];
#pragma unroll
    for (int q = 0; q < P::NPAIR; ++q) { const p8_d2 v = src[q * PIPE_SEG]; lv[2 * q] = v.x; lv[2 * q + 1] = v.y; }
#pragma unroll
    for (int e = 0; e < NI; ++e) arg[P::Arg::idx[e]] = lv[P::lin(e)];
    Coef vc;
    if constexpr (ND == 12) rhs12<PM, true>(arg, L.tp, dead, vc);
    else rhs14<PM, true>(arg, L.tp, dead, vc);
    if constexpr (P::LM_OFF) lm_acc = __builtin_fma(bw, dead[ND - 1], lm_acc);   // lambda_m_dot = -umag n / m at this stage
    const double* o = reinterpret_cast<const double*>(&vc);
    double* dst = s_coef + slab * SD;
    constexpr int NST = P::LM_OFF ? NC - 1 : NC;     // d lambda_m_dot / d lambda_m = 0 for these laws: never read
#pragma unroll
    for (int e = 0; e < NST; ++e) dst[CoefBySegment::at<P::NA>(e, seg)] = (e < 14 || e > 16) ? o[e] * as : o[e];
  };
  P8_WAIT_DECL;
  for (int p = 0; p < npairs + 1; ++p) {
    if (PIPE_ROLE_ON(a, 2)) {
      if (p >= 1 && 2 * p - 1 < steps) {
        build(2 * p - 1);
        p8_signal(&fl->coef_steps, 2 * p);
      }
      if (2 * p < steps) {
        if (PIPE_ROLE_ON(a, 1)) p8_wait_for(&fl->base_steps, 2 * p + 1, &fl->fail);
        build(2 * p);
      }
    }
    P8_SYNC();
  }
  P8_WAIT_REPORT(a);
  if constexpr (P::LM_OFF) {
    // lambda_m(t1) = lambda_m(t0) + sum over steps and stages of b_s h k_s: the four stage rows meet through LDS (same
    // wavefront: its LDS operations complete in order)
    s_lm[stage * PIPE_SEG + seg] = lm_acc;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
    if (L.in_range && stage == 0 && a.defect) {
      const double sum = (s_lm[seg] + s_lm[PIPE_SEG + seg]) + (s_lm[2 * PIPE_SEG + seg] + s_lm[3 * PIPE_SEG + seg]);
      const long r = (long)(ND - 1) * a.ldx + L.node;
      a.defect[(ND - 1) * a.ldd + L.s] = fl->fail ? __builtin_nan("") : (a.X[r] + sum) - a.X[r + 1];
    }
  }
}

// STM column `col` of the lane's segment to global memory: y carries 3^k Phi (stm_scale undoes it); columns the row does
// not integrate (col >= NA) are unit vectors.
template <int ND, int NA>
__device__ __forceinline__ void pipe8_store_column(const IndirectArgs& a, const PipeLane& L, const int col, const double (&y)[ND],
                                                   const bool fail) {
  if (L.in_range && col < ND) {
    const double sc = (col < NA) ? a.stm_scale : 1.0;
    const double poison = (col < NA) ? 0.0 : L.h - L.h;     // a unit column of a segment with a NaN (or infinite) span is NaN like the rest
#pragma unroll
    for (int r = 0; r < ND; ++r) a.Phi[(long)(col * ND + r) * a.ldp + L.s] = fail ? __builtin_nan("") : __builtin_fma(y[r], sc, poison);
  }
}

// ------------------------------------------------------------------------------ column role, both steps of every phase
template <int ND, int PM>
__device__ __forceinline__ void pipe8_role_columns(const IndirectArgs& a, const PipeLane& L, const int seg, const int col,
                                                   const double* s_coef, Pipe8Flags* fl, const int probe_bit) {
  using P = Pipe8<ND, PM>;
  constexpr int SD = P::SD;
  const int steps = a.steps, npairs = (steps + 1) >> 1;
  const ColStepConst k(L.h, L.w2);
  const double* rec = s_coef + CoefBySegment::lane_base(col, seg);
  double y[ND];
#pragma unroll
  for (int r = 0; r < ND; ++r) y[r] = (r == col) ? 1.0 : 0.0;
  P8_WAIT_DECL;
  for (int p = 0; p < npairs + 1; ++p) {
    if (p >= 1 && PIPE_ROLE_ON(a, 4) && PIPE_ROLE_ON(a, probe_bit)) {
      const int s0 = 2 * p - 2, s1 = 2 * p - 1;
      if (col < P::NA) col_dpp_step<ND, SD, P::Arg::LM, P::NA>(rec + ((s0 & 3) * 4) * SD, k, s0, y);   // spare lanes stay off: never DPP sources
      if (s1 < steps) {
        if (PIPE_ROLE_ON(a, 2)) p8_wait_for(&fl->coef_steps, s1 + 1, &fl->fail);
        if (col < P::NA) col_dpp_step<ND, SD, P::Arg::LM, P::NA>(rec + ((s1 & 3) * 4) * SD, k, s1, y);
      }
    }
    P8_SYNC();
  }
  P8_WAIT_REPORT(a);
  pipe8_store_column<ND, P::NA>(a, L, col, y, fl->fail != 0);
}

// --------------------------------------------------------------- column role of the alternating job (segments 12..15)
// ODD = false (w4): step 2p - 2 in phase p, then state -> s_hand, hand = p.  ODD = true (w5): waits for the coefficients
// of step 2p - 1 and for hand >= p, state <- s_hand, step 2p - 1, state -> s_hand.  The barrier at the end of the phase
// orders w5's stores before w4's loads of the next phase.  After the last phase w4 stores the STM columns from s_hand.
template <int ND, int PM, bool ODD>
__device__ __forceinline__ void pipe8_role_columns_alt(const IndirectArgs& a, const PipeLane& L, const int seg, const int col,
                                                       const double* s_coef, double* s_hand, Pipe8Flags* fl) {
  using P = Pipe8<ND, PM>;
  constexpr int SD = P::SD;
  const int steps = a.steps, npairs = (steps + 1) >> 1;
  const int lane = threadIdx.x & 63;
  const ColStepConst k(L.h, L.w2);
  const double* rec = s_coef + CoefBySegment::lane_base(col, seg);
  const bool on = col < P::NA;
  double y[ND];
#pragma unroll
  for (int r = 0; r < ND; ++r) y[r] = (r == col) ? 1.0 : 0.0;
  // the state travels as 16-byte pairs (rows 2q, 2q + 1 of a lane side by side): seven 128-bit LDS instructions each way instead
  // of fourteen 64-bit ones (round 4, tools/micro/lds_probe.hip: a 128-bit store costs 31 ticks of issue against 2 x 22.5, a load 18
  // against 2 x 17.5) -- the hand-over is 10 % of a step on the two SIMDs that share this job
  static_assert(ND % 2 == 0, "pairs of rows");
  p8_d2* hand2 = reinterpret_cast<p8_d2*>(s_hand);
  auto load = [&]() {
    if (on) {
#pragma unroll
      for (int q = 0; q < ND / 2; ++q) { const p8_d2 v = hand2[q * 64 + lane]; y[2 * q] = v.x; y[2 * q + 1] = v.y; }
    }
  };
  auto store = [&]() {
    if (on) {
#pragma unroll
      for (int q = 0; q < ND / 2; ++q) hand2[q * 64 + lane] = p8_d2{y[2 * q], y[2 * q + 1]};
    }
  };
  if (!ODD) __builtin_amdgcn_s_setprio(2);
  P8_WAIT_DECL;
  for (int p = 0; p < npairs + 1; ++p) {
    if (p >= 1 && PIPE_ROLE_ON(a, 4) && PIPE_ROLE_ON(a, 32)) {
      const int step = 2 * p - 2 + (ODD ? 1 : 0);
      if (!ODD) {
        if (p > 1) load();
        if (on) col_dpp_step<ND, SD, P::Arg::LM, P::NA>(rec + ((step & 3) * 4) * SD, k, step, y);
        store();
        p8_signal(&fl->hand, p);
      } else if (step < steps) {
        if (PIPE_ROLE_ON(a, 2)) p8_wait_for(&fl->coef_steps, step + 1, &fl->fail);
        p8_wait_for(&fl->hand, p, &fl->fail);
        load();
        if (on) col_dpp_step<ND, SD, P::Arg::LM, P::NA>(rec + ((step & 3) * 4) * SD, k, step, y);
        store();
      }
    }
    P8_SYNC();
  }
  P8_WAIT_REPORT(a);
  if (!ODD) {
    load();
    pipe8_store_column<ND, P::NA>(a, L, col, y, fl->fail != 0);
  }
}

// ------------------------------------------------------------------------------------------------------------- kernel
template <int ND, int PM>
__global__ __launch_bounds__(512) void k_indirect_pipe8(const IndirectArgs a) {
  using P = Pipe8<ND, PM>;
  __shared__ __attribute__((aligned(16))) double s_int[P::INT_DOUBLES];
  __shared__ double s_coef[P::COEF_DOUBLES];
  __shared__ __attribute__((aligned(16))) double s_hand[P::HAND_DOUBLES];
  __shared__ double s_lm[4 * PIPE_SEG];
  __shared__ Pipe8Flags s_fl;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // waves 0, 1, 7: column waves 0, 1, 2; waves 4, 5: the alternating column job 3; wave 2 base, wave 3 coefficients
  const bool col_wave = (wave != 2 && wave != 3);
  const int cw = (wave == 7) ? 2 : (wave >= 4) ? 3 : wave;
  constexpr bool PAIRED = (ND == 12) || P::LM_OFF;   // base role with paired stages: a segment is a DPP quad of the base wave
  const int seg = col_wave ? cw * 4 + (lane >> 4) : (PAIRED && wave == 2) ? (lane >> 2) : (lane & (PIPE_SEG - 1));
  const PipeLane L = pipe_lane<PM>(a, seg);
  if (threadIdx.x == 0) { s_fl.base_steps = 0; s_fl.coef_steps = 0; s_fl.hand = 0; s_fl.fail = 0; }
  if (!__syncthreads_or(L.mine)) return;         // workgroup-uniform
  if (wave == 6) return;                         // shares the base wave's SIMD: leaves before the first phase barrier
  if (wave == 2) {
    if constexpr (PAIRED) pipe8_role_base_paired<ND, PM>(a, L, seg, lane & 3, s_int, &s_fl);
    else pipe8_role_base<ND, PM>(a, L, seg, lane >> 4, s_int, &s_fl);
  }
  else if (wave == 3) pipe8_role_coef<ND, PM>(a, L, seg, lane >> 4, s_int, s_coef, s_lm, &s_fl);
  else if (wave == 4) pipe8_role_columns_alt<ND, PM, false>(a, L, seg, lane & 15, s_coef, s_hand, &s_fl);
  else if (wave == 5) pipe8_role_columns_alt<ND, PM, true>(a, L, seg, lane & 15, s_coef, s_hand, &s_fl);
  else pipe8_role_columns<ND, PM>(a, L, seg, lane & 15, s_coef, &s_fl, wave == 7 ? 64 : 8);   // probe build: role switches per wave
}

template <int ND, int PM>
static hipError_t launch_pipe8_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + PIPE_SEG - 1) / PIPE_SEG);
  hipLaunchKernelGGL((k_indirect_pipe8<ND, PM>), grid, dim3(512), 0, st, a);
  return hipGetLastError();
}

template <int ND>
static hipError_t launch_pipe8_pm(int pm, const IndirectArgs& a0, hipStream_t st) {
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_pipe8_one<ND, PM_P0>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_pipe8_one<ND, PM_P1>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_pipe8_one<ND, PM_P2>(a, st);
  if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_pipe8_one<ND, PM_PGEN>(a, st);
  return e;
}

// RK4 only; steps >= 1.
hipError_t launch_indirect_stm_pipe8(int ndim, int pm, const IndirectArgs& a, hipStream_t st) {
  if (a.S <= 0) return hipSuccess;
  if (a.steps < 1) return hipErrorInvalidValue;
  if (ndim == 12) return launch_pipe8_pm<12>(pm, a, st);
  if (ndim == 14) return launch_pipe8_pm<14>(pm, a, st);
  return hipErrorInvalidValue;
}

}  // namespace lto
