// kernels_indirect_pipe32.hip -- the three-role RK4 pipeline for batches BETWEEN one and a few rounds of the eight-wave form:
// 32 segments and twelve wavefronts per workgroup, one barrier per step.
//
// The eight-wave form (kernels_indirect_pipe8.hip) owns 16 segments per CU and cannot share a CU (two wavefronts per SIMD at 172-194
// registers fill the register file), so 4 097 ... 8 192 segments cost two rounds (2 x 70 us, 14-dim).  The 48- / 44-segment form
// (kernels_indirect_pipe48.hip) runs four wavefronts per SIMD at 128 registers; its 14-dim base role -- one lane per segment,
// the four stages one after the other -- is a dependent chain that sets the step time (189 us per round whatever shares its SIMD).
// This form takes the ROLES of the eight-wave kernel -- the paired-stage base role with four lanes per segment (two expensive
// evaluations per step on the chain), the coefficient role that also carries lambda_m for the always-thrust-limited laws, the DPP
// column role with the spare lanes switched off -- and the SYNCHRONISATION of the large-batch kernel: hand-overs double-buffered by
// the parity of the step, one __syncthreads() per step, steps + 2 phases, no flags, no polling, nothing that can wait forever.
//   waves 0, 1      base: segments 0..15 / 16..31, a DPP quad per segment (pipe32_role_base = pipe8_role_base_paired's step)
//   waves 2, 3      coefficients of the same halves, lane = (segment, RK stage), one step behind
//   waves 4..11     columns: four segments each (DPP row = segment), two steps behind
// Wave w runs on SIMD w mod 4: SIMDs 0 and 1 carry a base wave and two column waves (305 + 2 x 340 instructions per step, 14-dim),
// SIMDs 2 and 3 a coefficient wave and two column waves (180 + 2 x 340).  Three wavefronts per SIMD: 168 registers, no scratch.
// LDS: 2 x 4 stage slabs of 16-byte pairs (16 KB) + 2 x 4 coefficient slabs (68 KB).
// Built for the instantiations whose stages pair: 12-dim (every control law) and 14-dim with p = 0 / p = 1.
#include "pipe_common.hpp"

namespace lto {

constexpr int P32_SEG = 32;

template <int ND, int PM> struct Pipe32 {
  using Arg = PipeArg<ND, PM>;
  static constexpr int NI = Arg::N;
  static constexpr int NC = sizeof(typename PipeCoef<ND>::type) / sizeof(double);
  static constexpr bool LM_OFF = (ND == 14) && !Arg::LM;              // lambda_m integrated by the coefficient waves
  static_assert(ND == 12 || LM_OFF, "built for the laws whose RK4 stages pair");
  static constexpr int NB = LM_OFF ? ND - 1 : ND;                     // components the base waves integrate
  static constexpr int NA = LM_OFF ? ND - 1 : ND;                     // STM columns a row integrates (the rest: unit vectors)
  static constexpr int SD = P32_SEG * CoefBySegment::LD;              // doubles of one (step parity, stage) coefficient slab
  // stage arguments as 16-byte pairs: [step parity][stage][pair][segment][2], (r0 r1) (r2 l0) (l1 l2) [(m -)]
  static constexpr int NPAIR = (ND == 14) ? 4 : 3;
  static constexpr int SLABD = NPAIR * P32_SEG * 2;
  static constexpr int INT_DOUBLES = 2 * 4 * SLABD;
  static constexpr int COEF_DOUBLES = 2 * 4 * SD;
  __host__ __device__ static constexpr int lin(int e) { return (ND == 12) ? e : (e < 3 ? e : e == 3 ? 6 : e < 7 ? e - 1 : 7); }
};

typedef double p32_d2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double q32_from(const double x) {      // CTRL = quad_perm code
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// Lane q of a segment's quad: stage slot q & 1 (A: stages 1, 3; B: stages 2, 4), gravitating body q >> 1.  Everything that leaves
// an evaluation comes from lanes 0 (A) and 1 (B), the lanes of primary 1, whose operand order is that of rhs12_base / rhs14_base.
__device__ __forceinline__ double q32_from_a(const double x) { return q32_from<0x00>(x); }       // quad_perm:[0,0,0,0]
__device__ __forceinline__ double q32_from_b(const double x) { return q32_from<0x55>(x); }       // quad_perm:[1,1,1,1]
__device__ __forceinline__ double q32_other_body(const double x) { return q32_from<0x4E>(x); }   // quad_perm:[2,3,0,1]

template <int ND, int PM>
__device__ __forceinline__ void pipe32_role_base(const IndirectArgs& a, const PipeLane& L, const int seg, const int q,
                                                 double* s_int) {
  using P = Pipe32<ND, PM>;
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
  auto swap_body = [](const double v) { return q32_other_body(v); };
  const double gA = is_b ? h2 : 0.0;                 // round 1, own stage argument (rows r, lambda_v): y + gA k1
  const double al = is_b ? 0.0 : h2, be = is_b ? h : 0.0;   // round 2: y + al k2 + be k3
  const int own = is_b ? SLAB : 0;                   // own stage's slab relative to the round's first
  const double kt = L.tp.kappa_td;
  double y[ND];
#pragma unroll
  for (int c = 0; c < ND; ++c) y[c] = a.X[c * a.ldx + L.node];
  double inv_m = 0.0;
  if constexpr (M14) inv_m = rcp_nr(y[MI]);
  // one barrier per step, steps + 2 phases (the coefficient waves are one step behind, the column waves two)
  for (int step = 0; step < steps + 2; ++step) {
    if (step < steps) {
      {
        double* slab = s_int + ((step & 1) * 4) * SLAB + 2 * seg;     // stage s of this step: slab + s * SLAB; pair q of it at + q * 2 P32_SEG
        auto publish = [&](double* d, const double (&pr)[3], const double (&pl)[3]) {
          *reinterpret_cast<p32_d2*>(d) = p32_d2{pr[0], pr[1]};
          *reinterpret_cast<p32_d2*>(d + 2 * P32_SEG) = p32_d2{pr[2], pl[0]};
          *reinterpret_cast<p32_d2*>(d + 4 * P32_SEG) = p32_d2{pl[1], pl[2]};
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
        for (int i = 0; i < 3; ++i) { a1[i] = q32_from_a(o.av[i]); a2[i] = q32_from_b(o.av[i]); g1[i] = q32_from_a(o.gl[i]); g2[i] = q32_from_b(o.gl[i]); }
        double kv1[3], kv2[3], km1 = 0.0, km2 = 0.0, y2m = 0.0, y3m = 0.0, im2 = 0.0, im3 = 0.0, gf2 = 0.0;
        if constexpr (M14) {
          const double u1 = q32_from_a(o.gf) * inv_m;
          gf2 = q32_from_b(o.gf);
          km1 = -kt * q32_from_a(o.sc); km2 = -kt * q32_from_b(o.sc);
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
          slab[6 * P32_SEG] = y[MI]; slab[SLAB + 6 * P32_SEG] = y2m;       // pair 3, first half
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
        for (int i = 0; i < 3; ++i) { a3[i] = q32_from_a(o.av[i]); a4[i] = q32_from_b(o.av[i]); g3[i] = q32_from_a(o.gl[i]); g4[i] = q32_from_b(o.gl[i]); }
        double kv3[3], kv4[3], km3 = 0.0, km4 = 0.0, y4m = 0.0, mnew = 0.0, im4 = 0.0, gf4 = 0.0;
        if constexpr (M14) {
          const double u3 = q32_from_a(o.gf) * im3;
          gf4 = q32_from_b(o.gf);
          km3 = -kt * q32_from_a(o.sc); km4 = -kt * q32_from_b(o.sc);
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
          slab[2 * SLAB + 6 * P32_SEG] = y3m; slab[3 * SLAB + 6 * P32_SEG] = y4m;
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
      }
    }
    __syncthreads();
  }
  if (L.in_range && q == 0) {
    if (a.defect) {
#pragma unroll
      for (int c = 0; c < NB; ++c) a.defect[c * a.ldd + L.s] = y[c] - a.X[c * a.ldx + L.node + 1];
    }
    if (a.errors) a.errors[L.s] = 0.0;
    if (a.nacc) a.nacc[L.s] = steps;
    if (a.nrej) a.nrej[L.s] = 0;
  }
}


// ------------------------------------------------------------------------------------------------ coefficient role
// lane = (segment of this wave's half, RK stage); phase p builds step p - 1.  For the always-thrust-limited laws of the 14-dim
// system it also accumulates lambda_m (see kernels_indirect_pipe8.hip) and writes that row of the defect.
template <int ND, int PM>
__device__ __forceinline__ void pipe32_role_coef(const IndirectArgs& a, const PipeLane& L, const int seg, const int stage,
                                                 const double* s_int, double* s_coef, double* s_lm) {
  using P = Pipe32<ND, PM>;
  using Coef = typename PipeCoef<ND>::type;
  constexpr int NI = P::NI, NC = P::NC, SD = P::SD;
  const int steps = a.steps;
  const double as = (stage == 2) ? L.h : 0.5 * L.h;
  const double bw = (stage == 0 || stage == 3) ? L.h * (1.0 / 6.0) : L.h * (1.0 / 3.0);
  double lm_acc = 0.0;
  for (int p = 0; p < steps + 2; ++p) {
    if (p >= 1 && p <= steps) {
      const int slab = ((p - 1) & 1) * 4 + stage;
      double arg[ND], dead[ND];
#pragma unroll
      for (int c = 0; c < ND; ++c) arg[c] = 0.0;
      const p32_d2* src = reinterpret_cast<const p32_d2*>(s_int + slab * P::SLABD) + seg;
      double lv[2 * P::NPAIR];
#pragma unroll
      for (int q = 0; q < P::NPAIR; ++q) { const p32_d2 v = src[q * P32_SEG]; lv[2 * q] = v.x; lv[2 * q + 1] = v.y; }
#pragma unroll
      for (int e = 0; e < NI; ++e) arg[P::Arg::idx[e]] = lv[P::lin(e)];
      Coef vc;
      if constexpr (ND == 12) rhs12<PM, true>(arg, L.tp, dead, vc);
      else rhs14<PM, true>(arg, L.tp, dead, vc);
      if constexpr (P::LM_OFF) lm_acc = __builtin_fma(bw, dead[ND - 1], lm_acc);
      const double* o = reinterpret_cast<const double*>(&vc);
      double* dst = s_coef + slab * SD;
      constexpr int NST = P::LM_OFF ? NC - 1 : NC;
#pragma unroll
      for (int e = 0; e < NST; ++e) dst[CoefBySegment::at<P::NA>(e, seg)] = (e < 14 || e > 16) ? o[e] * as : o[e];
    }
    __syncthreads();
  }
  if constexpr (P::LM_OFF) {
    s_lm[stage * P32_SEG + seg] = lm_acc;
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the four stage rows of a segment are lanes of this wavefront
    if (L.in_range && stage == 0 && a.defect) {
      const double sum = (s_lm[seg] + s_lm[P32_SEG + seg]) + (s_lm[2 * P32_SEG + seg] + s_lm[3 * P32_SEG + seg]);
      const long r = (long)(ND - 1) * a.ldx + L.node;
      a.defect[(ND - 1) * a.ldd + L.s] = (a.X[r] + sum) - a.X[r + 1];
    }
  }
}

// --------------------------------------------------------------------------------------------------- column role
template <int ND, int PM>
__device__ __forceinline__ void pipe32_role_columns(const IndirectArgs& a, const PipeLane& L, const int seg, const int col,
                                                    const double* s_coef) {
  using P = Pipe32<ND, PM>;
  constexpr int SD = P::SD, NA = P::NA;
  const int steps = a.steps;
  const ColStepConst k(L.h, L.w2);
  const double* rec = s_coef + CoefBySegment::lane_base(col, seg);
  double y[ND];
#pragma unroll
  for (int r = 0; r < ND; ++r) y[r] = (r == col) ? 1.0 : 0.0;
  for (int p = 0; p < steps + 2; ++p) {
    if (p >= 2 && col < NA) col_dpp_step<ND, SD, P::Arg::LM, NA>(rec + ((p & 1) * 4) * SD, k, p - 2, y);   // spare lanes stay off: never DPP sources
    __syncthreads();
  }
  if (L.in_range && col < ND) {
    const double sc = (col < NA) ? a.stm_scale : 1.0;
    const double poison = (col < NA) ? 0.0 : L.h - L.h;     // a unit column of a segment with a NaN (or infinite) span is NaN like the rest
#pragma unroll
    for (int r = 0; r < ND; ++r) a.Phi[(long)(col * ND + r) * a.ldp + L.s] = __builtin_fma(y[r], sc, poison);
  }
}

template <int ND, int PM>
__global__ __launch_bounds__(768) void k_indirect_pipe32(const IndirectArgs a) {
  using P = Pipe32<ND, PM>;
  __shared__ __attribute__((aligned(16))) double s_int[P::INT_DOUBLES];
  __shared__ double s_coef[P::COEF_DOUBLES];
  __shared__ double s_lm[4 * P32_SEG];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int seg = (wave < 2) ? wave * 16 + (lane >> 2) : (wave < 4) ? (wave - 2) * 16 + (lane & 15) : (wave - 4) * 4 + (lane >> 4);
  const PipeLane L = pipe_lane<PM, P32_SEG>(a, seg);
  if (!__syncthreads_or(L.mine)) return;         // workgroup-uniform
  // the two roles with the long dependent streams issue first (three wavefronts share a SIMD and all meet at one barrier per step)
  if (wave < 2) { __builtin_amdgcn_s_setprio(3); pipe32_role_base<ND, PM>(a, L, seg, lane & 3, s_int); }
  else if (wave < 4) { __builtin_amdgcn_s_setprio(2); pipe32_role_coef<ND, PM>(a, L, seg, lane >> 4, s_int, s_coef, s_lm); }
  else pipe32_role_columns<ND, PM>(a, L, seg, lane & 15, s_coef);
}

template <int ND, int PM>
static hipError_t launch_pipe32_one(const IndirectArgs& a, hipStream_t st) {
  dim3 grid((a.S + P32_SEG - 1) / P32_SEG);
  hipLaunchKernelGGL((k_indirect_pipe32<ND, PM>), grid, dim3(768), 0, st, a);
  return hipGetLastError();
}

// Which (dimension, control-law classes) this form is built for.
bool indirect_stm_pipe32_available(int ndim, int pm) {
  if (ndim == 12) return true;
  return ndim == 14 && !(pm & ((1 << PM_P2) | (1 << PM_PGEN)));
}

// RK4 only; steps >= 1.
hipError_t launch_indirect_stm_pipe32(int ndim, int pm, const IndirectArgs& a0, hipStream_t st) {
  if (a0.S <= 0) return hipSuccess;
  if (a0.steps < 1 || !indirect_stm_pipe32_available(ndim, pm)) return hipErrorInvalidValue;
  IndirectArgs a = a0;
  a.class_filter = single_class(pm) ? 0 : 1;
  hipError_t e = hipSuccess;
  if (ndim == 12) {
    if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_pipe32_one<12, PM_P0>(a, st);
    if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_pipe32_one<12, PM_P1>(a, st);
    if (e == hipSuccess && (pm & (1 << PM_P2))) e = launch_pipe32_one<12, PM_P2>(a, st);
    if (e == hipSuccess && (pm & (1 << PM_PGEN))) e = launch_pipe32_one<12, PM_PGEN>(a, st);
  } else {
    if (e == hipSuccess && (pm & (1 << PM_P0))) e = launch_pipe32_one<14, PM_P0>(a, st);
    if (e == hipSuccess && (pm & (1 << PM_P1))) e = launch_pipe32_one<14, PM_P1>(a, st);
  }
  return e;
}

}  // namespace lto
