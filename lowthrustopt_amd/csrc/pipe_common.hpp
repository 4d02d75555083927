// pipe_common.hpp -- pieces shared by the three-role pipeline kernels (kernels_indirect_pipe8.hip: 16 segments and eight waves per
// workgroup, two RK4 steps per phase; kernels_indirect_pipe48.hip: 48 segments and sixteen waves, one step per phase).
#pragma once
#include "kernels.hpp"
#include <pipe_hooks.hpp>   // product: hooks/ (no-ops); `make probe`: tools/probe_hooks/

namespace lto {

constexpr int PIPE_SEG = 16;   // segments per workgroup

// The coefficients of an RK stage are functions of the stage argument's position, lambda_v (and, for ND = 14, mass and
// lambda_m) only: that is all the base wave publishes (6 / 8 doubles per stage -- an LDS store of a 64-lane wave costs
// ~25 issue cycles of the critical stream), in the order of PipeArg<ND, PM>::idx.
template <int ND> struct PipeCoef { using type = VarCoef12; };
template <> struct PipeCoef<14> { using type = VarCoef14; };
template <int ND, int PM> struct PipeArg {
  static constexpr bool LM = false;
  static constexpr int N = 6;
  static constexpr int idx[6] = {0, 1, 2, 9, 10, 11};
  using Coef = VarCoef12;
};
// ND = 14: lambda_m (the last entry) enters the coefficients only through the unclamped p > 1 law; for the
// always-thrust-limited laws (p = 0, p = 1) seven values are published.
template <int PM> struct PipeArg<14, PM> {
  static constexpr bool LM = !(PM == PM_P0 || PM == PM_P1);
  static constexpr int N = LM ? 8 : 7;
  static constexpr int idx[8] = {0, 1, 2, 6, 10, 11, 12, 13};
  using Coef = VarCoef14;
};

// Layout of the coefficient records in LDS, one record per (ring slot, stage, segment).
struct CoefBySegment {   // [segment][value], records padded to 33 doubles: the coefficient wave's stores (one record per
  static constexpr int LD = 33;   // lane) and the column rows' loads (one record per row) are both conflict-free
  static constexpr bool SCALED = true;
  template <int NC> static constexpr int stage_doubles() { return PIPE_SEG * LD; }
  // column lane j of a row reads doubles j and 16 + j of its segment's record; value e of VarCoef12 / 14 sits in the
  // first double of lane e for e < NA and in the second double of lane e - NA beyond (NA = active column lanes of a
  // row), so that only active lanes are ever DPP sources and the spare lanes can stay switched off
  template <int NA> __device__ static int at(int e, int seg) { return seg * LD + (e < NA ? e : 16 + e - NA); }
  __device__ static int lane_base(int lane, int seg) { return seg * LD + lane; }
};

// What a lane knows about its segment.
struct PipeLane {
  int s;            // segment (after the optional balanced order)
  long node;        // its first node in the SoA arrays
  bool in_range;    // stores allowed (not a shadow lane, and of this launch's control-law class)
  bool mine;
  double h, w2;
  TrajParams tp;
};

template <int PM, int SEGS = PIPE_SEG>
__device__ __forceinline__ PipeLane pipe_lane(const IndirectArgs& a, const int seg) {
  PipeLane L;
  const int s_raw = blockIdx.x * SEGS + seg;
  const int s_lin = s_raw < a.S ? s_raw : a.S - 1;             // shadow lanes repeat the last segment
  L.s = a.order ? a.order[s_lin] : s_lin;
  const int traj = L.s / a.seg_per_traj;
  const int i = L.s - traj * a.seg_per_traj;
  L.node = (long)traj * a.n_nodes + i;
  const long tg = (long)traj * a.t_stride + i;
  L.tp = a.tp[(long)traj * a.tp_stride];
  L.h = (a.t[tg + 1] - a.t[tg]) / (double)a.steps;
  L.w2 = 2.0 * L.tp.omega;
  // mixed-class batch: segments of another control-law class belong to that class's launch; here they run through the
  // barriers without storing
  L.mine = !a.class_filter || p_class(L.tp.p) == PM;
  L.in_range = (s_raw < a.S) && L.mine;
  return L;
}

// role switches: always on in the product build, a mask in IndirectArgs::max_steps in the probe build (pipe_hooks.hpp)
#define PIPE_ROLE_ON(a, bit) hook::role_on((a), (bit))

// ----------------------------------------------------------- column role, one column per lane, coefficients through DPP
// The 16 lanes of a DPP row are the 14 (12) STM columns of ONE segment; lane j of the row holds coefficients j and 16 + j
// of that segment (one ds_read2_b64 per stage instead of 13), and every product  coefficient x column entry  is a
// v_fmac_f64_dpp with row_newbcast:n -- the coefficient is read from lane n of the row inside the FMA, no move, no LDS.
// (Inline asm: the compiler has no pattern that folds a 64-bit DPP move into the FMA.  The DPP source registers are only
// ever written by the LDS loads below, never by a VALU instruction, so the VALU-write -> DPP-read hazard cannot arise.
// tools/micro/dpp_probe.hip checks semantics and issue rate on the device.)
template <int N>
__device__ __forceinline__ void fmac_b(double& acc, const double c, const double x) {      // acc += c[lane N of the row] * x
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(x), "n"(N));
}
template <int N>
__device__ __forceinline__ void fmac_bn(double& acc, const double c, const double x) {     // acc -= c[lane N of the row] * x
  asm("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(x), "n"(N));
}

// acc (+/-)= value E of the segment's VarCoef12 / 14 record (CoefBySegment placement) * x
template <int NA, int E, bool NEG = false>
__device__ __forceinline__ void fmac_c(double& acc, const double cA, const double cB, const double x) {
  static_assert(E < 2 * NA && NA <= 16, "record value outside the active lanes");
  if constexpr (E < NA) { if constexpr (NEG) fmac_bn<E>(acc, cA, x); else fmac_b<E>(acc, cA, x); }
  else { if constexpr (NEG) fmac_bn<E - NA>(acc, cB, x); else fmac_b<E - NA>(acc, cB, x); }
}

// One RK4 stage of one STM column with the coefficients spread over the row (cA, cB: this lane's two doubles of the record
//   [Gxx Gyy Gzz Gxy Gxz Gyz | Hxx Hyy Hzz Hxy Hxz Hyz | ua ub lx ly lz | umx umy umz mm mn Lm Ln Ll]   (VarCoef12 / 14),
// all but lhat pre-multiplied by the stage's weight a (CoefBySegment::SCALED), aw2 = a 2 omega:
//   out = init + a F(arg) arg
// The rows with coefficient products accumulate straight onto init (no separate slope, no separate RK update); the
// Coriolis rows start with a plain FMA, the others with a copy of init.  Same formulas as var_col12 / var_col14.
// LM = false (always-thrust-limited laws of the 14-dim system): d lambda_m_dot / d lambda_m = 0, that term is skipped.
template <int ND, bool LM = true, int NA = ND>
__device__ __forceinline__ void col_dpp_stage(const double cA, const double cB, const double aw2, const double a,
                                              const double (&arg)[ND], const double (&init)[ND], double (&out)[ND]) {
  constexpr int G = (ND == 14) ? 7 : 6;     // first lambda_r row
  constexpr int D = G + 3;                  // first lambda_v row
  enum { Gxx, Gyy, Gzz, Gxy, Gxz, Gyz, Hxx, Hyy, Hzz, Hxy, Hxz, Hyz, Ua, Ub, Lx, Ly, Lz, Umx, Umy, Umz, Mm, Mn, LLm, LLn, LLl };
  const double ax = arg[0], ay = arg[1], az = arg[2];
  const double dx = arg[D], dyv = arg[D + 1], dz = arg[D + 2];
#define FM(E, acc, x) fmac_c<NA, E, false>(acc, cA, cB, x)
#define FN(E, acc, x) fmac_c<NA, E, true>(acc, cA, cB, x)
  double ld = 0.0;
  FM(Lx, ld, dx); FM(Ly, ld, dyv); FM(Lz, ld, dz);
  double tl = 0.0;
  FM(Ub, tl, ld);                                            // a ub (lhat . d)
  out[0] = __builtin_fma(a, arg[3], init[0]); out[1] = __builtin_fma(a, arg[4], init[1]); out[2] = __builtin_fma(a, arg[5], init[2]);
  double o3 = __builtin_fma(aw2, arg[4], init[3]), o4 = __builtin_fma(-aw2, arg[3], init[4]), o5 = init[5];
  FM(Gxx, o3, ax); FM(Gxy, o3, ay); FM(Gxz, o3, az); FN(Ua, o3, dx); FM(Lx, o3, tl);
  FM(Gxy, o4, ax); FM(Gyy, o4, ay); FM(Gyz, o4, az); FN(Ua, o4, dyv); FM(Ly, o4, tl);
  FM(Gxz, o5, ax); FM(Gyz, o5, ay); FM(Gzz, o5, az); FN(Ua, o5, dz); FM(Lz, o5, tl);
  double o7 = init[G], o8 = init[G + 1], o9 = init[G + 2];   // lambda_r rows: minus (H a + G d)
  FN(Hxx, o7, ax); FN(Hxy, o7, ay); FN(Hxz, o7, az); FN(Gxx, o7, dx); FN(Gxy, o7, dyv); FN(Gxz, o7, dz);
  FN(Hxy, o8, ax); FN(Hyy, o8, ay); FN(Hyz, o8, az); FN(Gxy, o8, dx); FN(Gyy, o8, dyv); FN(Gyz, o8, dz);
  FN(Hxz, o9, ax); FN(Hyz, o9, ay); FN(Hzz, o9, az); FN(Gxz, o9, dx); FN(Gyz, o9, dyv); FN(Gzz, o9, dz);
  out[G] = o7; out[G + 1] = o8; out[G + 2] = o9;
  out[D] = __builtin_fma(aw2, dyv, __builtin_fma(-a, arg[G], init[D]));
  out[D + 1] = __builtin_fma(-aw2, dx, __builtin_fma(-a, arg[G + 1], init[D + 1]));
  out[D + 2] = __builtin_fma(-a, arg[G + 2], init[D + 2]);
  if constexpr (ND == 14) {
    const double mu = arg[6];
    FM(Umx, o3, mu); FM(Umy, o4, mu); FM(Umz, o5, mu);
    double o6 = init[6], o13 = init[13];
    FM(Mm, o6, mu); FM(Mn, o6, ld);
    FM(LLm, o13, mu); FM(LLn, o13, ld);
    if constexpr (LM) FM(LLl, o13, arg[13]);
    out[6] = o6; out[13] = o13;
  }
#undef FM
#undef FN
  out[3] = o3; out[4] = o4; out[5] = o5;
}

// One RK4 step of one STM column in the DPP form, coefficients of its four stages at rec[stage * SD] (this lane's first
// double) and rec[stage * SD + 16] (CoefBySegment, scaled: h/2, h/2, h, h/2).  The column equation is linear, so the lane
// carries u = 3^k y after k steps and one step is
//   V1 = u + (h/2) F1 u,  V2 = u + (h/2) F2 V1,  V3 = u + h F3 V2,   u+ = (V1 - u) + 2 V2 + V3 + (h/2) F4 V3  ( = 3 y+ )
// (RK4 written as y+ = -y/3 + Y1/3 + 2 Y2/3 + Y3/3 + (h/6) k4, times 3): every stage is one "init + a F arg"
// accumulation and the weights cost three instructions per component and step instead of four.  COL_RESCALE_EVERY
// steps the lane multiplies by 3^-COL_RESCALE_EVERY (u stays far inside the binary64 range); the caller applies the
// remaining 3^-(steps mod COL_RESCALE_EVERY) (IndirectArgs::stm_scale) when it stores the column.
constexpr int COL_RESCALE_EVERY = 256;
constexpr double COL_RESCALE = 7.193807159919265348769859e-123;   // 3^-256
struct ColStepConst {
  double h2, h, h2w, hw;
  __device__ __forceinline__ ColStepConst(double hh, double w2) : h2(0.5 * hh), h(hh), h2w(0.5 * hh * w2), hw(hh * w2) {}
};
template <int ND, int SD, bool LM = true, int NA = ND>
__device__ __forceinline__ void col_dpp_step(const double* rec, const ColStepConst& k, const int step, double (&y)[ND]) {
  double B[ND], V1[ND], V2[ND], V3[ND];
  col_dpp_stage<ND, LM, NA>(rec[0], rec[16], k.h2w, k.h2, y, y, V1);
#pragma unroll
  for (int c = 0; c < ND; ++c) B[c] = V1[c] - y[c];
  col_dpp_stage<ND, LM, NA>(rec[SD], rec[SD + 16], k.h2w, k.h2, V1, y, V2);
#pragma unroll
  for (int c = 0; c < ND; ++c) B[c] = __builtin_fma(2.0, V2[c], B[c]);
  col_dpp_stage<ND, LM, NA>(rec[2 * SD], rec[2 * SD + 16], k.hw, k.h, V2, y, V3);
#pragma unroll
  for (int c = 0; c < ND; ++c) B[c] += V3[c];
  col_dpp_stage<ND, LM, NA>(rec[3 * SD], rec[3 * SD + 16], k.h2w, k.h2, V3, B, y);
  if (((step + 1) & (COL_RESCALE_EVERY - 1)) == 0) {
#pragma unroll
    for (int c = 0; c < ND; ++c) y[c] *= COL_RESCALE;
  }
}

}  // namespace lto
