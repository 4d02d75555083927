// halves.hpp -- two lanes per 12-component state: the cross-lane pieces shared by kernels_indirect_coop2.hip and
// kernels_indirect_defect2.hip (device code only; the per-lane arithmetic, rhs12_base_half, is in dynamics.hpp).
//
// Lane layout inside a 16-lane DPP row: 4-lane banks A B A B; lane A of a segment owns (r, v), lane B -- four lanes up -- owns
// (lambda_v, lambda_r).  Own rows in global numbering: A (0..5), B (9, 10, 11, 6, 7, 8).
#pragma once
#include <hip/hip_runtime.h>
#include "dynamics.hpp"

namespace lto {

// src's value from the lane 4 below (CTRL = row_shr:4) or 4 above (row_shl:4) into the lanes of the banks in BANK; the other
// lanes keep old.
template <int CTRL, int BANK>
__device__ __forceinline__ double dpp_bank_merge(const double old, const double src) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, 0xF, BANK, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, 0xF, BANK, false);
  return __hiloint2double(hi, lo);
}
// the A lane's x in both lanes of a pair / the B lane's x in both lanes
__device__ __forceinline__ double from_lane_a(const double x) { return dpp_bank_merge<0x114, 0xA>(x, x); }   // row_shr:4 into the B banks
__device__ __forceinline__ double from_lane_b(const double x) { return dpp_bank_merge<0x104, 0x5>(x, x); }   // row_shl:4 into the A banks
// x_A + x_B, the same bits in both lanes
__device__ __forceinline__ double pair_sum(const double x) { return from_lane_a(x) + from_lane_b(x); }


// ---- four lanes per 12-component state (a DPP quad): lane 0 owns r, lane 1 v, lane 2 lambda_v, lane 3 lambda_r (global rows
// 0..2, 3..5, 9..11, 6..8).  quad_take<quad_perm> reads x from the named lane of the own quad.
constexpr int quad_perm(int a, int b, int c, int d) { return a | (b << 2) | (c << 4) | (d << 6); }
template <int CTRL>
__device__ __forceinline__ double quad_take(const double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// (x0 + x1) + (x2 + x3), the same bits in all four lanes
__device__ __forceinline__ double quad_sum(const double x) {
  const double h = quad_take<quad_perm(0, 0, 2, 2)>(x) + quad_take<quad_perm(1, 1, 3, 3)>(x);
  return quad_take<quad_perm(0, 0, 0, 0)>(h) + quad_take<quad_perm(2, 2, 2, 2)>(h);
}

// x summed over the four 16-lane rows of the wavefront (lanes l, l + 16, l + 32, l + 48), the same bits in all four: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange rows / halves between two registers, so with both operands = x one register
// ends up holding (r0 r0 r2 r2) and the other (r1 r1 r3 r3), then (lo lo) and (hi hi).  Five instructions per level and dword pair.
__device__ __forceinline__ double rows_sum(const double x) {
  const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double s = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
  const unsigned slo = (unsigned)__double2loint(s), shi = (unsigned)__double2hiint(s);
  const auto c = __builtin_amdgcn_permlane32_swap(slo, slo, false, false);
  const auto d = __builtin_amdgcn_permlane32_swap(shi, shi, false, false);
  return __hiloint2double((int)d[0], (int)c[0]) + __hiloint2double((int)d[1], (int)c[1]);
}

// ---- the base RHS with FOUR lanes per segment (a DPP quad; kernels_indirect_defect2.hip, quad kernel): lane 0 owns r, lane 1
// v, lane 2 lambda_v, lane 3 lambda_r -- three components and 13 x 3 slopes per lane, a quarter of the tableau arithmetic of
// the one-lane kernel -- and the three reciprocal square roots of an evaluation (the two primaries' distances, |lambda_v|) are
// ONE instruction stream: lane 0 evaluates primary 1, lane 1 primary 2, lanes 2 / 3 the norm, on the argument triple of the
// lane pair's first lane plus a per-lane x offset.  The lanes then trade kappa_b / d_b^{3/2} and e_b (rho_b . lambda_v) (lanes 0, 1),
// 1 / |lambda_v| (from lane 2), and lane 3 takes lambda_r' from lane 1, which holds everything it is made of.  Every sum in lane 1
// -- the lane whose v' and lambda_r' are used -- has the operand order of rhs12_base_parts (same bits).
struct QuadLane {
  double off1;      // x offset to primary 1 where the lane's argument is a position: MU, MU, 0, 0
  double off2;      // from there to the lane's own primary: 0, -1, 0, 0  (a = x + MU, b = a - 1: the reference's two roundings)
  double off2_o;    // ... and to the other one: -1, 0, 0, 0
  double kapb;      // 1 - MU, MU, 0, 0
  double floor;     // -inf for the distance lanes (no clamp, NaN kept), the zero-norm guard of inv_norm_guarded for lanes 2, 3
  double kap_lin, sg_lin;   // slope of a linear lane: lane 0 r' = v: (0, 1); lane 2 lambda_v' = 2 w J lambda_v - lambda_r: (2 w, -1)
  bool lane1, lane3;
};
__device__ __forceinline__ QuadLane quad_lane(const int q4, const TrajParams& tp) {
  QuadLane Q;
  Q.off1 = (q4 < 2) ? tp.MU : 0.0;
  Q.off2 = (q4 == 1) ? -1.0 : 0.0;
  Q.off2_o = (q4 == 0) ? -1.0 : 0.0;
  Q.kapb = (q4 == 0) ? 1.0 - tp.MU : (q4 == 1) ? tp.MU : 0.0;
  Q.floor = (q4 < 2) ? -__builtin_inf() : 9.33263618503218879e-302;
  Q.kap_lin = (q4 == 2) ? 2.0 * tp.omega : 0.0;
  Q.sg_lin = (q4 == 2) ? -1.0 : 1.0;
  Q.lane1 = (q4 == 1); Q.lane3 = (q4 == 3);
  return Q;
}
// by-products of an evaluation as the quad holds them: c, is in lanes 0 / 1 (primary 1 / 2), inv_n everywhere; ua, ub everywhere
// (cooperative kernel, WITH_PARTS: also a0 = the argument's x plus the lane's offset -- x + MU in lanes 0 / 1, lambda_v,x in lanes 2 / 3 --,
// e = 3 kappa_b / d_b^{5/2} and q = 5 e (rho_b . lambda_v) / d_b of the lane's own primary, es = sum_b e_b (rho_b . lambda_v),
// omc = 1 - sum_b kappa_b / d_b^{3/2}: what the column halves apply G, H through their dyadic structure from, dynamics.hpp)
struct QuadParts { double c, is, inv_n, n2, ua, ub, a0, e, q, es, omc; };
// w: the lane's argument triple; k: its slope; P: (r r lambda_v lambda_v) as the lanes hold it after the exchange (what the
// cooperative kernel publishes)
// WITH_PARTS: also the extra by-products listed at QuadParts (two instructions).  (Round 3 built the gravity-gradient block G
// here, in lane 1, and published it: 16 instructions and three stores on the sweep's longest chain.)
template <int PM, bool WITH_PARTS = false>
__device__ __forceinline__ void rhs12_base_quad(const double (&w)[3], const QuadLane& Q, const TrajParams& tp, double (&k)[3], QuadParts& bp,
                                                double (&P)[3]) {
  auto t0022 = [](const double v) { return quad_take<quad_perm(0, 0, 2, 2)>(v); };
  auto t1133 = [](const double v) { return quad_take<quad_perm(1, 1, 3, 3)>(v); };
  auto t2222 = [](const double v) { return quad_take<quad_perm(2, 2, 2, 2)>(v); };
  auto t1032 = [](const double v) { return quad_take<quad_perm(1, 0, 3, 2)>(v); };
  auto t1111 = [](const double v) { return quad_take<quad_perm(1, 1, 1, 1)>(v); };
  double S[3], L[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { P[j] = t0022(w[j]); S[j] = t1133(w[j]); L[j] = t2222(w[j]); }   // (r r lv lv), (v v lr lr), lambda_v
  const double w2 = 2.0 * tp.omega;
  const double a0 = P[0] + Q.off1;
  const double u0 = a0 + Q.off2, u0_o = a0 + Q.off2_o;
  const double yz2 = __builtin_fma(P[1], P[1], P[2] * P[2]);
  const double d = __builtin_fma(u0, u0, yz2);
  const double i = rsqrt_nr(fmax(d, Q.floor));
  const double is = i * i;
  const double c = Q.kapb * (is * i);
  const double e = 3.0 * c * is;
  const double n2 = t2222(d), inv_n = t2222(i);
  const double n = n2 * inv_n;
  double m, ua, ub;
  if constexpr (PM == PM_P1) {            // as rhs12_base_parts
    const double zz = fmin(fmax((1.0 - n) * tp.inv_rho, -700.0), 690.0);
    const double ez = exp_mid(zz);
    const double qq = rcp_nr(1.0 + ez);
    m = tp.accel_limit * qq;
    ua = m * inv_n;
    ub = __builtin_fma(-(tp.accel_limit * tp.inv_rho) * (ez * qq), qq, ua);
  } else {
    double un;
    bool tlim;
    control_dispatch<PM, true, true>(tp, tp.accel_limit, n, inv_n, m, ua, ub, un, tlim);
  }
  const double yzl = __builtin_fma(P[1], L[1], P[2] * L[2]);
  const double s = __builtin_fma(u0, L[0], yzl);
  const double t = e * s;
  const double c_o = t1032(c), t_o = t1032(t);
  // lane 1's view: own = primary 2 (c2, b), other = primary 1 (c1, a)
  const double cs = c_o + c, omc = 1.0 - cs;
  const double es = t_o + t;
  const double tA = __builtin_fma(t_o, u0_o, t * u0);
  const double x = P[0], yy = P[1], z = P[2];
  const double ax = __builtin_fma(-ua, L[0], __builtin_fma(-c_o, u0_o, __builtin_fma(-c, u0, __builtin_fma(w2, S[1], x))));
  const double ay = __builtin_fma(-ua, L[1], __builtin_fma(-cs, yy, __builtin_fma(-w2, S[0], yy)));
  const double az = __builtin_fma(-ua, L[2], -cs * z);
  const double gx = __builtin_fma(-omc, L[0], -tA);
  const double gy = __builtin_fma(-omc, L[1], -es * yy);
  const double gz = __builtin_fma(cs, L[2], -es * z);
  const double g3x = t1111(gx), g3y = t1111(gy), g3z = t1111(gz);
  const double lin0 = __builtin_fma(Q.kap_lin, P[1], Q.sg_lin * S[0]);
  const double lin1 = __builtin_fma(-Q.kap_lin, P[0], Q.sg_lin * S[1]);
  const double lin2 = Q.sg_lin * S[2];
  k[0] = Q.lane1 ? ax : (Q.lane3 ? g3x : lin0);
  k[1] = Q.lane1 ? ay : (Q.lane3 ? g3y : lin1);
  k[2] = Q.lane1 ? az : (Q.lane3 ? g3z : lin2);
  bp.c = c; bp.is = is; bp.inv_n = inv_n; bp.n2 = n2; bp.ua = ua; bp.ub = ub;
  if constexpr (WITH_PARTS) { bp.a0 = a0; bp.e = e; bp.q = 5.0 * t * is; bp.es = es; bp.omc = omc; }
}

// ---- 14-dim system (state + mass + costates + mass costate), always-thrust-limited laws (p = 0, p = 1): the quad of
// rhs12_base_quad with a fourth component in two of its lanes -- lane 1 owns (v, m), lane 3 owns (lambda_r, lambda_m); lanes 0 and 2
// carry a zero there.  The mass enters through accelLimit = cT / m alone (dynamics.hpp rhs14): one reciprocal, off the
// reciprocal-square-root chains, and
//   m' = -kappa_td umag m,   lambda_m' = -umag n / m          (GeneralCode/twoBody_stateCostate_mass_deriv.jl:57,76 in CRTBP units).
// By-products beyond QuadParts, for the variational terms of the mass (dynamics.hpp var_col14):
//   inv_m, un = d umag / d n, m = umag.
struct QuadParts14 { QuadParts q; double inv_m, mass, un, umag, n; };
template <int PM>
__device__ __forceinline__ void rhs14_base_quad(const double (&w)[4], const QuadLane& Q, const TrajParams& tp, double (&k)[4], QuadParts14& bp,
                                                double (&P)[3]) {
  static_assert(PM == PM_P0 || PM == PM_P1, "the 14-dim quad is built for the laws whose thrust is cT sigma(n) / mass");
  auto t0022 = [](const double v) { return quad_take<quad_perm(0, 0, 2, 2)>(v); };
  auto t1133 = [](const double v) { return quad_take<quad_perm(1, 1, 3, 3)>(v); };
  auto t2222 = [](const double v) { return quad_take<quad_perm(2, 2, 2, 2)>(v); };
  auto t1032 = [](const double v) { return quad_take<quad_perm(1, 0, 3, 2)>(v); };
  auto t1111 = [](const double v) { return quad_take<quad_perm(1, 1, 1, 1)>(v); };
  double S[3], L[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) { P[j] = t0022(w[j]); S[j] = t1133(w[j]); L[j] = t2222(w[j]); }   // (r r lv lv), (v v lr lr), lambda_v
  const double mass = t1111(w[3]);
  const double inv_m = rcp_nr(mass);
  const double aL = tp.cT * inv_m;
  const double w2 = 2.0 * tp.omega;
  const double a0 = P[0] + Q.off1;
  const double u0 = a0 + Q.off2, u0_o = a0 + Q.off2_o;
  const double yz2 = __builtin_fma(P[1], P[1], P[2] * P[2]);
  const double d = __builtin_fma(u0, u0, yz2);
  const double i = rsqrt_nr(fmax(d, Q.floor));
  const double is = i * i;
  const double c = Q.kapb * (is * i);
  const double e = 3.0 * c * is;
  const double n2 = t2222(d), inv_n = t2222(i);
  const double n = n2 * inv_n;
  double m, ua, ub, un;
  if constexpr (PM == PM_P1) {            // as rhs12_base_parts, with accelLimit = cT / m
    const double zz = fmin(fmax((1.0 - n) * tp.inv_rho, -700.0), 690.0);
    const double ez = exp_mid(zz);
    const double qq = rcp_nr(1.0 + ez);
    m = aL * qq;
    ua = m * inv_n;
    un = (aL * tp.inv_rho) * (ez * qq) * qq;
    ub = ua - un;
  } else {
    m = aL; ua = aL * inv_n; ub = ua; un = 0.0;
  }
  const double yzl = __builtin_fma(P[1], L[1], P[2] * L[2]);
  const double s = __builtin_fma(u0, L[0], yzl);
  const double t = e * s;
  const double c_o = t1032(c), t_o = t1032(t);
  const double cs = c_o + c, omc = 1.0 - cs;
  const double es = t_o + t;
  const double tA = __builtin_fma(t_o, u0_o, t * u0);
  const double x = P[0], yy = P[1], z = P[2];
  const double ax = __builtin_fma(-ua, L[0], __builtin_fma(-c_o, u0_o, __builtin_fma(-c, u0, __builtin_fma(w2, S[1], x))));
  const double ay = __builtin_fma(-ua, L[1], __builtin_fma(-cs, yy, __builtin_fma(-w2, S[0], yy)));
  const double az = __builtin_fma(-ua, L[2], -cs * z);
  const double gx = __builtin_fma(-omc, L[0], -tA);
  const double gy = __builtin_fma(-omc, L[1], -es * yy);
  const double gz = __builtin_fma(cs, L[2], -es * z);
  const double g3x = t1111(gx), g3y = t1111(gy), g3z = t1111(gz);
  const double lin0 = __builtin_fma(Q.kap_lin, P[1], Q.sg_lin * S[0]);
  const double lin1 = __builtin_fma(-Q.kap_lin, P[0], Q.sg_lin * S[1]);
  const double lin2 = Q.sg_lin * S[2];
  k[0] = Q.lane1 ? ax : (Q.lane3 ? g3x : lin0);
  k[1] = Q.lane1 ? ay : (Q.lane3 ? g3y : lin1);
  k[2] = Q.lane1 ? az : (Q.lane3 ? g3z : lin2);
  const double mdot = (-tp.kappa_td * m) * mass;                 // rhs14: dy[6]
  const double lmdot = -((m * n) * inv_m);                       // rhs14: dy[13], thrust-limited laws
  k[3] = Q.lane1 ? mdot : (Q.lane3 ? lmdot : 0.0);
  bp.q.c = c; bp.q.is = is; bp.q.inv_n = inv_n; bp.q.n2 = n2; bp.q.ua = ua; bp.q.ub = ub;
  bp.q.a0 = a0; bp.q.e = e; bp.q.q = 5.0 * t * is; bp.q.es = es; bp.q.omc = omc;
  bp.inv_m = inv_m; bp.mass = mass; bp.un = un; bp.umag = m; bp.n = n;
}

}  // namespace lto
