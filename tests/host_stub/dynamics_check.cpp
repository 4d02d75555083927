#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "dynamics.hpp"
using namespace lto;
// lean base RHS of the pipeline kernel (rhs*_base) against the one-piece rhs*<PM, false>
template <int PM> static void base12(const double (&y)[12], const TrajParams& tp, double* dy_a, double* dy_b) {
  double d[12]; VarCoef12 v; rhs12<PM, false>(y, tp, d, v);
  for (int i = 0; i < 12; ++i) dy_a[i] = d[i];
  rhs12_base<PM>(y, tp, d);
  for (int i = 0; i < 12; ++i) dy_b[i] = d[i];
}
template <int PM> static void base14(const double (&y)[14], const TrajParams& tp, double* dy_a, double* dy_b) {
  double d[14]; VarCoef14 v; rhs14<PM, false>(y, tp, d, v);
  for (int i = 0; i < 14; ++i) dy_a[i] = d[i];
  rhs14_base<PM>(y, tp, d);
  for (int i = 0; i < 14; ++i) dy_b[i] = d[i];
}
// by-products path of the cooperative kernel: rhs12_base_parts + coef12_from_parts against rhs12<PM, true>
template <int PM> static void parts12(const double (&y)[12], const TrajParams& tp, double* dy_a, double* vc_a, double* dy_b, double* vc_b) {
  double d[12]; VarCoef12 v; rhs12<PM, true>(y, tp, d, v);
  for (int i = 0; i < 12; ++i) dy_a[i] = d[i];
  std::memcpy(vc_a, &v, sizeof v);
  BaseParts12 bp; rhs12_base_parts<PM>(y, tp, d, bp);
  for (int i = 0; i < 12; ++i) dy_b[i] = d[i];
  VarCoef12 w; coef12_from_parts(y[0], y[1], y[2], y[9], y[10], y[11], bp, tp.MU, w);
  std::memcpy(vc_b, &w, sizeof w);
}
// two-lanes-per-state forms of the cooperative kernel: the base slopes assembled from lane A (r, v) and lane B (lambda_v,
// lambda_r) of rhs12_base_half, the column slopes from var_col12_top_g / var_col12_bottom_g, against the one-piece functions
template <int PM> static void halves12(const double (&y)[12], const double (&c)[12], const TrajParams& tp, double* dy_a, double* dc_a,
                                       double* dy_b, double* dc_b, double* parts_ab) {
  double d[12], dc[12]; VarCoef12 v; rhs12<PM, true>(y, tp, d, v);
  const double w2 = 2.0 * tp.omega;
  var_col12(v, w2, c, dc);
  for (int i = 0; i < 12; ++i) { dy_a[i] = d[i]; dc_a[i] = dc[i]; }
  const double R[3] = {y[0], y[1], y[2]}, L[3] = {y[9], y[10], y[11]};
  const double qa[3] = {y[3], y[4], y[5]}, qb[3] = {y[6], y[7], y[8]};
  double kp[3], kq[3]; BaseParts12 bpa, bpb;
  rhs12_base_half<PM>(R, L, qa, true, 1.0, 0.0, tp, kp, kq, bpa);          // lane A: (r', v')
  for (int j = 0; j < 3; ++j) { dy_b[j] = kp[j]; dy_b[3 + j] = kq[j]; }
  rhs12_base_half<PM>(R, L, qb, false, -1.0, w2, tp, kp, kq, bpb);         // lane B: (lambda_v', lambda_r')
  for (int j = 0; j < 3; ++j) { dy_b[9 + j] = kp[j]; dy_b[6 + j] = kq[j]; }
  std::memcpy(parts_ab, &bpa, sizeof bpa); std::memcpy(parts_ab + 7, &bpb, sizeof bpb);
  const double wt[6] = {c[0], c[1], c[2], c[3], c[4], c[5]}, wb[6] = {c[9], c[10], c[11], c[6], c[7], c[8]};
  const double dd[3] = {c[9], c[10], c[11]}, aa[3] = {c[0], c[1], c[2]};
  double ot[6], ob[6];
  // G as the base wave publishes it (the cooperative kernel's lane 1 builds it with the arithmetic of rhs12's VAR block)
  const CoefG12 g = {v.Gxx, v.Gyy, v.Gzz, v.Gxy, v.Gxz, v.Gyz};
  var_col12_top_g(g, y[9], y[10], y[11], bpa.inv_n, bpa.ua, bpa.ub, w2, wt, dd, ot);
  var_col12_bottom_g(g, y[0], y[1], y[2], y[9], y[10], y[11], bpa.c1, bpa.c2, bpa.i1s, bpa.i2s, tp.MU, w2, wb, aa, ob);
  for (int j = 0; j < 6; ++j) dc_b[j] = ot[j];
  for (int j = 0; j < 3; ++j) { dc_b[9 + j] = ob[j]; dc_b[6 + j] = ob[3 + j]; }
}
extern "C" {
void chk_halves12(const double* y, const double* col, const double* tpv, int pm, double* dy_a, double* dc_a, double* dy_b, double* dc_b,
                  double* parts_ab) {
  TrajParams tp; std::memcpy(&tp, tpv, sizeof tp);
  double yy[12], cc[12];
  for (int i = 0; i < 12; ++i) { yy[i] = y[i]; cc[i] = col[i]; }
  if (pm == PM_P1) halves12<PM_P1>(yy, cc, tp, dy_a, dc_a, dy_b, dc_b, parts_ab);
  else if (pm == PM_P2) halves12<PM_P2>(yy, cc, tp, dy_a, dc_a, dy_b, dc_b, parts_ab);
  else if (pm == PM_P0) halves12<PM_P0>(yy, cc, tp, dy_a, dc_a, dy_b, dc_b, parts_ab);
  else halves12<PM_PGEN>(yy, cc, tp, dy_a, dc_a, dy_b, dc_b, parts_ab);
}
// F*col for the 14-dim system via device formulas (host-compiled)
void chk_rhs14(const double* y, const double* tpv, int pm, double* dy, const double* col, double* dcol, double* dy_f, double* dcol_f) {
  TrajParams tp; std::memcpy(&tp, tpv, sizeof tp);
  double yy[14], d[14], c[14], dc[14];
  for (int i = 0; i < 14; ++i) { yy[i] = y[i]; c[i] = col[i]; }
  VarCoef14 vc;
  if (pm == PM_P1) rhs14<PM_P1, true>(yy, tp, d, vc);
  else if (pm == PM_P2) rhs14<PM_P2, true>(yy, tp, d, vc);
  else if (pm == PM_P0) rhs14<PM_P0, true>(yy, tp, d, vc);
  else rhs14<PM_PGEN, true>(yy, tp, d, vc);
  var_col14(vc, 2.0 * tp.omega, c, dc);
  for (int i = 0; i < 14; ++i) { dy[i] = d[i]; dcol[i] = dc[i]; }
  double y28[28], k28[28];
  for (int i = 0; i < 14; ++i) { y28[i] = y[i]; y28[14 + i] = col[i]; }
  if (pm == PM_P1) rhs14_fused1<PM_P1>(y28, tp, 2.0 * tp.omega, k28);
  else if (pm == PM_P2) rhs14_fused1<PM_P2>(y28, tp, 2.0 * tp.omega, k28);
  else if (pm == PM_P0) rhs14_fused1<PM_P0>(y28, tp, 2.0 * tp.omega, k28);
  else rhs14_fused1<PM_PGEN>(y28, tp, 2.0 * tp.omega, k28);
  for (int i = 0; i < 14; ++i) { dy_f[i] = k28[i]; dcol_f[i] = k28[14 + i]; }
}
void chk_rhs12(const double* y, const double* tpv, int pm, double* dy, const double* col, double* dcol, double* dy_f, double* dcol_f) {
  TrajParams tp; std::memcpy(&tp, tpv, sizeof tp);
  double yy[12], d[12], c[12], dc[12];
  for (int i = 0; i < 12; ++i) { yy[i] = y[i]; c[i] = col[i]; }
  VarCoef12 vc;
  if (pm == PM_P1) rhs12<PM_P1, true>(yy, tp, d, vc);
  else if (pm == PM_P2) rhs12<PM_P2, true>(yy, tp, d, vc);
  else if (pm == PM_P0) rhs12<PM_P0, true>(yy, tp, d, vc);
  else rhs12<PM_PGEN, true>(yy, tp, d, vc);
  var_col12(vc, 2.0 * tp.omega, c, dc);
  for (int i = 0; i < 12; ++i) { dy[i] = d[i]; dcol[i] = dc[i]; }
  double y24[24], k24[24];
  for (int i = 0; i < 12; ++i) { y24[i] = y[i]; y24[12 + i] = col[i]; }
  if (pm == PM_P1) rhs12_fused1<PM_P1>(y24, tp, 2.0 * tp.omega, k24);
  else if (pm == PM_P2) rhs12_fused1<PM_P2>(y24, tp, 2.0 * tp.omega, k24);
  else if (pm == PM_P0) rhs12_fused1<PM_P0>(y24, tp, 2.0 * tp.omega, k24);
  else rhs12_fused1<PM_PGEN>(y24, tp, 2.0 * tp.omega, k24);
  for (int i = 0; i < 12; ++i) { dy_f[i] = k24[i]; dcol_f[i] = k24[12 + i]; }
}
void chk_base(int ndim, const double* y, const double* tpv, int pm, double* dy_a, double* dy_b) {
  TrajParams tp; std::memcpy(&tp, tpv, sizeof tp);
  if (ndim == 12) {
    double yy[12]; for (int i = 0; i < 12; ++i) yy[i] = y[i];
    if (pm == PM_P1) base12<PM_P1>(yy, tp, dy_a, dy_b);
    else if (pm == PM_P2) base12<PM_P2>(yy, tp, dy_a, dy_b);
    else if (pm == PM_P0) base12<PM_P0>(yy, tp, dy_a, dy_b);
    else base12<PM_PGEN>(yy, tp, dy_a, dy_b);
    return;
  }
  double yy[14]; for (int i = 0; i < 14; ++i) yy[i] = y[i];
  if (pm == PM_P1) base14<PM_P1>(yy, tp, dy_a, dy_b);
  else if (pm == PM_P2) base14<PM_P2>(yy, tp, dy_a, dy_b);
  else if (pm == PM_P0) base14<PM_P0>(yy, tp, dy_a, dy_b);
  else base14<PM_PGEN>(yy, tp, dy_a, dy_b);
}
void chk_parts12(const double* y, const double* tpv, int pm, double* dy_a, double* vc_a, double* dy_b, double* vc_b) {
  TrajParams tp; std::memcpy(&tp, tpv, sizeof tp);
  double yy[12]; for (int i = 0; i < 12; ++i) yy[i] = y[i];
  if (pm == PM_P1) parts12<PM_P1>(yy, tp, dy_a, vc_a, dy_b, vc_b);
  else if (pm == PM_P2) parts12<PM_P2>(yy, tp, dy_a, vc_a, dy_b, vc_b);
  else if (pm == PM_P0) parts12<PM_P0>(yy, tp, dy_a, vc_a, dy_b, vc_b);
  else parts12<PM_PGEN>(yy, tp, dy_a, vc_a, dy_b, vc_b);
}
int chk_sizeof_tp() { return (int)sizeof(TrajParams); }
}
