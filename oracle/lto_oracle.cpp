/*
 * lto_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, single-threaded restatement of the reference's multiple-shooting hot path
 * (travelingspaceman/LowThrustOpt, Julia).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may build, load or call this file.  The product
 * (lowthrustopt_amd/ + liblto_hip.so) never includes, links or calls anything under oracle/.
 *
 * PARITY STATUS: *parity unpinned* against the reference itself.  The reference is Julia; no
 * `julia` binary exists in the build image or on the GPU box, the reference ships no tests and no
 * golden vectors, and its indirect path runs inside un-vendored third-party packages
 * (OrdinaryDiffEq 6.26.4 `Vern8`, ForwardDiff 0.10.32: Manifest.toml:987,419).  What pins this
 * file instead (tests/test_oracle_*.py, fixtures under tests/golden/):
 *   - RHS values against an independent 40-digit mpmath evaluation;
 *   - segment flows / STMs against scipy DOP853 @1e-13 and a 34-digit (binary128) integration;
 *   - the reference's own data files L2_Anderson_{1,2}.txt (closed halo orbits: column->column
 *     ballistic propagation, Jacobi constant);
 *   - invariants: symplectic 12x12 STM, det = 1, forward/backward round trip, order of accuracy.
 *
 * Each function cites the reference file:line it follows (paths relative to the reference root).
 * Scalar type T is a template parameter so the same text runs in binary64 (the reference's
 * arithmetic), binary128 (to mint converged fixtures) and forward-mode dual numbers (what the
 * reference's ForwardDiff.jacobian does, src/multiShoot_CRTBP_indirect.jl:103-121).
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <quadmath.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "dop853_tableau.h"

namespace {

typedef __float128 quad;

/* ------------------------------------------------------------------ scalar overloads */
inline double o_sqrt(double x) { return std::sqrt(x); }
inline quad o_sqrt(quad x) { return sqrtq(x); }
inline double o_pow(double x, double e) { return std::pow(x, e); }
inline quad o_pow(quad x, double e) { return powq(x, (quad)e); }
inline double o_tanh(double x) { return std::tanh(x); }
inline quad o_tanh(quad x) { return tanhq(x); }
inline double o_val(double x) { return x; }
inline double o_val(quad x) { return (double)x; }
inline bool o_isnan(double x) { return std::isnan(x); }
inline bool o_isnan(quad x) { return isnanq(x); }

/* Forward-mode dual number with N partials (ForwardDiff.Dual analogue). */
template <int N>
struct Dual {
  double v;
  double d[N];
  Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
  Dual(double x) : v(x) { for (int i = 0; i < N; ++i) d[i] = 0.0; }
};
template <int N> inline Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v / b.v; for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
template <int N> inline Dual<N> operator+(const Dual<N>& a, double b) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator+(double b, const Dual<N>& a) { Dual<N> r = a; r.v += b; return r; }
template <int N> inline Dual<N> operator-(const Dual<N>& a, double b) { Dual<N> r = a; r.v -= b; return r; }
template <int N> inline Dual<N> operator-(double b, const Dual<N>& a) { Dual<N> r = -a; r.v += b; return r; }
template <int N> inline Dual<N> operator*(const Dual<N>& a, double b) { Dual<N> r; r.v = a.v * b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b; return r; }
template <int N> inline Dual<N> operator*(double b, const Dual<N>& a) { return a * b; }
template <int N> inline Dual<N> operator/(const Dual<N>& a, double b) { Dual<N> r; r.v = a.v / b; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / b; return r; }
template <int N> inline Dual<N> operator/(double a, const Dual<N>& b) { return Dual<N>(a) / b; }
template <int N> inline Dual<N>& operator+=(Dual<N>& a, const Dual<N>& b) { a = a + b; return a; }
template <int N> inline bool operator>(const Dual<N>& a, const Dual<N>& b) { return a.v > b.v; }
template <int N> inline bool operator>(const Dual<N>& a, double b) { return a.v > b; }
template <int N> inline bool operator==(const Dual<N>& a, double b) { return a.v == b; }
template <int N> inline Dual<N> o_sqrt(const Dual<N>& a) { Dual<N> r; r.v = std::sqrt(a.v); for (int i = 0; i < N; ++i) r.d[i] = a.d[i] / (2.0 * r.v); return r; }
template <int N> inline Dual<N> o_pow(const Dual<N>& a, double e) { Dual<N> r; r.v = std::pow(a.v, e); double g = e * std::pow(a.v, e - 1.0); for (int i = 0; i < N; ++i) r.d[i] = g * a.d[i]; return r; }
template <int N> inline Dual<N> o_tanh(const Dual<N>& a) { Dual<N> r; r.v = std::tanh(a.v); double g = 1.0 - r.v * r.v; for (int i = 0; i < N; ++i) r.d[i] = g * a.d[i]; return r; }
template <int N> inline double o_val(const Dual<N>& a) { return a.v; }
template <int N> inline bool o_isnan(const Dual<N>& a) { return std::isnan(a.v); }

/* x^e with the exponent carried in the scalar type (binary128 runs keep 1/(p-1) in binary128) */
inline double o_powT(double x, double e) { return std::pow(x, e); }
inline quad o_powT(quad x, quad e) { return powq(x, e); }
template <int N> inline Dual<N> o_powT(const Dual<N>& a, const Dual<N>& e) { return o_pow(a, e.v); }

/* Euclidean norm of a 3-vector.  For duals at the origin the two-sided derivative does not exist; the
 * one-sided directional derivative |dc| is returned, which is what the reference's forward finite
 * difference sees when the nominal control is exactly zero (src/multiShoot_CRTBP_direct.jl:129-140). */
inline double o_norm3(double a, double b, double c) { return std::sqrt(a * a + b * b + c * c); }
inline quad o_norm3(quad a, quad b, quad c) { return sqrtq(a * a + b * b + c * c); }
template <int N> inline Dual<N> o_norm3(const Dual<N>& a, const Dual<N>& b, const Dual<N>& c) {
  if (a.v == 0.0 && b.v == 0.0 && c.v == 0.0) {
    Dual<N> r(0.0);
    for (int i = 0; i < N; ++i) r.d[i] = std::sqrt(a.d[i] * a.d[i] + b.d[i] * b.d[i] + c.d[i] * c.d[i]);
    return r;
  }
  return o_sqrt(a * a + b * b + c * c);
}

/* |x|_inf over value and (for duals) partials: DiffEqBase's default error norm includes the
 * partials of ForwardDiff duals, so the step controller also resolves the sensitivities. */
inline double o_absmax(double x) { return std::fabs(x); }
inline double o_absmax(quad x) { return (double)fabsq(x); }
template <int N> inline double o_absmax(const Dual<N>& a) { double m = std::fabs(a.v); for (int i = 0; i < N; ++i) m = std::max(m, std::fabs(a.d[i])); return m; }
/* number of scalar components / sum of squares of (x/scale), for rms norms */
inline int o_ncomp(double) { return 1; }
inline int o_ncomp(quad) { return 1; }
template <int N> inline int o_ncomp(const Dual<N>&) { return 1 + N; }
inline double o_sumsq(double x) { return x * x; }
inline double o_sumsq(quad x) { return (double)(x * x); }
template <int N> inline double o_sumsq(const Dual<N>& a) { double s = a.v * a.v; for (int i = 0; i < N; ++i) s += a.d[i] * a.d[i]; return s; }

/* ------------------------------------------------------------------ A1: 12-dim state+costate RHS
 * Follows src/CRTBP_stateCostate_deriv.jl:9-90 line by line (longhand costate rows :83-85).
 * prm = (MU, DU, TU, thrustLimit, mass, time_direction, p, rho)  (:13).
 * Returns 0, or 1 for the reference's error("Invalid value of p!") (:52). */
template <class T>
int rhs_state_costate(const T* y, const double* prm, T* dy) {
  /* MU is carried in T so that the binary128 instantiation evaluates (1-MU) etc. in binary128 */
  const T MU = T(prm[0]);
  const double DU = prm[1], TU = prm[2], thrustLimit = prm[3], mass = prm[4];
  const double time_direction = prm[5], p = prm[6], rho = prm[7];
  const T X1 = y[0], X2 = y[1], X3 = y[2], X4 = y[3], X5 = y[4], X6 = y[5];
  const T L1 = y[6], L2 = y[7], L3 = y[8], L4 = y[9], L5 = y[10], L6 = y[11];
  (void)X6;

  /* :33 -- same operation order as the reference; carried in T (identical in binary64) */
  const T accelLimit = T(thrustLimit) / mass / 1e3 * (T(TU) * T(TU)) / DU;
  const T nlv = o_sqrt(L4 * L4 + L5 * L5 + L6 * L6);                  /* norm(λv) */
  T umag;
  if (p == 0.0) {
    umag = accelLimit;                                                 /* :36-39 */
  } else if (p == 1.0) {
    T g = nlv - 1.0;                                                   /* :42 */
    umag = 0.5 * (1.0 + o_tanh(g / (2.0 * T(rho)))) * accelLimit;      /* :43 */
  } else if (p > 1.0) {
    umag = o_powT(T(1.0) / T(p) * nlv, T(1.0) / (T(p) - 1.0));         /* :46 */
    if (umag > accelLimit) umag = accelLimit;                          /* :48-50 */
  } else {
    return 1;                                                          /* :52 */
  }
  T c1 = -umag * L4 / nlv, c2 = -umag * L5 / nlv, c3 = -umag * L6 / nlv; /* :57 */
  if (o_isnan(c1)) { c1 = T(0.0); c2 = T(0.0); c3 = T(0.0); }          /* :59-64 */

  const T r1_3 = o_pow((X1 + MU) * (X1 + MU) + X2 * X2 + X3 * X3, 1.5);           /* :69 */
  const T r2_3 = o_pow((X1 + MU - 1.0) * (X1 + MU - 1.0) + X2 * X2 + X3 * X3, 1.5); /* :70 */
  const T temp1 = (MU + X1 - 1.0) * (MU + X1 - 1.0) + X2 * X2 + X3 * X3;          /* :72 */
  const T temp2 = (MU + X1) * (MU + X1) + X2 * X2 + X3 * X3;                      /* :73 */
  const T temp3 = 2.0 * MU + 2.0 * X1 - 2.0;                                      /* :74 */
  const T t1_52 = o_pow(temp1, 2.5), t2_52 = o_pow(temp2, 2.5);
  const T t1_32 = o_pow(temp1, 1.5), t2_32 = o_pow(temp2, 1.5);

  dy[0] = X4; dy[1] = X5; dy[2] = X6;                                              /* :78 */
  dy[3] = -(1.0 - MU) * (X1 + MU) / r1_3 - MU * (X1 - 1.0 + MU) / r2_3 + 2.0 * time_direction * X5 + X1 + c1; /* :79 */
  dy[4] = -(1.0 - MU) * X2 / r1_3 - MU * X2 / r2_3 - 2.0 * time_direction * X4 + X2 + c2;                    /* :80 */
  dy[5] = -(1.0 - MU) * X3 / r1_3 - MU * X3 / r2_3 + c3;                                                     /* :81 */

  /* :83 */
  dy[6] = -L5 * ((3.0 * MU * X2 * temp3) / (2.0 * t1_52) - (3.0 * X2 * (MU - 1.0) * (2.0 * MU + 2.0 * X1)) / (2.0 * t2_52))
          - L6 * ((3.0 * MU * X3 * temp3) / (2.0 * t1_52) - (3.0 * X3 * (MU - 1.0) * (2.0 * MU + 2.0 * X1)) / (2.0 * t2_52))
          - L4 * ((MU - 1.0) / t2_32 - MU / t1_32 + (3.0 * MU * (MU + X1 - 1.0) * temp3) / (2.0 * t1_52)
                  - (3.0 * (MU + X1) * (MU - 1.0) * (2.0 * MU + 2.0 * X1)) / (2.0 * t2_52) + 1.0);
  /* :84 */
  dy[7] = L6 * ((3.0 * X2 * X3 * (MU - 1.0)) / t2_52 - (3.0 * MU * X2 * X3) / t1_52)
          - L5 * ((MU - 1.0) / t2_32 - MU / t1_32 - (3.0 * X2 * X2 * (MU - 1.0)) / t2_52 + (3.0 * MU * X2 * X2) / t1_52 + 1.0)
          - L4 * ((3.0 * MU * X2 * (MU + X1 - 1.0)) / t1_52 - (3.0 * X2 * (MU + X1) * (MU - 1.0)) / t2_52);
  /* :85 */
  dy[8] = L6 * (MU / t1_32 - (MU - 1.0) / t2_32 + (3.0 * X3 * X3 * (MU - 1.0)) / t2_52 - (3.0 * MU * X3 * X3) / t1_52)
          + L5 * ((3.0 * X2 * X3 * (MU - 1.0)) / t2_52 - (3.0 * MU * X2 * X3) / t1_52)
          - L4 * ((3.0 * MU * X3 * (MU + X1 - 1.0)) / t1_52 - (3.0 * X3 * (MU + X1) * (MU - 1.0)) / t2_52);
  dy[9] = 2.0 * L5 * time_direction - L1;   /* :86 */
  dy[10] = -L2 - 2.0 * L4 * time_direction; /* :87 */
  dy[11] = -L3;                             /* :88 */
  return 0;
}

/* ------------------------------------------------------------------ 14-dim extension (NO reference output)
 * CRTBP state + mass + costates + mass costate, y = (r, v, m, lambda_r, lambda_v, lambda_m), the ordering of the
 * reference's two-body model GeneralCode/twoBody_stateCostate_mass_deriv.jl:17-20,59-76.  BASELINE configs[1]
 * names a 14-dim CRTBP system that the reference does not contain (its CRTBP RHS is 12-dim, constant mass:
 * src/CRTBP_stateCostate_deriv.jl:10).  Conventions chosen (DESIGN.md "14-dim extension"):
 *   accelLimit uses the CURRENT mass (two-body model :26) in CRTBP units (stateCostate_deriv.jl:33);
 *   control law and smoothing exactly as the CRTBP file (:36-53, tanh(g/(2 rho)));
 *   mdot = -td * (thrust in N) / (Isp * 9.81) * TU, thrust[N] = umag*m*1e3*DU/TU^2   (prop_EP_deriv.jl:41-42 units);
 *   lambda_m_dot = -dH/dm, H = lr.v + lv.(g + u) + lm*mdot; for every thrust-limited law this equals
 *   (lambda_v . u)/m, the form of the two-body model (:76) without its km<->m factor 1e3.
 * prm = (MU, DU, TU, thrustLimit, Isp, time_direction, p, rho): the `mass` slot of the tuple carries Isp. */
template <class T>
int rhs_state_costate_mass(const T* y, const double* prm, T* dy) {
  const T MU = T(prm[0]);
  const double DU = prm[1], TU = prm[2], thrustLimit = prm[3], Isp = prm[4];
  const double td = prm[5], p = prm[6], rho = prm[7];
  const T X1 = y[0], X2 = y[1], X3 = y[2], X4 = y[3], X5 = y[4], X6 = y[5], m = y[6];
  const T L1 = y[7], L2 = y[8], L3 = y[9], L4 = y[10], L5 = y[11], L6 = y[12], Lm = y[13];
  const T accelLimit = T(thrustLimit) / m / 1e3 * (T(TU) * T(TU)) / DU;
  const T nlv = o_sqrt(L4 * L4 + L5 * L5 + L6 * L6);
  T umag;
  bool thrust_limited = true;
  if (p == 0.0) {
    umag = accelLimit;
  } else if (p == 1.0) {
    T g = nlv - 1.0;
    umag = 0.5 * (1.0 + o_tanh(g / (2.0 * T(rho)))) * accelLimit;
  } else if (p > 1.0) {
    umag = o_powT(T(1.0) / T(p) * nlv, T(1.0) / (T(p) - 1.0));
    if (umag > accelLimit) umag = accelLimit; else thrust_limited = false;
  } else {
    return 1;
  }
  T c1 = -umag * L4 / nlv, c2 = -umag * L5 / nlv, c3 = -umag * L6 / nlv;
  if (o_isnan(c1)) { c1 = T(0.0); c2 = T(0.0); c3 = T(0.0); }
  const double kappa = 1e3 * DU / (TU * Isp * 9.81);
  const T d1 = (X1 + MU) * (X1 + MU) + X2 * X2 + X3 * X3;
  const T d2 = (X1 + MU - 1.0) * (X1 + MU - 1.0) + X2 * X2 + X3 * X3;
  const T r1_3 = o_pow(d1, 1.5), r2_3 = o_pow(d2, 1.5), r1_5 = o_pow(d1, 2.5), r2_5 = o_pow(d2, 2.5);
  dy[0] = X4; dy[1] = X5; dy[2] = X6;
  dy[3] = -(1.0 - MU) * (X1 + MU) / r1_3 - MU * (X1 - 1.0 + MU) / r2_3 + 2.0 * td * X5 + X1 + c1;
  dy[4] = -(1.0 - MU) * X2 / r1_3 - MU * X2 / r2_3 - 2.0 * td * X4 + X2 + c2;
  dy[5] = -(1.0 - MU) * X3 / r1_3 - MU * X3 / r2_3 + c3;
  dy[6] = -td * kappa * umag * m;
  /* lambda_r_dot = -G lambda_v with the tidal tensor G written out */
  const T a = X1 + MU, b = X1 + MU - 1.0;
  const T cs = (1.0 - MU) / r1_3 + MU / r2_3;
  const T e1 = 3.0 * (1.0 - MU) / r1_5, e2 = 3.0 * MU / r2_5;
  const T Gxx = 1.0 - cs + e1 * a * a + e2 * b * b, Gyy = 1.0 - cs + (e1 + e2) * X2 * X2, Gzz = (e1 + e2) * X3 * X3 - cs;
  const T Gxy = (e1 * a + e2 * b) * X2, Gxz = (e1 * a + e2 * b) * X3, Gyz = (e1 + e2) * X2 * X3;
  dy[7] = -(Gxx * L4 + Gxy * L5 + Gxz * L6);
  dy[8] = -(Gxy * L4 + Gyy * L5 + Gyz * L6);
  dy[9] = -(Gxz * L4 + Gyz * L5 + Gzz * L6);
  dy[10] = 2.0 * L5 * td - L1;
  dy[11] = -L2 - 2.0 * L4 * td;
  dy[12] = -L3;
  /* -dH/dm: thrust-limited laws have umag ~ 1/m (so d(mdot)/dm = 0); the unclamped p > 1 law does not depend on m */
  if (thrust_limited) dy[13] = -umag * nlv / m;
  else dy[13] = td * kappa * Lm * umag;
  return 0;
}

/* ------------------------------------------------------------------ A2: given-thrust RHS
 * Follows src/CRTBP_prop_EP_deriv.jl:8-61.  n = 6 (mass literal 1000.0, :20) or 7 (mass = x[6]).
 * `control` is in Newtons and may carry sensitivities (T), so it is typed T. */
template <class T>
void rhs_prop_ep(const T* s, int n, double MU, double DU, double TU, double Isp, const T* control,
                 double time_direction, T* ds) {
  const T x = s[0], y = s[1], z = s[2], xdot = s[3], ydot = s[4], zdot = s[5];
  T m = (n == 7) ? s[6] : T(1000.0);                               /* :17-21 */
  const T r1 = o_sqrt((x + MU) * (x + MU) + y * y + z * z);        /* :24 */
  const T r2 = o_sqrt((x + MU - 1.0) * (x + MU - 1.0) + y * y + z * z); /* :25 */
  const T r1_3 = r1 * r1 * r1, r2_3 = r2 * r2 * r2;                /* :28-29 */
  const T nc = o_norm3(control[0], control[1], control[2]);
  const T T_mag = nc / m / 1e3 * (TU * TU) / DU;                   /* :32 */
  T T1, T2, T3;
  /* :35-36 `T = control` when norm(control)==0.  Value-identical restatement that keeps the
   * sensitivity dT/dcontrol = T_mag/|c| continuous for dual numbers: */
  if (nc == 0.0) { const T k = 1.0 / m / 1e3 * (TU * TU) / DU; T1 = control[0] * k; T2 = control[1] * k; T3 = control[2] * k; }
  else { T1 = control[0] / nc * T_mag; T2 = control[1] / nc * T_mag; T3 = control[2] / nc * T_mag; } /* :38 */
  const double g0 = 9.81;                                           /* :41 */
  const T mdot = -time_direction * nc / (Isp * g0) * TU;            /* :42 */
  const double omega = time_direction;                              /* :45 */
  ds[0] = xdot; ds[1] = ydot; ds[2] = zdot;
  ds[3] = -(1.0 - MU) * (x + MU) / r1_3 - MU * (x - 1.0 + MU) / r2_3 + 2.0 * omega * ydot + x + T1; /* :48 */
  ds[4] = -(1.0 - MU) * y / r1_3 - MU * y / r2_3 - 2.0 * omega * xdot + y + T2;                     /* :49 */
  ds[5] = -(1.0 - MU) * z / r1_3 - MU * z / r2_3 + T3;                                              /* :50 */
  if (n == 7) ds[6] = mdot;                                         /* :53-55 */
}

/* ------------------------------------------------------------------ integrators
 * F is a callable  f(const T* y, T* dy).  Both reference RHS are autonomous (t unused), so the
 * integrators only see step sizes.  Time is carried in T so that binary128 runs use binary128 steps
 * and dual-number runs can differentiate with respect to the step (tf partial). */

/* Classical RK4, GeneralCode/ode.jl:21-73 (stage formulas :64-68); `nint` equal steps. */
template <class T, class F>
void ode4(F&& f, int n, T h, int nint, T* y) {
  std::vector<T> F1(n), F2(n), F3(n), F4(n), yt(n);
  for (int s = 0; s < nint; ++s) {
    f(y, F1.data());
    for (int i = 0; i < n; ++i) yt[i] = y[i] + 0.5 * h * F1[i];   /* :65 */
    f(yt.data(), F2.data());
    for (int i = 0; i < n; ++i) yt[i] = y[i] + 0.5 * h * F2[i];   /* :66 */
    f(yt.data(), F3.data());
    for (int i = 0; i < n; ++i) yt[i] = y[i] + h * F3[i];         /* :67 */
    f(yt.data(), F4.data());
    for (int i = 0; i < n; ++i) y[i] = y[i] + (h / 6.0) * (F1[i] + 2.0 * F2[i] + 2.0 * F3[i] + F4[i]); /* :68 */
  }
}

/* Fehlberg 7(8) coefficients, GeneralCode/ode.jl:875-892 (alpha_, beta_ 13x12, chi_, psi_). */
struct RKF78 {
  double alpha[12];
  double beta[13][12];
  double chi[13];
  double psi[13];
  RKF78() {
    const double a[12] = {2. / 27., 1. / 9, 1. / 6, 5. / 12, 0.5, 5. / 6, 1. / 6, 2. / 3, 1. / 3, 1, 0, 1};
    std::memcpy(alpha, a, sizeof a);
    std::memset(beta, 0, sizeof beta);
    /* beta[k][j]: weight of slope k (0-based) in the argument of slope j+1. */
    const double c1[1] = {2. / 27};
    const double c2[2] = {1. / 36, 1. / 12};
    const double c3[3] = {1. / 24, 0, 1. / 8};
    const double c4[4] = {5. / 12, 0, -25. / 16, 25. / 16};
    const double c5[5] = {0.05, 0, 0, 0.25, 0.2};
    const double c6[6] = {-25. / 108, 0, 0, 125. / 108, -65. / 27, 125. / 54};
    const double c7[7] = {31. / 300, 0, 0, 0, 61. / 225, -2. / 9, 13. / 900};
    const double c8[8] = {2, 0, 0, -53. / 6, 704. / 45, -107. / 9, 67. / 90, 3};
    const double c9[9] = {-91. / 108, 0, 0, 23. / 108, -976. / 135, 311. / 54, -19. / 60, 17. / 6, -1. / 12};
    const double c10[10] = {2383. / 4100, 0, 0, -341. / 164, 4496. / 1025, -301. / 82, 2133. / 4100, 45. / 82, 45. / 164, 18. / 41};
    const double c11[10] = {3. / 205, 0, 0, 0, 0, -6. / 41, -3. / 205, -3. / 41, 3. / 41, 6. / 41};
    const double c12[12] = {-1777. / 4100, 0, 0, -341. / 164, 4496. / 1025, -289. / 82, 2193. / 4100, 51. / 82, 33. / 164, 12. / 41, 0, 1};
    const double* cols[12] = {c1, c2, c3, c4, c5, c6, c7, c8, c9, c10, c11, c12};
    const int len[12] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 12};
    for (int j = 0; j < 12; ++j) for (int k = 0; k < len[j]; ++k) beta[k][j] = cols[j][k];
    const double ch[13] = {0, 0, 0, 0, 0, 34. / 105, 9. / 35, 9. / 35, 9. / 280, 9. / 280, 0, 41. / 840, 41. / 840};
    const double ps[13] = {1., 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, -1, -1};
    std::memcpy(chi, ch, sizeof ch);
    std::memcpy(psi, ps, sizeof ps);
  }
};
static const RKF78 kRKF78;

/* One RKF7(8) step: slopes (ode.jl:931-935), 8th-order update (:937), error term (:940-943).
 * Evaluation order follows the Julia expressions: `xi + hi*f*beta_[:,j]` is ((hi*f)*beta_). */
template <class T, class F>
double rkf78_step(F&& f, int n, T h, const T* x, T* xnew, std::vector<T>& fs /* n*13 */, std::vector<T>& xt) {
  f(x, &fs[0]);
  for (int j = 0; j < 12; ++j) {
    for (int i = 0; i < n; ++i) {
      T acc = T(0.0);
      for (int k = 0; k <= j; ++k) acc += (h * fs[k * n + i]) * kRKF78.beta[k][j];
      xt[i] = x[i] + acc;
    }
    f(xt.data(), &fs[(j + 1) * n]);
  }
  double delta = 0.0;
  for (int i = 0; i < n; ++i) {
    T acc = T(0.0), g = T(0.0);
    for (int k = 0; k < 13; ++k) {
      acc += (h * fs[k * n + i]) * kRKF78.chi[k];
      g += ((h * 41.0 / 840.0) * fs[k * n + i]) * kRKF78.psi[k];
    }
    xnew[i] = x[i] + acc;
    delta = std::max(delta, std::fabs(o_val(g)));  /* norm(gamma1, Inf), :943 */
  }
  return delta;
}

/* Fixed-grid RKF7(8): ode7_8, GeneralCode/ode.jl:773-953.  `npts` grid points -> npts-1 steps
 * (:904-907,:924); returns maxErr (:946-948).  y is overwritten with Xout[:,end]. */
template <class T, class F>
double ode7_8(F&& f, int n, T h, int npts, T* y) {
  std::vector<T> fs(n * 13), xt(n), xn(n);
  double maxErr = 0.0;
  for (int s = 1; s < npts; ++s) {
    double delta = rkf78_step<T>(f, n, h, y, xn.data(), fs, xt);
    for (int i = 0; i < n; ++i) y[i] = xn[i];
    if (delta > maxErr) maxErr = delta;
  }
  return maxErr;
}

/* Adaptive RKF7(8): ode78, GeneralCode/ode.jl:364-544 (loop :479-534).
 * h0 = span/50 (:471), hmax = span/2.5 (:464), hmin = span/1e7 (:470); accept if delta <= tau,
 * tau = tol*max(norm(x,Inf),1) (:497); h <- min(hmax, 0.8 h (tau/delta)^(1/8)) (:517-520).
 * Returns number of accepted steps (negative if the hmin "singularity" exit was taken, :524). */
template <class T, class F>
int ode78(F&& f, int n, T span, double tol, T* x, int* n_rejected) {
  std::vector<T> fs(n * 13), xt(n), xn(n);
  const double pw = 1.0 / 8.0;
  T t = T(0.0);
  const T hmax = span / 2.5, hmin = span / 1e7;
  T h = span / 50.0;
  int acc = 0, rej = 0;
  while (o_val(t) < o_val(span) && o_val(h) >= o_val(hmin)) {
    if (o_val(t + h) > o_val(span)) h = span - t;
    double delta = rkf78_step<T>(f, n, h, x, xn.data(), fs, xt);
    double nx = 0.0;
    for (int i = 0; i < n; ++i) nx = std::max(nx, std::fabs(o_val(x[i])));
    double tau = tol * std::max(nx, 1.0);
    if (delta <= tau) {
      t = t + h;
      for (int i = 0; i < n; ++i) x[i] = xn[i];
      ++acc;
    } else {
      ++rej;
    }
    if (delta == 0.0) delta = 1e-16;
    T hn = 0.8 * h * std::pow(tau / delta, pw);
    h = (o_val(hn) < o_val(hmax)) ? hn : hmax;
  }
  if (n_rejected) *n_rejected = rej;
  return (o_val(t) < o_val(span)) ? -acc : acc;
}

/* Adaptive DOP853 (Hairer's 8(5,3) pair) standing in for OrdinaryDiffEq's Vern8 at
 * reltol = abstol = 1e-13 (src/multiShoot_CRTBP_indirect.jl:79,110).  Controller: error norm
 * err = |h| e5^2 / sqrt((e5^2 + 0.01 e3^2) n) on scale = atol + rtol*max(|y|,|ynew|); accept if
 * err < 1; factor = min(10, 0.9 err^(-1/8)) (<=1 right after a rejection), reject factor =
 * max(0.2, 0.9 err^(-1/8)); initial step by Hairer's d0/d1/d2 rule.  For dual numbers the norms run
 * over values and partials (see o_sumsq). */
template <class T, class F>
int dop853(F&& f, int n, T span, double rtol, double atol, T* y, int* n_rejected, int max_steps) {
  const int NS = ODP_NSTAGES;
  std::vector<T> K((NS + 1) * n), yt(n), yn(n), f1(n);
  const double dir = (o_val(span) >= 0) ? 1.0 : -1.0;
  const double L = std::fabs(o_val(span));
  if (L == 0.0) { if (n_rejected) *n_rejected = 0; return 0; }
  int ncomp = 0;
  for (int i = 0; i < n; ++i) ncomp += o_ncomp(y[i]);
  /* initial step */
  f(y, &K[0]);
  double h_abs;
  {
    double d0 = 0, d1 = 0;
    for (int i = 0; i < n; ++i) {
      double sc = atol + std::fabs(o_val(y[i])) * rtol;
      d0 += o_sumsq(y[i] / sc); d1 += o_sumsq(K[i] / sc);
    }
    d0 = std::sqrt(d0 / ncomp); d1 = std::sqrt(d1 / ncomp);
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    for (int i = 0; i < n; ++i) yt[i] = y[i] + (h0 * dir) * K[i];
    f(yt.data(), f1.data());
    double d2 = 0;
    for (int i = 0; i < n; ++i) {
      double sc = atol + std::fabs(o_val(y[i])) * rtol;
      d2 += o_sumsq((f1[i] - K[i]) / sc);
    }
    d2 = std::sqrt(d2 / ncomp) / h0;
    double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? std::max(1e-6, h0 * 1e-3) : std::pow(0.01 / std::max(d1, d2), 1.0 / 9.0);
    h_abs = std::min(std::min(100 * h0, h1), L);
  }
  double t = 0.0;
  int acc = 0, rej = 0;
  bool have_f = true; /* K[0] holds f(y) */
  while (t < L && acc + rej < max_steps) {
    bool rejected = false;
    for (;;) {
      double h = h_abs;
      bool last = false;
      if (t + h >= L) { h = L - t; last = true; }
      const T hs = (last && t == 0.0) ? span : T(h * dir);
      if (!have_f) { f(y, &K[0]); have_f = true; }
      for (int s = 1; s < NS; ++s) {
        for (int i = 0; i < n; ++i) {
          T a = T(0.0);
          for (int k = 0; k < s; ++k) if (ODP_A[s][k] != 0.0) a += K[k * n + i] * ODP_A[s][k];
          yt[i] = y[i] + a * hs;
        }
        f(yt.data(), &K[s * n]);
      }
      for (int i = 0; i < n; ++i) {
        T a = T(0.0);
        for (int k = 0; k < NS; ++k) if (ODP_B[k] != 0.0) a += K[k * n + i] * ODP_B[k];
        yn[i] = y[i] + a * hs;
      }
      f(yn.data(), &K[NS * n]);
      double e5 = 0, e3 = 0;
      for (int i = 0; i < n; ++i) {
        double sc = atol + std::max(std::fabs(o_val(y[i])), std::fabs(o_val(yn[i]))) * rtol;
        T a5 = T(0.0), a3 = T(0.0);
        for (int k = 0; k <= NS; ++k) {
          if (ODP_E5[k] != 0.0) a5 += K[k * n + i] * ODP_E5[k];
          if (ODP_E3[k] != 0.0) a3 += K[k * n + i] * ODP_E3[k];
        }
        e5 += o_sumsq(a5 / sc); e3 += o_sumsq(a3 / sc);
      }
      double err;
      if (e5 == 0.0 && e3 == 0.0) err = 0.0;
      else err = std::fabs(h) * e5 / std::sqrt((e5 + 0.01 * e3) * ncomp);
      if (err < 1.0) {
        double factor = (err == 0.0) ? 10.0 : std::min(10.0, 0.9 * std::pow(err, -1.0 / 8.0));
        if (rejected) factor = std::min(1.0, factor);
        h_abs = h * factor;
        t = last ? L : t + h;
        for (int i = 0; i < n; ++i) { y[i] = yn[i]; K[i] = K[NS * n + i]; } /* FSAL */
        ++acc;
        break;
      } else {
        h_abs = h * std::max(0.2, 0.9 * std::pow(err, -1.0 / 8.0));
        rejected = true;
        ++rej;
        if (acc + rej >= max_steps) break;
      }
    }
  }
  if (n_rejected) *n_rejected = rej;
  return (t < L) ? -acc : acc;
}

/* integrator selector shared by the shooting sweeps (values mirror include/lto.h) */
enum { M_RK4 = 0, M_RKF78_FIXED = 1, M_RKF78_ADAPTIVE = 2, M_DOP853_ADAPTIVE = 3 };

template <class T, class F>
int integrate(F&& f, int n, T span, int method, int steps, double rtol, double atol, T* y, double* err_out, int* nacc, int* nrej) {
  if (err_out) *err_out = 0.0;
  if (nacc) *nacc = steps;
  if (nrej) *nrej = 0;
  switch (method) {
    case M_RK4: ode4<T>(f, n, span / (double)steps, steps, y); return 0;
    case M_RKF78_FIXED: { double e = ode7_8<T>(f, n, span / (double)steps, steps + 1, y); if (err_out) *err_out = e; return 0; }
    case M_RKF78_ADAPTIVE: { int r = 0; int a = ode78<T>(f, n, span, rtol, y, &r); if (nacc) *nacc = std::abs(a); if (nrej) *nrej = r; return a < 0 ? 2 : 0; }
    case M_DOP853_ADAPTIVE: { int r = 0; int a = dop853<T>(f, n, span, rtol, atol, y, &r, 100000); if (nacc) *nacc = std::abs(a); if (nrej) *nrej = r; return a < 0 ? 2 : 0; }
  }
  return -1;
}

} /* namespace */

/* =============================================================================== C API (ctypes) */
extern "C" {

/* A1, binary64 */
int lto_o_rhs_state_costate(const double* y, const double* prm, double* dy) { return rhs_state_costate<double>(y, prm, dy); }
/* A1, binary128 in/out carried as double pairs is overkill: evaluate in binary128, return rounded double
 * and the residual (hi, lo) so callers can compare beyond 1e-16. */
int lto_o_rhs_state_costate_q(const double* y, const double* prm, double* dy_hi, double* dy_lo) {
  quad yq[12], dq[12];
  for (int i = 0; i < 12; ++i) yq[i] = y[i];
  int rc = rhs_state_costate<quad>(yq, prm, dq);
  for (int i = 0; i < 12; ++i) { dy_hi[i] = (double)dq[i]; dy_lo[i] = (double)(dq[i] - (quad)dy_hi[i]); }
  return rc;
}
/* 12x12 Jacobian of A1 by dual numbers, row-major J[r*12+c] = d ydot_r / d y_c */
int lto_o_rhs_state_costate_jac(const double* y, const double* prm, double* J) {
  Dual<12> yd[12], dd[12];
  for (int i = 0; i < 12; ++i) { yd[i] = Dual<12>(y[i]); yd[i].d[i] = 1.0; }
  int rc = rhs_state_costate<Dual<12>>(yd, prm, dd);
  for (int r = 0; r < 12; ++r) for (int c = 0; c < 12; ++c) J[r * 12 + c] = dd[r].d[c];
  return rc;
}

/* A2, binary64; n = 6 or 7 */
void lto_o_rhs_prop_ep(const double* s, int n, double MU, double DU, double TU, double Isp, const double* control,
                       double td, double* ds) {
  rhs_prop_ep<double>(s, n, MU, DU, TU, Isp, control, td, ds);
}

/* Single-arc propagation of A1 over `span` from y (12) with the chosen integrator, binary64.
 * Returns integrator status; y overwritten. */
int lto_o_flow_state_costate(double* y, const double* prm, double span, int method, int steps, double rtol, double atol,
                             int* nacc, int* nrej) {
  int bad = 0;
  auto f = [&](const double* a, double* b) { bad |= rhs_state_costate<double>(a, prm, b); };
  int rc = integrate<double>(f, 12, span, method, steps, rtol, atol, y, nullptr, nacc, nrej);
  return bad ? 1 : rc;
}
/* Same in binary128 (fixed-step RKF7(8) with `steps` steps or RK4): converged reference flows. */
int lto_o_flow_state_costate_q(double* y, const double* prm, double span, int method, int steps, double* y_lo) {
  quad yq[12];
  for (int i = 0; i < 12; ++i) yq[i] = y[i];
  int bad = 0;
  auto f = [&](const quad* a, quad* b) { bad |= rhs_state_costate<quad>(a, prm, b); };
  int rc = integrate<quad>(f, 12, (quad)span, method, steps, 0, 0, yq, nullptr, nullptr, nullptr);
  for (int i = 0; i < 12; ++i) { y[i] = (double)yq[i]; if (y_lo) y_lo[i] = (double)(yq[i] - (quad)y[i]); }
  return bad ? 1 : rc;
}
/* Flow + STM of A1 by dual numbers pushed through the integrator (what ForwardDiff.jacobian does at
 * src/multiShoot_CRTBP_indirect.jl:121).  Phi column-major: Phi[c*12 + r] = d y_r(t1) / d y_c(t0). */
int lto_o_flow_stm_state_costate(double* y, const double* prm, double span, int method, int steps, double rtol, double atol,
                                 double* Phi, int* nacc, int* nrej) {
  Dual<12> yd[12];
  for (int i = 0; i < 12; ++i) { yd[i] = Dual<12>(y[i]); yd[i].d[i] = 1.0; }
  int bad = 0;
  auto f = [&](const Dual<12>* a, Dual<12>* b) { bad |= rhs_state_costate<Dual<12>>(a, prm, b); };
  int rc = integrate<Dual<12>>(f, 12, Dual<12>(span), method, steps, rtol, atol, yd, nullptr, nacc, nrej);
  for (int r = 0; r < 12; ++r) { y[r] = yd[r].v; for (int c = 0; c < 12; ++c) Phi[c * 12 + r] = yd[r].d[c]; }
  return bad ? 1 : rc;
}

/* 14-dim extension: RHS, flow and flow+STM (dual numbers), same conventions as the 12-dim calls. */
int lto_o_rhs_state_costate_mass(const double* y, const double* prm, double* dy) { return rhs_state_costate_mass<double>(y, prm, dy); }
int lto_o_flow_state_costate_mass(double* y, const double* prm, double span, int method, int steps, double rtol, double atol) {
  int bad = 0;
  auto f = [&](const double* a, double* b) { bad |= rhs_state_costate_mass<double>(a, prm, b); };
  int rc = integrate<double>(f, 14, span, method, steps, rtol, atol, y, nullptr, nullptr, nullptr);
  return bad ? 1 : rc;
}
int lto_o_flow_stm_state_costate_mass(double* y, const double* prm, double span, int method, int steps, double rtol, double atol,
                                      double* Phi) {
  Dual<14> yd[14];
  for (int i = 0; i < 14; ++i) { yd[i] = Dual<14>(y[i]); yd[i].d[i] = 1.0; }
  int bad = 0;
  auto f = [&](const Dual<14>* a, Dual<14>* b) { bad |= rhs_state_costate_mass<Dual<14>>(a, prm, b); };
  int rc = integrate<Dual<14>>(f, 14, Dual<14>(span), method, steps, rtol, atol, yd, nullptr, nullptr, nullptr);
  for (int r = 0; r < 14; ++r) { y[r] = yd[r].v; for (int c = 0; c < 14; ++c) Phi[c * 14 + r] = yd[r].d[c]; }
  return bad ? 1 : rc;
}
/* sweep: XC [14 x n_nodes], Phi [14 x 14 x S] (or NULL), defect [14 x S] */
int lto_o_indirect14(const double* XC, const double* t, int n_nodes, const double* prm, int method, int steps, double rtol,
                     double atol, double* Phi, double* defect) {
  int status = 0;
#pragma omp parallel for schedule(static) reduction(|:status)
  for (int i = 0; i < n_nodes - 1; ++i) {
    double y[14];
    std::memcpy(y, XC + 14 * i, sizeof y);
    int rc = Phi ? lto_o_flow_stm_state_costate_mass(y, prm, t[i + 1] - t[i], method, steps, rtol, atol, Phi + 196 * i)
                 : lto_o_flow_state_costate_mass(y, prm, t[i + 1] - t[i], method, steps, rtol, atol);
    if (rc) status = rc;
    for (int r = 0; r < 14; ++r) defect[14 * i + r] = y[r] - XC[14 * (i + 1) + r];
  }
  return status;
}

/* thread count for the OpenMP-annotated sweeps (CPU baseline only); returns the count in effect */
int lto_o_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

/* A7: indirect defectCalc, src/multiShoot_CRTBP_indirect.jl:63-90.
 * XC column-major [12 x n_nodes]; defect [12 x (n_nodes-1)]; errors[n_nodes-1] == 0 (:85).
 * method/steps/rtol/atol select the integrator (reference: adaptive order-8 pair @1e-13, :79). */
int lto_o_indirect_defect(const double* XC, const double* t, int n_nodes, const double* prm, int method, int steps,
                          double rtol, double atol, double* defect, double* errors) {
  int status = 0;
  for (int i = 0; i < n_nodes - 1; ++i) {
    double y[12];
    std::memcpy(y, XC + 12 * i, sizeof y);                       /* x0 = XC_all[:,i]  :75 */
    int rc = lto_o_flow_state_costate(y, prm, t[i + 1] - t[i], method, steps, rtol, atol, nullptr, nullptr); /* :76-79 */
    if (rc) status = rc;
    for (int r = 0; r < 12; ++r) defect[12 * i + r] = y[r] - XC[12 * (i + 1) + r]; /* :82 */
    if (errors) errors[i] = 0.0;                                 /* :85 */
  }
  return status;
}

/* A8: indirect jacobianCalc, src/multiShoot_CRTBP_indirect.jl:93-146, compact form only:
 * Phi[i] (column-major 12x12 per segment) = ForwardDiff.jacobian(f, x0) (:121); the caller forms
 * [Phi_i | -I] (:123) and the band scatter (:128-142).  Also returns defect. */
int lto_o_indirect_jacobian(const double* XC, const double* t, int n_nodes, const double* prm, int method, int steps,
                            double rtol, double atol, double* Phi, double* defect) {
  int status = 0;
  /* segments are independent (indirect.jl:116-124); the reference runs them serially -- the OpenMP pragma only
   * serves the all-cores CPU baseline of bench.py (thread count set by lto_o_set_threads, default 1) */
#pragma omp parallel for schedule(static) reduction(|:status)
  for (int i = 0; i < n_nodes - 1; ++i) {
    double y[12];
    std::memcpy(y, XC + 12 * i, sizeof y);
    int rc = lto_o_flow_stm_state_costate(y, prm, t[i + 1] - t[i], method, steps, rtol, atol, Phi + 144 * i, nullptr, nullptr);
    if (rc) status = rc;
    if (defect) for (int r = 0; r < 12; ++r) defect[12 * i + r] = y[r] - XC[12 * (i + 1) + r];
  }
  return status;
}

/* Dense band scatter of the indirect Jacobian, src/multiShoot_CRTBP_indirect.jl:128-142.
 * Jac_full column-major [12S x 12n]; row-block i gets [Phi_i | -I] at columns 12(i-1)+(1:24) (1-based),
 * then columns 1:6 and (end-11):(end-6) are zeroed. */
void lto_o_indirect_scatter_dense(const double* Phi, int n_nodes, double* Jac_full) {
  const int S = n_nodes - 1, R = 12 * S, C = 12 * n_nodes;
  std::memset(Jac_full, 0, sizeof(double) * (size_t)R * C);
  for (int i = 0; i < S; ++i)
    for (int r = 0; r < 12; ++r) {
      for (int c = 0; c < 12; ++c) Jac_full[(size_t)(12 * i + c) * R + 12 * i + r] = Phi[144 * i + c * 12 + r];
      Jac_full[(size_t)(12 * (i + 1) + r) * R + 12 * i + r] = -1.0;
    }
  for (int c = 0; c < 6; ++c) for (int r = 0; r < R; ++r) Jac_full[(size_t)c * R + r] = 0.0;
  for (int c = C - 12; c < C - 6; ++c) for (int r = 0; r < R; ++r) Jac_full[(size_t)c * R + r] = 0.0;
}

/* A4 on one segment with scalar type T: forward half from (x0,u0), backward half from (x1,u1) with
 * velocity flip, same step grid (src/multiShoot_CRTBP_direct.jl:77-105).  hseg = t_{i+1} - t_i. */
} /* extern "C" */

namespace {
template <class T>
double direct_segment(const T* x0, const T* x1, const T* u0, const T* u1, T hseg, int nstate, int nsteps,
                      double MU, double DU, double TU, double Isp, T* defect) {
  T xf[7] = {}, xb[7] = {};
  const T h = (hseg / 2.0) / (double)(nsteps - 1);  /* tspan = LinRange(t_i, t_mid, nsteps)  :70,:84 */
  for (int r = 0; r < nstate; ++r) { xf[r] = x0[r]; xb[r] = x1[r]; }
  for (int r = 3; r < 6; ++r) xb[r] = -xb[r];                                   /* :92 */
  auto ff = [&](const T* a, T* b) { rhs_prop_ep<T>(a, nstate, MU, DU, TU, Isp, u0, +1.0, b); };  /* :85-86 */
  auto fb = [&](const T* a, T* b) { rhs_prop_ep<T>(a, nstate, MU, DU, TU, Isp, u1, -1.0, b); };  /* :93-95 */
  double ef = ode7_8<T>(ff, nstate, h, nsteps, xf);
  double eb = ode7_8<T>(fb, nstate, h, nsteps, xb);
  for (int r = 3; r < 6; ++r) xb[r] = -xb[r];                                   /* :98 */
  for (int r = 0; r < nstate; ++r) defect[r] = xf[r] - xb[r];                   /* :101 */
  return std::max(ef, eb);                                                      /* :104 */
}
} /* namespace */

extern "C" {

/* A4: direct defectCalc, src/multiShoot_CRTBP_direct.jl:66-109.
 * X [nstate x n_nodes], U [3 x n_nodes] column-major; defect [nstate x S]; errors [S]. */
void lto_o_direct_defect(const double* X, const double* U, const double* t, int nstate, int n_nodes, int nsteps,
                         double MU, double DU, double TU, double Isp, double* defect, double* errors) {
  for (int i = 0; i < n_nodes - 1; ++i) {
    double d[7];
    double e = direct_segment<double>(X + nstate * i, X + nstate * (i + 1), U + 3 * i, U + 3 * (i + 1), t[i + 1] - t[i],
                                      nstate, nsteps, MU, DU, TU, Isp, d);
    for (int r = 0; r < nstate; ++r) defect[nstate * i + r] = d[r];
    if (errors) errors[i] = e;
  }
}

/* A5: direct jacobianCalc by forward finite differences, src/multiShoot_CRTBP_direct.jl:111-143,
 * compact form: Jac_temp block i is column-major [nstate x nvar], nvar = 2(nstate+3), variable order
 * XU = [x_i; x_{i+1}; u_i; u_{i+1}] (:125).  `defect` is the nominal defect (:140). */
void lto_o_direct_jacobian_fd(const double* X, const double* U, const double* t, const double* defect, int nstate,
                              int n_nodes, int nsteps, double MU, double DU, double TU, double Isp, double pert,
                              double* Jac_temp) {
  const int nvar = 2 * (nstate + 3);
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n_nodes - 1; ++i) {
    double XU[20];
    for (int r = 0; r < 2 * nstate; ++r) XU[r] = X[nstate * i + r];
    for (int r = 0; r < 6; ++r) XU[2 * nstate + r] = U[3 * i + r];
    for (int j = 0; j < nvar; ++j) {
      double M[20], d[7];
      std::memcpy(M, XU, sizeof(double) * nvar);
      M[j] = M[j] + pert;                                                        /* :129 */
      direct_segment<double>(M, M + nstate, M + 2 * nstate, M + 2 * nstate + 3, t[i + 1] - t[i], nstate, nsteps,
                             MU, DU, TU, Isp, d);                                /* :136 */
      for (int r = 0; r < nstate; ++r)
        Jac_temp[(size_t)i * nstate * nvar + j * nstate + r] = (d[r] - defect[nstate * i + r]) / pert; /* :140 */
    }
  }
}

/* Exact derivative of the discrete direct defect map by dual numbers (not in the reference; used to
 * check the HIP variational-equation Jacobian to ~1e-12).  Same layout as Jac_temp; also returns
 * d defect / d hseg per segment (nstate x S) for the tf column. */
void lto_o_direct_jacobian_dual(const double* X, const double* U, const double* t, int nstate, int n_nodes, int nsteps,
                                double MU, double DU, double TU, double Isp, double* Jac_temp, double* ddefect_dh,
                                double* defect) {
  const int nvar = 2 * (nstate + 3);
  typedef Dual<21> D;
  for (int i = 0; i < n_nodes - 1; ++i) {
    D XU[20], d[7];
    for (int r = 0; r < 2 * nstate; ++r) XU[r] = D(X[nstate * i + r]);
    for (int r = 0; r < 6; ++r) XU[2 * nstate + r] = D(U[3 * i + r]);
    for (int j = 0; j < nvar; ++j) XU[j].d[j] = 1.0;
    D h(t[i + 1] - t[i]);
    h.d[20] = 1.0;
    direct_segment<D>(XU, XU + nstate, XU + 2 * nstate, XU + 2 * nstate + 3, h, nstate, nsteps, MU, DU, TU, Isp, d);
    for (int r = 0; r < nstate; ++r) {
      for (int j = 0; j < nvar; ++j) Jac_temp[(size_t)i * nstate * nvar + j * nstate + r] = d[r].d[j];
      if (ddefect_dh) ddefect_dh[nstate * i + r] = d[r].d[20];
      if (defect) defect[nstate * i + r] = d[r].v;
    }
  }
}

/* A6: tf partial by central differences, src/multiShoot_CRTBP_direct.jl:503-516. */
void lto_o_direct_dtf_fd(const double* X, const double* U, const double* t, int nstate, int n_nodes, int nsteps,
                         double MU, double DU, double TU, double Isp, double pert_tf, double* ddefect_dt) {
  const int S = n_nodes - 1;
  std::vector<double> t1(n_nodes), t2(n_nodes), d1((size_t)nstate * S), d2((size_t)nstate * S);
  const double t0 = t[0], tf = t[n_nodes - 1];
  for (int k = 0; k < n_nodes; ++k) {
    double tau = (t[k] - t0) / (tf - t0) * 2.0 - 1.0;                          /* :479 */
    t1[k] = t0 + (tau + 1.0) / 2.0 * ((tf + pert_tf) - t0);                    /* :509 */
    t2[k] = t0 + (tau + 1.0) / 2.0 * ((tf - pert_tf) - t0);                    /* :510 */
  }
  lto_o_direct_defect(X, U, t1.data(), nstate, n_nodes, nsteps, MU, DU, TU, Isp, d1.data(), nullptr);
  lto_o_direct_defect(X, U, t2.data(), nstate, n_nodes, nsteps, MU, DU, TU, Isp, d2.data(), nullptr);
  for (size_t k = 0; k < (size_t)nstate * S; ++k) ddefect_dt[k] = (d1[k] - d2[k]) / (2.0 * pert_tf); /* :514 */
}

/* Dense band scatter of the direct Jacobian, src/multiShoot_CRTBP_direct.jl:146-162 (+ tf column :516).
 * Jac_full column-major [nstate*S x n(nstate+3)+1]. */
void lto_o_direct_scatter_dense(const double* Jac_temp, const double* ddefect_dt, int nstate, int n_nodes, double* Jac_full) {
  const int S = n_nodes - 1, R = nstate * S, C = n_nodes * (nstate + 3) + 1, nvar = 2 * (nstate + 3);
  std::memset(Jac_full, 0, sizeof(double) * (size_t)R * C);
  for (int i = 0; i < S; ++i)
    for (int r = 0; r < nstate; ++r) {
      const int row = nstate * i + r;
      for (int j = 0; j < 2 * nstate; ++j) Jac_full[(size_t)(nstate * i + j) * R + row] = Jac_temp[(size_t)i * nstate * nvar + j * nstate + r];
      for (int j = 0; j < 6; ++j) Jac_full[(size_t)(nstate * n_nodes + 3 * i + j) * R + row] = Jac_temp[(size_t)i * nstate * nvar + (2 * nstate + j) * nstate + r];
      Jac_full[(size_t)(C - 1) * R + row] = ddefect_dt ? ddefect_dt[row] : 0.0;
    }
}

/* Ballistic / given-thrust single arc of A2 (for halo-file and Jacobi-constant KATs). */
double lto_o_flow_prop_ep(double* x, int nstate, const double* control, double td, double span, int method, int steps,
                          double rtol, double atol, double MU, double DU, double TU, double Isp) {
  double err = 0.0;
  auto f = [&](const double* a, double* b) { rhs_prop_ep<double>(a, nstate, MU, DU, TU, Isp, control, td, b); };
  integrate<double>(f, nstate, span, method, steps, rtol, atol, x, &err, nullptr, nullptr);
  return err;
}

} /* extern "C" */
