// dynamics.hpp -- CRTBP right-hand sides and their variational coefficients, gfx950 device code.
//
// fp64 VALU only (no MFMA: every product here is a register-resident 3x3 against a 3-vector).
// Formulas: reference src/CRTBP_stateCostate_deriv.jl:9-90 (state+costate, "A1") and
// src/CRTBP_prop_EP_deriv.jl:8-61 (given thrust, "A2"), restated through the symmetric gravity
// gradient G (the reference's longhand costate rows :83-85 are exactly -G*lambda_v).  The variational
// coefficients (H = d(G lambda_v)/dr, U = du/dlambda_v) do not exist in the reference, which
// differentiates by ForwardDiff / finite differences (multiShoot_CRTBP_indirect.jl:121,
// multiShoot_CRTBP_direct.jl:123-143); they are the same mathematical object.
#pragma once
#include <hip/hip_runtime.h>

namespace lto {

// Control-law modes of CRTBP_stateCostate_deriv! (stateCostate_deriv.jl:36-53).
enum PMode : int { PM_P0 = 0, PM_P1 = 1, PM_P2 = 2, PM_PGEN = 3, PM_NCLASS = 4 };
// Not a class: "whatever law the trajectory has", a run-time switch over the four class bodies (control_dispatch).  For kernels that
// HBM bounds (kernels_indirect_stream.hip): there a kernel per class buys no time and costs four instantiations per dimension.
constexpr int PM_ANY = PM_NCLASS;
// Control-law class of an exponent p (valid p only: 0, 1 or > 1).  Every kernel is compiled for ONE class, so the law
// is straight-line code; a batch that mixes classes is swept by one launch per class present, each launch skipping the
// other classes' trajectories (IndirectArgs::class_filter).
__host__ __device__ inline int p_class(double p) { return (p == 1.0) ? PM_P1 : (p == 2.0) ? PM_P2 : (p == 0.0) ? PM_P0 : PM_PGEN; }

// Per-trajectory constants, precomputed on the host in the reference's operation order.
struct TrajParams {
  double accel_limit;  // thrustLimit / mass / 1e3 * TU^2 / DU        (stateCostate_deriv.jl:33); 12-dim only
  double inv_2rho;     // 1 / (2 rho)                                   (:43)
  double inv_rho;      // 1 / rho   (d umag / d|lambda_v| for p = 1 is accelLimit/rho * e q^2)
  double p;            // control-law exponent
  double inv_p;        // 1 / p
  double inv_pm1;      // 1 / (p - 1)   (p > 1 only)
  double omega;        // time_direction
  double MU;           // CRTBP mass ratio
  double cT;           // thrustLimit / 1e3 * TU^2 / DU: accelLimit = cT / m for the 14-dim (variable mass) system
  double kappa_td;     // time_direction * 1e3 * DU / (TU * Isp * 9.81): mdot = -kappa_td * umag * m  (14-dim)
};

// 1/sqrt(x) to ~1 ulp: v_rsq_f64 seed (relative error <~ 2^-26) + ONE third-order step
//   y <- y (1 + e/2 + 3 e^2/8),  e = 1 - x y^2      (residual error ~ 5/16 e^3)
// Arguments here are squared distances / squared norms of O(1e-6 .. 1e2): no range scaling needed.
__device__ __forceinline__ double rsqrt_nr(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-x * y, y, 1.0);
  const double t = __builtin_fma(0.375, e, 0.5);
  return __builtin_fma(y * e, t, y);
}
// 1/x to ~1 ulp: v_rcp_f64 seed + one third-order step  y <- y (1 + e + e^2), e = 1 - x y.
__device__ __forceinline__ double rcp_nr(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, __builtin_fma(e, e, e), y);
}
// Taylor coefficients 1/k!, k = 13 .. 3, of the exp polynomials (exp_neg's Horner branch, exp_mid).  They live in constant memory on purpose: a scalar
// load puts them into SGPR pairs, and `fma(p, r, <sgpr>)` is then a three-address v_fma_f64 the scheduler can interleave
// freely.  (As literals hipcc materialises them in VGPRs and turns part of the Horner chain into v_mov_b64 + v_fmac_f64,
// doubling its issue cost; inline asm with "s" operands fixes the encoding but fences the scheduler: the 13 dependent
// FMAs then run back to back at 8.5 ticks each instead of 5.2 interleaved with the gravity terms.)
__constant__ double kExpTaylor[11] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
                                      1.0 / 5040.0,       1.0 / 720.0,       1.0 / 120.0,      1.0 / 24.0,      1.0 / 6.0};

// exp(z) for z <= 0 to ~1 ulp without the special-case handling of the general routine:
// z = n ln2 + r, |r| <= ln2/2, degree-13 Taylor polynomial (remainder 4e-18), scaled by 2^n (underflows to 0).
// SHORT = false: Horner (13 dependent FMAs, fewest instructions: the per-lane kernels have other work to overlap).
// SHORT = true: four interleaved chains in r^4 (15 instructions, dependent depth 7) for the pipeline kernel's base wave,
// whose instruction stream IS the sweep's critical path.
template <bool SHORT = false>
__device__ __forceinline__ double exp_neg(double z) {
  z = fmax(z, -800.0);  // exp(-800) == 0 in binary64; keeps the reduction finite for rho -> 0
  const double n = __builtin_rint(z * 1.4426950408889634);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, z);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double p;
  if (SHORT) {
    const double r2 = r * r, r4 = r2 * r2;
    double p0 = __builtin_fma(1.0 / 479001600.0, r4, 1.0 / 40320.0);     // 1, r^4/4!, r^8/8!, r^12/12!
    double p1 = __builtin_fma(1.0 / 6227020800.0, r4, 1.0 / 362880.0);   // r, r^5/5!, r^9/9!, r^13/13!
    double p2 = __builtin_fma(1.0 / 3628800.0, r4, 1.0 / 720.0);         // r^2/2!, r^6/6!, r^10/10!
    double p3 = __builtin_fma(1.0 / 39916800.0, r4, 1.0 / 5040.0);       // r^3/3!, r^7/7!, r^11/11!
    p0 = __builtin_fma(p0, r4, 1.0 / 24.0);
    p1 = __builtin_fma(p1, r4, 1.0 / 120.0);
    p2 = __builtin_fma(p2, r4, 0.5);
    p3 = __builtin_fma(p3, r4, 1.0 / 6.0);
    p0 = __builtin_fma(p0, r4, 1.0);
    p1 = __builtin_fma(p1, r4, 1.0);
    const double lo = __builtin_fma(p1, r, p0), hi = __builtin_fma(p3, r, p2);
    p = __builtin_fma(hi, r2, lo);
  } else {
    p = kExpTaylor[0];
#pragma unroll
    for (int k = 1; k < 11; ++k) p = __builtin_fma(p, r, kExpTaylor[k]);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
  }
  return __builtin_ldexp(p, (int)n);
}

// Coefficients one STM column needs at an RK stage of the 12-dim system:
//   F = [0 I 0 0; G 2wJ 0 U; -H 0 0 -G; 0 0 -I 2wJ],  U d = -ua d + ub (l.d) l.
struct VarCoef12 {
  double Gxx, Gyy, Gzz, Gxy, Gxz, Gyz;
  double Hxx, Hyy, Hzz, Hxy, Hxz, Hyz;
  double ua, ub, lx, ly, lz;
};

// Thrust magnitude umag(n), n = |lambda_v|, for accelLimit aL, and (if VAR) un = d umag / dn.
//   m = umag, ua = umag / n, ub = ua - un  (so that  U d = -ua d + ub (lhat.d) lhat),
//   tlim = true when umag is proportional to aL (every law except the unclamped p > 1 branch).
// `inv_n` is 1/n (0 when n == 0, the reference's NaN guard :59-64).
template <int PM, bool VAR, bool SHORT = false>
__device__ __forceinline__ void control_law(const TrajParams& tp, const double aL, double n, double inv_n, double& m,
                                            double& ua, double& ub, double& un, bool& tlim) {
  tlim = true;
  un = 0.0;
  if (PM == PM_P0) {  // :36-39  umag = accelLimit
    m = aL;
    ua = aL * inv_n;
    ub = ua;
  } else if (PM == PM_P1) {  // :41-43  umag = 1/2 (1 + tanh(g / (2 rho))) accelLimit, g = n - 1
    // 1/2 (1 + tanh x) = 1 / (1 + exp(-2x)): evaluated without cancellation or overflow for any rho.
    const double x = (n - 1.0) * tp.inv_2rho;
    const double e = exp_neg<SHORT>(-2.0 * fabs(x));
    const double q = rcp_nr(1.0 + e);
    const double sig = (x >= 0.0) ? q : e * q;
    m = aL * sig;
    ua = m * inv_n;
    if (VAR) un = (aL * tp.inv_rho) * (e * q) * q;   // aL/(4 rho) sech^2 x = (aL / rho) e q^2
    ub = ua - un;
  } else if (PM == PM_P2) {  // :45-50 with p = 2: umag = n / 2, clamped at accelLimit
    const double mu = 0.5 * n;
    if (mu > aL) { m = aL; ua = aL * inv_n; ub = ua; }
    else { m = mu; ua = 0.5; ub = 0.0; un = 0.5; tlim = false; }  // u = -lambda_v / 2 => U = -I / 2 (also at n = 0)
  } else {  // PM_PGEN :45-50  umag = (n / p)^(1 / (p - 1)), clamped
    const double mu = pow(tp.inv_p * n, tp.inv_pm1);
    if (mu > aL) { m = aL; ua = aL * inv_n; ub = ua; }
    else { m = mu; ua = mu * inv_n; un = ua * tp.inv_pm1; ub = ua - un; tlim = false; }  // un = umag / ((p-1) n)
  }
}

template <int PM, bool VAR, bool SHORT = false>
__device__ __forceinline__ void control_dispatch(const TrajParams& tp, const double aL, double n, double inv_n, double& m,
                                                 double& ua, double& ub, double& un, bool& tlim) {
  static_assert(PM >= PM_P0 && PM <= PM_ANY, "kernels are compiled per control-law class (or PM_ANY)");
  if constexpr (PM == PM_ANY) {
    // the law of THIS trajectory, chosen at run time: the same control_law<class> code, so the same bits as the per-class kernels
    switch (p_class(tp.p)) {
      case PM_P0: control_law<PM_P0, VAR, SHORT>(tp, aL, n, inv_n, m, ua, ub, un, tlim); break;
      case PM_P1: control_law<PM_P1, VAR, SHORT>(tp, aL, n, inv_n, m, ua, ub, un, tlim); break;
      case PM_P2: control_law<PM_P2, VAR, SHORT>(tp, aL, n, inv_n, m, ua, ub, un, tlim); break;
      default: control_law<PM_PGEN, VAR, SHORT>(tp, aL, n, inv_n, m, ua, ub, un, tlim); break;
    }
  } else {
    control_law<PM, VAR, SHORT>(tp, aL, n, inv_n, m, ua, ub, un, tlim);
  }
}

// A1: ydot for y = (r, v, lambda_r, lambda_v); optionally the column coefficients.
// y, dy are fully unrolled register arrays.
template <int PM, bool VAR>
__device__ __forceinline__ void rhs12(const double (&y)[12], const TrajParams& tp, double (&dy)[12], VarCoef12& vc) {
  const double MU = tp.MU;
  const double x = y[0], yy = y[1], z = y[2];
  const double w2 = 2.0 * tp.omega;
  const double a = x + MU, b = a - 1.0;
  const double yz2 = __builtin_fma(yy, yy, z * z);
  const double d1 = __builtin_fma(a, a, yz2), d2 = __builtin_fma(b, b, yz2);
  const double i1 = rsqrt_nr(d1), i2 = rsqrt_nr(d2);
  const double i1s = i1 * i1, i2s = i2 * i2;
  const double c1 = (1.0 - MU) * (i1s * i1), c2 = MU * (i2s * i2);
  const double cs = c1 + c2;
  const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
  const double ee = e1 + e2;
  const double sa = e1 * a, tb = e2 * b;
  const double st = sa + tb;
  // gravity-gradient + centrifugal matrix G (symmetric)
  const double Gxx = __builtin_fma(sa, a, __builtin_fma(tb, b, 1.0 - cs));
  const double Gyy = __builtin_fma(ee * yy, yy, 1.0 - cs);
  const double Gzz = __builtin_fma(ee * z, z, -cs);
  const double Gxy = st * yy, Gxz = st * z, Gyz = ee * yy * z;

  // control  u = -m(n) lambda_v / n
  const double lx0 = y[9], ly0 = y[10], lz0 = y[11];
  const double n2 = __builtin_fma(lx0, lx0, __builtin_fma(ly0, ly0, lz0 * lz0));
  const double inv_n = (n2 > 0.0) ? rsqrt_nr(n2) : 0.0;
  const double n = n2 * inv_n;
  double m, ua, ub, un;
  bool tlim;
  control_dispatch<PM, VAR>(tp, tp.accel_limit, n, inv_n, m, ua, ub, un, tlim);
  const double lhx = lx0 * inv_n, lhy = ly0 * inv_n, lhz = lz0 * inv_n;

  dy[0] = y[3]; dy[1] = y[4]; dy[2] = y[5];                                   // :78
  dy[3] = __builtin_fma(-c1, a, __builtin_fma(-c2, b, __builtin_fma(w2, y[4], x))) - m * lhx;   // :79
  dy[4] = __builtin_fma(-cs, yy, __builtin_fma(-w2, y[3], yy)) - m * lhy;                       // :80
  dy[5] = __builtin_fma(-cs, z, -m * lhz);                                                     // :81
  // lambda_r dot = -G lambda_v                                                                 // :83-85
  dy[6] = -__builtin_fma(Gxx, lx0, __builtin_fma(Gxy, ly0, Gxz * lz0));
  dy[7] = -__builtin_fma(Gxy, lx0, __builtin_fma(Gyy, ly0, Gyz * lz0));
  dy[8] = -__builtin_fma(Gxz, lx0, __builtin_fma(Gyz, ly0, Gzz * lz0));
  dy[9] = __builtin_fma(w2, ly0, -y[6]);                                      // :86
  dy[10] = __builtin_fma(-w2, lx0, -y[7]);                                    // :87
  dy[11] = -y[8];                                                             // :88

  if (VAR) {
    vc.Gxx = Gxx; vc.Gyy = Gyy; vc.Gzz = Gzz; vc.Gxy = Gxy; vc.Gxz = Gxz; vc.Gyz = Gyz;
    vc.ua = ua; vc.ub = ub; vc.lx = lhx; vc.ly = lhy; vc.lz = lhz;
    // H = d(G lambda_v)/dr = sum_bodies e (s I + rho l^T + l rho^T) - f s rho rho^T,
    //   e = 3 kappa / d^{5/2}, f = 5 e / d, s = rho . lambda_v
    const double s1 = __builtin_fma(a, lx0, __builtin_fma(yy, ly0, z * lz0));
    const double s2 = __builtin_fma(b, lx0, __builtin_fma(yy, ly0, z * lz0));
    const double q1 = 5.0 * e1 * i1s * s1, q2 = 5.0 * e2 * i2s * s2;
    const double es = __builtin_fma(e1, s1, e2 * s2);   // e1 s1 + e2 s2
    const double qq = q1 + q2;
    const double qa = __builtin_fma(q1, a, q2 * b);     // q1 a + q2 b
    // diagonal: e(s + 2 rho_i l_i) - q rho_i^2
    vc.Hxx = es + 2.0 * st * lx0 - __builtin_fma(q1 * a, a, q2 * b * b);
    vc.Hyy = es + 2.0 * ee * yy * ly0 - qq * yy * yy;
    vc.Hzz = es + 2.0 * ee * z * lz0 - qq * z * z;
    // off-diagonal: e(rho_i l_j + rho_j l_i) - q rho_i rho_j
    vc.Hxy = __builtin_fma(st, ly0, ee * yy * lx0) - qa * yy;
    vc.Hxz = __builtin_fma(st, lz0, ee * z * lx0) - qa * z;
    vc.Hyz = ee * __builtin_fma(yy, lz0, z * ly0) - qq * yy * z;
  }
}

// Fused base + ONE STM column (the COLS = 1 mapping used while the chip is not yet full): G and H are
// applied through their dyadic structure  G = (1-cs) D - ... , H a = es a + (l.a)(e1 rho1 + e2 rho2) + (u1+u2) l - ...
// instead of being built entry by entry, which saves ~40 fp64 instructions per RK stage per lane.
//   y = (base[12], column[12]) -> k = (base_dot[12], column_dot[12])
template <int PM>
__device__ __forceinline__ void rhs12_fused1(const double (&y)[24], const TrajParams& tp, const double w2, double (&k)[24]) {
  const double MU = tp.MU;
  const double x = y[0], yy = y[1], z = y[2];
  const double A = x + MU, B = A - 1.0;
  const double yz2 = __builtin_fma(yy, yy, z * z);
  const double d1 = __builtin_fma(A, A, yz2), d2 = __builtin_fma(B, B, yz2);
  const double i1 = rsqrt_nr(d1), i2 = rsqrt_nr(d2);
  const double i1s = i1 * i1, i2s = i2 * i2;
  const double c1 = (1.0 - MU) * (i1s * i1), c2 = MU * (i2s * i2);
  const double cs = c1 + c2, omc = 1.0 - cs;
  const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
  const double ee = e1 + e2;
  const double st = __builtin_fma(e1, A, e2 * B);
  const double eey = ee * yy, eez = ee * z;

  const double lx = y[9], ly = y[10], lz = y[11];
  const double n2 = __builtin_fma(lx, lx, __builtin_fma(ly, ly, lz * lz));
  const double inv_n = (n2 > 0.0) ? rsqrt_nr(n2) : 0.0;
  const double n = n2 * inv_n;
  double m, ua, ub, un;
  bool tlim;
  control_dispatch<PM, true>(tp, tp.accel_limit, n, inv_n, m, ua, ub, un, tlim);
  const double lhx = lx * inv_n, lhy = ly * inv_n, lhz = lz * inv_n;

  // s_b = rho_b . lambda_v ;  G lambda_v = ((1-cs) lx + tA, (1-cs) ly + es y, -cs lz + es z)
  const double yzl = __builtin_fma(yy, ly, z * lz);
  const double s1 = __builtin_fma(A, lx, yzl), s2 = __builtin_fma(B, lx, yzl);
  const double t1 = e1 * s1, t2 = e2 * s2;
  const double es = t1 + t2;
  const double tA = __builtin_fma(t1, A, t2 * B);

  k[0] = y[3]; k[1] = y[4]; k[2] = y[5];
  k[3] = __builtin_fma(-c1, A, __builtin_fma(-c2, B, __builtin_fma(w2, y[4], x))) - m * lhx;
  k[4] = __builtin_fma(-cs, yy, __builtin_fma(-w2, y[3], yy)) - m * lhy;
  k[5] = __builtin_fma(-cs, z, -m * lhz);
  k[6] = -__builtin_fma(omc, lx, tA);
  k[7] = -__builtin_fma(omc, ly, es * yy);
  k[8] = -__builtin_fma(-cs, lz, es * z);
  k[9] = __builtin_fma(w2, ly, -y[6]);
  k[10] = __builtin_fma(-w2, lx, -y[7]);
  k[11] = -y[8];

  // ---- column c = (a, b, g, d)
  const double ax = y[12], ay = y[13], az = y[14];
  const double dx = y[21], dyv = y[22], dz = y[23];
  // G a
  const double yza = __builtin_fma(yy, ay, z * az);
  const double r1a = __builtin_fma(A, ax, yza), r2a = __builtin_fma(B, ax, yza);
  const double u1 = e1 * r1a, u2 = e2 * r2a;
  const double us = u1 + u2;
  const double uA = __builtin_fma(u1, A, u2 * B);
  const double Gax = __builtin_fma(omc, ax, uA), Gay = __builtin_fma(omc, ay, us * yy), Gaz = __builtin_fma(-cs, az, us * z);
  // G d
  const double yzd = __builtin_fma(yy, dyv, z * dz);
  const double r1d = __builtin_fma(A, dx, yzd), r2d = __builtin_fma(B, dx, yzd);
  const double v1 = e1 * r1d, v2 = e2 * r2d;
  const double vs = v1 + v2;
  const double vA = __builtin_fma(v1, A, v2 * B);
  const double Gdx = __builtin_fma(omc, dx, vA), Gdy = __builtin_fma(omc, dyv, vs * yy), Gdz = __builtin_fma(-cs, dz, vs * z);
  // H a = es a + (l.a)(st, ee y, ee z) + us l - (wA, ws y, ws z),  w_b = 5 e_b i_b^2 s_b (rho_b . a)
  const double la = __builtin_fma(lx, ax, __builtin_fma(ly, ay, lz * az));
  const double w1 = (5.0 * i1s) * (t1 * r1a), w2b = (5.0 * i2s) * (t2 * r2a);
  const double ws = w1 + w2b;
  const double wA = __builtin_fma(w1, A, w2b * B);
  const double Hax = __builtin_fma(es, ax, __builtin_fma(la, st, __builtin_fma(us, lx, -wA)));
  const double Hay = __builtin_fma(es, ay, __builtin_fma(la, eey, __builtin_fma(us, ly, -ws * yy)));
  const double Haz = __builtin_fma(es, az, __builtin_fma(la, eez, __builtin_fma(us, lz, -ws * z)));
  // U d = -ua d + ub (lhat . d) lhat
  const double ld = __builtin_fma(lhx, dx, __builtin_fma(lhy, dyv, lhz * dz));
  const double tl = ub * ld;
  k[12] = y[15]; k[13] = y[16]; k[14] = y[17];
  k[15] = Gax + __builtin_fma(w2, y[16], __builtin_fma(-ua, dx, tl * lhx));
  k[16] = Gay + __builtin_fma(-w2, y[15], __builtin_fma(-ua, dyv, tl * lhy));
  k[17] = Gaz + __builtin_fma(-ua, dz, tl * lhz);
  k[18] = -(Hax + Gdx);
  k[19] = -(Hay + Gdy);
  k[20] = -(Haz + Gdz);
  k[21] = __builtin_fma(w2, dyv, -y[18]);
  k[22] = __builtin_fma(-w2, dx, -y[19]);
  k[23] = -y[20];
}

// One STM column c = (a, b, g, d) of the 12-dim system: cdot = F c.
__device__ __forceinline__ void var_col12(const VarCoef12& vc, const double w2, const double (&c)[12], double (&dc)[12]) {
  const double ax = c[0], ay = c[1], az = c[2];
  const double dx = c[9], dyv = c[10], dz = c[11];
  dc[0] = c[3]; dc[1] = c[4]; dc[2] = c[5];
  const double ld = __builtin_fma(vc.lx, dx, __builtin_fma(vc.ly, dyv, vc.lz * dz));
  const double tl = vc.ub * ld;
  // b dot = G a + 2w J b + U d
  dc[3] = __builtin_fma(vc.Gxx, ax, __builtin_fma(vc.Gxy, ay, __builtin_fma(vc.Gxz, az,
          __builtin_fma(w2, c[4], __builtin_fma(-vc.ua, dx, tl * vc.lx)))));
  dc[4] = __builtin_fma(vc.Gxy, ax, __builtin_fma(vc.Gyy, ay, __builtin_fma(vc.Gyz, az,
          __builtin_fma(-w2, c[3], __builtin_fma(-vc.ua, dyv, tl * vc.ly)))));
  dc[5] = __builtin_fma(vc.Gxz, ax, __builtin_fma(vc.Gyz, ay, __builtin_fma(vc.Gzz, az,
          __builtin_fma(-vc.ua, dz, tl * vc.lz))));
  // g dot = -H a - G d
  dc[6] = -__builtin_fma(vc.Hxx, ax, __builtin_fma(vc.Hxy, ay, __builtin_fma(vc.Hxz, az,
           __builtin_fma(vc.Gxx, dx, __builtin_fma(vc.Gxy, dyv, vc.Gxz * dz)))));
  dc[7] = -__builtin_fma(vc.Hxy, ax, __builtin_fma(vc.Hyy, ay, __builtin_fma(vc.Hyz, az,
           __builtin_fma(vc.Gxy, dx, __builtin_fma(vc.Gyy, dyv, vc.Gyz * dz)))));
  dc[8] = -__builtin_fma(vc.Hxz, ax, __builtin_fma(vc.Hyz, ay, __builtin_fma(vc.Hzz, az,
           __builtin_fma(vc.Gxz, dx, __builtin_fma(vc.Gyz, dyv, vc.Gzz * dz)))));
  // d dot = -g + 2w J d
  dc[9] = __builtin_fma(w2, dyv, -c[6]);
  dc[10] = __builtin_fma(-w2, dx, -c[7]);
  dc[11] = -c[8];
}

// ------------------------------------------------------------------------------ 14-dim extension
// y = (r, v, m, lambda_r, lambda_v, lambda_m): CRTBP state+costate with mass flow and mass costate (BASELINE
// configs[1]).  The reference has NO such RHS (its CRTBP system is 12-dim, constant mass); the model follows
// GeneralCode/twoBody_stateCostate_mass_deriv.jl:11-78 re-expressed in CRTBP units, DESIGN.md "14-dim extension":
//   aL = cT / m,  umag = law(|lambda_v|) (CRTBP form),  u = -umag lhat,  mdot = -kappa_td umag m,
//   lambda_m_dot = -dH/dm = -umag n / m (thrust-limited laws)  or  kappa_td lambda_m umag (unclamped p > 1).
struct VarCoef14 {
  VarCoef12 c;              // G, H, U as in the 12-dim system
  double umx, umy, umz;     // d u / d m = -um lhat,  um = d umag / d m
  double mm, mn;            // d mdot / d m,  d mdot / d n   (n = |lambda_v|; d/d lambda_v = (.) lhat^T)
  double Lm, Ln, Ll;        // d lambda_m_dot / d m, / d n, / d lambda_m
};

template <int PM, bool VAR>
__device__ __forceinline__ void rhs14(const double (&y)[14], const TrajParams& tp, double (&dy)[14], VarCoef14& vc) {
  const double MU = tp.MU;
  const double x = y[0], yy = y[1], z = y[2], mass = y[6];
  const double w2 = 2.0 * tp.omega;
  const double a = x + MU, b = a - 1.0;
  const double yz2 = __builtin_fma(yy, yy, z * z);
  const double d1 = __builtin_fma(a, a, yz2), d2 = __builtin_fma(b, b, yz2);
  const double i1 = rsqrt_nr(d1), i2 = rsqrt_nr(d2);
  const double i1s = i1 * i1, i2s = i2 * i2;
  const double c1 = (1.0 - MU) * (i1s * i1), c2 = MU * (i2s * i2);
  const double cs = c1 + c2;
  const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
  const double ee = e1 + e2;
  const double sa = e1 * a, tb = e2 * b;
  const double st = sa + tb;
  const double Gxx = __builtin_fma(sa, a, __builtin_fma(tb, b, 1.0 - cs));
  const double Gyy = __builtin_fma(ee * yy, yy, 1.0 - cs);
  const double Gzz = __builtin_fma(ee * z, z, -cs);
  const double Gxy = st * yy, Gxz = st * z, Gyz = ee * yy * z;

  const double lx0 = y[10], ly0 = y[11], lz0 = y[12], lm = y[13];
  const double n2 = __builtin_fma(lx0, lx0, __builtin_fma(ly0, ly0, lz0 * lz0));
  const double inv_n = (n2 > 0.0) ? rsqrt_nr(n2) : 0.0;
  const double n = n2 * inv_n;
  const double inv_m = rcp_nr(mass);
  const double aL = tp.cT * inv_m;
  double m, ua, ub, un;
  bool tlim;
  control_dispatch<PM, true>(tp, aL, n, inv_n, m, ua, ub, un, tlim);
  const double lhx = lx0 * inv_n, lhy = ly0 * inv_n, lhz = lz0 * inv_n;
  const double kt = tp.kappa_td;

  dy[0] = y[3]; dy[1] = y[4]; dy[2] = y[5];
  dy[3] = __builtin_fma(-c1, a, __builtin_fma(-c2, b, __builtin_fma(w2, y[4], x))) - m * lhx;
  dy[4] = __builtin_fma(-cs, yy, __builtin_fma(-w2, y[3], yy)) - m * lhy;
  dy[5] = __builtin_fma(-cs, z, -m * lhz);
  dy[6] = -kt * m * mass;
  dy[7] = -__builtin_fma(Gxx, lx0, __builtin_fma(Gxy, ly0, Gxz * lz0));
  dy[8] = -__builtin_fma(Gxy, lx0, __builtin_fma(Gyy, ly0, Gyz * lz0));
  dy[9] = -__builtin_fma(Gxz, lx0, __builtin_fma(Gyz, ly0, Gzz * lz0));
  dy[10] = __builtin_fma(w2, ly0, -y[7]);
  dy[11] = __builtin_fma(-w2, lx0, -y[8]);
  dy[12] = -y[9];
  // tl = 1 for thrust-limited laws (umag ~ 1/m), 0 for the unclamped p > 1 law (umag independent of m); blended
  // arithmetically so that no per-lane boolean stays live across the (register-starved) 13-stage integrators.
  const double tl = tlim ? 1.0 : 0.0, ntl = 1.0 - tl;
  const double mn_over_m = (m * n) * inv_m;
  dy[13] = __builtin_fma(-tl, mn_over_m, ntl * (kt * lm * m));

  if (VAR) {
    VarCoef12& g = vc.c;
    g.Gxx = Gxx; g.Gyy = Gyy; g.Gzz = Gzz; g.Gxy = Gxy; g.Gxz = Gxz; g.Gyz = Gyz;
    g.ua = ua; g.ub = ub; g.lx = lhx; g.ly = lhy; g.lz = lhz;
    const double s1 = __builtin_fma(a, lx0, __builtin_fma(yy, ly0, z * lz0));
    const double s2 = __builtin_fma(b, lx0, __builtin_fma(yy, ly0, z * lz0));
    const double q1 = 5.0 * e1 * i1s * s1, q2 = 5.0 * e2 * i2s * s2;
    const double es = __builtin_fma(e1, s1, e2 * s2);
    const double qq = q1 + q2;
    const double qa = __builtin_fma(q1, a, q2 * b);
    g.Hxx = es + 2.0 * st * lx0 - __builtin_fma(q1 * a, a, q2 * b * b);
    g.Hyy = es + 2.0 * ee * yy * ly0 - qq * yy * yy;
    g.Hzz = es + 2.0 * ee * z * lz0 - qq * z * z;
    g.Hxy = __builtin_fma(st, ly0, ee * yy * lx0) - qa * yy;
    g.Hxz = __builtin_fma(st, lz0, ee * z * lx0) - qa * z;
    g.Hyz = ee * __builtin_fma(yy, lz0, z * ly0) - qq * yy * z;
    // thrust-limited: um = d umag/dm = -umag/m and um m + umag = 0;  otherwise um = 0
    const double um_neg = tl * (m * inv_m);                  // -um
    vc.umx = um_neg * lhx; vc.umy = um_neg * lhy; vc.umz = um_neg * lhz;   // d u / d m = -um lhat
    vc.mm = -kt * (ntl * m);                                 // d mdot / d m = -kt (um m + umag)
    vc.mn = -kt * mass * un;                                 // d mdot / d n
    vc.Lm = tl * (2.0 * mn_over_m * inv_m);
    vc.Ln = __builtin_fma(-tl, __builtin_fma(un, n, m) * inv_m, ntl * (kt * lm * un));
    vc.Ll = ntl * (kt * m);
  }
}

// One STM column c = (a, b, mu, g, d, nu) of the 14-dim system.
__device__ __forceinline__ void var_col14(const VarCoef14& v, const double w2, const double (&c)[14], double (&dc)[14]) {
  const VarCoef12& vc = v.c;
  const double ax = c[0], ay = c[1], az = c[2], mu = c[6];
  const double dx = c[10], dyv = c[11], dz = c[12];
  dc[0] = c[3]; dc[1] = c[4]; dc[2] = c[5];
  const double ld = __builtin_fma(vc.lx, dx, __builtin_fma(vc.ly, dyv, vc.lz * dz));
  const double tl = vc.ub * ld;
  dc[3] = __builtin_fma(vc.Gxx, ax, __builtin_fma(vc.Gxy, ay, __builtin_fma(vc.Gxz, az,
          __builtin_fma(w2, c[4], __builtin_fma(-vc.ua, dx, __builtin_fma(tl, vc.lx, v.umx * mu))))));
  dc[4] = __builtin_fma(vc.Gxy, ax, __builtin_fma(vc.Gyy, ay, __builtin_fma(vc.Gyz, az,
          __builtin_fma(-w2, c[3], __builtin_fma(-vc.ua, dyv, __builtin_fma(tl, vc.ly, v.umy * mu))))));
  dc[5] = __builtin_fma(vc.Gxz, ax, __builtin_fma(vc.Gyz, ay, __builtin_fma(vc.Gzz, az,
          __builtin_fma(-vc.ua, dz, __builtin_fma(tl, vc.lz, v.umz * mu)))));
  dc[6] = __builtin_fma(v.mm, mu, v.mn * ld);
  dc[7] = -__builtin_fma(vc.Hxx, ax, __builtin_fma(vc.Hxy, ay, __builtin_fma(vc.Hxz, az,
           __builtin_fma(vc.Gxx, dx, __builtin_fma(vc.Gxy, dyv, vc.Gxz * dz)))));
  dc[8] = -__builtin_fma(vc.Hxy, ax, __builtin_fma(vc.Hyy, ay, __builtin_fma(vc.Hyz, az,
           __builtin_fma(vc.Gxy, dx, __builtin_fma(vc.Gyy, dyv, vc.Gyz * dz)))));
  dc[9] = -__builtin_fma(vc.Hxz, ax, __builtin_fma(vc.Hyz, ay, __builtin_fma(vc.Hzz, az,
           __builtin_fma(vc.Gxz, dx, __builtin_fma(vc.Gyz, dyv, vc.Gzz * dz)))));
  dc[10] = __builtin_fma(w2, dyv, -c[7]);
  dc[11] = __builtin_fma(-w2, dx, -c[8]);
  dc[12] = -c[9];
  dc[13] = __builtin_fma(v.Lm, mu, __builtin_fma(v.Ln, ld, v.Ll * c[13]));
}

// Fused base + ONE STM column of the 14-dim system (the COLS = 1 mapping of the BASELINE configs[1] sweep): as
// rhs12_fused1, G and H are applied through their dyadic structure; the mass couplings of rhs14 / var_col14 are added.
//   y = (base[14], column[14]) -> k = (base_dot[14], column_dot[14]);  column = (a, b, mu, g, d, nu)
template <int PM>
__device__ __forceinline__ void rhs14_fused1(const double (&y)[28], const TrajParams& tp, const double w2, double (&k)[28]) {
  const double MU = tp.MU;
  const double x = y[0], yy = y[1], z = y[2], mass = y[6];
  const double A = x + MU, B = A - 1.0;
  const double yz2 = __builtin_fma(yy, yy, z * z);
  const double d1 = __builtin_fma(A, A, yz2), d2 = __builtin_fma(B, B, yz2);
  const double i1 = rsqrt_nr(d1), i2 = rsqrt_nr(d2);
  const double i1s = i1 * i1, i2s = i2 * i2;
  const double c1 = (1.0 - MU) * (i1s * i1), c2 = MU * (i2s * i2);
  const double cs = c1 + c2, omc = 1.0 - cs;
  const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
  const double ee = e1 + e2;
  const double st = __builtin_fma(e1, A, e2 * B);
  const double eey = ee * yy, eez = ee * z;

  const double lx = y[10], ly = y[11], lz = y[12], lm = y[13];
  const double n2 = __builtin_fma(lx, lx, __builtin_fma(ly, ly, lz * lz));
  const double inv_n = (n2 > 0.0) ? rsqrt_nr(n2) : 0.0;
  const double n = n2 * inv_n;
  const double inv_m = rcp_nr(mass);
  const double aL = tp.cT * inv_m;
  double m, ua, ub, un;
  bool tlim;
  control_dispatch<PM, true>(tp, aL, n, inv_n, m, ua, ub, un, tlim);
  const double lhx = lx * inv_n, lhy = ly * inv_n, lhz = lz * inv_n;
  const double kt = tp.kappa_td;

  const double yzl = __builtin_fma(yy, ly, z * lz);
  const double s1 = __builtin_fma(A, lx, yzl), s2 = __builtin_fma(B, lx, yzl);
  const double t1 = e1 * s1, t2 = e2 * s2;
  const double es = t1 + t2;
  const double tA = __builtin_fma(t1, A, t2 * B);

  k[0] = y[3]; k[1] = y[4]; k[2] = y[5];
  k[3] = __builtin_fma(-c1, A, __builtin_fma(-c2, B, __builtin_fma(w2, y[4], x))) - m * lhx;
  k[4] = __builtin_fma(-cs, yy, __builtin_fma(-w2, y[3], yy)) - m * lhy;
  k[5] = __builtin_fma(-cs, z, -m * lhz);
  k[6] = -kt * m * mass;
  k[7] = -__builtin_fma(omc, lx, tA);
  k[8] = -__builtin_fma(omc, ly, es * yy);
  k[9] = -__builtin_fma(-cs, lz, es * z);
  k[10] = __builtin_fma(w2, ly, -y[7]);
  k[11] = __builtin_fma(-w2, lx, -y[8]);
  k[12] = -y[9];
  const double tl = tlim ? 1.0 : 0.0, ntl = 1.0 - tl;      // thrust-limited law (umag ~ 1/m) or not, see rhs14
  const double m_over_m = m * inv_m;
  const double mn_over_m = m_over_m * n;
  k[13] = __builtin_fma(-tl, mn_over_m, ntl * (kt * lm * m));

  // ---- column (a, b, mu, g, d, nu)
  const double ax = y[14], ay = y[15], az = y[16], mu = y[20];
  const double dx = y[24], dyv = y[25], dz = y[26], nu = y[27];
  const double yza = __builtin_fma(yy, ay, z * az);
  const double r1a = __builtin_fma(A, ax, yza), r2a = __builtin_fma(B, ax, yza);
  const double u1 = e1 * r1a, u2 = e2 * r2a;
  const double us = u1 + u2;
  const double uA = __builtin_fma(u1, A, u2 * B);
  const double Gax = __builtin_fma(omc, ax, uA), Gay = __builtin_fma(omc, ay, us * yy), Gaz = __builtin_fma(-cs, az, us * z);
  const double yzd = __builtin_fma(yy, dyv, z * dz);
  const double r1d = __builtin_fma(A, dx, yzd), r2d = __builtin_fma(B, dx, yzd);
  const double v1 = e1 * r1d, v2 = e2 * r2d;
  const double vs = v1 + v2;
  const double vA = __builtin_fma(v1, A, v2 * B);
  const double Gdx = __builtin_fma(omc, dx, vA), Gdy = __builtin_fma(omc, dyv, vs * yy), Gdz = __builtin_fma(-cs, dz, vs * z);
  const double la = __builtin_fma(lx, ax, __builtin_fma(ly, ay, lz * az));
  const double w1 = (5.0 * i1s) * (t1 * r1a), w2b = (5.0 * i2s) * (t2 * r2a);
  const double ws = w1 + w2b;
  const double wA = __builtin_fma(w1, A, w2b * B);
  const double Hax = __builtin_fma(es, ax, __builtin_fma(la, st, __builtin_fma(us, lx, -wA)));
  const double Hay = __builtin_fma(es, ay, __builtin_fma(la, eey, __builtin_fma(us, ly, -ws * yy)));
  const double Haz = __builtin_fma(es, az, __builtin_fma(la, eez, __builtin_fma(us, lz, -ws * z)));
  // U d + (d u / d m) mu = -ua d + (ub (lhat.d) + um_neg mu) lhat,  um_neg = -d umag / d m = tl umag / m
  const double ld = __builtin_fma(lhx, dx, __builtin_fma(lhy, dyv, lhz * dz));
  const double tq = __builtin_fma(ub, ld, (tl * m_over_m) * mu);
  k[14] = y[17]; k[15] = y[18]; k[16] = y[19];
  k[17] = Gax + __builtin_fma(w2, y[18], __builtin_fma(-ua, dx, tq * lhx));
  k[18] = Gay + __builtin_fma(-w2, y[17], __builtin_fma(-ua, dyv, tq * lhy));
  k[19] = Gaz + __builtin_fma(-ua, dz, tq * lhz);
  // mu dot = (d mdot / d m) mu + (d mdot / d n) (lhat . d)
  k[20] = -kt * __builtin_fma(ntl * m, mu, (mass * un) * ld);
  k[21] = -(Hax + Gdx);
  k[22] = -(Hay + Gdy);
  k[23] = -(Haz + Gdz);
  k[24] = __builtin_fma(w2, dyv, -y[21]);
  k[25] = __builtin_fma(-w2, dx, -y[22]);
  k[26] = -y[23];
  // nu dot = Lm mu + Ln (lhat . d) + Ll nu   (coefficients as in rhs14)
  const double Lm = tl * (2.0 * mn_over_m * inv_m);
  const double Ln = __builtin_fma(-tl, __builtin_fma(un, n, m) * inv_m, ntl * (kt * lm * un));
  k[27] = __builtin_fma(Lm, mu, __builtin_fma(Ln, ld, (ntl * (kt * m)) * nu));
}

// ------------------------------------------------------------------------------ lean base RHS (pipeline kernels)
// The pipeline kernels run the base trajectory, the coefficient build and the STM columns in different wavefronts.  The
// base wave's instruction stream is the sweep's critical path and it is issue-bound (one dependent stream per SIMD), so
// this form of the RHS minimises the INSTRUCTION COUNT: G lambda_v through its dyadic structure (as rhs*_fused1), no
// per-lane selects, the thrust vector as -(umag / n) lambda_v, a short exp.  The coefficient wave rebuilds G, H, U from
// the stage argument with rhs12 / rhs14<PM, true>.

// exp(z) for z in [-700, 690], ~1 ulp: n = rint(z log2 e) through the 1.5 2^52 trick (the integer sits in the low word of
// the biased sum), Horner polynomial of degree 13 on |r| <= ln2 / 2 (remainder 4e-18), 2^n added to the exponent field.
// 20 instructions against 25 of exp_neg (no v_rndne, no v_cvt, no v_ldexp).
__device__ __forceinline__ double exp_mid(double z) {
  const double MAGIC = 6755399441055744.0;   // 1.5 * 2^52
  const double t = __builtin_fma(z, 1.4426950408889634, MAGIC);
  const double n = t - MAGIC;
  double r = __builtin_fma(n, -6.93147180369123816490e-01, z);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  // Horner, 13 dependent FMAs: measured ahead of an Estrin arrangement (13 FMAs + 3 products at depth 5: base chain 179 k ->
  // 186 k ticks per sweep): with the scheduler free to interleave, instruction count beats depth.
  double p = kExpTaylor[0];
#pragma unroll
  for (int k = 1; k < 11; ++k) p = __builtin_fma(p, r, kExpTaylor[k]);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  // 2^n: n (the low word of the biased sum, two's complement) added to the exponent field
  return __hiloint2double(__double2hiint(p) + (__double2loint(t) << 20), __double2loint(p));
}

// 1 / |lambda_v| without a select: n2 = 0 (the reference's NaN guard, stateCostate_deriv.jl:59-64: control = 0) gives a
// finite 2^500, and the thrust vector -(umag / n) lambda_v is then exactly 0.  Differs from the guarded form only for
// 0 < |lambda_v| < 2^-500.
__device__ __forceinline__ double inv_norm_guarded(double n2) { return rsqrt_nr(fmax(n2, 9.33263618503218879e-302)); }

// umag / n (`ua`) and umag (`m`) of the control law for the base wave; p = 1 evaluates the logistic directly,
//   1/2 (1 + tanh x) = 1 / (1 + e^{-2x}),  -2x = (1 - n) / rho clamped to [-700, 690]
// (beyond the clamp the value is 1 or < 1e-299 either way), so no per-lane branch on the sign of x is needed.
template <int PM>
__device__ __forceinline__ void control_base12(const TrajParams& tp, double n, double inv_n, double& m, double& ua) {
  if constexpr (PM == PM_P1) {
    const double z = fmin(fmax((1.0 - n) * tp.inv_rho, -700.0), 690.0);
    m = tp.accel_limit * rcp_nr(1.0 + exp_mid(z));
    ua = m * inv_n;
  } else {
    double ub, un;
    bool tlim;
    control_dispatch<PM, false, true>(tp, tp.accel_limit, n, inv_n, m, ua, ub, un, tlim);
  }
}

template <int PM>
__device__ __forceinline__ void rhs12_base(const double (&y)[12], const TrajParams& tp, double (&dy)[12]) {
  const double MU = tp.MU;
  const double x = y[0], yy = y[1], z = y[2];
  const double w2 = 2.0 * tp.omega;
  const double a = x + MU, b = a - 1.0;
  const double yz2 = __builtin_fma(yy, yy, z * z);
  const double d1 = __builtin_fma(a, a, yz2), d2 = __builtin_fma(b, b, yz2);
  const double i1 = rsqrt_nr(d1), i2 = rsqrt_nr(d2);
  const double i1s = i1 * i1, i2s = i2 * i2;
  const double c1 = (1.0 - MU) * (i1s * i1), c2 = MU * (i2s * i2);
  const double cs = c1 + c2, omc = 1.0 - cs;
  const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
  const double lx = y[9], ly = y[10], lz = y[11];
  const double n2 = __builtin_fma(lx, lx, __builtin_fma(ly, ly, lz * lz));
  const double inv_n = inv_norm_guarded(n2);
  const double n = n2 * inv_n;
  double m, ua;
  control_base12<PM>(tp, n, inv_n, m, ua);
  const double yzl = __builtin_fma(yy, ly, z * lz);
  const double s1 = __builtin_fma(a, lx, yzl), s2 = __builtin_fma(b, lx, yzl);
  const double t1 = e1 * s1, t2 = e2 * s2;
  const double es = t1 + t2;
  const double tA = __builtin_fma(t1, a, t2 * b);
  dy[0] = y[3]; dy[1] = y[4]; dy[2] = y[5];
  dy[3] = __builtin_fma(-ua, lx, __builtin_fma(-c1, a, __builtin_fma(-c2, b, __builtin_fma(w2, y[4], x))));
  dy[4] = __builtin_fma(-ua, ly, __builtin_fma(-cs, yy, __builtin_fma(-w2, y[3], yy)));
  dy[5] = __builtin_fma(-ua, lz, -cs * z);
  dy[6] = __builtin_fma(-omc, lx, -tA);
  dy[7] = __builtin_fma(-omc, ly, -es * yy);
  dy[8] = __builtin_fma(cs, lz, -es * z);
  dy[9] = __builtin_fma(w2, ly, -y[6]);
  dy[10] = __builtin_fma(-w2, lx, -y[7]);
  dy[11] = -y[8];
}

// By-products of the lean base RHS that the variational coefficients can be rebuilt from without a reciprocal square
// root, an exponential or a division (cooperative kernel: the base lane publishes them, the column lanes assemble G, H, U).
struct BaseParts12 {
  double c1, c2;      // kappa_b / d_b^{3/2}
  double i1s, i2s;    // 1 / d_b
  double ua, ub;      // umag / n and ua - d umag / d n   (U d = -ua d + ub (lhat.d) lhat)
  double inv_n;       // 1 / |lambda_v| (guarded)
};

// lambda_v = 0 exactly: the reference's guard sets the control to 0 (stateCostate_deriv.jl:59-64: a constant), so du / dlambda_v = 0
// there.  The slopes are right by themselves -- the guarded 1 / n is a finite 2^500 and the thrust -(umag / n) lambda_v an exact
// 0 -- but umag / n itself, which the column lanes build U from, would be ~1e148 for the laws whose umag does not vanish with n
// (p = 0, p = 1).  The unclamped p > 1 laws have umag / n -> a finite limit and keep their values (p = 2: U = -I / 2).
template <int PM>
__device__ __forceinline__ void parts_guard_zero_norm(const double n2, BaseParts12& bp) {
  if constexpr (PM == PM_P0 || PM == PM_P1) {
    const bool nz = n2 > 0.0;
    bp.ua = nz ? bp.ua : 0.0;
    bp.ub = nz ? bp.ub : 0.0;
  }
}

// rhs12_base plus the parts.  Same arithmetic for the slopes.
template <int PM>
__device__ __forceinline__ void rhs12_base_parts(const double (&y)[12], const TrajParams& tp, double (&dy)[12], BaseParts12& bp) {
  const double MU = tp.MU;
  const double x = y[0], yy = y[1], z = y[2];
  const double w2 = 2.0 * tp.omega;
  const double a = x + MU, b = a - 1.0;
  const double yz2 = __builtin_fma(yy, yy, z * z);
  const double d1 = __builtin_fma(a, a, yz2), d2 = __builtin_fma(b, b, yz2);
  const double i1 = rsqrt_nr(d1), i2 = rsqrt_nr(d2);
  const double i1s = i1 * i1, i2s = i2 * i2;
  const double c1 = (1.0 - MU) * (i1s * i1), c2 = MU * (i2s * i2);
  const double cs = c1 + c2, omc = 1.0 - cs;
  const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
  const double lx = y[9], ly = y[10], lz = y[11];
  const double n2 = __builtin_fma(lx, lx, __builtin_fma(ly, ly, lz * lz));
  const double inv_n = inv_norm_guarded(n2);
  const double n = n2 * inv_n;
  double m, ua, ub;
  if constexpr (PM == PM_P1) {
    const double zz = fmin(fmax((1.0 - n) * tp.inv_rho, -700.0), 690.0);
    const double e = exp_mid(zz);
    const double q = rcp_nr(1.0 + e);
    m = tp.accel_limit * q;
    ua = m * inv_n;
    ub = __builtin_fma(-(tp.accel_limit * tp.inv_rho) * (e * q), q, ua);   // ua - aL/(4 rho) sech^2 x
  } else {
    double un;
    bool tlim;
    control_dispatch<PM, true, true>(tp, tp.accel_limit, n, inv_n, m, ua, ub, un, tlim);
  }
  const double yzl = __builtin_fma(yy, ly, z * lz);
  const double s1 = __builtin_fma(a, lx, yzl), s2 = __builtin_fma(b, lx, yzl);
  const double t1 = e1 * s1, t2 = e2 * s2;
  const double es = t1 + t2;
  const double tA = __builtin_fma(t1, a, t2 * b);
  dy[0] = y[3]; dy[1] = y[4]; dy[2] = y[5];
  dy[3] = __builtin_fma(-ua, lx, __builtin_fma(-c1, a, __builtin_fma(-c2, b, __builtin_fma(w2, y[4], x))));
  dy[4] = __builtin_fma(-ua, ly, __builtin_fma(-cs, yy, __builtin_fma(-w2, y[3], yy)));
  dy[5] = __builtin_fma(-ua, lz, -cs * z);
  dy[6] = __builtin_fma(-omc, lx, -tA);
  dy[7] = __builtin_fma(-omc, ly, -es * yy);
  dy[8] = __builtin_fma(cs, lz, -es * z);
  dy[9] = __builtin_fma(w2, ly, -y[6]);
  dy[10] = __builtin_fma(-w2, lx, -y[7]);
  dy[11] = -y[8];
  bp.c1 = c1; bp.c2 = c2; bp.i1s = i1s; bp.i2s = i2s; bp.ua = ua; bp.ub = ub; bp.inv_n = inv_n;
  parts_guard_zero_norm<PM>(n2, bp);
}

// The base slopes of one HALF of the 12-dim state in a lane (cooperative kernel, two lanes per segment): lane A owns
// (r, v), lane B owns (lambda_v, lambda_r).  Both lanes hold R = r and L = lambda_v (three doubles each crossed over by DPP)
// and their own second triple q (A: v, B: lambda_r); the instruction stream is the same for both.
//   kp = slope of the lane's first triple:   A: r' = v                          B: lambda_v' = 2w J lambda_v - lambda_r
//        written as sg q + kap J L with per-lane (sg, kap) = (1, 0) / (-1, 2w): exact in both lanes;
//   kq = slope of the second triple:         A: v' (gravity, Coriolis, thrust)  B: lambda_r' = -(d a / d r)^T lambda_v
//        both evaluated, one kept.
// Same arithmetic per component as rhs12_base_parts (same by-products in bp).
template <int PM>
__device__ __forceinline__ void rhs12_base_half(const double (&R)[3], const double (&L)[3], const double (&q)[3], const bool is_a,
                                                const double sg, const double kap, const TrajParams& tp, double (&kp)[3],
                                                double (&kq)[3], BaseParts12& bp) {
  const double MU = tp.MU;
  const double x = R[0], yy = R[1], z = R[2];
  const double w2 = 2.0 * tp.omega;
  const double a = x + MU, b = a - 1.0;
  const double yz2 = __builtin_fma(yy, yy, z * z);
  const double d1 = __builtin_fma(a, a, yz2), d2 = __builtin_fma(b, b, yz2);
  const double i1 = rsqrt_nr(d1), i2 = rsqrt_nr(d2);
  const double i1s = i1 * i1, i2s = i2 * i2;
  const double c1 = (1.0 - MU) * (i1s * i1), c2 = MU * (i2s * i2);
  const double cs = c1 + c2, omc = 1.0 - cs;
  const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
  const double lx = L[0], ly = L[1], lz = L[2];
  const double n2 = __builtin_fma(lx, lx, __builtin_fma(ly, ly, lz * lz));
  const double inv_n = inv_norm_guarded(n2);
  const double n = n2 * inv_n;
  double m, ua, ub;
  if constexpr (PM == PM_P1) {
    const double zz = fmin(fmax((1.0 - n) * tp.inv_rho, -700.0), 690.0);
    const double e = exp_mid(zz);
    const double qq = rcp_nr(1.0 + e);
    m = tp.accel_limit * qq;
    ua = m * inv_n;
    ub = __builtin_fma(-(tp.accel_limit * tp.inv_rho) * (e * qq), qq, ua);
  } else {
    double un;
    bool tlim;
    control_dispatch<PM, true, true>(tp, tp.accel_limit, n, inv_n, m, ua, ub, un, tlim);
  }
  const double yzl = __builtin_fma(yy, ly, z * lz);
  const double s1 = __builtin_fma(a, lx, yzl), s2 = __builtin_fma(b, lx, yzl);
  const double t1 = e1 * s1, t2 = e2 * s2;
  const double es = t1 + t2;
  const double tA = __builtin_fma(t1, a, t2 * b);
  kp[0] = __builtin_fma(kap, ly, sg * q[0]);
  kp[1] = __builtin_fma(-kap, lx, sg * q[1]);
  kp[2] = sg * q[2];
  const double ax = __builtin_fma(-ua, lx, __builtin_fma(-c1, a, __builtin_fma(-c2, b, __builtin_fma(w2, q[1], x))));
  const double ay = __builtin_fma(-ua, ly, __builtin_fma(-cs, yy, __builtin_fma(-w2, q[0], yy)));
  const double az = __builtin_fma(-ua, lz, -cs * z);
  const double gx = __builtin_fma(-omc, lx, -tA);
  const double gy = __builtin_fma(-omc, ly, -es * yy);
  const double gz = __builtin_fma(cs, lz, -es * z);
  kq[0] = is_a ? ax : gx; kq[1] = is_a ? ay : gy; kq[2] = is_a ? az : gz;
  bp.c1 = c1; bp.c2 = c2; bp.i1s = i1s; bp.i2s = i2s; bp.ua = ua; bp.ub = ub; bp.inv_n = inv_n;
  parts_guard_zero_norm<PM>(n2, bp);
}

// The two halves of coef12_from_parts / var_col12 for a column split over two lanes in different waves (cooperative kernel):
// the TOP lane owns (a, b) = (delta r, delta v) and needs G and U, the BOTTOM lane owns (d, g) = (delta lambda_v, delta
// lambda_r) and needs G and H; each receives the other's first triple.  Same arithmetic per entry as the one-piece forms.
struct CoefG12 { double Gxx, Gyy, Gzz, Gxy, Gxz, Gyz; };
// top half: (a', b') from own (a, b) and the received d, with G as published by the base wave; lx0.. = lambda_v of the stage argument
__device__ __forceinline__ void var_col12_top_g(const CoefG12& g, const double lx0, const double ly0, const double lz0, const double inv_n,
                                                const double ua, const double ub, const double w2, const double (&w)[6],
                                                const double (&d)[3], double (&dw)[6]) {
  const double lx = lx0 * inv_n, ly = ly0 * inv_n, lz = lz0 * inv_n;
  const double ax = w[0], ay = w[1], az = w[2];
  const double dx = d[0], dyv = d[1], dz = d[2];
  dw[0] = w[3]; dw[1] = w[4]; dw[2] = w[5];
  const double ld = __builtin_fma(lx, dx, __builtin_fma(ly, dyv, lz * dz));
  const double tl = ub * ld;
  dw[3] = __builtin_fma(g.Gxx, ax, __builtin_fma(g.Gxy, ay, __builtin_fma(g.Gxz, az,
          __builtin_fma(w2, w[4], __builtin_fma(-ua, dx, tl * lx)))));
  dw[4] = __builtin_fma(g.Gxy, ax, __builtin_fma(g.Gyy, ay, __builtin_fma(g.Gyz, az,
          __builtin_fma(-w2, w[3], __builtin_fma(-ua, dyv, tl * ly)))));
  dw[5] = __builtin_fma(g.Gxz, ax, __builtin_fma(g.Gyz, ay, __builtin_fma(g.Gzz, az,
          __builtin_fma(-ua, dz, tl * lz))));
}
// bottom half: own w = (d, g) = (delta lambda_v, delta lambda_r), received a = delta r; returns (d', g').  G as published by the
// base wave, H from the parts (same arithmetic per entry as coef12_from_parts)
__device__ __forceinline__ void var_col12_bottom_g(const CoefG12& g, const double x, const double yy, const double z, const double lx0,
                                                   const double ly0, const double lz0, const double c1, const double c2, const double i1s,
                                                   const double i2s, const double MU, const double w2, const double (&w)[6],
                                                   const double (&av)[3], double (&dw)[6]) {
  const double a = x + MU, b = a - 1.0;
  const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
  const double ee = e1 + e2;
  const double st = e1 * a + e2 * b;
  const double yzl = __builtin_fma(yy, ly0, z * lz0);
  const double s1 = __builtin_fma(a, lx0, yzl), s2 = __builtin_fma(b, lx0, yzl);
  const double q1 = 5.0 * e1 * i1s * s1, q2 = 5.0 * e2 * i2s * s2;
  const double es = __builtin_fma(e1, s1, e2 * s2);
  const double qq = q1 + q2;
  const double qa = __builtin_fma(q1, a, q2 * b);
  const double Hxx = es + 2.0 * st * lx0 - __builtin_fma(q1 * a, a, q2 * b * b);
  const double Hyy = es + 2.0 * ee * yy * ly0 - qq * yy * yy;
  const double Hzz = es + 2.0 * ee * z * lz0 - qq * z * z;
  const double Hxy = __builtin_fma(st, ly0, ee * yy * lx0) - qa * yy;
  const double Hxz = __builtin_fma(st, lz0, ee * z * lx0) - qa * z;
  const double Hyz = ee * __builtin_fma(yy, lz0, z * ly0) - qq * yy * z;
  const double ax = av[0], ay = av[1], az = av[2];
  const double dx = w[0], dyv = w[1], dz = w[2];
  dw[0] = __builtin_fma(w2, dyv, -w[3]);
  dw[1] = __builtin_fma(-w2, dx, -w[4]);
  dw[2] = -w[5];
  dw[3] = -__builtin_fma(Hxx, ax, __builtin_fma(Hxy, ay, __builtin_fma(Hxz, az,
           __builtin_fma(g.Gxx, dx, __builtin_fma(g.Gxy, dyv, g.Gxz * dz)))));
  dw[4] = -__builtin_fma(Hxy, ax, __builtin_fma(Hyy, ay, __builtin_fma(Hyz, az,
           __builtin_fma(g.Gxy, dx, __builtin_fma(g.Gyy, dyv, g.Gyz * dz)))));
  dw[5] = -__builtin_fma(Hxz, ax, __builtin_fma(Hyz, ay, __builtin_fma(Hzz, az,
           __builtin_fma(g.Gxz, dx, __builtin_fma(g.Gyz, dyv, g.Gzz * dz)))));
}
// Round 4 forms of the two halves: no matrix is assembled or published any more, G, H and U are applied through their dyadic
// structure from what the base wave has at hand (A = x + MU, y, z; per primary e_b = 3 kappa_b / d_b^{5/2} and
// q_b = 5 e_b (rho_b . lambda_v) / d_b; omc = 1 - sum_b kappa_b / d_b^{3/2}; es = sum_b e_b (rho_b . lambda_v); lambda_v;
// ua = umag / n and ubn = (umag / n - umag') / n^2), rho_1 = (A, y, z), rho_2 = (A - 1, y, z):
//   G x = omc x - (0, 0, x_z) + sum_b e_b rho_b (rho_b . x)
//   H a = es a + sum_b [ e_b (rho_b (lambda_v . a) + lambda_v (rho_b . a)) - q_b rho_b (rho_b . a) ]
//   U d = -ua d + ubn (lambda_v . d) lambda_v
// -- the same matrices as coef12_from_parts builds entry by entry (Gxx = omc + e1 a^2 + e2 b^2, Hxx = es + 2 st lx - q1 a^2 -
// q2 b^2, ...), associated differently: the last bit or two may differ.  Top half 31 instructions per stage, bottom half 47
// (round 3: 92 - tableau and 135 - tableau with G from LDS and H built per lane), and the base wave neither builds nor stores G.
struct DyadParts { double A, yy, z, e1, e2, omc; };
__device__ __forceinline__ void dyad_G(const DyadParts& p, const double B, const double x0, const double x1, const double x2, double& g0,
                                       double& g1, double& g2) {
  const double yz = __builtin_fma(p.yy, x1, p.z * x2);
  const double k1 = p.e1 * __builtin_fma(p.A, x0, yz), k2 = p.e2 * __builtin_fma(B, x0, yz);
  const double ks = k1 + k2;
  g0 = __builtin_fma(p.omc, x0, __builtin_fma(k1, p.A, k2 * B));
  g1 = __builtin_fma(p.omc, x1, ks * p.yy);
  g2 = __builtin_fma(p.omc, x2, __builtin_fma(ks, p.z, -x2));
}
// top half: (a', b') from own w = (a, b) = (delta r, delta v) and the received d = delta lambda_v
__device__ __forceinline__ void var_col12_top_dy(const DyadParts& p, const double lx, const double ly, const double lz, const double ua,
                                                 const double ubn, const double w2, const double (&w)[6], const double (&d)[3],
                                                 double (&dw)[6]) {
  const double B = p.A - 1.0;
  double g0, g1, g2;
  dyad_G(p, B, w[0], w[1], w[2], g0, g1, g2);
  const double dx = d[0], dyv = d[1], dz = d[2];
  dw[0] = w[3]; dw[1] = w[4]; dw[2] = w[5];
  const double ld = __builtin_fma(lx, dx, __builtin_fma(ly, dyv, lz * dz));
  const double tl = ubn * ld;
  dw[3] = __builtin_fma(w2, w[4], __builtin_fma(-ua, dx, __builtin_fma(tl, lx, g0)));
  dw[4] = __builtin_fma(-w2, w[3], __builtin_fma(-ua, dyv, __builtin_fma(tl, ly, g1)));
  dw[5] = __builtin_fma(-ua, dz, __builtin_fma(tl, lz, g2));
}
// bottom half: own w = (d, g) = (delta lambda_v, delta lambda_r), received av = delta r; returns (d', g')
__device__ __forceinline__ void var_col12_bottom_dy(const DyadParts& p, const double lx, const double ly, const double lz, const double q1,
                                                    const double q2, const double es, const double w2, const double (&w)[6],
                                                    const double (&av)[3], double (&dw)[6]) {
  const double B = p.A - 1.0;
  const double ax = av[0], ay = av[1], az = av[2];
  const double dx = w[0], dyv = w[1], dz = w[2];
  double g0, g1, g2;
  dyad_G(p, B, dx, dyv, dz, g0, g1, g2);
  const double la = __builtin_fma(lx, ax, __builtin_fma(ly, ay, lz * az));
  const double yz = __builtin_fma(p.yy, ay, p.z * az);
  const double p1 = __builtin_fma(p.A, ax, yz), p2 = __builtin_fma(B, ax, yz);
  const double k1 = __builtin_fma(p.e1, la, -q1 * p1), k2 = __builtin_fma(p.e2, la, -q2 * p2);
  const double kl = __builtin_fma(p.e1, p1, p.e2 * p2);
  const double ks = k1 + k2;
  const double hx = __builtin_fma(es, ax, __builtin_fma(kl, lx, __builtin_fma(k1, p.A, __builtin_fma(k2, B, g0))));
  const double hy = __builtin_fma(es, ay, __builtin_fma(kl, ly, __builtin_fma(ks, p.yy, g1)));
  const double hz = __builtin_fma(es, az, __builtin_fma(kl, lz, __builtin_fma(ks, p.z, g2)));
  dw[0] = __builtin_fma(w2, dyv, -w[3]);
  dw[1] = __builtin_fma(-w2, dx, -w[4]);
  dw[2] = -w[5];
  dw[3] = -hx; dw[4] = -hy; dw[5] = -hz;
}
// 14-dim forms of the two column halves (always-thrust-limited laws; kernels_indirect_coop2_14.hip): top w = (a, b, mu) =
// (delta r, delta v, delta m) with the received d = delta lambda_v; bottom w = (d, g, nu) = (delta lambda_v, delta lambda_r,
// delta lambda_m) with the received (a, mu).  The 12-dim part is var_col12_top_dy / _bottom_dy; the mass couplings of var_col14
// with the unit vector lhat written as inv_n lambda_v and the factor folded into the coefficients the base wave publishes:
//   b'  += umn lambda_v mu,       umn = (umag / mass) / n                (d u / d m = (umag / m) lhat)
//   mu'  = mnn (lambda_v . d),    mnn = -kappa_td mass umag' / n         (d mdot / d lambda_v)
//   nu'  = Lm mu + Lnn (lambda_v . d),   Lm = 2 umag n / mass^2,  Lnn = -(umag' n + umag) / (mass n)
// (nothing depends on nu for these laws, so the lambda_m column of the STM stays the unit vector and is not integrated).
__device__ __forceinline__ void var_col14_top_dy(const DyadParts& p, const double lx, const double ly, const double lz, const double ua,
                                                 const double ubn, const double umn, const double mnn, const double w2, const double (&w)[7],
                                                 const double (&d)[3], double (&dw)[7]) {
  const double B = p.A - 1.0;
  double g0, g1, g2;
  dyad_G(p, B, w[0], w[1], w[2], g0, g1, g2);
  const double dx = d[0], dyv = d[1], dz = d[2];
  dw[0] = w[3]; dw[1] = w[4]; dw[2] = w[5];
  const double ld = __builtin_fma(lx, dx, __builtin_fma(ly, dyv, lz * dz));
  const double tl = __builtin_fma(umn, w[6], ubn * ld);       // the two multiples of lambda_v: U's dyad and d u / d m
  dw[3] = __builtin_fma(w2, w[4], __builtin_fma(-ua, dx, __builtin_fma(tl, lx, g0)));
  dw[4] = __builtin_fma(-w2, w[3], __builtin_fma(-ua, dyv, __builtin_fma(tl, ly, g1)));
  dw[5] = __builtin_fma(-ua, dz, __builtin_fma(tl, lz, g2));
  dw[6] = mnn * ld;
}
__device__ __forceinline__ void var_col14_bottom_dy(const DyadParts& p, const double lx, const double ly, const double lz, const double q1,
                                                    const double q2, const double es, const double Lm, const double Lnn, const double w2,
                                                    const double (&w)[7], const double (&av)[3], const double mu, double (&dw)[7]) {
  const double w6[6] = {w[0], w[1], w[2], w[3], w[4], w[5]};
  double d6[6];
  var_col12_bottom_dy(p, lx, ly, lz, q1, q2, es, w2, w6, av, d6);
#pragma unroll
  for (int j = 0; j < 6; ++j) dw[j] = d6[j];
  const double ld = __builtin_fma(lx, w[0], __builtin_fma(ly, w[1], lz * w[2]));
  dw[6] = __builtin_fma(Lm, mu, Lnn * ld);
}

// G, H, U of the 12-dim system from the base argument's position r, lambda_v and the base lane's by-products: the
// VAR block of rhs12 without its reciprocal square roots and control law.
__device__ __forceinline__ void coef12_from_parts(const double x, const double yy, const double z, const double lx0, const double ly0,
                                                  const double lz0, const BaseParts12& bp, const double MU, VarCoef12& vc) {
  const double a = x + MU, b = a - 1.0;
  const double c1 = bp.c1, c2 = bp.c2, i1s = bp.i1s, i2s = bp.i2s;
  const double cs = c1 + c2;
  const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
  const double ee = e1 + e2;
  const double sa = e1 * a, tb = e2 * b;
  const double st = sa + tb;
  vc.Gxx = __builtin_fma(sa, a, __builtin_fma(tb, b, 1.0 - cs));
  vc.Gyy = __builtin_fma(ee * yy, yy, 1.0 - cs);
  vc.Gzz = __builtin_fma(ee * z, z, -cs);
  vc.Gxy = st * yy; vc.Gxz = st * z; vc.Gyz = ee * yy * z;
  vc.ua = bp.ua; vc.ub = bp.ub;
  vc.lx = lx0 * bp.inv_n; vc.ly = ly0 * bp.inv_n; vc.lz = lz0 * bp.inv_n;
  const double yzl = __builtin_fma(yy, ly0, z * lz0);
  const double s1 = __builtin_fma(a, lx0, yzl), s2 = __builtin_fma(b, lx0, yzl);
  const double q1 = 5.0 * e1 * i1s * s1, q2 = 5.0 * e2 * i2s * s2;
  const double es = __builtin_fma(e1, s1, e2 * s2);
  const double qq = q1 + q2;
  const double qa = __builtin_fma(q1, a, q2 * b);
  vc.Hxx = es + 2.0 * st * lx0 - __builtin_fma(q1 * a, a, q2 * b * b);
  vc.Hyy = es + 2.0 * ee * yy * ly0 - qq * yy * yy;
  vc.Hzz = es + 2.0 * ee * z * lz0 - qq * z * z;
  vc.Hxy = __builtin_fma(st, ly0, ee * yy * lx0) - qa * yy;
  vc.Hxz = __builtin_fma(st, lz0, ee * z * lx0) - qa * z;
  vc.Hyz = ee * __builtin_fma(yy, lz0, z * ly0) - qq * yy * z;
}

// LM = false: lambda_m_dot is not evaluated (dy[13] = 0).  For the always-thrust-limited laws (p = 0, p = 1) nothing else
// depends on lambda_m, and the eight-wave pipeline kernel integrates it in the coefficient wave, off the critical stream.
template <int PM, bool LM = true>
__device__ __forceinline__ void rhs14_base(const double (&y)[14], const TrajParams& tp, double (&dy)[14]) {
  static_assert(LM || PM == PM_P0 || PM == PM_P1, "lambda_m feeds back into the unclamped p > 1 law");
  const double MU = tp.MU;
  const double x = y[0], yy = y[1], z = y[2], mass = y[6];
  const double w2 = 2.0 * tp.omega;
  const double a = x + MU, b = a - 1.0;
  const double yz2 = __builtin_fma(yy, yy, z * z);
  const double d1 = __builtin_fma(a, a, yz2), d2 = __builtin_fma(b, b, yz2);
  const double i1 = rsqrt_nr(d1), i2 = rsqrt_nr(d2);
  const double i1s = i1 * i1, i2s = i2 * i2;
  const double c1 = (1.0 - MU) * (i1s * i1), c2 = MU * (i2s * i2);
  const double cs = c1 + c2, omc = 1.0 - cs;
  const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
  const double lx = y[10], ly = y[11], lz = y[12], lm = y[13];
  const double n2 = __builtin_fma(lx, lx, __builtin_fma(ly, ly, lz * lz));
  const double inv_n = inv_norm_guarded(n2);
  const double n = n2 * inv_n;
  double inv_m, m, ua;
  bool tlim = true;
  if constexpr (PM == PM_P1) {
    // umag = (cT / mass) / (1 + e^{-2x}): ONE reciprocal of (1 + e) mass serves both 1 / mass and the logistic
    // (v_rcp_f64 is a quarter-rate instruction on the critical stream)
    const double zz = fmin(fmax((1.0 - n) * tp.inv_rho, -700.0), 690.0);
    const double ope = 1.0 + exp_mid(zz);
    const double r = rcp_nr(ope * mass);
    inv_m = r * ope;
    m = tp.cT * r;
    ua = m * inv_n;
  } else {
    inv_m = rcp_nr(mass);
    double ub, un;
    control_dispatch<PM, false, true>(tp, tp.cT * inv_m, n, inv_n, m, ua, ub, un, tlim);
  }
  const double kt = tp.kappa_td;
  const double yzl = __builtin_fma(yy, ly, z * lz);
  const double s1 = __builtin_fma(a, lx, yzl), s2 = __builtin_fma(b, lx, yzl);
  const double t1 = e1 * s1, t2 = e2 * s2;
  const double es = t1 + t2;
  const double tA = __builtin_fma(t1, a, t2 * b);
  dy[0] = y[3]; dy[1] = y[4]; dy[2] = y[5];
  dy[3] = __builtin_fma(-ua, lx, __builtin_fma(-c1, a, __builtin_fma(-c2, b, __builtin_fma(w2, y[4], x))));
  dy[4] = __builtin_fma(-ua, ly, __builtin_fma(-cs, yy, __builtin_fma(-w2, y[3], yy)));
  dy[5] = __builtin_fma(-ua, lz, -cs * z);
  dy[6] = (-kt * m) * mass;
  dy[7] = __builtin_fma(-omc, lx, -tA);
  dy[8] = __builtin_fma(-omc, ly, -es * yy);
  dy[9] = __builtin_fma(cs, lz, -es * z);
  dy[10] = __builtin_fma(w2, ly, -y[7]);
  dy[11] = __builtin_fma(-w2, lx, -y[8]);
  dy[12] = -y[9];
  if constexpr (!LM) {
    dy[13] = 0.0;
  } else if constexpr (PM == PM_P0 || PM == PM_P1) {   // always thrust-limited: lambda_m_dot = -umag n / m, no blend needed
    dy[13] = -(m * n) * inv_m;
  } else {
    const double tl = tlim ? 1.0 : 0.0, ntl = 1.0 - tl;
    dy[13] = __builtin_fma(-tl, (m * n) * inv_m, ntl * (kt * lm * m));
  }
}

// ----------------------------------------------------- the base RHS split for the paired-stage base role (pipe8)
// (and, inside a stage, the two gravitating bodies split over two lanes)
// In classical RK4 on this system the arguments r and lambda_v of stage 2 do not depend on the expensive part of stage 1
// (gravity, control law): r_2 = r + h/2 v, lambda_v,2 = lambda_v + h/2 (2 w J lambda_v - lambda_r); likewise stage 4's on stage
// 3's.  Two lanes of a segment therefore evaluate the expensive parts of stages (1, 2) and then (3, 4) side by side.  What a
// lane contributes for ITS stage, from that stage's r and lambda_v alone:
//   av = gravity + centrifugal part of v_dot, for ND = 12 with the thrust -(umag / n) lambda_v (the caller adds the
//        Coriolis term 2 w J v, which needs the stage's velocity, and for ND = 14 the thrust, which needs the stage's mass);
//   gl = lambda_r_dot = -(d a / d r)^T lambda_v, complete;
//   ND = 14 (always-thrust-limited laws): sc = umag * mass (mass-free: cT sigma) and gf = sc / n.
// Same arithmetic per term as rhs12_base / rhs14_base.
struct StageOwn {
  double av[3], gl[3];
  double gf, sc;
};
// `swap_body` returns its argument as held by the lane that evaluates the OTHER gravitating body of the same stage (a DPP move in
// the caller's lane layout).  A lane evaluates ONE body's inverse-distance powers: body_off = MU (primary 1) or MU - 1 (primary
// 2), body_kap = 1 - MU or MU, body_sgn = -1 or +1 (the other body's x offset relative to this one's); the two lanes swap
// c_b = kappa_b / d_b^{3/2} and t_b = e_b (rho_b . lambda_v).  In the lanes of primary 1 every sum below has the operand order of
// rhs12_base / rhs14_base (same bits); the lanes of primary 2 differ from them by round-off and are never used as a source.
template <int ND, int PM, class Swap>
__device__ __forceinline__ void base_stage_own(const double x, const double yy, const double z, const double lx, const double ly,
                                               const double lz, const TrajParams& tp, const double body_off, const double body_kap,
                                               const double body_sgn, Swap&& swap_body, StageOwn& o) {
  static_assert(ND == 12 || PM == PM_P0 || PM == PM_P1, "ND = 14: only the laws whose thrust is cT sigma(n) / mass");
  const double rx = x + body_off, rx_o = rx + body_sgn;
  const double yz2 = __builtin_fma(yy, yy, z * z);
  const double d = __builtin_fma(rx, rx, yz2);
  const double i = rsqrt_nr(d);
  const double is = i * i;
  const double c = body_kap * (is * i);
  const double e = 3.0 * c * is;
  const double n2 = __builtin_fma(lx, lx, __builtin_fma(ly, ly, lz * lz));
  const double inv_n = inv_norm_guarded(n2);
  const double n = n2 * inv_n;
  const double yzl = __builtin_fma(yy, ly, z * lz);
  const double t = e * __builtin_fma(rx, lx, yzl);
  const double c_o = swap_body(c), t_o = swap_body(t);
  const double cs = c + c_o, omc = 1.0 - cs;
  const double es = t + t_o;
  const double tA = __builtin_fma(t, rx, t_o * rx_o);
  o.gl[0] = __builtin_fma(-omc, lx, -tA);
  o.gl[1] = __builtin_fma(-omc, ly, -es * yy);
  o.gl[2] = __builtin_fma(cs, lz, -es * z);
  if constexpr (ND == 12) {
    double m, ua;
    control_base12<PM>(tp, n, inv_n, m, ua);
    o.av[0] = __builtin_fma(-ua, lx, __builtin_fma(-c, rx, __builtin_fma(-c_o, rx_o, x)));
    o.av[1] = __builtin_fma(-ua, ly, __builtin_fma(-cs, yy, yy));
    o.av[2] = __builtin_fma(-ua, lz, -cs * z);
    o.gf = 0.0; o.sc = 0.0;
  } else {
    if constexpr (PM == PM_P1) {
      const double zz = fmin(fmax((1.0 - n) * tp.inv_rho, -700.0), 690.0);
      o.sc = tp.cT * rcp_nr(1.0 + exp_mid(zz));
    } else {
      o.sc = tp.cT;
    }
    o.gf = o.sc * inv_n;
    o.av[0] = __builtin_fma(-c, rx, __builtin_fma(-c_o, rx_o, x));
    o.av[1] = __builtin_fma(-cs, yy, yy);
    o.av[2] = -cs * z;
  }
}

// ------------------------------------------------------------------------------ A2 (direct path)
// Per-lane constants of one half-segment propagation (prop_EP_deriv.jl:8-61).
struct DirectLane {
  double MU;
  double w2;       // 2 * time_direction
  double cx, cy, cz;  // control [N]
  double tx, ty, tz;  // NS = 6 (mass literal 1000.0, :20): the thrust acceleration control * (kk * 1e-3), constant along the arc
  double kk;       // TU^2 / DU / 1e3: thrust [N] / mass [kg] -> DU/TU^2        (:32)
  double mdot;     // -time_direction * |control| / (Isp * 9.81) * TU           (:41-42)
};

struct VarCoef6 {
  double Gxx, Gyy, Gzz, Gxy, Gxz, Gyz;
  double k_over_m;      // dv/dcontrol = kk / m
  double dvdm_x, dvdm_y, dvdm_z;  // dv/dm = -control * kk / m^2 (7-state only)
};

// xdot for x = (r, v[, m]).  NS = 6: mass literal 1000.0 (:20); NS = 7: mass = x[6].
template <int NS, bool VAR>
__device__ __forceinline__ void rhs_direct(const double (&x)[NS], const DirectLane& L, double (&dx)[NS], VarCoef6& vc) {
  const double X = x[0], Y = x[1], Z = x[2];
  const double a = X + L.MU, b = a - 1.0;
  const double yz2 = __builtin_fma(Y, Y, Z * Z);
  const double d1 = __builtin_fma(a, a, yz2), d2 = __builtin_fma(b, b, yz2);
  const double i1 = rsqrt_nr(d1), i2 = rsqrt_nr(d2);
  const double i1s = i1 * i1, i2s = i2 * i2;
  const double c1 = (1.0 - L.MU) * (i1s * i1), c2 = L.MU * (i2s * i2);
  const double cs = c1 + c2;
  const double inv_m = (NS == 7) ? rcp_nr(x[NS - 1]) : 1e-3;  // 1 / 1000.0
  const double k = L.kk * inv_m;
  dx[0] = x[3]; dx[1] = x[4]; dx[2] = x[5];
  // NS = 6: control * k does not change along the arc (L.tx = L.cx * (L.kk * 1e-3), the same product): kept in the lane record so
  // that the control itself need not stay in registers through the step loops
  const double tx = (NS == 7) ? L.cx * k : L.tx, ty = (NS == 7) ? L.cy * k : L.ty, tz = (NS == 7) ? L.cz * k : L.tz;
  dx[3] = __builtin_fma(-c1, a, __builtin_fma(-c2, b, __builtin_fma(L.w2, x[4], X))) + tx;        // :48
  dx[4] = __builtin_fma(-cs, Y, __builtin_fma(-L.w2, x[3], Y)) + ty;                              // :49
  dx[5] = __builtin_fma(-cs, Z, tz);                                                              // :50
  if (NS == 7) dx[NS - 1] = L.mdot;
  if (VAR) {
    const double e1 = 3.0 * c1 * i1s, e2 = 3.0 * c2 * i2s;
    const double ee = e1 + e2;
    const double sa = e1 * a, tb = e2 * b;
    const double st = sa + tb;
    vc.Gxx = __builtin_fma(sa, a, __builtin_fma(tb, b, 1.0 - cs));
    vc.Gyy = __builtin_fma(ee * Y, Y, 1.0 - cs);
    vc.Gzz = __builtin_fma(ee * Z, Z, -cs);
    vc.Gxy = st * Y; vc.Gxz = st * Z; vc.Gyz = ee * Y * Z;
    vc.k_over_m = k;
    if (NS == 7) {
      const double km = -k * inv_m;
      vc.dvdm_x = L.cx * km; vc.dvdm_y = L.cy * km; vc.dvdm_z = L.cz * km;
    }
  }
}

// One sensitivity column of the direct system: c = d x / d p for p an initial state component or a
// control component.  forcing_v = d vdot / d p (explicit), forcing_m = d mdot / d p (explicit).
template <int NS>
__device__ __forceinline__ void var_col_direct(const VarCoef6& vc, const double w2, const double (&c)[NS],
                                               const double fvx, const double fvy, const double fvz, const double fm,
                                               double (&dc)[NS]) {
  dc[0] = c[3]; dc[1] = c[4]; dc[2] = c[5];
  double bx = __builtin_fma(vc.Gxx, c[0], __builtin_fma(vc.Gxy, c[1], __builtin_fma(vc.Gxz, c[2], __builtin_fma(w2, c[4], fvx))));
  double by = __builtin_fma(vc.Gxy, c[0], __builtin_fma(vc.Gyy, c[1], __builtin_fma(vc.Gyz, c[2], __builtin_fma(-w2, c[3], fvy))));
  double bz = __builtin_fma(vc.Gxz, c[0], __builtin_fma(vc.Gyz, c[1], __builtin_fma(vc.Gzz, c[2], fvz)));
  if (NS == 7) {
    bx = __builtin_fma(vc.dvdm_x, c[NS - 1], bx);
    by = __builtin_fma(vc.dvdm_y, c[NS - 1], by);
    bz = __builtin_fma(vc.dvdm_z, c[NS - 1], bz);
    dc[NS - 1] = fm;
  }
  dc[3] = bx; dc[4] = by; dc[5] = bz;
}

}  // namespace lto
