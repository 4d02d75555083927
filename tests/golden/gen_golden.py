#!/usr/bin/env python3
"""Mint the golden vectors under tests/golden/ (run once in the build container; output committed).

The reference (Julia) cannot run here and ships no tests or golden vectors, so these vectors come from
restatements that are INDEPENDENT of both the C++ oracle (oracle/lto_oracle.cpp) and the HIP kernels:

  rhs_state_costate.json   CRTBP_stateCostate_deriv! (src/CRTBP_stateCostate_deriv.jl:9-90) evaluated with
                           mpmath at 40 digits.  The costate rows are NOT transcribed from the reference's
                           longhand (:83-85) nor from the G-matrix form used on the GPU: they are obtained as
                           lambda_r_dot = -grad_r( grad(Omega)(r) . lambda_v ) by high-order numerical
                           differentiation of the CRTBP pseudo-potential Omega.
  rhs_prop_ep.json         CRTBP_prop_EP_deriv (src/CRTBP_prop_EP_deriv.jl:8-61), mpmath 40 digits.
  flows_taylor.json        final states (and one STM by central differences) of a few demo-sized segments
                           integrated with mpmath's Taylor-series ODE solver at 30 digits.
  stm_taylor.json          (round 6) the 12x12 STM of one demo segment per control-law class -- p = 2 unclamped and clamped, p = 0,
                           p = 1.5, p = 1 at rho = 1e-2 -- by 4th-order central differences of the same 30-digit Taylor flow: the
                           independent pin of jacobianCalc's blocks (indirect.jl:121) for every branch of :36-53.
  direct_jacobian_ld.json  (round 6) the direct path's 6 x 18 Jacobian blocks and tf column of 8 segments: the numpy Fehlberg march in
                           80-bit precision, Richardson central differences -- the independent pin of jacobianCalc
                           (multiShoot_CRTBP_direct.jl:111-166) and of the tf partial (:503-516).
  flows_scipy.json         32 segments with scipy DOP853 at rtol = atol = 1e-13 (the reference's tolerance,
                           src/multiShoot_CRTBP_indirect.jl:79).
  direct_numpy.json        mid-point defects and maxErr of 8 segments from a numpy transliteration of ode7_8
                           (GeneralCode/ode.jl:875-952, with the same `f*beta_[:,j]` matrix-vector form) and of
                           the two-sided shooting of src/multiShoot_CRTBP_direct.jl:77-105.
  halo_kat.json            facts about the reference's own data files (Jacobi constant per column,
                           column->column propagation residuals) computed with scipy.

Usage: python tests/golden/gen_golden.py
"""
import json
from fractions import Fraction as F
import os
import sys

import mpmath as mp
import numpy as np
from scipy.integrate import solve_ivp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from lowthrustopt_amd import synth  # noqa: E402
from lowthrustopt_amd.constants import MU, DU, TU  # noqa: E402

mp.mp.dps = 40


# ---------------------------------------------------------------------------------------------- mpmath RHS
def mp_control(lv, thrustLimit, mass, p, rho):
    """Control law, stateCostate_deriv.jl:33-64."""
    accelLimit = mp.mpf(thrustLimit) / mass / 1000 * mp.mpf(TU) ** 2 / mp.mpf(DU)
    n = mp.sqrt(sum(x * x for x in lv))
    if p == 0:
        umag = accelLimit
    elif p == 1:
        umag = mp.mpf(1) / 2 * (1 + mp.tanh((n - 1) / (2 * mp.mpf(rho)))) * accelLimit
    elif p > 1:
        umag = (n / mp.mpf(p)) ** (1 / (mp.mpf(p) - 1))
        if umag > accelLimit:
            umag = accelLimit
    else:
        raise ValueError("Invalid value of p!")
    if n == 0:
        return [mp.mpf(0)] * 3
    return [-umag * x / n for x in lv]


def mp_grad_omega(r, mu):
    """Gradient of the CRTBP pseudo-potential Omega = (x^2+y^2)/2 + (1-mu)/r1 + mu/r2."""
    x, y, z = r
    r1 = mp.sqrt((x + mu) ** 2 + y ** 2 + z ** 2)
    r2 = mp.sqrt((x + mu - 1) ** 2 + y ** 2 + z ** 2)
    return [x - (1 - mu) * (x + mu) / r1 ** 3 - mu * (x + mu - 1) / r2 ** 3,
            y - (1 - mu) * y / r1 ** 3 - mu * y / r2 ** 3,
            -(1 - mu) * z / r1 ** 3 - mu * z / r2 ** 3]


def mp_hessian_omega(r, mu):
    """Hessian of Omega in closed form (textbook tidal tensor), used where mp.diff would be too slow/inexact."""
    x, y, z = r
    out = [[mp.mpf(0)] * 3 for _ in range(3)]
    out[0][0] = mp.mpf(1); out[1][1] = mp.mpf(1)
    for kappa, rho in ((1 - mu, [x + mu, y, z]), (mu, [x + mu - 1, y, z])):
        d = mp.sqrt(sum(v * v for v in rho))
        for i in range(3):
            out[i][i] -= kappa / d ** 3
            for j in range(3):
                out[i][j] += 3 * kappa * rho[i] * rho[j] / d ** 5
    return out


def mp_rhs_state_costate(y, prm, numeric_hessian=True):
    MUq, _, _, thrustLimit, mass, td, p, rho = prm
    mu = mp.mpf(MUq)
    y = [mp.mpf(v) for v in y]
    r, v, lr, lv = y[0:3], y[3:6], y[6:9], y[9:12]
    g = mp_grad_omega(r, mu)
    u = mp_control(lv, thrustLimit, mass, p, rho)
    td = mp.mpf(td)
    dv = [g[0] + 2 * td * v[1] + u[0], g[1] - 2 * td * v[0] + u[1], g[2] + u[2]]

    def phi(*rr):
        gg = mp_grad_omega(list(rr), mu)
        return gg[0] * lv[0] + gg[1] * lv[1] + gg[2] * lv[2]

    if numeric_hessian:
        # numerical differentiation at 3x the working precision: error far below 1e-25
        old = mp.mp.dps
        mp.mp.dps = 3 * old
        try:
            dlr = [-mp.diff(phi, tuple(r), n=tuple(1 if k == i else 0 for k in range(3))) for i in range(3)]
        finally:
            mp.mp.dps = old
        dlr = [+v for v in dlr]
    else:
        Hs = mp_hessian_omega(r, mu)
        dlr = [-(Hs[i][0] * lv[0] + Hs[i][1] * lv[1] + Hs[i][2] * lv[2]) for i in range(3)]
    dlv = [2 * lv[1] * td - lr[0], -lr[1] - 2 * lv[0] * td, -lr[2]]
    return v + dv + dlr + dlv


def mp_rhs_prop_ep(s, Isp, control, td):
    s = [mp.mpf(v) for v in s]
    c = [mp.mpf(v) for v in control]
    mu = mp.mpf(MU)
    m = s[6] if len(s) == 7 else mp.mpf(1000)
    g = mp_grad_omega(s[0:3], mu)
    nc = mp.sqrt(sum(x * x for x in c))
    Tmag = nc / m / 1000 * mp.mpf(TU) ** 2 / mp.mpf(DU)
    T = c if nc == 0 else [x / nc * Tmag for x in c]
    td = mp.mpf(td)
    out = s[3:6] + [g[0] + 2 * td * s[4] + T[0], g[1] - 2 * td * s[3] + T[1], g[2] + T[2]]
    if len(s) == 7:
        out.append(-td * nc / (mp.mpf(Isp) * mp.mpf("9.81")) * mp.mpf(TU))
    return out


def f64(v):
    return [float(x) for x in v]


# ---------------------------------------------------------------------------------------------- fixtures
def gen_rhs():
    rng = np.random.default_rng(11)
    H1, H2 = synth.halo_orbits()
    cases = []
    combos = [(1.0, 1.0, 0.05, 1.0), (1.0, 1e-2, 0.05, 1.0), (1.0, 1e-4, 0.05, 1.0), (2.0, 1.0, 10.0, 1.0),
              (2.0, 1.0, 0.05, 1.0), (1.5, 1.0, 10.0, 1.0), (1.5, 1.0, 0.001, 1.0), (0.0, 1.0, 0.05, 1.0),
              (1.0, 1.0, 0.05, -1.0), (2.0, 1.0, 10.0, -1.0), (1.0, 1e-3, 10.0, 1.0), (1.2, 0.5, 10.0, -1.0)]
    for k, (p, rho, thr, td) in enumerate(combos):
        for rep in range(2):
            tab = H1 if (k + rep) % 2 == 0 else H2
            y = np.concatenate([tab[:, rng.integers(0, 99)] + 1e-3 * rng.standard_normal(6),
                                (0.1 if rep == 0 else 1.2) * rng.standard_normal(6)])
            prm = [MU, DU, TU, thr, 1000.0, td, p, rho]
            cases.append({"y": f64(y), "prm": prm, "dy": [mp.nstr(v, 25) for v in mp_rhs_state_costate(y, prm)]})
    # lambda_v == 0 guard (stateCostate_deriv.jl:59-64) and |lambda_v| == 1 exactly (tanh argument 0)
    y = np.concatenate([H1[:, 5], [0.01, -0.02, 0.03, 0.0, 0.0, 0.0]])
    for p in (0.0, 1.0, 2.0):
        prm = [MU, DU, TU, 0.05, 1000.0, 1.0, p, 1.0]
        cases.append({"y": f64(y), "prm": prm, "dy": [mp.nstr(v, 25) for v in mp_rhs_state_costate(y, prm)]})
    y = np.concatenate([H2[:, 40], [0.3, 0.1, -0.2, 0.6, 0.0, 0.8]])
    prm = [MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1e-3]
    cases.append({"y": f64(y), "prm": prm, "dy": [mp.nstr(v, 25) for v in mp_rhs_state_costate(y, prm)]})
    json.dump({"doc": "mpmath 40-digit CRTBP_stateCostate_deriv!, 25 significant digits", "cases": cases},
              open(os.path.join(HERE, "rhs_state_costate.json"), "w"), indent=0)

    cases = []
    for k in range(10):
        tab = H1 if k % 2 == 0 else H2
        n = 6 if k < 6 else 7
        s = list(tab[:, rng.integers(0, 99)] + 1e-3 * rng.standard_normal(6))
        if n == 7:
            s.append(1000.0 - 3.0 * k)
        control = [0.0, 0.0, 0.0] if k in (2, 7) else list(0.05 * rng.standard_normal(3))
        td = 1.0 if k % 3 else -1.0
        cases.append({"s": f64(s), "Isp": 2000.0, "control": f64(control), "td": td,
                      "ds": [mp.nstr(v, 25) for v in mp_rhs_prop_ep(s, 2000.0, control, td)]})
    json.dump({"doc": "mpmath 40-digit CRTBP_prop_EP_deriv, 25 significant digits", "cases": cases},
              open(os.path.join(HERE, "rhs_prop_ep.json"), "w"), indent=0)


def gen_flows_taylor():
    mp.mp.dps = 30
    XC, T = synth.indirect_problem(30, seed=3)
    span = float(T[1, 0] - T[0, 0])
    cases = []
    for (node, p, rho, thr) in ((4, 1.0, 1.0, 0.05), (20, 2.0, 1.0, 10.0), (11, 1.0, 1e-2, 0.05)):
        y0 = XC[:, node, 0]
        prm = [MU, DU, TU, thr, 1000.0, 1.0, p, rho]

        def flow(yy):
            sol = mp.odefun(lambda t, y: mp_rhs_state_costate(y, prm, numeric_hessian=False), 0, [mp.mpf(float(v)) for v in yy], tol=mp.mpf(10) ** -26)
            return sol(mp.mpf(span))

        yf = flow(y0)
        case = {"y0": f64(y0), "prm": prm, "span": span, "yf": [mp.nstr(v, 22) for v in yf]}
        if node == 4:  # STM by 4th-order central differences of the Taylor flow
            d = 1e-6
            Phi = np.zeros((12, 12))
            for c in range(12):
                fs = {}
                for k in (-2, -1, 1, 2):
                    yp = np.array(y0, dtype=np.float64)
                    yp[c] = yp[c] + k * d
                    fs[k] = (flow(yp), mp.mpf(float(yp[c])) - mp.mpf(float(y0[c])))
                for r in range(12):
                    # Richardson on possibly slightly unequal perturbations (they are exact doubles)
                    d1 = (fs[1][0][r] - fs[-1][0][r]) / (fs[1][1] - fs[-1][1])
                    d2 = (fs[2][0][r] - fs[-2][0][r]) / (fs[2][1] - fs[-2][1])
                    Phi[r, c] = float((4 * d1 - d2) / 3)
            case["Phi_rowmajor"] = Phi.reshape(-1).tolist()
        cases.append(case)
        print("taylor flow node", node, "done", flush=True)
    mp.mp.dps = 40
    json.dump({"doc": "mpmath Taylor-series flows of CRTBP_stateCostate_deriv! over one demo segment, 22 digits",
               "cases": cases}, open(os.path.join(HERE, "flows_taylor.json"), "w"), indent=0)


def taylor_flow_and_stm(y0, prm, span, d=1e-6):
    """(final state, 12x12 STM) of one segment: mpmath Taylor flow at the working precision, STM by Richardson-extrapolated central
    differences of that flow (perturbations are exact doubles)."""
    def flow(yy):
        sol = mp.odefun(lambda t, y: mp_rhs_state_costate(y, prm, numeric_hessian=False), 0, [mp.mpf(float(v)) for v in yy], tol=mp.mpf(10) ** -26)
        return sol(mp.mpf(span))
    yf = flow(y0)
    Phi = np.zeros((12, 12))
    for c in range(12):
        fs = {}
        for k in (-2, -1, 1, 2):
            yp = np.array(y0, dtype=np.float64)
            yp[c] = yp[c] + k * d
            fs[k] = (flow(yp), mp.mpf(float(yp[c])) - mp.mpf(float(y0[c])))
        for r in range(12):
            d1 = (fs[1][0][r] - fs[-1][0][r]) / (fs[1][1] - fs[-1][1])
            d2 = (fs[2][0][r] - fs[-2][0][r]) / (fs[2][1] - fs[-2][1])
            Phi[r, c] = float((4 * d1 - d2) / 3)
    return yf, Phi


def gen_stm_taylor():
    """One segment per branch of the control law (stateCostate_deriv.jl:36-53) away from its kinks: the clamp of p > 1 is either far
    off (thrust limit 10 N, the demo's first setting, indirect_demo.jl:179) or firmly on (0.05 N with |lambda_v| ~ 1)."""
    mp.mp.dps = 30
    cases = []
    for (name, node, p, rho, thr, lam) in (("p2_unclamped", 7, 2.0, 1.0, 10.0, 0.1), ("p2_clamped", 9, 2.0, 1.0, 0.05, 1.0), ("p0", 13, 0.0, 1.0, 0.05, 0.1),
                                           ("p1.5_unclamped", 16, 1.5, 1.0, 10.0, 0.3), ("p1_rho1e-2", 23, 1.0, 1e-2, 0.05, 1.0)):
        XC, T = synth.indirect_problem(30, seed=5, lam_sigma=lam)
        span = float(T[1, 0] - T[0, 0])
        y0 = XC[:, node, 0]
        prm = [MU, DU, TU, thr, 1000.0, 1.0, p, rho]
        yf, Phi = taylor_flow_and_stm(y0, prm, span)
        cases.append({"name": name, "y0": f64(y0), "prm": prm, "span": span, "yf": [mp.nstr(v, 22) for v in yf], "Phi_rowmajor": Phi.reshape(-1).tolist()})
        print("taylor STM", name, "done", flush=True)
    mp.mp.dps = 40
    json.dump({"doc": "mpmath Taylor-series flow and STM (Richardson central differences of the flow) of CRTBP_stateCostate_deriv! over one demo "
                      "segment per control-law class", "cases": cases}, open(os.path.join(HERE, "stm_taylor.json"), "w"), indent=0)


def np_rhs_state_costate(y, prm):
    """numpy restatement through the pseudo-potential Hessian (independent of the oracle's longhand rows)."""
    MUq, DUq, TUq, thrustLimit, mass, td, p, rho = prm
    x, yy, z = y[0:3]
    lv = y[9:12]
    aL = thrustLimit / mass / 1e3 * TUq ** 2 / DUq
    n = np.linalg.norm(lv)
    if p == 0:
        umag = aL
    elif p == 1:
        umag = 0.5 * (1 + np.tanh((n - 1) / (2 * rho))) * aL
    else:
        umag = min((n / p) ** (1 / (p - 1)), aL)
    u = -umag * lv / n if n > 0 else np.zeros(3)
    r1v = np.array([x + MUq, yy, z]); r2v = np.array([x + MUq - 1, yy, z])
    r1 = np.linalg.norm(r1v); r2 = np.linalg.norm(r2v)
    g = np.array([x, yy, 0.0]) - (1 - MUq) * r1v / r1 ** 3 - MUq * r2v / r2 ** 3
    Hs = (np.diag([1.0, 1.0, 0.0]) - ((1 - MUq) / r1 ** 3 + MUq / r2 ** 3) * np.eye(3)
          + 3 * (1 - MUq) * np.outer(r1v, r1v) / r1 ** 5 + 3 * MUq * np.outer(r2v, r2v) / r2 ** 5)
    dv = g + 2 * td * np.array([y[4], -y[3], 0.0]) + u
    dlr = -Hs @ lv
    dlv = np.array([2 * lv[1] * td - y[6], -y[7] - 2 * lv[0] * td, -y[8]])
    return np.concatenate([y[3:6], dv, dlr, dlv])


def gen_flows_scipy():
    XC, T = synth.indirect_problem(33, seed=5)
    cases = []
    combos = [(1.0, 1.0, 0.05), (2.0, 1.0, 10.0), (1.0, 1e-2, 0.05), (1.5, 1.0, 10.0), (0.0, 1.0, 0.05), (1.0, 1e-3, 10.0)]
    for i in range(32):
        p, rho, thr = combos[i % len(combos)]
        prm = [MU, DU, TU, thr, 1000.0, 1.0, p, rho]
        y0 = XC[:, i, 0]
        span = float(T[i + 1, 0] - T[i, 0])
        sol = solve_ivp(lambda t, y: np_rhs_state_costate(y, prm), (0.0, span), y0, method="DOP853", rtol=1e-13, atol=1e-13)
        cases.append({"y0": f64(y0), "prm": prm, "span": span, "yf": f64(sol.y[:, -1]), "nfev": int(sol.nfev)})
    json.dump({"doc": "scipy DOP853 rtol=atol=1e-13 flows of the numpy pseudo-potential restatement", "cases": cases},
              open(os.path.join(HERE, "flows_scipy.json"), "w"), indent=0)


# ---------------------------------------------------------------------------------------------- direct path (numpy)
def np_prop_ep(t, state, MUq, DUq, TUq, Isp, control, td):
    """CRTBP_prop_EP_deriv, src/CRTBP_prop_EP_deriv.jl:8-61, statement by statement."""
    x, y, z, xdot, ydot, zdot = state[0:6]
    m = state[6] if len(state) == 7 else 1000.0
    r1 = np.sqrt((x + MUq) ** 2 + y ** 2 + z ** 2)
    r2 = np.sqrt((x + MUq - 1) ** 2 + y ** 2 + z ** 2)
    r1_3 = r1 ** 3
    r2_3 = r2 ** 3
    nc = np.linalg.norm(control)
    T_mag = nc / m / 1e3 * TUq ** 2 / DUq
    T = control if nc == 0 else control / nc * T_mag
    mdot = -td * nc / (Isp * 9.81) * TUq
    xdd = -(1 - MUq) * (x + MUq) / r1_3 - MUq * (x - 1 + MUq) / r2_3 + 2 * td * ydot + x + T[0]
    ydd = -(1 - MUq) * y / r1_3 - MUq * y / r2_3 - 2 * td * xdot + y + T[1]
    zdd = -(1 - MUq) * z / r1_3 - MUq * z / r2_3 + T[2]
    out = [xdot, ydot, zdot, xdd, ydd, zdd]
    if len(state) == 7:
        out.append(mdot)
    return np.array(out)


class FixedGridFehlberg78:
    """Fehlberg's 13-stage 7(8) pair marched over a GIVEN grid, no step-size control: an independent statement of what the reference's
    `ode7_8` computes (GeneralCode/ode.jl:773-953), used only to mint direct_numpy.json.

    The tableau is Fehlberg's published one (NASA TR R-287, 1968, table of the RK7(8) formula), kept here stage by stage as
    {earlier stage: coefficient}; the dense 13 x 13 array `self.lower` has stage s's row as its COLUMN s, so that a stage argument is
    ONE matrix-vector product over all 13 slope columns (zeros included), x + (h K) @ lower[:, s] -- the association the reference
    uses (`x + h*(F*beta[:, j])` with F scaled first), which is what makes the vectors reproducible to the bit.
    The 8th-order solution uses stages 5 ... 9 and 11, 12; the error indicator is 41/840 h (k0 + k10 - k11 - k12), largest component.
    """
    NODES = (0, F(2, 27), F(1, 9), F(1, 6), F(5, 12), F(1, 2), F(5, 6), F(1, 6), F(2, 3), F(1, 3), 1, 0, 1)
    STAGES = {
        1: {0: F(2, 27)},
        2: {0: F(1, 36), 1: F(1, 12)},
        3: {0: F(1, 24), 2: F(1, 8)},
        4: {0: F(5, 12), 2: F(-25, 16), 3: F(25, 16)},
        5: {0: F(1, 20), 3: F(1, 4), 4: F(1, 5)},
        6: {0: F(-25, 108), 3: F(125, 108), 4: F(-65, 27), 5: F(125, 54)},
        7: {0: F(31, 300), 4: F(61, 225), 5: F(-2, 9), 6: F(13, 900)},
        8: {0: 2, 3: F(-53, 6), 4: F(704, 45), 5: F(-107, 9), 6: F(67, 90), 7: 3},
        9: {0: F(-91, 108), 3: F(23, 108), 4: F(-976, 135), 5: F(311, 54), 6: F(-19, 60), 7: F(17, 6), 8: F(-1, 12)},
        10: {0: F(2383, 4100), 3: F(-341, 164), 4: F(4496, 1025), 5: F(-301, 82), 6: F(2133, 4100), 7: F(45, 82), 8: F(45, 164), 9: F(18, 41)},
        11: {0: F(3, 205), 5: F(-6, 41), 6: F(-3, 205), 7: F(-3, 41), 8: F(3, 41), 9: F(6, 41)},
        12: {0: F(-1777, 4100), 3: F(-341, 164), 4: F(4496, 1025), 5: F(-289, 82), 6: F(2193, 4100), 7: F(51, 82), 8: F(33, 164), 9: F(12, 41), 11: 1},
    }
    WEIGHTS8 = {5: F(34, 105), 6: F(9, 35), 7: F(9, 35), 8: F(9, 280), 9: F(9, 280), 11: F(41, 840), 12: F(41, 840)}
    INDICATOR = {0: 1, 10: 1, 11: -1, 12: -1}

    def __init__(self):
        # numerator / denominator in binary64, as the reference's literals `2383/4100` are evaluated
        q = lambda v: float(v.numerator) / float(v.denominator) if isinstance(v, F) else float(v)   # noqa: E731
        self.nodes = [q(c) for c in self.NODES]
        self.lower = np.zeros((13, 13))
        for s, row in self.STAGES.items():
            for k, v in row.items():
                self.lower[k, s] = q(v)
        self.weights = np.zeros(13)
        for k, v in self.WEIGHTS8.items():
            self.weights[k] = q(v)
        self.indicator = np.zeros(13)
        for k, v in self.INDICATOR.items():
            self.indicator[k] = q(v)

    def step(self, rhs, t, x, h, args):
        """One step from (t, x): (x_next, largest component of the error indicator)."""
        K = np.zeros((len(x), 13))
        K[:, 0] = rhs(t, x, *args)
        for s in range(1, 13):
            K[:, s] = rhs(t + self.nodes[s] * h, x + (h * K) @ self.lower[:, s], *args)
        x_next = x + (h * K) @ self.weights
        return x_next, np.linalg.norm(((h * 41 / 840) * K) @ self.indicator, np.inf)

    def march(self, rhs, grid, x0, *args):
        """States at every grid point [n x len(grid)] and the largest indicator met on the way."""
        out = np.zeros((len(x0), len(grid)))
        out[:, 0] = x0
        worst = 0.0
        for k in range(1, len(grid)):
            out[:, k], e = self.step(rhs, grid[k - 1], out[:, k - 1].copy(), grid[k] - grid[k - 1], args)
            worst = max(worst, e)
        return out, worst


_RKF78 = FixedGridFehlberg78()


def np_ode7_8(odefun, tspan, x0, *args):
    return _RKF78.march(odefun, tspan, x0, *args)


def np_direct_defect(X_all, u_all, t_TU, nstate, n_nodes, nsteps, Isp):
    """defectCalc, src/multiShoot_CRTBP_direct.jl:66-109."""
    t_mid = t_TU[:-1] + np.diff(t_TU) / 2
    defect = np.zeros((nstate, n_nodes - 1))
    errors = np.zeros(n_nodes - 1)
    for i in range(n_nodes - 1):
        tspan = np.linspace(t_TU[i], t_mid[i], nsteps)
        sf, ef = np_ode7_8(np_prop_ep, tspan, X_all[:, i].copy(), MU, DU, TU, Isp, u_all[:, i].copy(), 1.0)
        x0 = X_all[:, i + 1].copy()
        x0[3:6] = -x0[3:6]
        sb, eb = np_ode7_8(np_prop_ep, tspan, x0, MU, DU, TU, Isp, u_all[:, i + 1].copy(), -1.0)
        xb = sb[:, -1].copy()
        xb[3:6] = -xb[3:6]
        defect[:, i] = sf[:, -1] - xb
        errors[i] = max(ef, eb)
    return defect, errors


def gen_direct():
    out = {"doc": "independent numpy statement of the fixed-grid Fehlberg 7(8) march (what ode7_8 computes) + the direct two-sided defect, nsteps=10", "cases": []}
    for nstate, seed in ((6, 0), (7, 1)):
        X, U, T = synth.direct_problem(9, seed=seed, nstate=nstate)
        X, U, t = X[:, :, 0], U[:, :, 0], T[:, 0]
        if nstate == 6:
            U[:, 3] = 0.0  # exercise the zero-control branch (prop_EP_deriv.jl:35-36)
        d, e = np_direct_defect(X, U, t, nstate, 9, 10, 2000.0)
        out["cases"].append({"nstate": nstate, "nsteps": 10, "Isp": 2000.0, "X": X.T.tolist(), "U": U.T.tolist(),
                             "t": t.tolist(), "defect": d.T.tolist(), "errors": e.tolist()})
    json.dump(out, open(os.path.join(HERE, "direct_numpy.json"), "w"), indent=0)


def gen_direct_jacobian():
    """(round 6) The direct path's Jacobian blocks and tf column, independently: the numpy restatement above run in 80-bit extended
    precision (np.longdouble, eps 1.1e-19) and differentiated by Richardson-extrapolated central differences (steps 1e-5 and 2e-5 in
    every variable: truncation ~1e-15, rounding ~1e-14) -- jacobianCalc's 18 columns (src/multiShoot_CRTBP_direct.jl:111-166; the
    reference itself uses forward differences with pert 1e-8, i.e. ~1e-8 of noise) and the tf partial (:503-516; there a central
    difference of +-1e-3).  Same inputs as direct_numpy.json's 6-state case."""
    LD = np.longdouble
    X, U, T = synth.direct_problem(9, seed=0, nstate=6)
    X, U, t = X[:, :, 0], U[:, :, 0], T[:, 0]
    U[:, 3] = 0.0
    n, S = 9, 8

    def march(x0, grid, Isp, ctrl, td):
        x = x0.astype(LD)
        for k in range(1, len(grid)):
            h = grid[k] - grid[k - 1]
            K = np.zeros((6, 13), dtype=LD)
            K[:, 0] = np_prop_ep_ld(x, ctrl, td)
            for s_ in range(1, 13):
                K[:, s_] = np_prop_ep_ld(x + (h * K) @ _RKF78.lower[:, s_].astype(LD), ctrl, td)
            x = x + (h * K) @ _RKF78.weights.astype(LD)
        return x

    def np_prop_ep_ld(state, control, td):
        x, y, z, xdot, ydot, zdot = state
        MUq, DUq, TUq = LD(MU), LD(DU), LD(TU)
        m = LD(1000.0)
        r1 = np.sqrt((x + MUq) ** 2 + y ** 2 + z ** 2)
        r2 = np.sqrt((x + MUq - 1) ** 2 + y ** 2 + z ** 2)
        Tacc = control / m / LD(1e3) * TUq ** 2 / DUq          # = control / |control| * T_mag of prop_EP_deriv.jl:32-39, which is linear in control
        xdd = -(1 - MUq) * (x + MUq) / r1 ** 3 - MUq * (x - 1 + MUq) / r2 ** 3 + 2 * td * ydot + x + Tacc[0]
        ydd = -(1 - MUq) * y / r1 ** 3 - MUq * y / r2 ** 3 - 2 * td * xdot + y + Tacc[1]
        zdd = -(1 - MUq) * z / r1 ** 3 - MUq * z / r2 ** 3 + Tacc[2]
        return np.array([xdot, ydot, zdot, xdd, ydd, zdd], dtype=LD)

    def seg_defect(xi, xj, ui, uj, ti, tj):
        grid = np.linspace(LD(ti), LD(ti) + (LD(tj) - LD(ti)) / 2, 10, dtype=LD)
        xf = march(xi, grid, 2000.0, ui.astype(LD), LD(1.0))
        x0 = xj.astype(LD).copy(); x0[3:6] = -x0[3:6]
        xb = march(x0, grid, 2000.0, uj.astype(LD), LD(-1.0))
        xb[3:6] = -xb[3:6]
        return xf - xb

    jac = np.zeros((S, 6, 18)); dtf = np.zeros((S, 6)); dfc = np.zeros((S, 6))
    span = LD(t[-1]) - LD(t[0])
    for i in range(S):
        base = [X[:, i].astype(LD), X[:, i + 1].astype(LD), U[:, i].astype(LD), U[:, i + 1].astype(LD)]
        dfc[i] = seg_defect(*base, t[i], t[i + 1]).astype(np.float64)
        col = 0
        for blk, size, h in ((0, 6, LD(1e-5)), (1, 6, LD(1e-5)), (2, 3, LD(1e-5)), (3, 3, LD(1e-5))):
            for c in range(size):
                def at(k):
                    v = [b.copy() for b in base]
                    v[blk][c] = v[blk][c] + k * h
                    return seg_defect(*v, t[i], t[i + 1])
                d1 = (at(1) - at(-1)) / (2 * h)
                d2 = (at(2) - at(-2)) / (4 * h)
                jac[i, :, col] = ((4 * d1 - d2) / 3).astype(np.float64)
                col += 1
        # tf partial: every interval scales with (tf - t0); d defect / d tf by the same Richardson scheme on the scale factor
        def at_tf(k, e=LD(1e-5)):
            sc = (span + k * e) / span
            ti, tj = LD(t[0]) + (LD(t[i]) - LD(t[0])) * sc, LD(t[0]) + (LD(t[i + 1]) - LD(t[0])) * sc
            return seg_defect(*base, ti, tj)
        d1 = (at_tf(1) - at_tf(-1)) / (2 * LD(1e-5)); d2 = (at_tf(2) - at_tf(-2)) / (4 * LD(1e-5))
        dtf[i] = ((4 * d1 - d2) / 3).astype(np.float64)
        print("direct jacobian segment", i, "done", flush=True)
    json.dump({"doc": "direct two-sided defect (nsteps = 10, Isp 2000, 6-state), its 6 x 18 Jacobian wrt [x_i; x_{i+1}; u_i; u_{i+1}] and its tf partial "
                      "from the numpy Fehlberg march in 80-bit precision + Richardson central differences",
               "nsteps": 10, "Isp": 2000.0, "X": X.T.tolist(), "U": U.T.tolist(), "t": t.tolist(), "defect": dfc.tolist(), "jac": jac.tolist(),
               "dtf": dtf.tolist()}, open(os.path.join(HERE, "direct_jacobian_ld.json"), "w"), indent=0)


def gen_halo():
    H = synth.halo_orbits()
    out = {"doc": "facts about the reference's L2_Anderson_{1,2}.txt", "orbits": []}
    for k in (0, 1):
        tab = H[k]
        r1 = np.sqrt((tab[0] + MU) ** 2 + tab[1] ** 2 + tab[2] ** 2)
        r2 = np.sqrt((tab[0] + MU - 1) ** 2 + tab[1] ** 2 + tab[2] ** 2)
        C = tab[0] ** 2 + tab[1] ** 2 + 2 * (1 - MU) / r1 + 2 * MU / r2 - (tab[3] ** 2 + tab[4] ** 2 + tab[5] ** 2)
        res = []
        for c in range(0, 99, 11):
            sol = solve_ivp(lambda t, s: np_prop_ep(t, s, MU, DU, TU, 2000.0, np.zeros(3), 1.0), (0, synth.HALO_DT[k]),
                            tab[:, c], method="DOP853", rtol=1e-13, atol=1e-13)
            res.append(float(np.abs(sol.y[:, -1] - tab[:, c + 1]).max()))
        out["orbits"].append({"dt": synth.HALO_DT[k], "closure": float(np.abs(tab[:, 0] - tab[:, 99]).max()),
                              "jacobi_mean": float(C.mean()), "jacobi_spread": float(C.max() - C.min()),
                              "col_to_col_residual_max": max(res)})
    json.dump(out, open(os.path.join(HERE, "halo_kat.json"), "w"), indent=0)


if __name__ == "__main__":
    which = sys.argv[1:] or ["rhs", "scipy", "direct", "halo", "taylor"]
    if "rhs" in which:
        gen_rhs(); print("rhs done", flush=True)
    if "scipy" in which:
        gen_flows_scipy(); print("scipy done", flush=True)
    if "direct" in which:
        gen_direct(); print("direct done", flush=True)
    if "halo" in which:
        gen_halo(); print("halo done", flush=True)
    if "taylor" in which:
        gen_flows_taylor(); print("taylor done", flush=True)
    if "direct_jacobian" in which:     # (round 6; likewise)
        gen_direct_jacobian(); print("direct_jacobian done", flush=True)
    if "stm_taylor" in which:          # (round 6; not part of the default list: the older files are not regenerated)
        gen_stm_taylor(); print("stm_taylor done", flush=True)
