"""CPU: pin the oracle (oracle/lto_oracle.cpp) against the golden vectors of tests/golden/ and the
reference-independent known-answer tests of SURVEY.md section 8c.  No GPU involved."""
import json
import os

import numpy as np
import pytest

from lowthrustopt_amd import synth
from lowthrustopt_amd.constants import MU, DU, TU

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def test_rhs_state_costate_vs_mpmath(oracle):
    """A1 (longhand transcription of stateCostate_deriv.jl:9-90) vs a 40-digit evaluation that derives the
    costate rows from the pseudo-potential instead: tolerance 2e-13 relative to the row scale (the longhand
    `^(5/2)` chain loses a few ulps near the Moon)."""
    cases = load("rhs_state_costate.json")["cases"]
    assert len(cases) >= 25
    worst = 0.0
    for c in cases:
        dy = oracle.rhs_state_costate(c["y"], c["prm"])
        ref = np.array([float(v) for v in c["dy"]])
        scale = np.maximum(np.abs(ref), np.abs(ref).max() * 1e-3) + 1e-300
        worst = max(worst, float((np.abs(dy - ref) / scale).max()))
        # the binary128 evaluation of the same text must agree much tighter (rules out a wrong transcription
        # hiding behind binary64 round-off)
        hi, lo = oracle.rhs_state_costate_q(c["y"], c["prm"])
        assert np.all(np.abs(hi - ref) <= 4e-16 * scale + 1e-300)
    assert worst < 2e-13, worst


def test_rhs_lambda_v_zero_guard(oracle):
    """stateCostate_deriv.jl:59-64: norm(lambda_v) == 0 -> control set to zero, no NaN."""
    y = np.array([1.1, 0.05, 0.02, 0.01, -0.1, 0.03, 0.2, -0.1, 0.3, 0.0, 0.0, 0.0])
    for p in (0.0, 1.0, 2.0, 1.5):
        dy = oracle.rhs_state_costate(y, [MU, DU, TU, 0.05, 1000.0, 1.0, p, 1.0])
        assert np.all(np.isfinite(dy))


def test_rhs_invalid_p(oracle):
    """stateCostate_deriv.jl:52: error("Invalid value of p!") for p < 0 and 0 < p < 1."""
    y = np.ones(12)
    for p in (-1.0, 0.5):
        with pytest.raises(ValueError):
            oracle.rhs_state_costate(y, [MU, DU, TU, 0.05, 1000.0, 1.0, p, 1.0])


def test_rhs_prop_ep_vs_mpmath(oracle):
    cases = load("rhs_prop_ep.json")["cases"]
    for c in cases:
        ds = oracle.rhs_prop_ep(c["s"], MU, DU, TU, c["Isp"], c["control"], c["td"])
        ref = np.array([float(v) for v in c["ds"]])
        assert np.all(np.abs(ds - ref) <= 5e-15 * np.maximum(np.abs(ref), 1.0)), (ds - ref)


def test_rhs_jacobian_dual_vs_fd(oracle):
    """Dual-number Jacobian of A1 vs central differences; trace F = 0 (Hamiltonian flow)."""
    rng = np.random.default_rng(2)
    H1 = synth.halo_orbits()[0]
    for p, rho, thr in ((1.0, 1.0, 0.05), (2.0, 1.0, 10.0), (1.5, 1.0, 10.0), (1.0, 1e-2, 0.05), (0.0, 1.0, 0.05)):
        y = np.concatenate([H1[:, 17], rng.standard_normal(6)])
        prm = [MU, DU, TU, thr, 1000.0, 1.0, p, rho]
        J = oracle.rhs_state_costate_jac(y, prm)
        Jfd = np.zeros((12, 12))
        for c in range(12):
            d = 1e-6
            yp = y.copy(); yp[c] += d
            ym = y.copy(); ym[c] -= d
            Jfd[:, c] = (oracle.rhs_state_costate(yp, prm) - oracle.rhs_state_costate(ym, prm)) / (yp[c] - ym[c])
        assert np.abs(J - Jfd).max() < 1e-6 * max(1.0, np.abs(J).max())
        assert abs(np.trace(J)) < 1e-12


def test_flows_vs_scipy_dop853(oracle):
    """Oracle adaptive integrators at the reference tolerance (1e-13) vs scipy DOP853 on an independent
    numpy restatement: 32 demo-sized segments, agreement 5e-13 absolute."""
    cases = load("flows_scipy.json")["cases"]
    assert len(cases) == 32
    for c in cases:
        for method in (oracle.DOP853_ADAPTIVE, oracle.RKF78_ADAPTIVE):
            yf, rc, na, nr = oracle.flow_state_costate(c["y0"], c["prm"], c["span"], method)
            assert rc == 0
            assert np.abs(yf - np.array(c["yf"])).max() < 5e-13
        # fixed-step RKF7(8), 20 steps per segment, is converged to 2e-12 even near the Moon
        yf, rc, _, _ = oracle.flow_state_costate(c["y0"], c["prm"], c["span"], oracle.RKF78_FIXED, 20)
        assert np.abs(yf - np.array(c["yf"])).max() < 2e-12


def test_flows_vs_taylor(oracle):
    """Oracle flows vs mpmath Taylor-series integration (30 digits) of the pseudo-potential form."""
    cases = load("flows_taylor.json")["cases"]
    for c in cases:
        ref = np.array([float(v) for v in c["yf"]])
        yq, lo = oracle.flow_state_costate_q(c["y0"], c["prm"], c["span"], oracle.RKF78_FIXED, 40)
        assert np.abs(yq - ref).max() < 1e-15 * max(1.0, np.abs(ref).max()) * 4
        ya, rc, _, _ = oracle.flow_state_costate(c["y0"], c["prm"], c["span"], oracle.DOP853_ADAPTIVE)
        assert np.abs(ya - ref).max() < 3e-13
        if "Phi_rowmajor" in c:
            Phi_ref = np.array(c["Phi_rowmajor"]).reshape(12, 12)
            _, Phi, rc, _, _ = oracle.flow_stm_state_costate(c["y0"], c["prm"], c["span"], oracle.DOP853_ADAPTIVE)
            assert np.abs(Phi - Phi_ref).max() < 1e-9 * np.abs(Phi_ref).max()


def test_stm_vs_taylor_per_control_law_class(oracle):
    """Round 6: the oracle's STM (the same solve on dual numbers, what ForwardDiff.jacobian does at indirect.jl:121) for one demo
    segment per branch of the control law against tests/golden/stm_taylor.json -- central differences of a 30-digit mpmath Taylor flow of
    an independent mpmath restatement of the RHS."""
    cases = load("stm_taylor.json")["cases"]
    assert [c["name"] for c in cases] == ["p2_unclamped", "p2_clamped", "p0", "p1.5_unclamped", "p1_rho1e-2"]
    for c in cases:
        ref = np.array([float(v) for v in c["yf"]])
        Phi_ref = np.array(c["Phi_rowmajor"]).reshape(12, 12)
        ya, Phi, rc, _, _ = oracle.flow_stm_state_costate(c["y0"], c["prm"], c["span"], oracle.DOP853_ADAPTIVE)
        assert rc == 0 and np.abs(ya - ref).max() < 3e-13
        assert np.abs(Phi - Phi_ref).max() < 1e-12 * np.abs(Phi_ref).max(), c["name"]          # measured: 9e-15 ... 4e-14
        y4, Phi4, rc, _, _ = oracle.flow_stm_state_costate(c["y0"], c["prm"], c["span"], oracle.RK4, 256)
        assert np.abs(Phi4 - Phi_ref).max() < 1e-8 * np.abs(Phi_ref).max(), c["name"]
        # the flow is symplectic for every branch away from its kinks: Phi^T Omega Phi = Omega
        Om = np.zeros((12, 12)); Om[:6, 6:] = np.eye(6); Om[6:, :6] = -np.eye(6)
        assert np.abs(Phi_ref.T @ Om @ Phi_ref - Om).max() < 1e-7 * max(1.0, np.abs(Phi_ref).max() ** 2)


def test_rk4_order(oracle):
    """RK4 (ode.jl:64-68) shows 4th-order convergence on a demo segment."""
    c = load("flows_scipy.json")["cases"][0]
    ref, _ = oracle.flow_state_costate_q(c["y0"], c["prm"], c["span"], oracle.RKF78_FIXED, 40)
    errs = []
    for n in (16, 32, 64, 128):
        y, _, _, _ = oracle.flow_state_costate(c["y0"], c["prm"], c["span"], oracle.RK4, n)
        errs.append(np.linalg.norm(y - ref) / np.linalg.norm(ref))
    rates = [np.log2(errs[i] / errs[i + 1]) for i in range(3)]
    assert all(3.7 < r < 4.3 for r in rates[:2]), (errs, rates)
    assert errs[2] < 1e-10   # the C2 configuration (64 steps) is inside the 1e-10 defect budget


def test_stm_symplectic_and_det(oracle):
    """12x12 Phi of the state+costate flow is symplectic and has det 1."""
    XC, T = synth.indirect_problem(30, seed=7)
    Om = np.block([[np.zeros((6, 6)), np.eye(6)], [-np.eye(6), np.zeros((6, 6))]])
    for p, rho, thr in ((1.0, 1.0, 0.05), (1.0, 1e-2, 0.05), (2.0, 1.0, 10.0)):
        prm = [MU, DU, TU, thr, 1000.0, 1.0, p, rho]
        Phi, defect, rc = oracle.indirect_jacobian(XC[:, :, 0], T[:, 0], prm, oracle.DOP853_ADAPTIVE)
        assert rc == 0
        for i in range(Phi.shape[2]):
            P = Phi[:, :, i]
            assert np.abs(P.T @ Om @ P - Om).max() < 5e-11 * max(1.0, np.abs(P).max() ** 2)
            assert abs(np.linalg.det(P) - 1.0) < 1e-9


def test_stm_dual_vs_fd_of_discrete_map(oracle):
    """Dual-number STM through fixed-step RK4 equals the derivative of the same discrete map (FD, 1e-7)."""
    XC, T = synth.indirect_problem(4, seed=9)
    prm = [MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1.0]
    y0 = XC[:, 1, 0]; span = T[2, 0] - T[1, 0]
    _, Phi, rc, _, _ = oracle.flow_stm_state_costate(y0, prm, span, oracle.RK4, 64)
    for c in range(12):
        d = 1e-6
        yp = y0.copy(); yp[c] += d
        ym = y0.copy(); ym[c] -= d
        fp, _, _, _ = oracle.flow_state_costate(yp, prm, span, oracle.RK4, 64)
        fm, _, _, _ = oracle.flow_state_costate(ym, prm, span, oracle.RK4, 64)
        assert np.abs((fp - fm) / (yp[c] - ym[c]) - Phi[:, c]).max() < 2e-7 * max(1.0, np.abs(Phi).max())


def test_halo_files_known_answers(oracle):
    """The reference's own data: column -> column ballistic propagation (1e-8) and Jacobi constant."""
    kat = load("halo_kat.json")["orbits"]
    for k, tab in enumerate(synth.halo_orbits()):
        assert tab.shape == (6, 100)
        assert np.abs(tab[:, 0] - tab[:, 99]).max() < 2e-9      # closed orbit
        for c in (0, 33, 77):
            x, _ = oracle.flow_prop_ep(tab[:, c], [0, 0, 0], 1.0, synth.HALO_DT[k], oracle.RKF78_FIXED, 9, MU, DU, TU, 2000.0)
            assert np.abs(x - tab[:, c + 1]).max() < 1e-8
        # Jacobi constant (src/HelperFunctions.jl:10-15) is conserved along a ballistic arc
        s0 = tab[:, 10]
        s1, _ = oracle.flow_prop_ep(s0, [0, 0, 0], 1.0, 1.0, oracle.DOP853_ADAPTIVE, 0, MU, DU, TU, 2000.0)

        def jac(s):
            r1 = np.sqrt((s[0] + MU) ** 2 + s[1] ** 2 + s[2] ** 2)
            r2 = np.sqrt((s[0] + MU - 1) ** 2 + s[1] ** 2 + s[2] ** 2)
            return s[0] ** 2 + s[1] ** 2 + 2 * (1 - MU) / r1 + 2 * MU / r2 - np.dot(s[3:6], s[3:6])
        assert abs(jac(s1) - jac(s0)) < 1e-12
        assert abs(jac(s0) - kat[k]["jacobi_mean"]) < 5e-8


def test_direct_defect_vs_numpy_transliteration(oracle):
    """A3/A4 (ode7_8 + two-sided shooting) vs the numpy transliteration: 1e-14."""
    for c in load("direct_numpy.json")["cases"]:
        X = np.array(c["X"]).T; U = np.array(c["U"]).T; t = np.array(c["t"])
        d, e = oracle.direct_defect(X, U, t, c["nsteps"], MU, DU, TU, c["Isp"])
        assert np.abs(d - np.array(c["defect"]).T).max() < 1e-14
        assert np.abs(e - np.array(c["errors"])).max() < 1e-17 + 1e-3 * np.abs(c["errors"]).max()


def test_direct_forward_backward_round_trip(oracle):
    """direct.jl:90-98: propagating the forward result backward (td=-1, flipped velocity) returns to the
    start; so the defect vanishes when node i+1 is the exact propagation of node i."""
    X, U, T = synth.direct_problem(2, seed=4)
    x0 = X[:, 0, 0]; u = U[:, 0, 0]; span = T[1, 0] - T[0, 0]
    x1, _ = oracle.flow_prop_ep(x0, u, 1.0, span, oracle.RKF78_FIXED, 18, MU, DU, TU, 2000.0)
    Xp = np.stack([x0, x1], axis=1); Up = np.stack([u, u], axis=1)
    d, e = oracle.direct_defect(Xp, Up, T[:, 0], 10, MU, DU, TU, 2000.0)
    assert np.abs(d).max() < 1e-13


def test_direct_jacobian_fd_vs_dual(oracle):
    """A5 forward differences (pert 1e-8, direct.jl:123-143) vs exact dual-number derivative of the same
    discrete map: FD noise only (1e-6 relative); A6 tf partial vs d/dh."""
    for nstate in (6, 7):
        X, U, T = synth.direct_problem(8, seed=2, nstate=nstate)
        X, U, t = X[:, :, 0], U[:, :, 0], T[:, 0]
        d, e = oracle.direct_defect(X, U, t, 10, MU, DU, TU, 2000.0)
        Jfd = oracle.direct_jacobian_fd(X, U, t, d, 10, MU, DU, TU, 2000.0)
        Jd, dh, dd = oracle.direct_jacobian_dual(X, U, t, 10, MU, DU, TU, 2000.0)
        assert np.abs(dd - d).max() < 1e-15
        assert np.abs(Jfd[:6] - Jd[:6]).max() < 2e-6 * max(1.0, np.abs(Jd).max())
        if nstate == 7:
            # mass row: the forward-difference quotient cancels at the magnitude of the mass itself
            # (1000 kg * eps / 1e-8 ~ 1e-5 .. 1e-4), the dual-number derivative does not
            assert np.abs(Jfd[6] - Jd[6]).max() < 5e-4
        dtf = oracle.direct_dtf_fd(X, U, t, 10, MU, DU, TU, 2000.0)
        hseg = np.diff(t)
        dtf_exact = dh * (hseg / (t[-1] - t[0]))[None, :]
        assert np.abs(dtf - dtf_exact).max() < 1e-6


def test_direct_jacobian_vs_extended_precision_golden(oracle):
    """Round 6: the oracle's derivative of the direct two-sided defect (dual numbers through the RKF7(8) march) and its tf partial
    (d/dh) against tests/golden/direct_jacobian_ld.json -- an independent numpy restatement run in 80-bit precision and differentiated
    by Richardson central differences.  The reference's own forward differences (pert 1e-8, direct.jl:123-143) sit 1e-7 from both."""
    g = load("direct_jacobian_ld.json")
    X, U, t = np.array(g["X"]).T, np.array(g["U"]).T, np.array(g["t"])
    d, e = oracle.direct_defect(X, U, t, g["nsteps"], MU, DU, TU, g["Isp"])
    assert np.abs(d - np.array(g["defect"]).T).max() < 1e-14
    J_ref = np.array(g["jac"]).transpose(1, 2, 0)
    Jd, dh, _ = oracle.direct_jacobian_dual(X, U, t, g["nsteps"], MU, DU, TU, g["Isp"])
    assert np.abs(Jd - J_ref).max() < 1e-12 * np.abs(J_ref).max()              # measured 1.4e-14
    dtf = dh * (np.diff(t) / (t[-1] - t[0]))[None, :]
    assert np.abs(dtf - np.array(g["dtf"]).T).max() < 1e-12                    # measured 1.2e-14
    Jfd = oracle.direct_jacobian_fd(X, U, t, d, g["nsteps"], MU, DU, TU, g["Isp"])
    assert 1e-10 < np.abs(Jfd - J_ref).max() < 2e-6 * np.abs(J_ref).max()      # the reference's method: FD noise


def test_indirect_scatter_dense_shape_and_mask(oracle):
    """jacobianCalc band scatter (indirect.jl:128-142): shape, [Phi | -I] placement, zeroed end-state columns."""
    XC, T = synth.indirect_problem(5, seed=1)
    prm = [MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1.0]
    Phi, defect, rc = oracle.indirect_jacobian(XC[:, :, 0], T[:, 0], prm, oracle.RK4, 16)
    J = oracle.indirect_scatter_dense(Phi)
    assert J.shape == (48, 60)
    assert np.all(J[:, 0:6] == 0) and np.all(J[:, 48:54] == 0)
    assert np.array_equal(J[12:24, 12:24], Phi[:, :, 1])
    assert np.array_equal(J[12:24, 24:36], -np.eye(12))
    assert np.array_equal(J[0:12, 6:12], Phi[:, 6:12, 0])
    assert np.count_nonzero(J[0:12, 36:]) == 0


def test_direct_errors_output_is_conditioned_at_the_rounding_level(oracle):
    """`errors` (direct.jl:104; ode.jl:940-943) at the demo's step size is ~1e-19 ... 1e-15: h*41/840 times a difference of O(1)
    slopes that cancel to the 1e-13 level.  Moving ONE input component of every node by one ulp (a change of 1e-16 in the slopes)
    moves the output by > 1e-5 of its maximum and single entries by tens of percent, while the defect moves by 1e-15.  So two
    correct evaluations of the reference's formula -- another BLAS order in `f*psi_`, `^(3/2)` against a reciprocal square root --
    agree on `errors` to about 1e-3 of its maximum and no better: the GPU tests hold the kernels to that, not to 1e-9 (VERDICT
    round 5, item 6: "or document why not"); what the kernels DO follow is the reference's order of the last seven operations
    (rk.hpp rkf78_err_term)."""
    X, U, T = synth.direct_problem(30, seed=1)
    X, U, t = X[:, :, 0], U[:, :, 0], T[:, 0]
    d, e = oracle.direct_defect(X, U, t, 10, MU, DU, TU, 2000.0)
    X2 = X.copy()
    X2[0, :] = np.nextafter(X2[0, :], np.inf)
    d2, e2 = oracle.direct_defect(X2, U, t, 10, MU, DU, TU, 2000.0)
    assert e.max() < 1e-13 and e.min() > 0
    assert np.abs(d2 - d).max() < 1e-14
    assert np.abs(e2 - e).max() > 1e-5 * e.max()
    assert (np.abs(e2 - e) / e).max() > 0.1
    assert np.abs(e2 - e).max() < 1e-2 * e.max()           # ... and that is all a one-ulp change does: 1e-3 of the maximum is a real test
