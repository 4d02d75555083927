"""Host drivers (lowthrustopt_amd/drivers.py) mirror multiShoot_CRTBP_indirect / reduceFuel_indirect.
CPU: driver logic with an injected oracle back end.  GPU: the same problems through the HIP library."""
import numpy as np
import pytest

import lowthrustopt_amd as lto
from lowthrustopt_amd import drivers, synth
from lowthrustopt_amd.constants import MU, DU, TU


class OracleOps:
    """Test-only propagation back end (the product's is drivers.HipOps)."""

    def __init__(self, O, method=None):
        self.O = O
        self.method = O.DOP853_ADAPTIVE if method is None else method

    def _prm(self, p):
        return [p.MU, p.DU, p.TU, p.thrustLimit, p.mass, p.time_direction, p.p, p.rho]

    def defect(self, XC, t, params):
        d, _, rc = self.O.indirect_defect(XC, t, self._prm(params), self.method)
        return d

    def stm(self, XC, t, params):
        Phi, d, rc = self.O.indirect_jacobian(XC, t, self._prm(params), self.method)
        return Phi, d

    def defect_batch_sumsq(self, XC_batch, t, params):
        return np.array([np.sum(self.defect(XC_batch[:, :, b], t, params) ** 2) for b in range(XC_batch.shape[2])])


def consistent_problem(O, n_nodes=8, p=2.0, rho=1.0, thrust=10.0, seed=0, pert=1e-3):
    """Nodes sampled from ONE exact trajectory (zero defect), interior nodes then perturbed."""
    rng = np.random.default_rng(seed)
    H1 = synth.halo_orbits()[0]
    y = np.concatenate([H1[:, 20], 0.05 * rng.standard_normal(6)])
    prm = [MU, DU, TU, thrust, 1000.0, 1.0, p, rho]
    t = np.arange(n_nodes) * 0.1
    XC = np.zeros((12, n_nodes))
    XC[:, 0] = y
    for k in range(1, n_nodes):
        y, rc, _, _ = O.flow_state_costate(y, prm, 0.1, O.DOP853_ADAPTIVE)
        XC[:, k] = y
    exact = XC.copy()
    XC[:, 1:-1] += pert * rng.standard_normal((12, n_nodes - 2))
    XC[6:, 0] += pert * rng.standard_normal(6)
    XC[6:, -1] += pert * rng.standard_normal(6)
    return XC, t, exact


def test_driver_converges_with_injected_backend(oracle):
    XC, t, exact = consistent_problem(oracle)
    out, defect, status = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, 8, 1000.0, 10.0, False, False, 20, 2.0, 1.0,
                                                            ops=OracleOps(oracle), verbose=False)
    assert status == 0 and np.abs(defect).max() <= 1e-10
    assert np.array_equal(out[:6, 0], XC[:6, 0]) and np.array_equal(out[:6, -1], XC[:6, -1])   # end states pinned
    assert np.abs(out - exact).max() < 1e-6      # the fixed-endpoint problem has the sampled trajectory as solution


def test_driver_status_flags(oracle):
    XC, t, _ = consistent_problem(oracle, pert=1e-2)
    out, defect, status = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, 8, 1000.0, 10.0, False, False, 1, 2.0, 1.0,
                                                            ops=OracleOps(oracle), verbose=False)
    assert status == 1                      # maxIter reached (indirect.jl:282-285)
    XC[0, 3] = np.nan
    out, defect, status = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, 8, 1000.0, 10.0, False, False, 5, 2.0, 1.0,
                                                            ops=OracleOps(oracle), verbose=False)
    assert status == 2                      # NaN (indirect.jl:339-341)


def test_driver_rejects_the_14_row_extension(oracle):
    """The driver is the reference's 12-row loop (end-state pinning of rows 1:6, 12x12 blocks); 14 rows used to run into
    LTO_EUNSUPPORTED on the first Newton step with the mass slot silently read as Isp."""
    XC, t, _ = consistent_problem(oracle)
    X14 = np.zeros((14, XC.shape[1]))
    X14[:6] = XC[:6]; X14[6] = 1000.0; X14[7:13] = XC[6:]
    with pytest.raises(ValueError, match="12-row"):
        drivers.multiShoot_CRTBP_indirect(X14, t, MU, DU, TU, 8, 1000.0, 10.0, False, False, 5, 2.0, 1.0, ops=OracleOps(oracle), verbose=False)


def test_adjoints_only_mask_and_linesearch(oracle):
    XC, t, exact = consistent_problem(oracle, pert=0.0)
    rng = np.random.default_rng(5)
    XC[6:, :] += 1e-3 * rng.standard_normal((6, 8))          # good states, poor adjoints
    ops = OracleOps(oracle)
    prm = lto.make_params(MU, DU, TU, 10.0, 1000.0, 1.0, 2.0, 1.0)
    d0 = ops.defect(XC, t, prm)
    Phi, _ = ops.stm(XC, t, prm)
    upd = drivers.optimizeTraj_OLS(XC, t, d0, Phi, 6, 8, prm, True, ops)
    assert np.all(upd[:6, :7] == 0.0)        # state columns of nodes 1..n-1 removed (indirect.jl:169-178)
    assert np.abs(upd[6:, :]).max() > 0
    alpha = drivers.lineSearch(XC, upd, t, prm, ops)
    assert 0.1 <= alpha <= 1.0
    d1 = ops.defect(XC + alpha * upd, t, prm)
    assert np.sum(d1 ** 2) < np.sum(d0 ** 2)


@pytest.mark.gpu
def test_driver_on_gpu_matches_cpu_driver(gpu_ctx, oracle):
    XC, t, exact = consistent_problem(oracle, n_nodes=12, seed=3)
    ops = drivers.HipOps(gpu_ctx)
    out, defect, status = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, 12, 1000.0, 10.0, False, False, 20, 2.0, 1.0,
                                                            ops=ops, verbose=False)
    assert status == 0 and np.abs(defect).max() <= 1e-10
    out_c, defect_c, status_c = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, 12, 1000.0, 10.0, False, False, 20, 2.0,
                                                                  1.0, ops=OracleOps(oracle), verbose=False)
    assert status_c == 0 and np.abs(out - out_c).max() < 1e-8


@pytest.mark.gpu
def test_driver_device_newton_equals_host_linear_algebra(gpu_ctx, oracle):
    """The device-resident Newton step (N1) and the host sparse least squares produce the same iterates."""
    XC, t, exact = consistent_problem(oracle, n_nodes=20, seed=9)
    ops_dev = drivers.HipOps(gpu_ctx)
    ops_host = drivers.HipOps(gpu_ctx)
    ops_host.device_newton = False
    a = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, 20, 1000.0, 10.0, False, False, 20, 2.0, 1.0, ops=ops_dev, verbose=False)
    b = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, 20, 1000.0, 10.0, False, False, 20, 2.0, 1.0, ops=ops_host, verbose=False)
    assert a[2] == 0 and b[2] == 0
    assert np.abs(a[0] - b[0]).max() < 1e-9
    # adjoints-only iterations (host least squares on the masked Jacobian) followed by full iterations
    XC2 = XC.copy()
    XC2[:6] = exact[:6]
    c = drivers.multiShoot_CRTBP_indirect(XC2, t, MU, DU, TU, 20, 1000.0, 10.0, False, True, 10, 2.0, 1.0, ops=ops_dev, verbose=False)
    assert c[2] == 0 and np.abs(c[0][:6] - exact[:6]).max() == 0.0     # states untouched


@pytest.mark.gpu
def test_reduce_fuel_continuation_on_gpu(gpu_ctx, oracle):
    """rho continuation 1 -> 0.25 on a p = 1 problem built from an exact trajectory at rho = 1."""
    XC, t, exact = consistent_problem(oracle, n_nodes=10, p=1.0, rho=1.0, thrust=0.05, seed=4, pert=1e-5)
    ops = drivers.HipOps(gpu_ctx)
    out, defect, status = drivers.reduceFuel_indirect(XC, t, MU, DU, TU, 10, 1000.0, 0.05, 1.0, 0.25, ops=ops, verbose=False)
    assert status == 0 and np.abs(defect).max() <= 1e-10
    # the result satisfies the rho = 0.25 problem according to the oracle as well
    d_o, _, rc = oracle.indirect_defect(out, t, [MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 0.25], oracle.DOP853_ADAPTIVE)
    assert np.abs(d_o).max() < 1e-9


@pytest.mark.gpu
def test_homotopy_levels_one_launch(gpu_ctx):
    B = 16
    XC, T = synth.indirect_problem(33, n_batch=B, seed=2)
    rhos = synth.homotopy_rhos(B)
    mx, ss, d = drivers.homotopy_defect_sweep(XC, T, MU, DU, TU, 1000.0, 0.05, rhos, ops=drivers.HipOps(gpu_ctx))
    assert mx.shape == (B,) and ss.shape == (B,) and d.shape == (12, 32, B)
    for b in (0, 7, 15):
        d1, _ = lto.indirect_defectCalc(XC[:, :, b], T[:, b], lto.make_params(MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, rhos[b]),
                                        ctx=gpu_ctx)
        assert np.abs(d[:, :, b] - d1).max() < 1e-13


# ------------------------------------------------------------------------------------- meshRefine_direct (direct.jl:597-680)
class OracleDirectOps:
    """Test-only direct back end (the product's is drivers.HipDirectOps)."""

    def __init__(self, O, Isp):
        self.O, self.Isp, self.sweeps = O, Isp, 0

    def defect(self, X, U, t, nsteps):
        self.sweeps += 1
        return self.O.direct_defect(X, U, t, nsteps, MU, DU, TU, self.Isp)

    def defect_batch_sumsq(self, Xb, Ub, t, nsteps):
        return np.array([np.sum(self.defect(Xb[:, :, b], Ub[:, :, b], t, nsteps)[0] ** 2) for b in range(Xb.shape[2])])

    def midpoints(self, X, U, t):
        out = np.zeros((X.shape[0], X.shape[1] - 1), order="F")
        for i in range(X.shape[1] - 1):         # `ode7` over [t_i, t_new] = ONE RKF7(8) step (ode.jl:154, direct.jl:651-656)
            out[:, i], _ = self.O.flow_prop_ep(X[:, i], U[:, i], 1.0, (t[i + 1] - t[i]) / 2, self.O.RKF78_FIXED, 1,
                                               MU, DU, TU, self.Isp)
        return out


def mesh_problem(nstate):
    X, U, T = synth.direct_problem(12, seed=3, nstate=nstate, dt_seg=0.4)
    return X[:, :, 0], U[:, :, 0], T[:, 0]


@pytest.mark.parametrize("nstate", [6, 7])
def test_meshRefine_direct_with_injected_backend(oracle, nstate):
    X, U, t = mesh_problem(nstate)
    _, e0 = oracle.direct_defect(X, U, t, 10, MU, DU, TU, 2000.0)
    tol_min, tol_max = 1e-16, 1e-13
    assert e0.min() < tol_min and e0.max() > tol_max          # both phases are exercised
    res = {}
    for batched in (True, False):
        ops = OracleDirectOps(oracle, 2000.0)
        res[batched] = drivers.meshRefine_direct(X, U, t, nstate, 12, 10, 2000.0, MU, DU, TU, tol_min=tol_min,
                                                 tol_max=tol_max, batched=batched, ops=ops, verbose=False) + (ops.sweeps,)
    Xb, Ub, tb, nb, sweeps_b = res[True]
    Xs, Us, ts, n_s, sweeps_s = res[False]
    # all-at-once bisection == the reference's one-node-per-pass loop, in fewer sweeps
    assert nb == n_s and np.array_equal(tb, ts) and np.array_equal(Xb, Xs) and np.array_equal(Ub, Us)
    assert sweeps_b < sweeps_s
    assert Xb.shape == (nstate, nb) and Ub.shape == (3, nb) and tb.shape == (nb,)
    assert np.all(np.diff(tb) > 0) and tb[0] == t[0] and tb[-1] == t[-1]
    assert np.array_equal(Xb[:, 0], X[:, 0]) and np.array_equal(Xb[:, -1], X[:, -1])
    _, e1 = oracle.direct_defect(Xb, Ub, tb, 10, MU, DU, TU, 2000.0)
    assert e1.max() <= tol_max
    # inserted nodes lie on the forward arc of their parent segment: the defect of the left half is one step's error
    d1, _ = oracle.direct_defect(Xb, Ub, tb, 10, MU, DU, TU, 2000.0)
    new = np.flatnonzero(~np.isin(tb, t))
    assert len(new) > 0
    # max_nodes stops a refinement whose tolerance is out of reach
    Xc, Uc, tc, nc = drivers.meshRefine_direct(X, U, t, nstate, 12, 10, 2000.0, MU, DU, TU, tol_min=0.0, tol_max=1e-30,
                                               max_nodes=40, ops=OracleDirectOps(oracle, 2000.0), verbose=False)
    assert nc == 40 and np.all(np.diff(tc) > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("nstate", [6, 7])
def test_meshRefine_direct_gpu(gpu_ctx, oracle, nstate):
    """Mid-point sweep == the oracle's single RKF7(8) step; the GPU-driven refinement reproduces the oracle-driven mesh."""
    X, U, t = mesh_problem(nstate)
    for nsteps in (2, 10):
        xm, d, e = lto.direct_midpoints(X, U, t, nsteps, MU, DU, TU, 2000.0, ctx=gpu_ctx)
        d_ref, e_ref = lto.direct_defectCalc(X, U, t, nsteps, MU, DU, TU, 2000.0, ctx=gpu_ctx)
        assert np.array_equal(d, d_ref) and np.array_equal(e, e_ref)
        for i in range(11):
            ref, _ = oracle.flow_prop_ep(X[:, i], U[:, i], 1.0, (t[i + 1] - t[i]) / 2, oracle.RKF78_FIXED, nsteps - 1,
                                         MU, DU, TU, 2000.0)
            assert np.abs(xm[:, i] - ref).max() < 1e-13
    kw = dict(tol_min=1e-16, tol_max=1e-13, verbose=False)
    gb = drivers.meshRefine_direct(X, U, t, nstate, 12, 10, 2000.0, MU, DU, TU, batched=True,
                                   ops=drivers.HipDirectOps(MU, DU, TU, 2000.0, ctx=gpu_ctx), **kw)
    gs = drivers.meshRefine_direct(X, U, t, nstate, 12, 10, 2000.0, MU, DU, TU, batched=False,
                                   ops=drivers.HipDirectOps(MU, DU, TU, 2000.0, ctx=gpu_ctx), **kw)
    ob = drivers.meshRefine_direct(X, U, t, nstate, 12, 10, 2000.0, MU, DU, TU, ops=OracleDirectOps(oracle, 2000.0), **kw)
    assert gb[3] == gs[3] and all(np.array_equal(a, b) for a, b in zip(gb[:3], gs[:3]))
    assert gb[3] == ob[3] and np.allclose(gb[2], ob[2], rtol=0, atol=1e-15)
    assert np.abs(gb[0] - ob[0]).max() < 1e-12 and np.abs(gb[1] - ob[1]).max() < 1e-15


def _direct_linesearch_case(nstate):
    X, U, T = synth.direct_problem(16, seed=8, nstate=nstate)
    X, U, t = X[:, :, 0], U[:, :, 0], T[:, 0]
    rng = np.random.default_rng(4)
    dx = 1e-3 * rng.standard_normal(X.shape); du = 1e-3 * rng.standard_normal(U.shape)
    return X + dx, -dx / 0.6, U + du, -du / 0.6, t


@pytest.mark.parametrize("nstate", [6, 7])
def test_lineSearch_direct_with_injected_backend(oracle, nstate):
    X, xu, U, uu, t = _direct_linesearch_case(nstate)
    a = drivers.lineSearch_direct(X, xu, U, uu, t, nstate, 16, 10, 2000.0, MU, DU, TU, ops=OracleDirectOps(oracle, 2000.0))
    er = [np.sum(oracle.direct_defect(X + xu * al, U + uu * al, t, 10, MU, DU, TU, 2000.0)[0] ** 2)
          for al in np.linspace(0.1, 1.0, 10)]
    assert a == np.linspace(0.1, 1.0, 10)[int(np.argmin(er))]


@pytest.mark.gpu
@pytest.mark.parametrize("nstate", [6, 7])
def test_lineSearch_direct_gpu(gpu_ctx, oracle, nstate):
    X, xu, U, uu, t = _direct_linesearch_case(nstate)
    ops = drivers.HipDirectOps(MU, DU, TU, 2000.0, ctx=gpu_ctx)
    a = drivers.lineSearch_direct(X, xu, U, uu, t, nstate, 16, 10, 2000.0, MU, DU, TU, ops=ops)
    al = np.linspace(0.1, 1.0, 10)
    Xt = np.asfortranarray(X[:, :, None] + xu[:, :, None] * al); Ut = np.asfortranarray(U[:, :, None] + uu[:, :, None] * al)
    er_gpu = ops.defect_batch_sumsq(Xt, Ut, t, 10)
    er_o = OracleDirectOps(oracle, 2000.0).defect_batch_sumsq(Xt, Ut, t, 10)
    assert np.allclose(er_gpu, er_o, rtol=1e-9, atol=1e-24) and a == al[int(np.argmin(er_o))]


def test_controlLaw_cart_matches_the_propagated_dynamics(oracle):
    """controlLaw_cart (indirect.jl:389-440) in N == the control acceleration inside the RHS (thrust acceleration =
    v_dot minus the ballistic v_dot), for every law; NaN for a zero primer vector; invalid p raises."""
    rng = np.random.default_rng(2)
    H1 = synth.halo_orbits()[0]
    for p, rho, thr in ((1.0, 1.0, 0.05), (1.0, 1e-3, 0.05), (2.0, 1.0, 10.0), (2.0, 1.0, 0.05), (1.5, 1.0, 10.0), (0.0, 1.0, 0.05)):
        lam = rng.standard_normal((3, 7)) * (1.0 if p == 1.0 else 0.3)
        u = drivers.controlLaw_cart(lam, thr, p, rho, 1000.0)
        assert u.shape == (3, 7)
        for k in range(7):
            y = np.concatenate([H1[:, 10 + k], rng.standard_normal(3), lam[:, k]])
            dy = oracle.rhs_state_costate(y, [MU, DU, TU, thr, 1000.0, 1.0, p, rho])
            y0 = y.copy(); y0[9:] = 0.0                                   # lambda_v = 0 -> control zeroed (:59-64)
            dy0 = oracle.rhs_state_costate(y0, [MU, DU, TU, thr, 1000.0, 1.0, p, rho])
            acc = (dy[3:6] - dy0[3:6]) * 1000.0 * DU * 1e3 / TU ** 2       # DU/TU^2 -> N
            assert np.abs(u[:, k] - acc).max() < 1e-9 * max(1.0, np.abs(acc).max())
        assert np.allclose(drivers.controlLaw_cart(lam[:, 0], thr, p, rho, 1000.0), u[:, 0], rtol=0, atol=0)
    assert np.all(np.isnan(drivers.controlLaw_cart(np.zeros(3), 0.05, 1.0, 1.0, 1000.0)))
    with pytest.raises(ValueError):
        drivers.controlLaw_cart(np.ones(3), 0.05, 0.5, 1.0, 1000.0)


@pytest.mark.gpu
@pytest.mark.parametrize("adjoints_only", [False, True])
def test_device_newton_loop_equals_python_loop(gpu_ctx, oracle, adjoints_only):
    """lto_indirect_solve (whole multiShoot_CRTBP_indirect loop in one call, trajectory resident on the device) against
    the Python mirror of the reference loop driving the same device operators: same status, same iteration history,
    same converged trajectory; maxIter and NaN paths report the reference's status flags."""
    XC, t, exact = consistent_problem(oracle, n_nodes=20, pert=1e-3, seed=5)
    prm = lto.make_params(MU, DU, TU, 10.0, 1000.0, 1.0, 2.0, 1.0)
    XC_d, def_d, st_d, it_d, hist = lto.indirect_solve(XC, t, prm, None, adjoints_only, 25, ctx=gpu_ctx)
    XC_p, def_p, st_p = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, 20, 1000.0, 10.0, False, adjoints_only, 25, 2.0, 1.0,
                                                          ops=drivers.HipOps(gpu_ctx), verbose=False)
    assert st_d == st_p
    assert np.array_equal(XC_d[:6, 0], XC[:6, 0]) and np.array_equal(XC_d[:6, -1], XC[:6, -1])
    if st_d == 0:
        assert np.abs(def_d).max() <= 1e-10 and hist.shape == (it_d, 2) and hist[-1, 0] <= 1e-10
        assert np.abs(XC_d - XC_p).max() < 1e-8
        if not adjoints_only:
            assert np.abs(XC_d - exact).max() < 1e-6
    assert len(hist) >= 1 and np.all(hist[:3, 1] == 1.0)     # no line search before iteration 4 (indirect.jl:300)
    # the returned defect is defectCalc at the returned trajectory, bit for bit -- also when the last iteration took it from the
    # line search's sweep at the chosen trial point instead of sweeping again (:328), and the history holds its maximum
    d_again, _ = lto.indirect_defectCalc(XC_d, t, prm, None, ctx=gpu_ctx)
    assert np.array_equal(d_again, def_d) and hist[-1, 0] == np.abs(def_d).max()
    # a step that blows the defect up past 1e3 aborts the way the reference does: iterCount += 100 -> status 1 (:333-336)
    XCb, _, _ = consistent_problem(oracle, n_nodes=20, pert=3e-3, seed=5)
    _, _, stb, itb, hb = lto.indirect_solve(XCb, t, prm, None, False, 25, ctx=gpu_ctx)
    _, _, stb_p = drivers.multiShoot_CRTBP_indirect(XCb, t, MU, DU, TU, 20, 1000.0, 10.0, False, False, 25, 2.0, 1.0,
                                                    ops=drivers.HipOps(gpu_ctx), verbose=False)
    assert stb == stb_p
    if hb[-1, 0] > 1e3:
        assert stb == 1 and itb > 100
    # the driver's default path is the device loop
    XC_q, def_q, st_q = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, 20, 1000.0, 10.0, False, adjoints_only, 25, 2.0, 1.0,
                                                          verbose=False)
    assert st_q == st_d and np.array_equal(XC_q, XC_d) and np.array_equal(def_q, def_d)
    # maxIter reached -> status 1 with iterCount = maxIter + 1 (indirect.jl:282-286)
    _, _, st1, it1, h1 = lto.indirect_solve(XC, t, prm, None, adjoints_only, 1, ctx=gpu_ctx)
    assert st1 == 1 and it1 == 2 and h1.shape == (1, 2)
    # NaN in the trajectory -> status 2 (indirect.jl:339-341)
    XCn = XC.copy(); XCn[0, 0] = np.nan
    _, dn, st2, _, _ = lto.indirect_solve(XCn, t, prm, None, adjoints_only, 5, ctx=gpu_ctx)
    assert st2 == 2 and np.isnan(dn).any()
    with pytest.raises(lto.LtoError) as ei:
        lto.indirect_solve(XC, t, lto.make_params(MU, DU, TU, 10.0, 1000.0, 1.0, 0.5, 1.0), None, False, 5, ctx=gpu_ctx)
    assert ei.value.code == 2


@pytest.mark.gpu
def test_device_newton_loop_large_adaptive_problem_rebalances(gpu_ctx):
    """8 299 segments with the adaptive integrator: lto_indirect_solve orders the lanes of each sweep by the previous
    sweep's step counts (a pure scheduling change); two iterations agree with the Python loop on the host API."""
    n = 8300
    XC, T = synth.indirect_problem(n, seed=81, dt_range=(0.005, 0.04), lam_sigma=0.02)
    XC, t = XC[:, :, 0], T[:, 0]
    prm = lto.make_params(MU, DU, TU, 10.0, 1000.0, 1.0, 2.0, 1.0)
    Xd, dd, sd, itd, hist = lto.indirect_solve(XC, t, prm, None, False, 2, ctx=gpu_ctx)
    Xp, dp, sp = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, 1000.0, 10.0, False, False, 2, 2.0, 1.0,
                                                   ops=drivers.HipOps(gpu_ctx), verbose=False)
    assert sd == sp and len(hist) == min(itd, 2)
    assert np.all(np.isfinite(Xd)) and np.abs(Xd - Xp).max() < 1e-9 * max(1.0, np.abs(Xp).max())
    assert np.abs(dd - dp).max() < 1e-9 * max(1.0, np.abs(dp).max())


@pytest.mark.gpu
def test_batched_newton_loop_equals_single_loops(gpu_ctx, oracle):
    """lto_indirect_solve_batch: independent problems with their own grids, control laws and smoothing run side by side
    and finish at different iterations; each equals its own lto_indirect_solve."""
    probs = [consistent_problem(oracle, n_nodes=16, p=pp, rho=rho, thrust=thr, seed=s, pert=pert)
             for pp, rho, thr, s, pert in ((2.0, 1.0, 10.0, 1, 1e-3), (1.0, 1.0, 0.05, 2, 1e-4), (2.0, 1.0, 10.0, 3, 1e-6),
                                          (1.0, 0.5, 0.05, 4, 3e-4), (2.0, 1.0, 10.0, 5, 3e-2))]
    prms = [lto.make_params(MU, DU, TU, thr, 1000.0, 1.0, pp, rho)
            for pp, rho, thr in ((2.0, 1.0, 10.0), (1.0, 1.0, 0.05), (2.0, 1.0, 10.0), (1.0, 0.5, 0.05), (2.0, 1.0, 10.0))]
    XC = np.stack([q[0] for q in probs], axis=2)
    T = np.stack([q[1] * (1.0 + 0.01 * k) for k, q in enumerate(probs)], axis=1)       # per-trajectory grids
    Xb, Db, stb, itb, hb = lto.indirect_solve_batch(XC, T, prms, None, False, 12, ctx=gpu_ctx)
    assert Xb.shape == XC.shape and Db.shape == (12, 15, 5) and len(hb) == 5
    for k in range(5):
        X1, D1, st1, it1, h1 = lto.indirect_solve(XC[:, :, k], T[:, k], prms[k], None, False, 12, ctx=gpu_ctx)
        assert stb[k] == st1 and itb[k] == it1, (k, stb, itb, st1, it1)
        assert np.array_equal(Xb[:, :, k], X1) and np.array_equal(Db[:, :, k], D1)
        assert np.array_equal(hb[k], h1)
    assert len(set(int(v) for v in itb)) > 1                      # the trajectories really left the loop at different times
    # shared grid + shared parameters
    Xs, Ds, sts, its, _ = lto.indirect_solve_batch(XC[:, :, [0, 2]], T[:, 0], prms[0], None, False, 12, ctx=gpu_ctx)
    X1, D1, st1, it1, _ = lto.indirect_solve(XC[:, :, 2], T[:, 0], prms[0], None, False, 12, ctx=gpu_ctx)
    assert sts[1] == st1 and np.array_equal(Xs[:, :, 1], X1)


@pytest.mark.gpu
def test_homotopy_solve_concurrent_ladder(gpu_ctx):
    """The rho ladder of the demo solved concurrently: same converged trajectories as the sequential
    reduceFuel_indirect continuation at the levels both visit."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("demo", os.path.join(os.path.dirname(__file__), "..", "examples", "halo_transfer_demo.py"))
    demo = importlib.util.module_from_spec(spec); spec.loader.exec_module(demo)
    n = 30
    X, t = demo.stacked_guess(n)
    rng = np.random.default_rng(0)
    XC = np.vstack([X, 0.1 * rng.standard_normal((6, n))])
    XC[:, 1:-1] += 1e-10 * rng.standard_normal((12, n - 2))
    XC, _, f = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, 1e3, 10.0, False, True, 10, 2.0, 1.0, verbose=False)
    XC, _, f = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, 1e3, 10.0, False, False, 50, 2.0, 1.0, verbose=False)
    assert f == 0
    XC1, _, f1 = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, 1e3, 0.05, False, False, 30, 1.0, 1.0, verbose=False)
    assert f1 == 0
    rhos = 2.0 ** -np.arange(1, 8)                                   # 0.5 ... 1/128: the halving ladder of :179
    Xl, Dl, st, waves = drivers.homotopy_solve(XC1, t, MU, DU, TU, 1e3, 0.05, rhos, ctx=gpu_ctx, verbose=False)
    assert np.all(st == 0) and np.abs(Dl).max() <= 1e-10 and waves <= 7
    Xseq, dseq, fseq = drivers.reduceFuel_indirect(XC1, t, MU, DU, TU, n, 1e3, 0.05, 1.0, float(rhos[-1]), verbose=False)
    assert fseq == 0
    assert np.abs(Xl[:, :, -1] - Xseq).max() < 1e-6 * max(1.0, np.abs(Xseq).max())
