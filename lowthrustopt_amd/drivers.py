"""Host-side mirror of the reference's indirect shooting drivers, calling the HIP hot path.

  multiShoot_CRTBP_indirect   src/multiShoot_CRTBP_indirect.jl:58-61, :254-345   (Newton loop, status flags)
  optimizeTraj_OLS            :149-218   (least-squares step, adjoints-only mask, second-order correction)
  lineSearch                  :221-246   (20 alphas; here ONE batched device launch instead of 20 sweeps -- SURVEY N2)
  reduceFuel_indirect         src/HelperFunctions.jl:105-193   (rho continuation -- SURVEY N3)
  meshRefine_direct           src/multiShoot_CRTBP_direct.jl:597-680   (errors-driven mesh refinement -- SURVEY N4)
  lineSearch_direct           src/multiShoot_CRTBP_direct.jl:405-430   (10 alphas in ONE batched launch -- SURVEY N2)
  homotopy_solve              concurrent form of the rho continuation (HelperFunctions.jl:105-193): every level of the
                              ladder is a trajectory of ONE batched device Newton loop (lto_indirect_solve_batch)
  controlLaw_cart             src/multiShoot_CRTBP_indirect.jl:389-440 (costates -> thrust vectors in N: the u_all format
                                                                        of the direct transcription; host post-processing)

Same signatures, return tuples and status flags as the Julia functions (the reference is Julia; this mirror exists
because no `julia` binary is available to run julia/LowThrustOptHIP.jl -- see INTEGRATION.md).  The propagation
(defects, STM blocks) always runs on the GPU through the C ABI; only the small sparse least-squares solve is on the
host, as in the reference (`-Jac_sparse \\ defect_vec`, :182).  The direct driver's JuMP/Ipopt QP
(src/multiShoot_CRTBP_direct.jl:248-403) is out of scope; its hot-path closures are in hotpath.py.

`ops` lets the CPU unit tests inject a different propagation back end; the product default is the HIP library.
"""
import numpy as np

from . import hotpath


class HipOps:
    """Hot-path operators backed by liblto_hip.so (the product path)."""

    def __init__(self, ctx=None, integ=None):
        self.ctx = ctx or hotpath.default_context()
        self.integ = integ or hotpath.integrator()     # adaptive order 8 @ 1e-13 = the reference's Vern8 setting

    def defect(self, XC, t, params):
        d, _ = hotpath.indirect_defectCalc(XC, t, params, self.integ, ctx=self.ctx)
        return d

    def stm(self, XC, t, params):
        return hotpath.indirect_stm(XC, t, params, self.integ, ctx=self.ctx)

    def newton_step(self, XC, t, params, flag_adjointsOnly=False):
        """jacobianCalc + least-squares step (incl. the adjoints-only mask) + second-order correction in one
        device-resident call (indirect.jl:290-296): Phi never leaves HBM."""
        upd, _ = hotpath.indirect_newton_step(XC, t, params, self.integ, ctx=self.ctx, flag_adjointsOnly=flag_adjointsOnly)
        return upd

    def defect_batch_sumsq(self, XC_batch, t, params):
        """sum(defect.^2) per trial trajectory: one batched launch (+ on-device reduction when torch is present)."""
        d, _ = hotpath.indirect_defectCalc(XC_batch, t, params, self.integ, ctx=self.ctx)
        return np.sum(d * d, axis=(0, 1))


def _solve_ls(J, rhs):
    """x = -J \\ rhs (least squares).  The systems here are square or over-determined once the fixed-end-state
    columns are dropped, so the solution is unique and independent of the factorisation used."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    if sp.issparse(J):
        keep = np.flatnonzero(np.diff(J.tocsc().indptr) > 0)
        Jk = J.tocsc()[:, keep]
        x = np.zeros(J.shape[1])
        if Jk.shape[0] == Jk.shape[1]:
            try:
                x[keep] = spl.spsolve(Jk, -rhs)
                if np.all(np.isfinite(x)):
                    return x
            except Exception:
                pass
        sol = spl.lsmr(Jk, -rhs, atol=1e-15, btol=1e-15, conlim=1e14, maxiter=20 * Jk.shape[1])[0]
        x[keep] = sol
        return x
    return np.linalg.lstsq(J, -rhs, rcond=None)[0]


def optimizeTraj_OLS(XC_all, t_TU, defect, Phi, nstate, n_nodes, params, flag_adjointsOnly, ops):
    """Least-squares Newton step with second-order correction (indirect.jl:149-218)."""
    nd = 2 * nstate
    defect_vec = defect.reshape(-1, order="F")
    J = hotpath.indirect_scatter(Phi, sparse=True)
    temp = np.ones(J.shape[1], dtype=bool)
    if flag_adjointsOnly:                                   # :169-178: drop the state columns of nodes 1..n-1
        for ind in range(n_nodes - 1):
            temp[ind * nd: ind * nd + nstate] = False
    Jm = J[:, np.flatnonzero(temp)]
    upd = np.zeros(J.shape[1])
    upd[temp] = _solve_ls(Jm, defect_vec)                   # :182
    xc_update = upd.reshape(nd, n_nodes, order="F")
    if np.abs(xc_update).max() < 1e-1:                      # :190  SOC: same Jacobian, defect at the trial point
        d_soc = ops.defect(XC_all + xc_update, t_TU, params)
        upd2 = np.zeros(J.shape[1])
        upd2[temp] = _solve_ls(Jm, d_soc.reshape(-1, order="F"))
        xc_update = xc_update + upd2.reshape(nd, n_nodes, order="F")
    return xc_update


def lineSearch(XC_all, xc_update, t_TU, params, ops):
    """alpha in LinRange(0.1, 1, 20) minimising sum(defect.^2) (indirect.jl:221-246), all 20 trial trajectories in
    one batched sweep."""
    alpha_all = np.linspace(0.1, 1.0, 20)
    trial = XC_all[:, :, None] + xc_update[:, :, None] * alpha_all[None, None, :]
    er = ops.defect_batch_sumsq(np.asfortranarray(trial), t_TU, params)
    return float(alpha_all[int(np.argmin(er))])             # first minimiser, as `alpha[er .== minimum(er)][1]`


def multiShoot_CRTBP_indirect(XC_all, t_TU, MU, DU, TU, n_nodes, mass0, thrustLimit, plot_yn, flag_adjointsOnly,
                              maxIter, p, rho, ops=None, verbose=True):
    """Indirect multiple shooting with fixed end states (indirect.jl:58-61, :254-345).
    Returns (XC_all, defect, status_flag): 0 converged, 1 maxIter reached, 2 NaN.
    The driver is the reference's: 12 rows (state + costate, constant mass `mass0`).  The 14-dim extension (mass and mass
    costate as states, Isp in the parameter tuple's mass slot) exists for the sweeps only (indirect_defectCalc /
    indirect_stm with ndim = 14) -- the reference's loop pins XC_all[1:6] and solves 12x12 blocks (indirect.jl:324-325)."""
    if np.asarray(XC_all).shape[0] != 12:
        raise ValueError("multiShoot_CRTBP_indirect drives the reference's 12-row state+costate system; got %d rows "
                         "(the 14-dim extension is available through indirect_defectCalc / indirect_stm only)" % np.asarray(XC_all).shape[0])
    if ops is None:
        # product default: the whole loop below is one library call with the trajectory resident on the device
        # (lto_indirect_solve); the Python loop remains for injected back ends
        params = hotpath.make_params(MU, DU, TU, thrustLimit, mass0, 1.0, p, rho)
        XC_out, defect, status_flag, iterCount, hist = hotpath.indirect_solve(XC_all, t_TU, params, None, flag_adjointsOnly, maxIter)
        if verbose:
            for k, (er, alpha) in enumerate(hist):
                print("Iter %d. Max defect = %.2e. alpha = %.3f." % (k + 1, er, alpha))
                if not (er <= 1e3):
                    print("Not likely to converge. Aborting.")
            if status_flag == 1:
                print("Reached max iteration count at %d iterations" % iterCount)
        return XC_out, defect, status_flag
    ops = ops or HipOps()
    XC_all = np.array(XC_all, dtype=np.float64, order="F")
    t_TU = np.array(t_TU, dtype=np.float64)
    nstate = XC_all.shape[0] // 2                            # :255
    params = hotpath.make_params(MU, DU, TU, thrustLimit, mass0, 1.0, p, rho)   # :258-260
    status_flag = 0
    state_0 = XC_all[:nstate, 0].copy()
    state_f = XC_all[:nstate, -1].copy()
    defect = ops.defect(XC_all, t_TU, params)                # :274
    iterCount = 0
    er = 1.0
    while er > 1e-10:                                        # :280
        iterCount += 1
        if iterCount > maxIter:
            if verbose:
                print("Reached max iteration count at %d iterations" % iterCount)
            status_flag = 1
            break
        if hasattr(ops, "newton_step") and getattr(ops, "device_newton", True):
            xc_update = ops.newton_step(XC_all, t_TU, params, flag_adjointsOnly)   # :290-296 on the device
        else:
            Phi, _ = ops.stm(XC_all, t_TU, params)               # jacobianCalc, :290
            xc_update = optimizeTraj_OLS(XC_all, t_TU, defect, Phi, nstate, n_nodes, params, flag_adjointsOnly, ops)
        alpha = 1.0
        if iterCount > 3:                                    # :300
            alpha = lineSearch(XC_all, xc_update, t_TU, params, ops)
        XC_all = XC_all + xc_update * alpha
        XC_all[:nstate, 0] = state_0                         # :324-325
        XC_all[:nstate, -1] = state_f
        defect = ops.defect(XC_all, t_TU, params)            # :328
        er = float(np.max(np.abs(defect))) if np.all(np.isfinite(defect)) else float("nan")
        if verbose:
            print("Iter %d. Max defect = %.2e. alpha = %.3f." % (iterCount, er, alpha))
        if not (er <= 1e3):                                  # :333-336 (also leaves the loop on NaN)
            if verbose:
                print("Not likely to converge. Aborting.")
            iterCount += 100
            if er != er:
                break
    if np.isnan(XC_all[0, 0]) or not np.all(np.isfinite(defect)):
        status_flag = 2                                      # :339-341
    return XC_all, defect, status_flag


def reduceFuel_indirect(XC_all, t_TU, MU, DU, TU, n_nodes, mass, thrustLimit, rho_current, rho_target, ops=None,
                        verbose=True, rng=None):
    """rho continuation (HelperFunctions.jl:105-193): halve rho on success, back off on failure.
    status_flag 3 = continuation exhausted (:161)."""
    rng = rng or np.random.default_rng(0)
    if rho_target > rho_current:
        rho_target = rho_current
    p = 1.0
    rho_temp = rho_current
    maxIter = 10

    def run(X, rho):
        return multiShoot_CRTBP_indirect(X, t_TU, MU, DU, TU, n_nodes, mass, thrustLimit, False, False, maxIter, p, rho,
                                         ops=ops, verbose=verbose)

    XC_new, defect, status = run(XC_all, rho_temp)
    if status == 0 and rho_current == rho_target:
        return XC_new, defect, status
    while status != 0 and rho_temp < 1:                      # :133-143
        rho_temp = min(rho_temp * 5, 1.0)
        XC_new, defect, status = run(XC_all, rho_temp)
    if rho_temp == 1 and status != 0:
        return XC_new, defect, status
    if status == 0:
        XC_all = XC_new.copy()
    count = 0
    while rho_temp > rho_target or status != 0:              # :158-188
        count += 1
        if count > 100:
            return XC_new, defect, 3
        if status == 0:
            XC_all = XC_new.copy()
            rho_temp = max(rho_temp / 2, rho_target)
        else:
            rho_temp *= 3 * (1 + rng.random())
        XC_new, defect, status = run(XC_all, rho_temp)
    return XC_new, defect, status


def homotopy_defect_sweep(XC_levels, t_TU, MU, DU, TU, mass, thrustLimit, rhos, ops=None):
    """All continuation levels evaluated concurrently (BASELINE configs[3]): level l has its own node array
    XC_levels[:, :, l] and smoothing rho_l; ONE launch returns max|defect| and sum(defect^2) per level."""
    ops = ops or HipOps()
    prms = [hotpath.make_params(MU, DU, TU, thrustLimit, mass, 1.0, 1.0, r) for r in rhos]
    d, _ = hotpath.indirect_defectCalc(np.asfortranarray(XC_levels), t_TU, prms, ops.integ, ctx=ops.ctx)
    return np.max(np.abs(d), axis=(0, 1)), np.sum(d * d, axis=(0, 1)), d


class HipDirectOps:
    """Direct-method hot-path operators backed by liblto_hip.so."""

    def __init__(self, MU, DU, TU, Isp, ctx=None):
        self.ctx = ctx or hotpath.default_context()
        self.MU, self.DU, self.TU, self.Isp = MU, DU, TU, Isp

    def defect(self, X, U, t, nsteps):
        return hotpath.direct_defectCalc(X, U, t, nsteps, self.MU, self.DU, self.TU, self.Isp, ctx=self.ctx)

    def midpoints(self, X, U, t):
        """x(t_i + h_i/2) from node i with u_i, one RKF7(8) step (`ode7` over [t_i, t_new], direct.jl:651-656)."""
        return hotpath.direct_midpoints(X, U, t, 2, self.MU, self.DU, self.TU, self.Isp, ctx=self.ctx)[0]

    def defect_batch_sumsq(self, X_batch, U_batch, t, nsteps):
        """sum(defect.^2) of every trial trajectory X_batch[:, :, b], U_batch[:, :, b]: one batched launch."""
        d, _ = hotpath.direct_defectCalc(X_batch, U_batch, t, nsteps, self.MU, self.DU, self.TU, self.Isp, ctx=self.ctx)
        return np.sum(d * d, axis=(0, 1))


def lineSearch_direct(X_all, x_update, u_all, u_update, t_TU, nstate, n_nodes, nsteps, Isp, MU, DU, TU, ops=None):
    """alpha in LinRange(0.1, 1, 10) minimising sum(defect.^2) of the direct transcription (direct.jl:405-430); the ten
    trial trajectories are one batched sweep instead of ten."""
    ops = ops or HipDirectOps(MU, DU, TU, Isp)
    alpha_all = np.linspace(0.1, 1.0, 10)
    X_all = np.asarray(X_all, dtype=np.float64); x_update = np.asarray(x_update, dtype=np.float64)
    u_all = np.asarray(u_all, dtype=np.float64); u_update = np.asarray(u_update, dtype=np.float64)
    Xt = X_all[:, :, None] + x_update[:, :, None] * alpha_all[None, None, :]
    Ut = u_all[:, :, None] + u_update[:, :, None] * alpha_all[None, None, :]
    er = ops.defect_batch_sumsq(np.asfortranarray(Xt), np.asfortranarray(Ut), t_TU, nsteps)
    return float(alpha_all[int(np.argmin(er))])             # first minimiser, as `alpha[1]` (:428-429)


def meshRefine_direct(X_all, u_all, t_TU, nstate, n_nodes, nsteps, Isp, MU, DU, TU, tol_min=1e-20, tol_max=1e-18,
                      max_nodes=1 << 20, batched=True, ops=None, verbose=True):
    """Errors-driven mesh refinement of the direct transcription (direct.jl:597-680): nodes are removed while the
    smallest RKF7(8) error estimate of a segment is below tol_min, then segments are bisected while the largest is above
    tol_max (new state = forward propagation to the segment's middle, new control = mean of its two controls).
    Returns (X_all, u_all, t_TU, n_nodes).

    Re-specified where the reference cannot run as written: `find(errors .== minimum(errors))[1]` (removed from Julia
    1.x) is the first arg-min / arg-max; MU, DU, TU are arguments instead of globals; the removal phase stops at two
    nodes and the addition phase at max_nodes (the reference loops forever if the estimate never reaches tol_max).

    batched=True bisects EVERY segment above tol_max in one pass: one error sweep + one mid-point sweep on the GPU per
    pass instead of one full sweep per inserted node.  The result is identical to the reference's one-node-per-pass
    loop, because a segment's error estimate depends only on its own two nodes, controls and times (direct.jl:77-105),
    so splitting one segment never changes the decision for another; batched=False runs the literal loop."""
    ops = ops or HipDirectOps(MU, DU, TU, Isp)
    X = np.array(X_all, dtype=np.float64, order="F")
    U = np.array(u_all, dtype=np.float64, order="F")
    t = np.array(t_TU, dtype=np.float64)
    n_nodes = int(n_nodes)
    n_start = n_nodes
    if verbose:
        print("Starting with %d nodes." % n_nodes)
    _, errors = ops.defect(X, U, t, nsteps)
    # ---- remove nodes that only make things messy (:611-628); inherently sequential: a removal merges two segments
    while n_nodes > 2 and np.min(errors) < tol_min:
        k = int(np.argmin(errors))
        if k == 0:
            k = 1                                            # never remove the first node (:616-618)
        X = np.delete(X, k, axis=1)
        U = np.delete(U, k, axis=1)
        t = np.delete(t, k)
        n_nodes -= 1
        _, errors = ops.defect(X, U, t, nsteps)
    if verbose and n_nodes != n_start:
        print("Removed nodes, now n_nodes = %d" % n_nodes)
    n_mid = n_nodes
    # ---- add nodes where the estimate is too large (:635-668)
    while np.max(errors) > tol_max and n_nodes < max_nodes:
        if batched:
            split = np.flatnonzero(errors > tol_max)[: max_nodes - n_nodes]
        else:
            split = np.array([int(np.argmax(errors))])
        x_mid = ops.midpoints(X, U, t)                       # every segment's mid-point state in one sweep
        t_new = t[split] + (t[split + 1] - t[split]) / 2     # :644
        u_new = (U[:, split] + U[:, split + 1]) / 2          # :659
        X = np.insert(X, split + 1, x_mid[:, split], axis=1)
        U = np.insert(U, split + 1, u_new, axis=1)
        t = np.insert(t, split + 1, t_new)
        n_nodes += len(split)
        _, errors = ops.defect(X, U, t, nsteps)
    if verbose:
        if n_nodes != n_mid:
            print("Added nodes, now n_nodes = %d" % n_nodes)
        if n_nodes == n_start:
            print("Did not need to add or remove nodes.")
        else:
            print("Refined the mesh. Now have %d nodes." % n_nodes)
    return np.asfortranarray(X), np.asfortranarray(U), t, n_nodes


def controlLaw_cart(lambda_v, thrustLimit, p, rho, mass, DU=None, TU=None):
    """Thrust vector(s) in N from the velocity costate(s) (indirect.jl:389-440): lambda_v [3] or [3 x n] ->
    control of the same shape.  Same law as the propagated dynamics (stateCostate_deriv.jl:33-64); like the reference,
    a zero primer vector yields NaN (0/0 in `lambda_v ./ norm(lambda_v)`, :439) and an invalid p raises."""
    from .constants import DU as _DU, TU as _TU
    DU = _DU if DU is None else DU
    TU = _TU if TU is None else TU
    lam = np.asarray(lambda_v, dtype=np.float64)
    n = np.sqrt(np.sum(lam * lam, axis=0))
    accelLimit = thrustLimit / mass / 1e3 * TU ** 2 / DU            # N -> DU/TU^2 (:412)
    if p == 0:
        umag = np.full_like(n, accelLimit)
    elif p == 1:
        umag = 0.5 * (1.0 + np.tanh((n - 1.0) / (2.0 * rho))) * accelLimit
    elif p > 1:
        umag = np.minimum((n / p) ** (1.0 / (p - 1.0)), accelLimit)
    else:
        raise ValueError("Invalid value of p!")
    umag = np.where(np.isnan(umag), 0.0, umag)                      # :431-433
    with np.errstate(invalid="ignore", divide="ignore"):
        return -umag * lam / n * mass * DU * 1e3 / TU ** 2          # DU/TU^2 -> N (:439)


def homotopy_solve(XC_all, t_TU, MU, DU, TU, mass, thrustLimit, rhos, p=1.0, maxIter=10, max_waves=12, ctx=None, verbose=True):
    """Solve the whole smoothing ladder rho_0 > rho_1 > ... concurrently.  reduceFuel_indirect walks the
    ladder one level at a time, halving rho after each success (HelperFunctions.jl:158-188); here every unsolved level is
    one trajectory of a batched device Newton loop.  Wave 1 starts all levels from XC_all (a solution at or above
    rho_0); each later wave restarts the levels that failed from the converged solution of the nearest level with a
    larger rho.  Stops when every level has converged, a wave makes no progress, or after max_waves.

    Returns (XC_levels [12 x n x L], defect [12 x (n-1) x L], status [L] (0 converged, else the last status_flag of the
    level, 3 = never converged: HelperFunctions.jl:161), waves)."""
    rhos = np.asarray(rhos, dtype=np.float64)
    order = np.argsort(-rhos)                                 # descending: neighbours in the list are neighbours in rho
    L = len(rhos)
    XC0 = np.array(XC_all, dtype=np.float64, order="F")
    n = XC0.shape[1]
    X = np.repeat(XC0[:, :, None], L, axis=2)
    D = np.full((12, n - 1, L), np.nan)
    status = np.full(L, 3, dtype=np.int32)
    solved = np.zeros(L, dtype=bool)
    waves = 0
    while not solved.all() and waves < max_waves:
        waves += 1
        todo = [k for k in order if not solved[k]]
        guess = np.empty((12, n, len(todo)), order="F")
        for j, k in enumerate(todo):                          # nearest converged level with a larger rho, else the input
            better = [q for q in order if solved[q] and rhos[q] > rhos[k]]
            guess[:, :, j] = X[:, :, better[-1]] if better else XC0
        prms = [hotpath.make_params(MU, DU, TU, thrustLimit, mass, 1.0, p, float(rhos[k])) for k in todo]
        Xo, Do, st, it, _ = hotpath.indirect_solve_batch(guess, t_TU, prms, None, False, maxIter, ctx=ctx)
        progress = 0
        for j, k in enumerate(todo):
            if st[j] == 0:
                X[:, :, k], D[:, :, k], status[k], solved[k] = Xo[:, :, j], Do[:, :, j], 0, True
                progress += 1
            elif status[k] == 3:
                D[:, :, k] = Do[:, :, j]
        if verbose:
            print("wave %d: %d of %d levels converged (%d remaining)" % (waves, progress, len(todo), int((~solved).sum())))
        if progress == 0:
            break
    return np.asfortranarray(X), np.asfortranarray(D), status, waves
