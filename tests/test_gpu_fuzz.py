"""Seeded random configurations of the indirect sweep (size, batch, dimension, integrator, control-law class, smoothing,
thrust, segment lengths, time direction, kernel family, columns per lane) against the oracle.  Complements the
structured cases of test_gpu_parity.py: kernels that are miscompiled or mis-dispatched only for some template
combination show up here (see DESIGN.md "Compiler hazards")."""
import os

import numpy as np
import pytest

import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
from lowthrustopt_amd.constants import MU, DU, TU

pytestmark = pytest.mark.gpu

METHODS = [(lto.RK4, 24), (lto.RKF78_FIXED, 5), (lto.RKF78_ADAPTIVE, 0), (lto.DOP853_ADAPTIVE, 0)]


def make_case(seed):
    rng = np.random.default_rng(1000 + seed)
    ndim = int(rng.choice([12, 14]))
    method, steps = METHODS[int(rng.integers(0, 4))]
    adaptive = method in (lto.RKF78_ADAPTIVE, lto.DOP853_ADAPTIVE)
    B = int(rng.choice([1, 1, 2, 5]))
    n = int(rng.integers(2, 90))
    lo = 10.0 ** rng.uniform(-3, -1)
    hi = lo * 10.0 ** rng.uniform(0, 0.8)
    ps = [float(rng.choice([0.0, 1.0, 2.0, 1.5, 3.0])) for _ in range(B)]
    # adaptive controllers are compared at smooth settings (the sharp-switch behaviour of ode78 has its own test)
    rhos = [10.0 ** rng.uniform(-1.0 if adaptive else -4.0, 0.0) for _ in range(B)]
    thr = [float(rng.choice([0.05, 10.0])) for _ in range(B)]
    td = float(rng.choice([1.0, 1.0, -1.0]))
    lam = float(rng.choice([0.1, 0.5, 1.0]))
    kernel = int(rng.integers(0, 3))
    cols = int(rng.integers(0, 4))
    if method == lto.RK4:        # the pipeline kernels exist for fixed-step RK4 only (own generator: the other draws stay put)
        kernel = int(np.random.default_rng(7000 + seed).choice([kernel, 3, 4, 5, 7]))
        kernel = 5 if kernel in (3, 4) else kernel   # selectors 3 / 4 (four- / six-wave forms) were removed in round 3: those draws take their successor
    if method == lto.RK4 and ndim == 12:
        # round 5 (own generator: the other draws stay put): a third of the 12-dim RK4 cases run the whole-segment lanes
        # (LTO_KERNEL_LANE), a further sixth the one-step sweep with the whole STM in the segment's lane (cols_per_lane = 12)
        pick = int(np.random.default_rng(11000 + seed).integers(0, 6))
        if pick < 2:
            kernel = 9
        elif pick == 2:
            kernel, cols, steps = 1, 12, 1
    XC, T = synth.indirect_problem(n, n_batch=B, seed=seed, dt_range=(lo, hi), lam_sigma=lam)
    if ndim == 14:
        X = np.zeros((14, n, B), order="F")
        X[:6] = XC[:6]; X[6] = rng.uniform(500.0, 1500.0); X[7:13] = XC[6:]; X[13] = rng.uniform(-0.5, 0.5)
        slot = 2000.0
    else:
        X, slot = XC, 1000.0
    prm_l = [[MU, DU, TU, thr[b], slot, td, ps[b], rhos[b]] for b in range(B)]
    # lanes per segment of the defect-only sweep with the reference's setting (own generator: the other draws stay put)
    lanes = int(np.random.default_rng(9000 + seed).choice([0, 1, 2, 4])) if (ndim == 12 and method == lto.DOP853_ADAPTIVE) else 0
    # round 6 (own generator): 14-dim DOP853 batches of the always-thrust-limited laws have the quad defect kernel and the 7 + 7
    # two-lane cooperative kernel (AUTO's choice there; selector 6 names it)
    if ndim == 14 and method == lto.DOP853_ADAPTIVE and all(p in (0.0, 1.0) for p in ps):
        g14 = np.random.default_rng(13000 + seed)
        lanes = int(g14.choice([0, 1, 4]))
        kernel = int(g14.choice([kernel, 6, 0]))
    return dict(ndim=ndim, method=method, steps=steps, adaptive=adaptive, B=B, n=n, X=X, T=T, prm_l=prm_l, kernel=kernel,
                cols=cols, lanes=lanes)


@pytest.mark.parametrize("seed", range(int(os.environ.get("LTO_FUZZ_SEEDS", "96"))))
def test_indirect_random_configuration_vs_oracle(gpu_ctx, oracle, seed):
    import torch
    c = make_case(seed)
    ndim, n, B, S = c["ndim"], c["n"], c["B"], c["n"] - 1
    plan = lto.IndirectPlan(gpu_ctx, n, B, [lto.make_params(*q) for q in c["prm_l"]], lto.integrator(c["method"], steps=c["steps"]),
                            ndim=ndim)
    plan.set_kernel(c["kernel"])
    plan.set_defect_lanes(c["lanes"])
    if (ndim == 14 and c["cols"] == 3) or (ndim == 12 and c["cols"] == 2):
        # 14 columns do not split into groups of 3, and the 12-dim pairs left the library in round 6: refused, auto is kept
        with pytest.raises(lto.LtoError) as ei:
            plan.set_cols_per_lane(c["cols"])
        assert ei.value.code == -3
    else:
        plan.set_cols_per_lane(c["cols"])
    Xd = torch.from_numpy(synth.to_soa_nodes(c["X"])).cuda()
    td = torch.from_numpy(np.ascontiguousarray(c["T"].T.reshape(-1))).cuda()
    J = S * B
    Phi = torch.full((ndim * ndim, J), 3.0, dtype=torch.float64, device="cuda")
    d = torch.full((ndim, J), 3.0, dtype=torch.float64, device="cuda")
    d0 = torch.full((ndim, J), 3.0, dtype=torch.float64, device="cuda")
    plan.jacobian(Xd, n * B, td, B, Phi, J, d, J)
    plan.defect(Xd, n * B, td, B, d0, J)
    torch.cuda.synchronize()
    Pn = Phi.cpu().numpy().reshape(ndim, ndim, J).transpose(1, 0, 2)
    dn, d0n = d.cpu().numpy(), d0.cpu().numpy()
    assert np.all(np.isfinite(Pn)) and np.all(np.isfinite(dn)) and np.all(np.isfinite(d0n))
    tol_d = 1e-10
    tol_P = 1e-7 if c["adaptive"] else 1e-10
    for b in range(B):
        tol_d, tol_P = 1e-10, (1e-7 if c["adaptive"] else 1e-10)
        if c["method"] == lto.RKF78_ADAPTIVE and c["prm_l"][b][6] > 1.0:
            # ode78 (error control on the base state's infinity norm only, ode.jl:492-497) across the clamp of the p > 1 law:
            # two implementations whose step sizes differ in the last bit land on different sides of the kink and agree to the
            # method's TRUE error there, not to round-off (seeds 149, 239, 2982 of an extended run; cf. the rho = 1e-4 case of
            # test_indirect_defect_vs_oracle); the STM, which ode78 does not control at all, only to ~1e-3 (seed 1388) ... 6e-2 (seed 1373 of
            # round 6's extended run, after the error term took the reference's operation order and the last bits of every step size
            # moved).  DOP853, the setting the indirect path uses, holds the tight bars.
            tol_d, tol_P = 1e-6, 2e-1
        Xb, tb, sl = c["X"][:, :, b], c["T"][:, b], slice(b * S, (b + 1) * S)
        if ndim == 12:
            P_o, d_o, rc = oracle.indirect_jacobian(Xb, tb, c["prm_l"][b], c["method"], c["steps"])
        else:
            P_o, d_o, rc = oracle.indirect14(Xb, tb, c["prm_l"][b], c["method"], c["steps"])
        assert rc == 0
        scale = np.linalg.norm(d_o + Xb[:, 1:])
        what = "seed %d: ndim %d method %d B %d n %d kernel %d cols %d lanes %d p %g" % (seed, ndim, c["method"], B, n, c["kernel"],
                                                                                       c["cols"], c["lanes"], c["prm_l"][b][6])
        assert np.linalg.norm(dn[:, sl] - d_o) < tol_d * scale, what
        assert np.linalg.norm(d0n[:, sl] - d_o) < tol_d * scale, what
        assert np.abs(Pn[:, :, sl] - P_o).max() < tol_P * np.abs(P_o).max(), what


@pytest.mark.parametrize("seed", range(int(os.environ.get("LTO_FUZZ_SEEDS_DIRECT", "32"))))
def test_direct_random_configuration_vs_oracle(gpu_ctx, oracle, seed):
    """Direct transcription: random state size, node count, batch, steps per half segment, Isp, thrust scale (incl.
    zero-control nodes), segment lengths and Jacobian kernel family against the oracle: defect, RKF7(8) error estimate,
    Jacobian blocks (exact derivative of the discrete map by dual numbers), tf partial, mid-point states."""
    import torch
    rng = np.random.default_rng(5000 + seed)
    nstate = int(rng.choice([6, 7]))
    n = int(rng.integers(2, 70))
    B = int(rng.choice([1, 1, 3]))
    nsteps = int(rng.integers(2, 13))
    Isp = float(rng.choice([300.0, 2000.0, 3000.0]))
    kern = int(rng.integers(0, 3))
    kern = int(np.random.default_rng(9000 + seed).choice([kern, 3]))   # + the pipelined kernel (own generator: the other draws stay put)
    kern = 3 if kern == 2 else kern              # selector 2 (wave-specialised form) was removed in round 3
    X, U, T = synth.direct_problem(n, n_batch=B, seed=seed, nstate=nstate, dt_seg=10.0 ** rng.uniform(-2.5, -0.4),
                                   thrust_sigma=float(rng.choice([0.0, 0.03, 1.0])))
    if n > 3:
        U[:, int(rng.integers(0, n)), 0] = 0.0                      # a zero-control node (prop_EP_deriv.jl:35-36)
    S = n - 1
    nvar = 2 * (nstate + 3)
    plan = lto.DirectPlan(gpu_ctx, nstate, n, B, nsteps, MU, DU, TU, Isp)
    plan.set_kernel(kern)
    Xs = torch.from_numpy(synth.to_soa_nodes(X)).cuda(); Us = torch.from_numpy(synth.to_soa_nodes(U)).cuda()
    td = torch.from_numpy(np.ascontiguousarray(T.T.reshape(-1))).cuda()
    J = S * B
    Jac = torch.full((nstate * nvar, J), 3.0, dtype=torch.float64, device="cuda")
    dtf = torch.full((nstate, J), 3.0, dtype=torch.float64, device="cuda")
    d = torch.full((nstate, J), 3.0, dtype=torch.float64, device="cuda")
    e = torch.full((J,), 3.0, dtype=torch.float64, device="cuda")
    d0 = torch.full((nstate, J), 3.0, dtype=torch.float64, device="cuda")
    e0 = torch.full((J,), 3.0, dtype=torch.float64, device="cuda")
    xm = torch.full((nstate, J), 3.0, dtype=torch.float64, device="cuda")
    plan.jacobian(Xs, n * B, Us, n * B, td, B, Jac, J, dtf, d, J, e)
    plan.midpoints(Xs, n * B, Us, n * B, td, B, xm, J, d0, J, e0)
    torch.cuda.synchronize()
    Jg = Jac.cpu().numpy().reshape(nvar, nstate, J).transpose(1, 0, 2)
    dtfg, dg, eg, d0g, e0g, xmg = (v.cpu().numpy() for v in (dtf, d, e, d0, e0, xm))
    what = "seed %d: nstate %d n %d B %d nsteps %d Isp %g kernel %d" % (seed, nstate, n, B, nsteps, Isp, kern)
    for b in range(B):
        Xb, Ub, tb, sl = X[:, :, b], U[:, :, b], T[:, b], slice(b * S, (b + 1) * S)
        d_o, e_o = oracle.direct_defect(Xb, Ub, tb, nsteps, MU, DU, TU, Isp)
        Jd, dh, dd = oracle.direct_jacobian_dual(Xb, Ub, tb, nsteps, MU, DU, TU, Isp)
        tol = 1e-12 * max(1.0, np.abs(Xb).max() * 1e-2)      # the mass row (~1e3 kg) carries round-off of its own size
        assert np.abs(dg[:, sl] - d_o).max() < tol and np.abs(d0g[:, sl] - d_o).max() < tol, what
        # the estimate is a difference of nearly equal slopes: round-off level noise differs between kernels
        assert np.abs(eg[sl] - e_o).max() < 1e-3 * e_o.max() + 1e-16 and np.abs(e0g[sl] - e_o).max() < 1e-3 * e_o.max() + 1e-16, what
        assert np.abs(Jg[:, :, sl] - Jd).max() < 1e-10 * max(1.0, np.abs(Jd).max()), what
        dtf_exact = dh * (np.diff(tb) / (tb[-1] - tb[0]))[None, :]
        # the tf partial is the continuous formula (f_f - R f_b) h/(tf - t0); against the derivative of the discrete
        # map it differs by the RKF7(8) truncation error, which the error estimate bounds
        assert np.abs(dtfg[:, sl] - dtf_exact).max() < 1e-8 * max(1.0, np.abs(dtf_exact).max()) + 1e3 * e_o.max(), what
        for i in range(0, S, max(1, S // 5)):
            ref, _ = oracle.flow_prop_ep(Xb[:, i], Ub[:, i], 1.0, (tb[i + 1] - tb[i]) / 2, oracle.RKF78_FIXED, nsteps - 1,
                                         MU, DU, TU, Isp)
            assert np.abs(xmg[:, b * S + i] - ref).max() < 1e-12, what


@pytest.mark.parametrize("seed", range(24))
def test_device_newton_solve_random_sizes_vs_dense(gpu_ctx, seed):
    """Structured orthogonal cyclic reduction (square system and adjoints-only least squares, fused tail included) at
    random node counts / batch sizes against numpy's dense least squares on the scattered Jacobian
    (`-Jac_sparse \\ defect_vec`, indirect.jl:169-182), plus the stored-factorisation re-solve."""
    import torch
    rng = np.random.default_rng(9000 + seed)
    n_nodes = int(rng.integers(2, 140))
    n_batch = int(rng.choice([1, 1, 2, 4]))
    adj = bool(rng.integers(0, 2))
    XC, T = synth.indirect_problem(n_nodes, n_batch=n_batch, seed=seed, dt_range=(0.03, 0.25))
    prm = lto.make_params(MU, DU, TU, float(rng.choice([0.05, 10.0])), 1000.0, 1.0, float(rng.choice([1.0, 2.0])), 1.0)
    S, J = (n_nodes - 1) * n_batch, n_nodes * n_batch
    plan = lto.IndirectPlan(gpu_ctx, n_nodes, n_batch, prm, lto.integrator(lto.RKF78_FIXED, steps=4))
    X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T.T)).cuda()
    Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
    d = torch.zeros(12, S, dtype=torch.float64, device="cuda")
    delta = torch.full((12, J), float("nan"), dtype=torch.float64, device="cuda")
    delta2 = torch.full((12, J), float("nan"), dtype=torch.float64, device="cuda")
    plan.jacobian(X, J, t, n_batch, Phi, S, d, S)
    plan.newton_solve(Phi, S, d, S, delta, J, adjoints_only=adj)
    d2 = d * -0.3 + 0.02
    plan.newton_solve(None, 0, d2, S, delta2, J, adjoints_only=adj)
    torch.cuda.synchronize()
    Pn = Phi.cpu().numpy().reshape(12, 12, n_batch, n_nodes - 1).transpose(1, 0, 3, 2)
    what = "seed %d: n_nodes %d n_batch %d adjoints_only %s" % (seed, n_nodes, n_batch, adj)
    for dd, de in ((d, delta), (d2, delta2)):
        dn = dd.cpu().numpy().reshape(12, n_batch, n_nodes - 1).transpose(0, 2, 1)
        den = de.cpu().numpy().reshape(12, n_batch, n_nodes).transpose(0, 2, 1)
        assert np.all(np.isfinite(den)), what
        for b in range(n_batch):
            Jd = lto.indirect_scatter(np.asfortranarray(Pn[:, :, :, b]))
            keep = np.ones(Jd.shape[1], bool)
            keep[:6] = False; keep[12 * (n_nodes - 1):12 * (n_nodes - 1) + 6] = False       # fixed end states (:141-142)
            if adj:                                                                          # :169-178
                for k in range(n_nodes - 1):
                    keep[12 * k:12 * k + 6] = False
            rhs = -dn[:, :, b].reshape(-1, order="F")
            ref = np.zeros(Jd.shape[1])
            A = Jd[:, keep]                                                                  # square without the adjoints-only rows: LU
            ref[keep] = np.linalg.solve(A, rhs) if A.shape[0] == A.shape[1] else (lambda q, r: np.linalg.solve(r, q.T @ rhs))(*np.linalg.qr(A))
            got = den[:, :, b].reshape(-1, order="F")
            assert np.all(got[~keep] == 0.0), what
            assert np.abs(got - ref).max() < 1e-7 * max(1.0, np.abs(ref).max()), what


@pytest.mark.parametrize("seed", range(int(os.environ.get("LTO_FUZZ_SEEDS_LANE", "32"))))
def test_whole_segment_kernels_random_configuration_vs_oracle(gpu_ctx, oracle, seed):
    """The two kernels of round 5 whose lane owns a whole segment -- LTO_KERNEL_LANE (RK4, any number of steps) and the one-step
    sweep with cols_per_lane = 12 -- on random 12-dim configurations: node counts that leave wavefronts ragged or straddle
    trajectories, batches with mixed control-law classes, sharp and smooth switches, both time directions, one RK4 step up to
    more than the columns' rescaling period; defect and STM against the oracle's dual-number derivative of the same discrete map,
    and the defect-only sweep of the same plan."""
    import torch
    rng = np.random.default_rng(13000 + seed)
    one_step = bool(seed % 3 == 0)
    steps = 1 if one_step else int(rng.choice([1, 2, 5, 24, 64, 257]))
    B = int(rng.choice([1, 2, 5]))
    n = int(rng.integers(2, 150))
    lo = 10.0 ** rng.uniform(-3, -1.3)
    hi = lo * 10.0 ** rng.uniform(0, 0.6)
    ps = [float(rng.choice([0.0, 1.0, 2.0, 1.5, 3.0])) for _ in range(B)]
    rhos = [10.0 ** rng.uniform(-4.0, 0.0) for _ in range(B)]
    thr = [float(rng.choice([0.05, 10.0])) for _ in range(B)]
    td = float(rng.choice([1.0, 1.0, -1.0]))
    XC, T = synth.indirect_problem(n, n_batch=B, seed=500 + seed, dt_range=(lo, hi), lam_sigma=float(rng.choice([0.1, 0.5, 1.0])))
    prm_l = [[MU, DU, TU, thr[b], 1000.0, td, ps[b], rhos[b]] for b in range(B)]
    S = n - 1
    J = S * B
    plan = lto.IndirectPlan(gpu_ctx, n, B, [lto.make_params(*q) for q in prm_l], lto.integrator(lto.RK4, steps=steps))
    if one_step:
        plan.set_cols_per_lane(12)
    else:
        plan.set_kernel(plan.KERNEL_LANE)
    Xd = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    tdv = torch.from_numpy(np.ascontiguousarray(T.T.reshape(-1))).cuda()
    Phi = torch.full((144, J), 3.0, dtype=torch.float64, device="cuda")
    d = torch.full((12, J), 3.0, dtype=torch.float64, device="cuda")
    d0 = torch.full((12, J), 3.0, dtype=torch.float64, device="cuda")
    plan.jacobian(Xd, n * B, tdv, B, Phi, J, d, J)
    plan.defect(Xd, n * B, tdv, B, d0, J)
    torch.cuda.synchronize()
    assert plan.last_kernel() == ("per-lane" if one_step else "segment-lane")
    plan.close()
    Pn = Phi.cpu().numpy().reshape(12, 12, J).transpose(1, 0, 2)
    dn, d0n = d.cpu().numpy(), d0.cpu().numpy()
    for b in range(B):
        Xb, tb, sl = XC[:, :, b], T[:, b], slice(b * S, (b + 1) * S)
        P_o, d_o, rc = oracle.indirect_jacobian(Xb, tb, prm_l[b], oracle.RK4, steps)
        assert rc == 0
        scale = np.linalg.norm(d_o + Xb[:, 1:])
        what = "seed %d: steps %d B %d n %d one_step %s p %g rho %g td %g" % (seed, steps, B, n, one_step, ps[b], rhos[b], td)
        assert np.linalg.norm(dn[:, sl] - d_o) < 1e-10 * scale, what
        assert np.linalg.norm(d0n[:, sl] - d_o) < 1e-10 * scale, what
        assert np.abs(Pn[:, :, sl] - P_o).max() < 1e-10 * np.abs(P_o).max(), what
