"""14-dim extension (CRTBP state + mass + costates + mass costate, BASELINE configs[1]).  The reference has no such
RHS, so there is no reference parity to claim: the tests pin (i) reduction to the reference's 12-dim system when
the mass flow is switched off, (ii) HIP == oracle on the same discrete map (defect 1e-10, STM vs dual numbers
1e-10), (iii) the model's invariants (mass decreases at thrust/(Isp g0); volume-preserving flow)."""
import numpy as np
import pytest

import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
from lowthrustopt_amd.constants import MU, DU, TU

IDX12 = [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12]


def problem14(n_nodes, seed=1, lam=0.1, lam_m=0.3, mass=1000.0):
    XC, T = synth.indirect_problem(n_nodes, seed=seed, lam_sigma=lam)
    XC, t = XC[:, :, 0], T[:, 0]
    X14 = np.zeros((14, n_nodes), order="F")
    X14[:6] = XC[:6]
    X14[6] = mass - 0.05 * np.arange(n_nodes)
    X14[7:13] = XC[6:]
    X14[13] = lam_m
    return X14, XC, t


def test_oracle14_reduces_to_reference_system(oracle):
    """Isp -> infinity and constant mass: components (r, v, lambda_r, lambda_v) and their 12x12 STM block equal the
    12-dim oracle (which follows src/CRTBP_stateCostate_deriv.jl line by line)."""
    X14, XC, t = problem14(6)
    X14[6] = 1000.0
    for p, rho, thr in ((1.0, 1.0, 0.05), (2.0, 1.0, 10.0), (2.0, 1.0, 0.05), (0.0, 1.0, 0.05), (1.5, 1.0, 10.0)):
        P14, d14, rc = oracle.indirect14(X14, t, [MU, DU, TU, thr, 1e30, 1.0, p, rho], oracle.RK4, 32)
        P12, d12, rc2 = oracle.indirect_jacobian(XC, t, [MU, DU, TU, thr, 1000.0, 1.0, p, rho], oracle.RK4, 32)
        assert rc == 0 and rc2 == 0
        assert np.abs(d14[IDX12] - d12).max() < 1e-14
        assert np.abs(P14[np.ix_(IDX12, IDX12)] - P12).max() < 1e-12 * np.abs(P12).max()
        assert np.abs(d14[6]).max() < 1e-20          # no mass flow


def test_oracle14_mass_flow_and_volume(oracle):
    X14, XC, t = problem14(5)
    X14[6] = 1000.0
    prm = [MU, DU, TU, 0.05, 2000.0, 1.0, 0.0, 1.0]         # p = 0: thrust always on at the limit
    P, d, rc = oracle.indirect14(X14, t, prm, oracle.DOP853_ADAPTIVE)
    dm_expected = -0.05 / (2000.0 * 9.81) * TU * np.diff(t)  # -T/(Isp g0) per second
    assert np.abs(d[6] - dm_expected).max() < 1e-12
    for i in range(4):
        assert abs(np.linalg.det(P[:, :, i]) - 1.0) < 1e-9   # trace F = 0 for thrust-limited laws


CASES = {"p1_rho1": (1.0, 1.0, 0.05, 0.1), "p1_rho1e-2": (1.0, 1e-2, 0.05, 1.0), "p2_unclamped": (2.0, 1.0, 10.0, 0.1),
         "p2_clamped": (2.0, 1.0, 0.05, 1.0), "p1.5": (1.5, 1.0, 10.0, 0.3), "p0": (0.0, 1.0, 0.05, 0.1)}


@pytest.mark.gpu
@pytest.mark.parametrize("pcase", list(CASES))
@pytest.mark.parametrize("method,steps", [(lto.RK4, 32), (lto.RKF78_FIXED, 6), (lto.DOP853_ADAPTIVE, 0)])
def test_gpu14_vs_oracle(gpu_ctx, oracle, pcase, method, steps):
    p, rho, thr, lam = CASES[pcase]
    X14, XC, t = problem14(30, seed=2, lam=lam)
    prm = lto.make_params(MU, DU, TU, thr, 2000.0, 1.0, p, rho)          # mass slot = Isp for ndim = 14
    integ = lto.integrator(method, steps=steps)
    Phi, d = lto.indirect_stm(X14, t, prm, integ, ctx=gpu_ctx)
    d2, _ = lto.indirect_defectCalc(X14, t, prm, integ, ctx=gpu_ctx)
    Phi_o, d_o, rc = oracle.indirect14(X14, t, [MU, DU, TU, thr, 2000.0, 1.0, p, rho], method, steps)
    assert rc == 0 and Phi.shape == (14, 14, 29)
    xn = np.linalg.norm(d_o + X14[:, 1:])
    assert np.linalg.norm(d - d_o) / xn < 1e-10 and np.linalg.norm(d2 - d_o) / xn < 1e-10
    tol = 1e-7 if method == lto.DOP853_ADAPTIVE else 1e-10     # adaptive: different step sequences, same flow
    assert np.abs(Phi - Phi_o).max() < tol * np.abs(Phi_o).max()


@pytest.mark.gpu
def test_gpu14_adaptive_and_reduction(gpu_ctx, oracle):
    X14, XC, t = problem14(20, seed=3)
    prm14 = lto.make_params(MU, DU, TU, 0.05, 2000.0, 1.0, 1.0, 1.0)
    Phi, d = lto.indirect_stm(X14, t, prm14, lto.integrator(lto.DOP853_ADAPTIVE), ctx=gpu_ctx)
    Phi_o, d_o, rc = oracle.indirect14(X14, t, [MU, DU, TU, 0.05, 2000.0, 1.0, 1.0, 1.0], oracle.DOP853_ADAPTIVE)
    assert np.linalg.norm(d - d_o) / np.linalg.norm(d_o + X14[:, 1:]) < 1e-10
    assert np.abs(Phi - Phi_o).max() < 1e-8 * np.abs(Phi_o).max()
    # Isp -> infinity: the GPU 14-dim result contains the GPU 12-dim result
    X14[6] = 1000.0
    integ = lto.integrator(lto.RK4, steps=32)
    Phi14, d14 = lto.indirect_stm(X14, t, lto.make_params(MU, DU, TU, 0.05, 1e30, 1.0, 1.0, 1.0), integ, ctx=gpu_ctx)
    Phi12, d12 = lto.indirect_stm(XC, t, lto.make_params(MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1.0), integ, ctx=gpu_ctx)
    assert np.abs(d14[IDX12] - d12).max() < 1e-13
    assert np.abs(Phi14[np.ix_(IDX12, IDX12)] - Phi12).max() < 1e-11 * np.abs(Phi12).max()


@pytest.mark.gpu
def test_gpu14_reference_integrator_kernels_for_the_thrust_limited_laws(gpu_ctx, oracle):
    """Round 6: the reference's integrator setting (adaptive order 8 @ 1e-13) on the 14-dim system runs the two-lanes-per-state
    cooperative kernel (STM sweep: states split 7 + 7, thirteen columns, the lambda_m column the unit vector) and the four-lanes-per-
    segment defect kernel for batches of the always-thrust-limited laws (p = 0, p = 1).  A ragged batch with both classes in one
    workgroup: STM and defect against the oracle's dual-number flow (same error norm: 1e-9 of max |Phi|), the quad defect sweep
    against the one-lane kernel, the lambda_m column exactly e_13, det Phi = 1; p = 2 keeps the one-piece cooperative kernel."""
    import torch
    n, B = 41, 3
    S = (n - 1) * B
    XC, T = synth.indirect_problem(n, n_batch=B, seed=4, dt_range=(0.05, 0.35))
    X = np.zeros((14, n, B), order="F")
    X[:6] = XC[:6]; X[6] = 1000.0 - 0.03 * np.arange(n)[:, None]; X[7:13] = XC[6:]; X[13] = 0.25
    ps, rhos = [1.0, 0.0, 1.0], [1.0, 1.0, 1e-2]
    prm_l = [[MU, DU, TU, 0.05, 2000.0, 1.0, ps[b], rhos[b]] for b in range(B)]
    plan = lto.IndirectPlan(gpu_ctx, n, B, [lto.make_params(*q) for q in prm_l], lto.integrator(), ndim=14)
    Xd = torch.from_numpy(synth.to_soa_nodes(X)).cuda()
    td = torch.from_numpy(np.ascontiguousarray(T.T.reshape(-1))).cuda()
    f64 = dict(dtype=torch.float64, device="cuda")
    Phi = torch.zeros(196, S, **f64); d = torch.zeros(14, S, **f64); d4 = torch.zeros(14, S, **f64); d1 = torch.zeros(14, S, **f64)
    plan.jacobian(Xd, n * B, td, B, Phi, S, d, S)
    assert plan.last_kernel() == "cooperative2"
    plan.defect(Xd, n * B, td, B, d4, S)                 # AUTO: four lanes per segment
    plan.set_defect_lanes(1); plan.defect(Xd, n * B, td, B, d1, S)
    plan.set_defect_lanes(4)
    with pytest.raises(lto.LtoError):
        plan.set_defect_lanes(2)                          # the pair form is 12-dim only
    torch.cuda.synchronize()
    acc, rej = plan.step_counts()
    assert acc.min() >= 1 and (acc + rej).max() < 200
    plan.close()
    P = Phi.cpu().numpy().reshape(14, 14, S).transpose(1, 0, 2)
    dn, d4n, d1n = d.cpu().numpy(), d4.cpu().numpy(), d1.cpu().numpy()
    assert np.abs(d4n - d1n).max() < 1e-12 and np.abs(d4n - dn).max() < 1e-11
    e13 = np.zeros(14); e13[13] = 1.0
    assert np.array_equal(P[:, 13, :], np.repeat(e13[:, None], S, axis=1))
    for b in range(B):
        sl = slice(b * (n - 1), (b + 1) * (n - 1))
        P_o, d_o, rc = oracle.indirect14(X[:, :, b], T[:, b], prm_l[b], oracle.DOP853_ADAPTIVE)
        assert rc == 0
        xn = np.linalg.norm(d_o + X[:, 1:, b])
        assert np.linalg.norm(dn[:, sl] - d_o) / xn < 1e-10 and np.linalg.norm(d4n[:, sl] - d_o) / xn < 1e-10
        assert np.abs(P[:, :, sl] - P_o).max() < 1e-9 * np.abs(P_o).max()
    for s in (0, 17, S - 1):
        assert abs(np.linalg.det(P[:, :, s]) - 1.0) < 1e-8
    p2 = lto.IndirectPlan(gpu_ctx, 8, 1, lto.make_params(MU, DU, TU, 10.0, 2000.0, 1.0, 2.0, 1.0), lto.integrator(), ndim=14)
    with pytest.raises(lto.LtoError):
        p2.set_kernel(p2.KERNEL_COOP2)
    with pytest.raises(lto.LtoError):
        p2.set_defect_lanes(4)
    p2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("pcase", ["p1_rho1", "p1_rho1e-2", "p2_unclamped", "p2_clamped", "p1.5", "p0"])
def test_gpu14_one_step_whole_segment_lanes(gpu_ctx, oracle, pcase):
    """Round 6: RK4 with ONE step per segment and the whole 14x14 STM in the segment's own lane (k_indirect_stream<14, PM>,
    cols_per_lane = 14) -- the 14-dim form of SURVEY 8d's HBM-bound corner.  Defect and Phi equal the oracle's dual-number
    derivative of the same one-step map and the per-(segment, column group) kernel's (same functions on the same operands: round-off),
    on a ragged batch of three trajectories x 333 segments with a mixed law in one of them; every control-law class."""
    import torch
    p, rho, thr, lam = CASES[pcase]
    n, B = 334, 3
    XC, T = synth.indirect_problem(n, n_batch=B, seed=31, dt_seg=0.01, lam_sigma=lam)
    S1, S = n - 1, (n - 1) * B
    X = np.zeros((14, n, B), order="F")
    X[:6] = XC[:6]; X[6] = 1000.0 - 0.03 * np.arange(n)[:, None]; X[7:13] = XC[6:]; X[13] = 0.25
    ps = [p, p, 1.0]                                              # the third trajectory always smooth-switch: two classes in one launch unless p = 1
    prm_l = [[MU, DU, TU, thr, 2000.0, 1.0, ps[b], rho] for b in range(B)]
    plan = lto.IndirectPlan(gpu_ctx, n, B, [lto.make_params(*q) for q in prm_l], lto.integrator(lto.RK4, steps=1), ndim=14)
    Xd = torch.from_numpy(synth.to_soa_nodes(X)).cuda()
    td = torch.from_numpy(np.ascontiguousarray(T.T.reshape(-1))).cuda()
    out = {}
    with pytest.raises(lto.LtoError):
        plan.set_cols_per_lane(12)                               # the 12-dim plans' value
    for cols in (14, 2):
        plan.set_cols_per_lane(cols)
        Phi = torch.full((196, S), float("nan"), dtype=torch.float64, device="cuda")
        d = torch.full((14, S), float("nan"), dtype=torch.float64, device="cuda")
        plan.jacobian(Xd, n * B, td, B, Phi, S, d, S)
        torch.cuda.synchronize()
        assert plan.last_kernel() == "per-lane"
        out[cols] = (Phi.cpu().numpy(), d.cpu().numpy())
    plan.close()
    P14, d14 = out[14]
    assert np.all(np.isfinite(P14)) and np.all(np.isfinite(d14))
    assert np.abs(P14 - out[2][0]).max() <= 1e-14 * np.abs(out[2][0]).max()
    assert np.abs(d14 - out[2][1]).max() <= 1e-13                # (the mass row is O(1000))
    for b in range(B):
        P_o, d_o, rc = oracle.indirect14(X[:, :, b], T[:, b], prm_l[b], oracle.RK4, 1)
        assert rc == 0
        sl = slice(b * S1, (b + 1) * S1)
        Pg = P14[:, sl].reshape(14, 14, S1).transpose(1, 0, 2)
        assert np.linalg.norm(d14[:, sl] - d_o) / np.linalg.norm(d_o + X[:, 1:, b]) < 1e-13
        assert np.abs(Pg - P_o).max() < 1e-12 * np.abs(P_o).max()
