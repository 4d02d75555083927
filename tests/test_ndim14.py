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
