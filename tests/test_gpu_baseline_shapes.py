"""GPU parity at BASELINE.json's stated shapes: each test runs the HIP path through the C ABI at
the full configured size, compares an oracle sample of it (bit-level for integer-like properties, <= 1e-10 for floating
point: same tableau and step grid on both sides, so differences are round-off), and checks a size-independent property.

  configs[1]  indirect 14-dim + 14x14 STM, 4 096 segments, RK4 x 64, AUTO kernel
  configs[2]  direct 6-state, 16 384 segments, RKF7(8) nsteps = 10, Jacobian blocks
  configs[3]  homotopy sweep, 256 rho levels x 1 024 segments, 12-dim + STM, RK4 x 64
  reference data: L2_Anderson_{1,2} halo tables fed to the HIP direct defect (CRTBP_Multishoot_indirect_demo.jl:66-70)
"""
import numpy as np
import pytest

import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
from lowthrustopt_amd.constants import MU, DU, TU

pytestmark = pytest.mark.gpu


def test_configs3_homotopy_sweep_256_levels_x_1024_segments_with_stm(gpu_ctx, oracle):
    """BASELINE configs[3] as written: 256 rho levels (1 -> 1e-4, reduceFuel_indirect's range) x 1 024 segments each
    (n = 1025, B = 256), defect + 12x12 STM in ONE batched launch: finite; equals per-level launches bit for bit (a level's
    result does not depend on its neighbours in the batch); a sample of segments on several levels equals the oracle's
    dual-number STM; levels that share nodes but differ in rho differ."""
    import torch
    n, B = 1025, 256
    S = (n - 1) * B
    XC1, T1 = synth.indirect_problem(n, seed=10)
    XC = np.asfortranarray(np.repeat(XC1, B, axis=2))             # every level starts from the same node set
    T = np.asfortranarray(np.repeat(T1, B, axis=1))
    rhos = synth.homotopy_rhos(B)
    prm_l = [[MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, float(r)] for r in rhos]
    prms = [lto.make_params(*q) for q in prm_l]
    integ = lto.integrator(lto.RK4, steps=64)
    X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T.T)).cuda()          # [level][node]
    plan = lto.IndirectPlan(gpu_ctx, n, B, prms, integ)
    Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
    d = torch.zeros(12, S, dtype=torch.float64, device="cuda")
    plan.jacobian(X, n * B, t, B, Phi, S, d, S)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(Phi).all()) and bool(torch.isfinite(d).all())
    kernel = plan.last_kernel()
    plan.close()
    Xs = torch.from_numpy(synth.to_soa_nodes(XC1)).cuda()
    ts = torch.from_numpy(np.ascontiguousarray(T1[:, 0])).cuda()
    S1 = n - 1
    for b in (0, 1, 100, 255):
        p1 = lto.IndirectPlan(gpu_ctx, n, 1, prms[b], integ)
        p1.set_kernel({"per-lane": p1.KERNEL_PER_LANE, "pipeline8": p1.KERNEL_PIPE8, "cooperative": p1.KERNEL_COOP, "cooperative2": p1.KERNEL_COOP2, "pipeline48": p1.KERNEL_PIPE48, "segment-lane": p1.KERNEL_LANE}[kernel])    # the family the batch ran
        if kernel == "per-lane":
            p1.set_cols_per_lane(3)           # what AUTO picks for the 262 144-segment batch (kernels_indirect.hip)
        Phi1 = torch.zeros(144, S1, dtype=torch.float64, device="cuda")
        d1 = torch.zeros(12, S1, dtype=torch.float64, device="cuda")
        p1.jacobian(Xs, n, ts, 1, Phi1, S1, d1, S1)
        torch.cuda.synchronize()
        sl = slice(b * S1, (b + 1) * S1)
        assert torch.equal(Phi[:, sl], Phi1) and torch.equal(d[:, sl], d1), "level %d" % b
        p1.close()
        Pn = Phi1.cpu().numpy().reshape(12, 12, S1).transpose(1, 0, 2)
        dn = d1.cpu().numpy()
        for i in (0, 511, 1023):
            y, P_o, rc, _, _ = oracle.flow_stm_state_costate(XC1[:, i, 0], prm_l[b], T1[i + 1, 0] - T1[i, 0], oracle.RK4, 64)
            assert rc == 0
            assert np.abs(Pn[:, :, i] - P_o).max() < 1e-10 * np.abs(P_o).max()
            assert np.linalg.norm(dn[:, i] - (y - XC1[:, i + 1, 0])) < 1e-10 * np.linalg.norm(y)
    assert float((d[:, :S1] - d[:, 255 * S1:]).abs().max()) > 1e-6       # rho = 1 vs rho = 1e-4 on the same nodes


def test_configs1_14dim_4096_segments_auto_kernel_vs_oracle(gpu_ctx, oracle):
    """BASELINE configs[1] as benchmarked: 14-dim state + mass + costates with the 14x14 STM, 4 096 segments, RK4 x 64,
    LTO_KERNEL_AUTO (the eight-wave pipeline kernel).  Four blocks of 16 consecutive segments spread over the sweep (64
    segments) equal the oracle's dual-number STM and defect; det Phi = 1 for every segment (the vector field is
    divergence-free: trace F = 0 also with the mass rows)."""
    import torch
    S = 4096
    n = S + 1
    XC, T = synth.indirect_problem(n, seed=0)
    X = np.zeros((14, n), order="F")
    X[:6] = XC[:6, :, 0]; X[6] = 1000.0 - 0.01 * np.arange(n); X[7:13] = XC[6:, :, 0]; X[13] = 0.1
    t = np.ascontiguousarray(T[:, 0])
    prm_l = [MU, DU, TU, 0.05, 2000.0, 1.0, 1.0, 1.0]
    plan = lto.IndirectPlan(gpu_ctx, n, 1, lto.make_params(*prm_l), lto.integrator(lto.RK4, steps=64), ndim=14)
    Xd = torch.from_numpy(synth.to_soa_nodes(X)).cuda()
    td = torch.from_numpy(t).cuda()
    Phi = torch.zeros(196, S, dtype=torch.float64, device="cuda")
    d = torch.zeros(14, S, dtype=torch.float64, device="cuda")
    plan.jacobian(Xd, n, td, 1, Phi, S, d, S)
    torch.cuda.synchronize()
    assert plan.last_kernel() == "pipeline8"
    P = Phi.cpu().numpy().reshape(14, 14, S).transpose(1, 0, 2)
    dn = d.cpu().numpy()
    assert np.all(np.isfinite(P)) and np.all(np.isfinite(dn))
    n_checked = 0
    for i0 in (0, 1357, 2048, S - 16):
        P_o, d_o, rc = oracle.indirect14(X[:, i0:i0 + 17], t[i0:i0 + 17], prm_l, oracle.RK4, 64)
        assert rc == 0
        assert np.linalg.norm(dn[:, i0:i0 + 16] - d_o) / np.linalg.norm(d_o + X[:, i0 + 1:i0 + 17]) < 1e-10
        assert np.abs(P[:, :, i0:i0 + 16] - P_o).max() < 1e-10 * np.abs(P_o).max()
        n_checked += 16
    assert n_checked >= 64
    dets = np.linalg.det(P.transpose(2, 0, 1))
    assert np.abs(dets - 1.0).max() < 1e-7
    # d/d lambda_m(t0) of everything but lambda_m itself is zero for p = 1 (always thrust-limited): the unit column
    assert np.array_equal(P[:, 13, :], np.repeat(np.eye(14)[:, 13:14], S, axis=1))


def test_configs2_direct_16384_segments_oracle_sample(gpu_ctx, oracle):
    """BASELINE configs[2] size (16 384 segments, 6-state, RKF7(8) nsteps = 10, on-device Jacobian blocks): blocks of 8
    consecutive segments spread over the sweep equal the oracle -- defect and errors of the reference's two-sided shooting
    (direct.jl:66-109), Jacobian blocks against the dual-number derivative of the same discrete map, tf column against
    the oracle's d/dh."""
    import torch
    S = 16384
    n = S + 1
    X, U, T = synth.direct_problem(n, seed=3)
    Xs = torch.from_numpy(synth.to_soa_nodes(X)).cuda(); Us = torch.from_numpy(synth.to_soa_nodes(U)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
    plan = lto.DirectPlan(gpu_ctx, 6, n, 1, 10, MU, DU, TU, 2000.0)
    d = torch.zeros(6, S, dtype=torch.float64, device="cuda"); e = torch.zeros(S, dtype=torch.float64, device="cuda")
    Jac = torch.zeros(108, S, dtype=torch.float64, device="cuda"); dtf = torch.zeros(6, S, dtype=torch.float64, device="cuda")
    plan.jacobian(Xs, n, Us, n, t, 1, Jac, S, dtf, d, S, e)
    torch.cuda.synchronize()
    dn, en = d.cpu().numpy(), e.cpu().numpy()
    J = Jac.cpu().numpy().reshape(18, 6, S).transpose(1, 0, 2)       # [row, var, s]
    dtfn = dtf.cpu().numpy()
    Xh, Uh, th = X[:, :, 0], U[:, :, 0], T[:, 0]
    span = th[-1] - th[0]
    for i0 in (0, 4099, 8192, 12345, S - 8):
        sl = slice(i0, i0 + 9)
        d_o, e_o = oracle.direct_defect(Xh[:, sl], Uh[:, sl], th[sl], 10, MU, DU, TU, 2000.0)
        Jd, dh, _ = oracle.direct_jacobian_dual(Xh[:, sl], Uh[:, sl], th[sl], 10, MU, DU, TU, 2000.0)
        assert np.abs(dn[:, i0:i0 + 8] - d_o).max() < 1e-12
        assert np.abs(en[i0:i0 + 8] - e_o).max() < 1e-3 * e_o.max() + 1e-18
        assert np.abs(J[:, :, i0:i0 + 8] - Jd).max() < 1e-11 * max(1.0, np.abs(Jd).max())
        dtf_exact = dh * (np.diff(th[sl]) / span)[None, :]
        assert np.abs(dtfn[:, i0:i0 + 8] - dtf_exact).max() < 1e-9


@pytest.mark.parametrize("which", [0, 1])
def test_reference_halo_tables_through_the_hip_direct_defect(gpu_ctx, which):
    """The only data the reference ships (L2_Anderson_1.txt / _2.txt: two closed Earth-Moon L2 halo orbits, 100 equally
    spaced columns, loaded by CRTBP_Multishoot_indirect_demo.jl:66-70).  Consecutive columns as shooting nodes with zero
    control are a ballistic trajectory, so the HIP direct defect (two-sided RKF7(8) shooting, direct.jl:66-109) must
    vanish to the tables' precision: |defect| <= 1e-8 on all 99 segments, and the RKF 8th-order error estimate is tiny."""
    tab = synth.halo_orbits()[which]
    n = tab.shape[1]
    t = synth.HALO_DT[which] * np.arange(n)
    U = np.zeros((3, n), order="F")
    d, e = lto.direct_defectCalc(np.asfortranarray(tab), U, t, 10, MU, DU, TU, 2000.0, ctx=gpu_ctx)
    assert d.shape == (6, n - 1) and e.shape == (n - 1,)
    assert np.abs(d).max() <= 1e-8
    assert e.max() < 1e-12


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("mname", ["rk4x64", "dop853"])
def test_reference_halo_tables_through_the_hip_indirect_kernels(gpu_ctx, which, mname):
    """The reference's halo tables as shooting nodes of the INDIRECT transcription with zero costates: lambda_v = 0 takes the
    lambda_v = 0 guard of CRTBP_stateCostate_deriv! (stateCostate_deriv.jl:59-64: control = 0), so the state rows are the
    ballistic trajectory of the table -- |defect| <= 1e-8, its precision, on all 99 segments -- and the costate rows of the
    defect are exactly 0 (lambda_dot = 0 at lambda = 0).  Fixed-step RK4 x 64 (pipeline kernels) and the reference's integrator
    setting (adaptive order 8 @ 1e-13: the two-lane cooperative / defect kernels), STM sweep and defect-only sweep."""
    tab = synth.halo_orbits()[which]
    n = tab.shape[1]
    t = synth.HALO_DT[which] * np.arange(n)
    XC = np.zeros((12, n), order="F")
    XC[:6] = tab
    prm = lto.make_params(MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    integ = lto.integrator(lto.RK4, steps=64) if mname == "rk4x64" else lto.integrator()
    Phi, d = lto.indirect_stm(XC, t, prm, integ, ctx=gpu_ctx)
    d0, _ = lto.indirect_defectCalc(XC, t, prm, integ, ctx=gpu_ctx)
    for dd in (d, d0):
        assert dd.shape == (12, n - 1) and np.all(np.isfinite(dd))
        assert np.abs(dd[:6]).max() <= 1e-8
        assert np.all(dd[6:] == 0.0)
    assert np.all(np.isfinite(Phi))
    # the ballistic 6x6 block of Phi has determinant 1 (Liouville); the costates do not feed the state at lambda = 0 exactly
    # only through U = du/dlambda_v, so the full 12x12 determinant is 1 as well
    dets = np.linalg.det(np.transpose(Phi, (2, 0, 1)))
    assert np.abs(dets - 1.0).max() < 1e-8


@pytest.mark.parametrize("kernel,want", [("auto", "pipeline32"), ("pipe48", "pipeline48")])
def test_configs3_size_14dim_large_batch_pipelines_oracle_sample(gpu_ctx, oracle, kernel, want):
    """262 144 segments of the 14-dim system (256 trajectories x 1 024 segments), RK4 x 64, defect + 14x14 STM: AUTO resolves to
    the 32-segment pipeline (round 4; 3.97 ms against 4.08 for the 48-segment form, which is forced in the second case: its base
    role keeps the step's base point and RK4 sum in LDS); blocks of segments spread over the batch equal the oracle's dual-number
    STM and defect."""
    import torch
    n, B = 1025, 256
    S1 = n - 1
    S = S1 * B
    XC1, T1 = synth.indirect_problem(n, n_batch=4, seed=21)
    reps = B // 4
    X14 = np.zeros((14, n, 4), order="F")
    X14[:6] = XC1[:6]; X14[6] = (1000.0 - 0.01 * np.arange(n))[:, None]; X14[7:13] = XC1[6:]; X14[13] = 0.1
    XC = np.asfortranarray(np.tile(X14, (1, 1, reps)))             # trajectory b = copy of trajectory b % 4
    T = np.asfortranarray(np.tile(T1, (1, reps)))
    prm_l = [MU, DU, TU, 0.05, 2000.0, 1.0, 1.0, 1.0]
    plan = lto.IndirectPlan(gpu_ctx, n, B, lto.make_params(*prm_l), lto.integrator(lto.RK4, steps=64), ndim=14)
    if kernel != "auto":
        plan.set_kernel(plan.KERNEL_PIPE48)
    X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T.T)).cuda()
    Phi = torch.zeros(196, S, dtype=torch.float64, device="cuda")
    d = torch.zeros(14, S, dtype=torch.float64, device="cuda")
    plan.jacobian(X, n * B, t, B, Phi, S, d, S)
    torch.cuda.synchronize()
    assert plan.last_kernel() == want
    assert bool(torch.isfinite(Phi).all()) and bool(torch.isfinite(d).all())
    plan.close()
    for b, i0 in ((0, 0), (1, 500), (130, 47), (255, S1 - 8)):
        tr = b % 4
        P_o, d_o, rc = oracle.indirect14(X14[:, i0:i0 + 9, tr], T1[i0:i0 + 9, tr], prm_l, oracle.RK4, 64)
        assert rc == 0
        sl = slice(b * S1 + i0, b * S1 + i0 + 8)
        Pn = Phi[:, sl].cpu().numpy().reshape(14, 14, 8).transpose(1, 0, 2)
        dn = d[:, sl].cpu().numpy()
        assert np.linalg.norm(dn - d_o) / np.linalg.norm(d_o + X14[:, i0 + 1:i0 + 9, tr]) < 1e-10
        assert np.abs(Pn - P_o).max() < 1e-10 * np.abs(P_o).max()
    # copies of a trajectory inside the batch give the same bits wherever they sit
    assert torch.equal(Phi[:, :S1], Phi[:, 4 * S1:5 * S1]) and torch.equal(d[:, 3 * S1:4 * S1], d[:, 255 * S1:256 * S1])


# ---- the per-rank batches of an N-GPU run of the sharded configs (tests/per_rank_sizes.py) ---------------------
from per_rank_sizes import C4 as PER_RANK_C4, C5_STM as PER_RANK_C5     # noqa: E402


@pytest.mark.parametrize("world", [2, 4, 8])
def test_configs3_homotopy_sweep_per_rank_batches(gpu_ctx, oracle, world):
    """configs[3] as `bench.py --workload c4 --gpus N` shards it: the LAST rank's block of 256 / N rho levels (ending at rho = 1e-4) x
    1 024 segments.  AUTO runs the family tests/per_rank_sizes.py lists for that batch (the whole-segment lanes down to 65 536
    segments, the 44-segment pipeline at 32 768) and an oracle sample on the block's first and last level agrees to 1e-10.  (N = 1 is
    test_configs3_homotopy_sweep_256_levels_x_1024_segments_with_stm.)"""
    import torch
    S, family = PER_RANK_C4[world]
    n, B = 1025, S // 1024
    assert B * world == 256
    XC, T = synth.indirect_problem(n, n_batch=B, seed=10 + world - 1)
    rhos = synth.homotopy_rhos(256)[(world - 1) * B:]
    prm_l = [[MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, float(r)] for r in rhos]
    plan = lto.IndirectPlan(gpu_ctx, n, B, [lto.make_params(*q) for q in prm_l], lto.integrator(lto.RK4, steps=64))
    X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T.T)).cuda()
    Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
    d = torch.zeros(12, S, dtype=torch.float64, device="cuda")
    plan.jacobian(X, n * B, t, B, Phi, S, d, S)
    torch.cuda.synchronize()
    assert plan.last_kernel() == family == lto.auto_kernel(12, lto.RK4, 64, 1.0, S)
    plan.close()
    assert bool(torch.isfinite(Phi).all()) and bool(torch.isfinite(d).all())
    for b in (0, B - 1):
        for i in (0, 500, 1023):
            s = b * 1024 + i
            y, P_o, rc, _, _ = oracle.flow_stm_state_costate(XC[:, i, b], prm_l[b], T[i + 1, b] - T[i, b], oracle.RK4, 64)
            assert rc == 0
            P_g = Phi[:, s].cpu().numpy().reshape(12, 12).T
            assert np.abs(P_g - P_o).max() < 1e-10 * np.abs(P_o).max()
            assert np.linalg.norm(d[:, s].cpu().numpy() - (y - XC[:, i + 1, b])) < 1e-10 * np.linalg.norm(y)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_c5_per_rank_batches_with_stm(gpu_ctx, oracle, world):
    """configs[4] as `bench.py --workload c5 | c5_stm --gpus N` shards it: 65 536 / N segments of the C5 inputs (dt ~ U[0.05, 0.5],
    rho = 1e-3), adaptive order 8 @ 1e-13, lanes ordered after a first sweep as the bench does.  STM sweep (AUTO: the two-lanes-per-
    state cooperative kernel at every one of these sizes) and defect-only sweep against the oracle's converged flow on a sample;
    ordered and natural sweeps agree bit for bit.  (N = 1 is test_indirect_adaptive_full_size_properties[c5].)"""
    import torch
    S, family = PER_RANK_C5[world]
    n = S + 1
    XC, T = synth.indirect_problem(n, seed=1 + world - 1, dt_range=(0.05, 0.5))
    prm_l = [MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1e-3]
    plan = lto.IndirectPlan(gpu_ctx, n, 1, lto.make_params(*prm_l), lto.integrator())
    X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
    Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda"); Phi2 = torch.zeros_like(Phi)
    d = torch.zeros(12, S, dtype=torch.float64, device="cuda"); d2 = torch.zeros_like(d); d0 = torch.zeros_like(d)
    plan.jacobian(X, n, t, 1, Phi, S, d, S)
    assert plan.last_kernel() == family == lto.auto_kernel(12, lto.DOP853_ADAPTIVE, 0, 1.0, S)
    plan.rebalance()
    plan.jacobian(X, n, t, 1, Phi2, S, d2, S)
    plan.defect(X, n, t, 1, d0, S)
    torch.cuda.synchronize()
    assert plan.last_kernel() == family
    assert torch.equal(Phi, Phi2) and torch.equal(d, d2)
    acc, rej = plan.step_counts()
    assert acc.min() >= 1 and (acc + rej).max() <= 400
    plan.close()
    heavy = int(np.argmax(acc + rej))
    for s in sorted({0, 1, S // 2, S - 1, heavy}):
        y, P_o, rc, _, _ = oracle.flow_stm_state_costate(XC[:, s, 0], prm_l, T[s + 1, 0] - T[s, 0], oracle.DOP853_ADAPTIVE, 0)
        assert rc == 0
        P_g = Phi[:, s].cpu().numpy().reshape(12, 12).T
        assert np.abs(P_g - P_o).max() < 1e-9 * np.abs(P_o).max()
        ref = y - XC[:, s + 1, 0]
        assert np.linalg.norm(d[:, s].cpu().numpy() - ref) < 1e-10 * np.linalg.norm(y)
        assert np.linalg.norm(d0[:, s].cpu().numpy() - ref) < 1e-10 * np.linalg.norm(y)
