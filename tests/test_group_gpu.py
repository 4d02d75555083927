"""lto_group_*: several device contexts behind one host process (include/lto.h, lowthrustopt_amd/csrc/lto_group.hip).
The GPU box has one device, so the groups here repeat device 0 -- that exercises the whole sharding path (partition,
one-node halo, per-shard host threads and contexts, slab placement in the caller's arrays); only the placement of the
shards on different physical GPUs is not covered."""
import numpy as np
import pytest

import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
from lowthrustopt_amd.constants import MU, DU, TU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def groups():
    return {g: lto.Group([0] * g) for g in (1, 2, 3, 4)}


@pytest.mark.parametrize("g", [1, 2, 3, 4])
@pytest.mark.parametrize("mname", ["rk4", "dop853"])
def test_group_indirect_single_trajectory_is_sharded_by_segments(gpu_ctx, groups, g, mname):
    """n_batch = 1: contiguous segment blocks with a one-node halo, ragged (29 segments over 1..4 shards)."""
    assert len(groups[g]) == g
    XC, T = synth.indirect_problem(30, seed=61, dt_range=(0.05, 0.3))
    XC, t = XC[:, :, 0], T[:, 0]
    prm = lto.make_params(MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 0.1)
    integ = lto.integrator(lto.RK4, steps=16) if mname == "rk4" else lto.integrator()
    d0, e0 = lto.indirect_defectCalc(XC, t, prm, integ, ctx=gpu_ctx)
    P0, dd0 = lto.indirect_stm(XC, t, prm, integ, ctx=gpu_ctx)
    d1, e1 = lto.indirect_defectCalc(XC, t, prm, integ, ctx=groups[g])
    P1, dd1 = lto.indirect_stm(XC, t, prm, integ, ctx=groups[g])
    assert np.array_equal(d0, d1) and np.array_equal(e0, e1)
    assert np.array_equal(P0, P1) and np.array_equal(dd0, dd1)
    assert np.array_equal(lto.indirect_jacobianCalc(XC, t, prm, integ, ctx=gpu_ctx),
                          lto.indirect_jacobianCalc(XC, t, prm, integ, ctx=groups[g]))


@pytest.mark.parametrize("g", [2, 3, 4])
def test_group_indirect_batch_is_sharded_by_trajectory(gpu_ctx, groups, g):
    """n_batch = 5 with per-trajectory grids and parameters (mixed control-law classes) over 2..4 shards; a shared
    grid and a single parameter tuple are passed through unsplit."""
    B, n = 5, 17
    XC, T = synth.indirect_problem(n, n_batch=B, seed=62, dt_range=(0.05, 0.3))
    prms = [lto.make_params(MU, DU, TU, 0.05 * (1 + b), 1000.0, 1.0, [1.0, 2.0, 0.0, 1.5, 1.0][b], 0.5 ** b) for b in range(B)]
    integ = lto.integrator(lto.RKF78_FIXED, steps=4)
    d0, _ = lto.indirect_defectCalc(XC, T, prms, integ, ctx=gpu_ctx)
    P0, _ = lto.indirect_stm(XC, T, prms, integ, ctx=gpu_ctx)
    d1, _ = lto.indirect_defectCalc(XC, T, prms, integ, ctx=groups[g])
    P1, _ = lto.indirect_stm(XC, T, prms, integ, ctx=groups[g])
    assert np.array_equal(d0, d1) and np.array_equal(P0, P1)
    d2, _ = lto.indirect_defectCalc(XC, T[:, 0], prms[0], integ, ctx=gpu_ctx)
    d3, _ = lto.indirect_defectCalc(XC, T[:, 0], prms[0], integ, ctx=groups[g])
    assert np.array_equal(d2, d3)


@pytest.mark.parametrize("nstate", [6, 7])
@pytest.mark.parametrize("g", [2, 3])
def test_group_direct(gpu_ctx, groups, g, nstate):
    """Direct defect / Jacobian blocks / tf partial: segment shards of one trajectory (the tf partial is rescaled from
    the shard's span to the whole trajectory's, direct.jl:506-510) and trajectory shards of a batch."""
    X, U, T = synth.direct_problem(24, seed=63, nstate=nstate)
    Xs, Us, t = X[:, :, 0], U[:, :, 0], T[:, 0]
    a0 = lto.direct_jacobian_blocks(Xs, Us, t, 10, MU, DU, TU, 2000.0, ctx=gpu_ctx)
    a1 = lto.direct_jacobian_blocks(Xs, Us, t, 10, MU, DU, TU, 2000.0, ctx=groups[g])
    assert np.array_equal(a0[0], a1[0]) and np.array_equal(a0[2], a1[2]) and np.array_equal(a0[3], a1[3])
    assert np.abs(a0[1] - a1[1]).max() <= 4e-16 * np.abs(a0[1]).max()
    d0, e0 = lto.direct_defectCalc(Xs, Us, t, 10, MU, DU, TU, 2000.0, ctx=gpu_ctx)
    d1, e1 = lto.direct_defectCalc(Xs, Us, t, 10, MU, DU, TU, 2000.0, ctx=groups[g])
    assert np.array_equal(d0, d1) and np.array_equal(e0, e1)
    Xb, Ub, Tb = synth.direct_problem(12, n_batch=4, seed=64, nstate=nstate)
    b0 = lto.direct_jacobian_blocks(Xb, Ub, Tb, 6, MU, DU, TU, 2000.0, ctx=gpu_ctx)
    b1 = lto.direct_jacobian_blocks(Xb, Ub, Tb, 6, MU, DU, TU, 2000.0, ctx=groups[g])
    assert all(np.array_equal(p, q) for p, q in zip(b0, b1))


def test_group_errors(gpu_ctx, groups):
    XC, T = synth.indirect_problem(10, seed=65)
    XC, t = XC[:, :, 0], T[:, 0]
    integ = lto.integrator(lto.RK4, steps=4)
    with pytest.raises(lto.LtoError) as ei:       # the reference's error("Invalid value of p!") surfaces from a shard
        lto.indirect_defectCalc(XC, t, lto.make_params(MU, DU, TU, 0.05, 1000.0, 1.0, 0.5, 1.0), integ, ctx=groups[3])
    assert ei.value.code == 2 and "Invalid value of p" in str(ei.value)
    with pytest.raises(lto.LtoError) as ei:
        lto.indirect_defectCalc(np.zeros((13, 10)), t, lto.make_params(MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1.0), integ, ctx=groups[2])
    assert ei.value.code == -1
    with pytest.raises(lto.LtoError) as ei:       # entry points without a group form say so
        lto.indirect_newton_step(XC, t, lto.make_params(MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1.0), integ, ctx=groups[2])
    assert ei.value.code == -3
    # more shards than segments: 2 segments over 4 contexts
    d0, _ = lto.indirect_defectCalc(XC[:, :3], t[:3], lto.make_params(MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1.0), integ, ctx=gpu_ctx)
    d1, _ = lto.indirect_defectCalc(XC[:, :3], t[:3], lto.make_params(MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1.0), integ, ctx=groups[4])
    assert np.array_equal(d0, d1)
    with pytest.raises(lto.LtoError):
        lto.Group([99])
