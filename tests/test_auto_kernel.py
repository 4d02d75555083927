"""LTO_KERNEL_AUTO as a pure function (lto_indirect_auto_kernel: no context, no device) -- checked on the CPU: the choice on both sides
of every round boundary (the same cases tests/test_gpu_parity.py::test_indirect_auto_kernel_choice runs on the device, which also
asserts that the plan and this function agree), and the per-rank batches of an N-GPU bench run (tests/per_rank_sizes.py)."""
import importlib.util
import os

import pytest

import lowthrustopt_amd as lto
from per_rank_sizes import C2, C4, C5_STM, WORLDS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module_for_auto", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

# (ndim, segments, method, steps, family) on 256 CUs
BOUNDARY_CASES = [
    (12, 29, lto.RK4, 64, "pipeline8"), (12, 4096, lto.RK4, 64, "pipeline8"), (14, 4096, lto.RK4, 64, "pipeline8"),
    (14, 4097, lto.RK4, 64, "pipeline32"), (14, 8192, lto.RK4, 64, "pipeline32"), (12, 8192, lto.RK4, 64, "pipeline32"),
    (12, 12288, lto.RK4, 8, "pipeline48"), (12, 16384, lto.RK4, 8, "pipeline32"), (14, 29, lto.RK4, 2, "per-lane"),
    (14, 12288, lto.RK4, 6, "pipeline48"), (14, 16384, lto.RK4, 6, "pipeline32"), (14, 20480, lto.RK4, 6, "pipeline8"),
    (12, 11264, lto.RK4, 6, "pipeline48"), (12, 32768, lto.RK4, 8, "pipeline48"), (14, 24576, lto.RK4, 6, "pipeline48"),
    (12, 24576, lto.RK4, 6, "pipeline48"), (12, 36864, lto.RK4, 64, "pipeline48"), (12, 36865, lto.RK4, 64, "segment-lane"),
    (12, 45057, lto.RK4, 64, "segment-lane"), (12, 65536, lto.RK4, 64, "segment-lane"), (12, 65537, lto.RK4, 64, "pipeline48"),
    (12, 78848, lto.RK4, 64, "pipeline48"), (12, 78849, lto.RK4, 64, "segment-lane"), (12, 90113, lto.RK4, 64, "segment-lane"),
    (12, 262144, lto.RK4, 64, "segment-lane"), (14, 262144, lto.RK4, 64, "pipeline32"),
    (12, 29, lto.DOP853_ADAPTIVE, 0, "cooperative2"), (14, 29, lto.DOP853_ADAPTIVE, 0, "cooperative2"),
    (12, 29, lto.RKF78_ADAPTIVE, 0, "cooperative"), (14, 29, lto.RKF78_FIXED, 4, "cooperative"),
    (12, 65535, lto.RK4, 1, "per-lane"), (12, 1048576, lto.RK4, 1, "per-lane"), (12, 262144, lto.RK4, 3, "segment-lane"),
]


@pytest.mark.parametrize("ndim,S,method,steps,want", BOUNDARY_CASES)
def test_auto_kernel_at_the_round_boundaries(ndim, S, method, steps, want):
    assert lto.auto_kernel(ndim, method, steps, 1.0, S) == want


def test_auto_kernel_depends_on_what_the_families_are_built_for():
    # the 32-segment pipeline pairs stages: 14-dim only for the always-thrust-limited laws (p = 0, 1); general p takes the other forms
    assert lto.auto_kernel(14, lto.RK4, 64, 1.0, 8192) == "pipeline32" and lto.auto_kernel(14, lto.RK4, 64, 0.0, 8192) == "pipeline32"
    assert lto.auto_kernel(14, lto.RK4, 64, 2.0, 8192) == "pipeline8" and lto.auto_kernel(14, lto.RK4, 64, 1.5, 12288) == "pipeline48"
    assert lto.auto_kernel(12, lto.RK4, 64, 1.5, 8192) == "pipeline32"
    # 14-dim DOP853: the two-lanes-per-state form for the always-thrust-limited laws, the one-piece cooperative kernel for the others
    assert lto.auto_kernel(14, lto.DOP853_ADAPTIVE, 0, 0.0, 4096) == "cooperative2" and lto.auto_kernel(14, lto.DOP853_ADAPTIVE, 0, 2.0, 4096) == "cooperative"
    assert lto.auto_kernel(14, lto.DOP853_ADAPTIVE, 0, 1.5, 4096) == "cooperative" and lto.auto_kernel(14, lto.RKF78_ADAPTIVE, 0, 1.0, 4096) == "cooperative"
    # ordered sweeps (a lane order from lto_indirect_plan_rebalance) keep the pipelines: the whole-segment lanes read nodes in place
    assert lto.auto_kernel(12, lto.RK4, 64, 1.0, 262144, ordered=True) == "pipeline48"
    # fewer compute units: the rounds shrink with the device
    assert lto.auto_kernel(12, lto.RK4, 64, 1.0, 4096, n_cus=128) == "pipeline32" and lto.auto_kernel(12, lto.RK4, 64, 1.0, 2048, n_cus=128) == "pipeline8"
    for bad in ((13, lto.RK4, 64, 1.0, 100), (12, 7, 64, 1.0, 100), (12, lto.RK4, 64, 0.5, 100), (12, lto.RK4, 64, 1.0, 0)):
        with pytest.raises(lto.LtoError):
            lto.auto_kernel(*bad)


@pytest.mark.parametrize("world", WORLDS)
def test_per_rank_batches_of_an_n_gpu_run_resolve_to_families_the_gpu_suite_checks_at_that_size(world):
    """bench.py's sharding gives every rank exactly the batch tests/per_rank_sizes.py lists, and AUTO resolves there to the family the
    GPU parity tests of that table run (test_configs3_homotopy_sweep_per_rank_batches, test_c5_per_rank_batches_with_stm)."""
    S4, fam4 = C4[world]
    assert bench.default_segments("c4", world) == S4 and lto.auto_kernel(12, lto.RK4, 64, 1.0, S4) == fam4
    S5, fam5 = C5_STM[world]
    assert bench.default_segments("c5_stm", world) == S5 == bench.default_segments("c5", world)
    assert lto.auto_kernel(12, lto.DOP853_ADAPTIVE, 0, 1.0, S5) == fam5 and lto.auto_kernel(12, lto.DOP853_ADAPTIVE, 0, 1.0, S5, ordered=True) == fam5
    S2, fam2 = C2[world]
    assert bench.default_segments("c2", world) == S2 and lto.auto_kernel(14, lto.RK4, 64, 1.0, S2) == fam2
    src = open(os.path.join(ROOT, "tests", "test_gpu_baseline_shapes.py")).read()
    for name in ("def test_configs3_homotopy_sweep_per_rank_batches", "def test_c5_per_rank_batches_with_stm", "per_rank_sizes import"):
        assert name in src, name
