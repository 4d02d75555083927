"""bench.py's bookkeeping, checked without a GPU: which stored counter profile a roofline object quotes (an
8 192-segment line quoted the 4 096-segment profile), the BASELINE sizes of the workloads, the work model's consistency with
SURVEY 8d, and that every compact config leg names a workload bench.py knows."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_pmc_key_carries_dimension_batch_size_and_integrator():
    assert bench.pmc_key("c2", 14) == "c2" and bench.pmc_key("c2", 14, None, 4096) == "c2"
    assert bench.pmc_key("c2", 14, None, 8192) == "c2_8192"                      # the round-4 bug: this used to be "c2"
    assert bench.pmc_key("c2", 12, "dop853", 4096) == "c2_ndim12_dop853"
    assert bench.pmc_key("c2", 12, None, 8192) == "c2_ndim12_8192"
    assert bench.pmc_key("c4", 12, None, 262144) == "c4" and bench.pmc_key("c4", 12, None, 131072) == "c4_131072"
    assert bench.pmc_key("c5", 12, None, 65536) == "c5" and bench.pmc_key("c5_stm", 12, None, 8192) == "c5_stm_8192"
    assert bench.pmc_key("hbm", 12, None, bench.HBM_SEGMENTS) == "hbm_ndim12"
    assert bench.pmc_key("c3", 12, None, 16384) == "c3"


def test_default_sizes_are_the_baseline_configs_and_shard_a_fixed_global_size():
    assert bench.default_segments("c2") == 4096 and bench.default_segments("c3") == 16384
    assert bench.default_segments("c4") == 256 * 1024 and bench.default_segments("c5") == 65536 == bench.default_segments("c5_stm")
    for world in (1, 2, 4, 8):                                                   # c4 / c5: strong scaling, the others weak
        assert bench.default_segments("c4", world) * world == 262144
        assert bench.default_segments("c5", world) * world == 65536
        assert bench.default_segments("c2", world) == 4096
    assert bench.default_ndim("c2") == 14 and bench.default_ndim("c4") == 12 and bench.default_ndim("hbm") == 12


def test_work_model_follows_survey_8d():
    # flops = steps x (stages x F_rhs + C_tab x dim); bytes: both nodes + the two times + defect + STM
    assert bench.WORK[("c2", 14)] == (64 * (4 * 1490 + 14 * 210), 2 * 14 * 8 + 16 + 14 * 8 + 196 * 8)
    assert bench.WORK[("c2", 12)] == (64 * (4 * 1070 + 14 * 156), 2 * 12 * 8 + 16 + 12 * 8 + 144 * 8)
    assert bench.WORK[("c4", 12)] == bench.WORK[("c2", 12)]
    assert bench.WORK[("hbm", 12)] == (4 * 1070 + 14 * 156, 1456)
    assert bench.dop853_flops(1.0, 1070, 156) == 12 * 1070 + 148 * 156


def test_every_config_leg_is_a_known_workload_with_a_stored_profile():
    keys = [k for k, _, _, _ in bench.CONFIG_LEGS]
    assert keys == ["c3", "c4", "c5", "c5_stm", "hbm"]
    for key, wl, steps, warmup in bench.CONFIG_LEGS:
        assert steps >= 5 and warmup >= 2
        prof = os.path.join(ROOT, "profiles", "pmc_%s.json" % bench.pmc_key(wl, bench.default_ndim(wl), None, bench.default_segments(wl)))
        assert os.path.exists(prof), prof
        rec = json.load(open(prof))
        assert rec["hbm_bytes_per_launch"] > 0


def test_roofline_object_quotes_the_profile_of_its_own_batch_size():
    r = bench.roofline("c2", 14, 8192, 0.127)
    assert "pmc_c2_8192.json" in r["traffic_from"] and abs(r["traffic"] - 15.44e6) < 0.2e6
    r = bench.roofline("c2", 14, 4096, 0.0705)
    assert "pmc_c2.json" in r["traffic_from"] and abs(r["frac"] - 569600 * 4096 / 70.5e-6 / 78.6e12) < 1e-3
    r = bench.roofline("c2", 14, 5000, 0.1)
    assert r["traffic"] is None and "no counter profile" in r["traffic_from"]
    h = bench.roofline("hbm", 12, bench.HBM_SEGMENTS, 0.254)
    assert h["bound"] == "hbm" and abs(h["frac"] - 1456 * bench.HBM_SEGMENTS / 0.254e-3 / 8e12) < 1e-6
