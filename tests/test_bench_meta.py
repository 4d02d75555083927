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


def _canned_long_form():
    """A long form with every leg the default run produces: the long form of the builder's round-6 run of the driver's command
    (`--detail`), 20 KB -- the size of round 5's default line, which the driver could not parse."""
    with open(os.path.join(ROOT, "profiles", "r06z_bench_c2_verbose.json")) as fh:
        return json.loads(fh.read().strip().splitlines()[-1])


def test_default_line_budget():
    """the default line is compact_line(long form) -- under LINE_BUDGET (6 000) bytes, a JSON object that
    round-trips, with the contract keys and roofline / cpu_baseline / parity at the top level, numbers and short tokens only."""
    long_form = _canned_long_form()
    assert len(json.dumps(long_form)) > 15000
    line = bench.compact_line(long_form)
    text = json.dumps(line)
    assert len(text) < bench.LINE_BUDGET == 6000, len(text)
    back = json.loads(text)
    assert back == line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity", "configs", "ok", "detail"):
        assert k in back, k
    assert back["metric"] == long_form["metric"] and back["steps"] == 20 and back["warmup"] == 5 and back["n_gpus"] == 1
    assert abs(back["value"] / long_form["value"] - 1) < 1e-5 and abs(back["ms_per_step"] / long_form["ms_per_step"] - 1) < 1e-5
    r = back["roofline"]
    assert set(r) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and r["bound"] in ("hbm", "mfma")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and abs(r["frac"] / long_form["roofline"]["frac"] - 1) < 1e-3
    assert r["algorithmic"] == 1920 * 4096 and r["traffic"] > 0
    cb = back["cpu_baseline"]
    assert set(cb) >= {"value", "unit", "cores", "kind", "sample"} and cb["kind"] == "port" and cb["cores"] == 1
    assert list(back["configs"]) == ["c1", "c3", "c4", "c5", "c5_stm", "hbm"]
    c1 = back["configs"].pop("c1")
    assert c1["segments"] == 29 and c1["ok"] is True and c1["stm_us"] > c1["defect_us"] > 0
    for key, leg in back["configs"].items():
        assert set(leg) >= {"ms_per_step", "value", "kernel", "kernel_ms", "frac", "bound", "traffic", "algorithmic", "parity_defect",
                            "parity_stm", "cpu_value", "ok"}, key
        assert leg["ok"] is True and leg["frac"] > 0
    assert back["ok"] is True and "failed" not in back

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(s) for s in strings(back)) <= 100            # tokens, not prose


def test_line_reports_a_failed_leg_and_a_parity_figure_out_of_tolerance():
    """Advisor finding, round 5: a crashed leg or a parity regression must not look like success."""
    long_form = _canned_long_form()
    long_form["configs"]["c4"] = {"error": "RuntimeError: planted"}
    long_form["configs"]["c5"]["parity"]["defect_rel_l2"] = 3e-9
    long_form["reference_integrator"]["parity"]["stm_rel_max"] = 1e-6
    assert bench.legs_failed(long_form) == ["reference_integrator", "c4", "c5"]
    line = bench.compact_line(long_form)
    assert line["ok"] is False and line["failed"] == ["reference_integrator", "c4", "c5"]
    assert line["configs"]["c4"]["ok"] is False and line["configs"]["c5"]["ok"] is False and line["refint"]["ok"] is False
    assert len(json.dumps(line)) < bench.LINE_BUDGET
    long_form = _canned_long_form()
    long_form["parity"]["defect_rel_l2"] = float("nan")
    assert bench.legs_failed(long_form) == ["main"] and bench.compact_line(long_form)["parity"]["ok"] is False


def test_multi_rank_line_stays_compact_and_names_both_transports():
    """N > 1: no config legs, no per-rank prose; the line shows the time of both transports, which one carried the timed legs, and how
    many ranks RCCL saw."""
    long_form = _canned_long_form()
    for k in ("configs", "reference_system_12dim", "reference_integrator", "newton_iteration", "host_api", "cpu_baseline", "parity",
              "cpu_baseline_all_cores", "cpu_baseline_reference_algorithm"):
        long_form.pop(k, None)
    long_form["n_gpus"] = 8
    long_form["config"].update(collective_token="windows", stream="side", devices_token="distinct", slab_ok=True, rccl_ranks=8,
                               transports={"windows_ms": 0.081, "rccl_ms": 0.094, "chosen": "windows"},
                               policy={"serial_ms": 0.09, "overlap_ms": 0.081, "chosen": "overlap"}, global_segments=8 * 4096)
    line = bench.compact_line(long_form)
    assert len(json.dumps(line)) < 2000
    c = line["config"]
    assert c["collective"] == "windows" and c["rccl_ranks"] == 8 and c["transports"]["chosen"] == "windows" and c["policy"]["chosen"] == "overlap"
    assert c["stream"] == "side" and c["devices_token"] == "distinct" and c["slab_ok"] is True


def test_design_table_is_the_one_generated_from_the_profiles():
    """DESIGN.md section 6's table of measurements is generated from the committed profiles
    (tools/design_table.py: bench lines, rocprofv3 kernel-trace averages, counter summaries) -- regenerate it and compare, so that the
    prose cannot drift from the traces; and every fraction in it is reproducible from the kernel-stats average it names."""
    spec_t = importlib.util.spec_from_file_location("design_table", os.path.join(ROOT, "tools", "design_table.py"))
    dt = importlib.util.module_from_spec(spec_t)
    spec_t.loader.exec_module(dt)
    text = dt.table("r06z")
    doc = open(os.path.join(ROOT, "DESIGN.md")).read()
    block = doc[doc.index(dt.BEGIN) + len(dt.BEGIN):doc.index(dt.END)].strip()
    assert block == text.strip()
    assert open(os.path.join(ROOT, "profiles", "r06z_table.md")).read().strip() == text.strip()
    rows = [ln for ln in text.splitlines()[2:]]
    assert len(rows) == len(dt.ROWS) == 11 and all("--" not in r.split("|")[4] for r in rows)      # every row has its rocprofv3 average
    # the contract row's fraction from its rocprofv3 average: 4 096 segments x 569 600 flops / average / 78.6 TFLOP/s
    avg_us, calls, name = dt.kernel_avg("r06z", "c2", "k_indirect_pipe8<14")
    assert calls > 1000 and "pipe8<14" in name
    frac = 4096 * 569600 / (avg_us * 1e-6) / 78.6e12
    import re
    shown = float(re.search(r"FP64 ([0-9.]+)", rows[0]).group(1))
    assert abs(shown - frac) < 2e-3, (shown, frac)              # (the table rescales the bench line's rounded fraction by the two times)


def test_optional_legs_that_raise_or_run_out_of_time_do_not_cost_the_headline():
    """one exception path was all that protected the headline.  Every optional leg now runs behind
    `optional()`: a leg that raises is named in `failed` with `ok: false` in its place, a leg not started because --time-budget was
    used up is named in `skipped` (its `ok` is null and the line's `ok` is not touched), a failed counter pass is neither."""
    long_form = _canned_long_form()
    long_form["configs"]["c4"] = {"error": "RuntimeError: planted"}
    long_form["configs"]["hbm"] = {"error": "skipped: --time-budget 240 s used up"}
    long_form.pop("newton_iteration")
    long_form["leg_errors"] = {"c4": "RuntimeError: planted", "hbm": "skipped: --time-budget 240 s used up", "newton": "ValueError: planted",
                               "live_traffic": "OSError: no rocprofv3"}
    assert bench.legs_failed(long_form) == ["c4", "newton"]
    line = bench.compact_line(long_form)
    assert line["ok"] is False and line["failed"] == ["c4", "newton"] and line["skipped"] == ["hbm"]
    assert line["configs"]["c4"]["ok"] is False and line["configs"]["hbm"]["ok"] is None and "newton_us" not in line
    assert line["value"] > 0 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0 and line["parity"]["ok"] is True
    assert len(json.dumps(line)) < bench.LINE_BUDGET
    long_form = _canned_long_form()
    long_form["leg_errors"] = {"hbm": "skipped: --time-budget 240 s used up"}
    long_form["configs"]["hbm"] = {"error": "skipped: --time-budget 240 s used up"}
    line = bench.compact_line(long_form)
    assert line["ok"] is True and "failed" not in line and line["skipped"] == ["hbm"]
