"""GPU end-to-end: the indirect part of the reference demo (CRTBP_Multishoot_indirect_demo.jl) -- 30-node L2 halo ->
halo transfer, p = 2 then p = 1 at 0.05 N, rho continuation -- converges with every defect / Jacobian / Newton solve
on the device."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_halo_transfer_demo_converges():
    spec = importlib.util.spec_from_file_location("halo_demo", os.path.join(ROOT, "examples", "halo_transfer_demo.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.main(seed=0, verbose=False, rho_target=1e-2)
    assert res["p2"][0] == 0 and res["p2"][1] <= 1e-10          # indirect.jl:280 convergence threshold
    assert res["p1"][0] == 0 and res["p1"][1] <= 1e-10
    assert res["rho"][0] == 0 and res["rho"][1] <= 1e-10


@pytest.mark.gpu
def test_halo_transfer_demo_continuation_to_the_reference_target():
    """The demo's continuation run to the reference's own target, rho = 1e-4 (CRTBP_Multishoot_indirect_demo.jl:277-281:
    `reduceFuel_indirect(..., rho_current = 1, rho_target = 1e-4)`), by both routes the library offers:
      * sequentially -- drivers.reduceFuel_indirect, the mirror of HelperFunctions.jl:105-193 (halving with back-off), every
        multiShoot_CRTBP_indirect call one lto_indirect_solve on the device;
      * concurrently -- drivers.homotopy_solve: a ladder of 28 levels from 0.5 down to 1e-4 as one batch (lto_indirect_solve_batch).
    Both end with status 0 and max |defect| <= 1e-10 (the reference's convergence threshold, indirect.jl:280), and their rho = 1e-4
    trajectories agree to 1e-6."""
    import time
    import numpy as np
    import lowthrustopt_amd as lto
    from lowthrustopt_amd import drivers
    from lowthrustopt_amd.constants import MU, DU, TU
    spec = importlib.util.spec_from_file_location("halo_demo", os.path.join(ROOT, "examples", "halo_transfer_demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    t0 = time.perf_counter()
    n = 30
    X, t = demo.stacked_guess(n)
    rng = np.random.default_rng(0)
    XC = np.vstack([X, 0.1 * rng.standard_normal((6, n))])
    XC[:, 1:-1] += 1e-10 * rng.standard_normal((12, n - 2))
    XC, _, f = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, 1e3, 10.0, False, True, 10, 2.0, 1.0, verbose=False)
    XC, _, f = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, 1e3, 10.0, False, False, 50, 2.0, 1.0, verbose=False)
    assert f == 0
    XC1, d1, f1 = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, 1e3, 0.05, False, False, 30, 1.0, 1.0, verbose=False)
    assert f1 == 0 and np.abs(d1).max() <= 1e-10
    rho_target = 1e-4
    Xseq, dseq, fseq = drivers.reduceFuel_indirect(XC1, t, MU, DU, TU, n, 1e3, 0.05, 1.0, rho_target, verbose=False)
    assert fseq == 0 and np.abs(dseq).max() <= 1e-10
    rhos = np.geomspace(0.5, rho_target, 28)
    Xl, Dl, st, waves = drivers.homotopy_solve(XC1, t, MU, DU, TU, 1e3, 0.05, rhos, ctx=lto.default_context(0), verbose=False)
    assert np.all(st == 0) and np.abs(Dl).max() <= 1e-10
    assert np.abs(Xl[:, :, -1] - Xseq).max() <= 1e-6 * max(1.0, np.abs(Xseq).max())
    # the result is a bang-bang-like profile: at rho = 1e-4 the throttle 1/2 (1 + tanh((|lambda_v| - 1) / (2 rho))) is within 1e-6 of
    # 0 or 1 at all but a few of the nodes
    lam = np.linalg.norm(Xseq[9:12], axis=0)
    thr = 0.5 * (1.0 + np.tanh((lam - 1.0) / (2.0 * rho_target)))
    assert np.mean((thr < 1e-6) | (thr > 1.0 - 1e-6)) >= 0.8
    assert time.perf_counter() - t0 < 30.0


@pytest.mark.gpu
def test_default_bench_line_is_compact_complete_and_every_leg_within_tolerance():
    """The line the driver parses (advisor finding on bench.py:1371): `bench.py` with the driver's flags prints
    ONE line last on stdout, under 6 000 bytes, with roofline / cpu_baseline / parity at the top level and the five BASELINE config
    legs; every leg's oracle sample is COMPARED with its tolerance (`ok`), and `--strict` turns a failed leg into a non-zero exit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0.4",
                        "--live-traffic", "off", "--strict"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = p.stdout.strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 6000, (len(lines), len(lines[-1]))
    out = json.loads(lines[0])
    assert out["ok"] is True and "failed" not in out
    assert out["steps"] == 20 and out["warmup"] == 5 and out["n_gpus"] == 1 and out["dtype"] == "f64"
    assert out["parity"]["ok"] is True and out["parity"]["defect_rel_l2"] < 1e-10
    assert out["roofline"]["bound"] == "mfma" and 0.2 < out["roofline"]["frac"] < 1.0 and out["roofline"]["traffic"] > 0
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0
    assert list(out["configs"]) == ["c1", "c3", "c4", "c5", "c5_stm", "hbm"]
    c1 = out["configs"].pop("c1")                    # configs[0]: the demo's 29 segments -- latencies, all segments against the oracle
    assert c1["segments"] == 29 and c1["ok"] is True and 20 < c1["defect_us"] < c1["stm_us"] < 400 and c1["rk4_stm_us"] > 20 and c1["cpu_value"] > 0
    for key, leg in out["configs"].items():
        assert leg["ok"] is True and leg["frac"] > 0 and leg["kernel_ms"] > 0 and leg["cpu_value"] > 0, (key, leg)
    assert out["configs"]["hbm"]["bound"] == "hbm"
    assert out["ref12"]["ok"] is True and out["refint"]["ok"] is True


@pytest.mark.gpu
def test_bench_headline_survives_a_spent_time_budget():
    """`--time-budget 0`: every optional leg is skipped (named in `skipped`), the contract leg with its roofline, cpu_baseline and
    parity still runs and the line is printed -- what protects the headline on a slow box."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-seconds", "0.3",
                        "--time-budget", "0", "--strict"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["ok"] is True and out["value"] > 1e7 and out["roofline"]["frac"] > 0.2 and out["cpu_baseline"]["value"] > 0 and out["parity"]["ok"] is True
    assert set(out["skipped"]) >= {"ref12", "refint", "host_api", "newton", "c1", "c3", "c4", "c5", "c5_stm", "hbm"}
    assert all(leg["ok"] is None for leg in out["configs"].values()) and "ref12" not in out and "newton_us" not in out
