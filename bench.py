#!/usr/bin/env python3
"""bench.py -- segment-integrations/s of the multiple-shooting hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched as
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, one rank per GPU (RCCL).
Rank 0 prints ONE JSON line.

Workload (default `c2`, BASELINE.json configs[1] "Indirect 14-dim state+costate, 4 096 segments, fixed-step RK4
fp64"): indirect method, 14-dim state + mass + costates PLUS the 14x14 STM (the metric's "state+costate+STM" unit),
4 096 shooting segments per GPU, RK4 with 64 steps per segment, fp64, p = 1, rho = 1, thrust 0.05 N -- i.e. one
`jacobianCalc` sweep (which also emits the defect) of src/multiShoot_CRTBP_indirect.jl:93-146 per step.  Inputs are
synthetic halo->halo stacked trajectories (lowthrustopt_amd/synth.py) resident in HBM (struct-of-arrays) before the
timed region starts.  The reference's own CRTBP system is 12-dim (constant mass): the same sweep on that
system -- the one reference parity is claimed for -- is timed in the same run and reported as `reference_system_12dim`
(`--ndim 12` makes it the main line).
N > 1: weak scaling -- every rank sweeps its own 4 096 segments, then one RCCL all-gather of the per-rank
defect slabs (ndim x 4096 doubles) gives every rank the full defect vector.

Timing: the K steps of the timed region are enqueued back to back with NO event in between (an event pair per launch
serialises the queue and costs ~12 us per step at the contract size); `value` = segments x K / that wall time, so it does
not depend on K.  The dominant kernel's launch duration (`roofline.kernel_ms`) comes from a separate pass right after the
timed region, at the same ramped clocks: one HIP event pair on the launch stream around a burst of back-to-back launches,
divided by their number (so the ~1.5 us between consecutive launches is counted against the kernel).  N_SAMPLE isolated
launches with an event pair each are reported beside it (`kernel_ms_isolated`): the device drops its clocks between them.

Device clocks: the device ramps its clocks over the first ~20 ms of load, so a 25-launch run measures the ramp (91 us per
step) and not the sweep (80 us).  The W + K region is therefore run twice: from cold clocks first (`cold_clocks` in the JSON
line), then again after `--device-warmup-ms` (default 30) of untimed sweeps -- `value` is the second run.

Other workloads (`--workload c3|c4|c5|hbm`) are measurement aids for DESIGN.md, not the contract line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
from ctypes import c_void_p as C_void_p

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_TFLOPS = 78.6   # MI355X vector (= matrix) FP64 peak
PEAK_HBM_GBS = 8000.0     # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
N_SAMPLE = 8              # isolated launches timed with an event pair each, after the timed region
N_BURST = 200             # back-to-back launches inside one event pair (fewer for millisecond kernels)

# Algorithmic work per unit (DESIGN.md "Roofline"): flops = steps*(stages*F_rhs + C_tab*dim)
# F_rhs (model flops: + - x / sqrt tanh = 1, FMA = 2): 12-dim 95, + 12x12 variational 1070; 14-dim 110,
# + 14x14 variational 1490 = 116 base + 86 coefficient build + 14 columns x 92 (counted on dynamics.hpp).
WORK = {
    # (name, ndim): (flops per segment, algorithmic bytes per segment)
    ("c2", 12): (64 * (4 * 1070 + 14 * 156), 1456),       # 12-dim + 12x12 STM, RK4 x 64
    ("c2", 14): (64 * (4 * 1490 + 14 * 210), 1920),       # 14-dim + 14x14 STM, RK4 x 64  (BASELINE configs[1])
    ("c2_defect", 12): (64 * (4 * 95 + 14 * 12), 304),
    ("c2_defect", 14): (64 * (4 * 110 + 14 * 14), 352),
    ("c3", 12): (18 * (13 * 232 + 132 * 60), 1080 + 48),  # direct 6-dim + Phi + Psi, RKF7(8) 9 steps x 2 halves
    ("c4", 12): (64 * (4 * 1070 + 14 * 156), 1456),       # homotopy sweep: the 12-dim + STM unit, per-level rho
    ("hbm", 12): (1 * (4 * 1070 + 14 * 156), 1456),       # 1 RK4 step + full STM output: the HBM evidence point
    ("hbm", 14): (1 * (4 * 1490 + 14 * 210), 1920),
}


def to14(XC):
    """(r, v, lambda_r, lambda_v) -> (r, v, m, lambda_r, lambda_v, lambda_m) with m = 1000 kg, lambda_m = 0.1; the
    params' mass slot carries Isp for the 14-dim system (include/lto.h)."""
    X = np.zeros((14,) + XC.shape[1:], order="F")
    X[:6] = XC[:6]; X[6] = 1000.0; X[7:13] = XC[6:]; X[13] = 0.1
    return X


HBM_SEGMENTS = 1048576    # the HBM evidence point's batch: enough segments for every CU to stream


def default_ndim(wl):
    """14 = BASELINE configs[1]'s system (c2, c2_defect); 12 = the reference's own system (everything else, incl. the HBM point)."""
    return 14 if wl in ("c2", "c2_defect") else 12


def default_segments(wl, world=1):
    """Segments per GPU of a workload at its BASELINE size.  c4 / c5 shard a FIXED global size over the ranks (strong scaling);
    the others give every rank the same batch (weak scaling)."""
    if wl in ("c5", "c5_stm"):
        return 65536 // max(world, 1)
    if wl == "c4":
        return (256 // max(world, 1)) * 1024
    return {"c3": 16384, "hbm": HBM_SEGMENTS}.get(wl, 4096)


def make_workload(wl, lto, synth, torch, ctx, st, dev, ndim=0, segments=0, method="", rank=0, world=1, cols=0, kernel=0):
    """Synthetic inputs of one workload resident in HBM (struct-of-arrays), its plan, its outputs and `sweep(defect_buffer)` = one
    step of the hot path on stream `st`.  Used by the contract leg of main() and by the compact `configs` legs."""
    import types
    nd = ndim or default_ndim(wl)
    f64 = dict(dtype=torch.float64, device=dev)
    w = types.SimpleNamespace(wl=wl, ndim=nd, method=method, Phi=None, XC=None, T=None, levels=0, extra={}, blocks=False)
    prm1 = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    c5 = wl in ("c5", "c5_stm")
    if wl in ("c2", "c2_defect", "hbm") or c5:
        S = segments or default_segments(wl, world)
        n = S + 1
        if c5:
            XC, T = synth.indirect_problem(n, seed=1 + rank, dt_range=(0.05, 0.5))
            prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1e-3)
            integ = lto.integrator(lto.DOP853_ADAPTIVE, rtol=1e-13, atol=1e-13)
            desc = "C5: indirect 12-dim defect%s, adaptive DOP853 rtol=atol=1e-13, dt_seg~U[0.05,0.5], rho=1e-3" % (
                " + 12x12 STM" if wl == "c5_stm" else "")
        else:
            XC, T = synth.indirect_problem(n, seed=rank)
            prm = prm1
            steps = 1 if wl == "hbm" else 64
            integ = lto.integrator(lto.RK4, steps=steps)
            if method == "rkf78":
                integ = lto.integrator(lto.RKF78_FIXED, steps=4)
            elif method == "dop853":
                integ = lto.integrator(lto.DOP853_ADAPTIVE, rtol=1e-13, atol=1e-13)
            dd = (nd, nd, nd)
            desc = {"c2": "C2: indirect %d-dim state+costate + %dx%d STM, RK4 x 64 steps, fp64, p=1 rho=1 thrust 0.05 N" % dd,
                    "c2_defect": "C2 (defect only): indirect %d-dim state+costate, RK4 x 64 steps" % nd,
                    "hbm": "HBM evidence point: indirect %d-dim + %dx%d STM, ONE RK4 step per segment" % dd}[wl]
        if nd == 14:   # mass + mass costate: (r, v, m, lambda_r, lambda_v, lambda_m); params' mass slot = Isp
            XC = to14(XC)
            prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 2000.0, 1.0, 1.0, prm.rho)
        X = torch.from_numpy(synth.to_soa_nodes(XC)).to(dev)
        t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).to(dev)
        plan = lto.IndirectPlan(ctx, n, 1, prm, integ, ndim=nd)
        if cols:
            plan.set_cols_per_lane(cols)
        if kernel:
            plan.set_kernel(kernel)
        defect = torch.zeros(nd, S, **f64)
        Phi = torch.zeros(nd * nd, S, **f64)
        if wl == "c5_stm" and nd == 12 and os.environ.get("LTO_BENCH_C5_LAYOUT", "blocks") == "blocks":
            # the reference's own layout (one 12x12 block per segment, indirect.jl:121-123; defect[12 x S] column-major): the ordered
            # sweep then writes its records straight into the caller's arrays -- no record arrays, no transposes (lto.h LTO_LAYOUT_BLOCKS)
            plan.set_output_layout(plan.LAYOUT_BLOCKS)
            w.blocks = True
        if wl in ("c2", "hbm", "c5_stm"):
            def sweep(dbuf):
                plan.jacobian(X, n, t, 1, Phi, S, dbuf, S, stream=st)
        else:
            def sweep(dbuf):
                plan.defect(X, n, t, 1, dbuf, S, stream=st)
        gather_rows = nd
        if method:
            desc += " [integrator %s]" % method
    elif wl == "c4":
        spt = 1024
        levels = max(1, (segments or default_segments(wl, world)) // spt)
        n = spt + 1
        S = levels * spt
        XC, T = synth.indirect_problem(n, n_batch=levels, seed=10 + rank)
        rhos = synth.homotopy_rhos(256)[rank * levels:(rank + 1) * levels] if world * levels == 256 else synth.homotopy_rhos(levels)
        prm = [lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, r) for r in rhos]
        integ = lto.integrator(lto.RK4, steps=64)
        X = torch.from_numpy(synth.to_soa_nodes(XC)).to(dev)
        t = torch.from_numpy(np.ascontiguousarray(T.T)).to(dev)   # [levels][n]
        plan = lto.IndirectPlan(ctx, n, levels, prm, integ)
        if cols:
            plan.set_cols_per_lane(cols)
        defect = torch.zeros(12, S, **f64)
        Phi = torch.zeros(144, S, **f64)

        def sweep(dbuf):
            plan.jacobian(X, n * levels, t, levels, Phi, S, dbuf, S, stream=st)
        desc = "C4: homotopy sweep, %d rho levels x 1024 segments per GPU, 12-dim + STM, RK4 x 64" % levels
        gather_rows = 12
        w.levels, w.extra["rhos"] = levels, rhos
    else:  # c3
        S = segments or default_segments(wl, world)
        n = S + 1
        Xd, Ud, Td = synth.direct_problem(n, seed=rank)
        X = torch.from_numpy(synth.to_soa_nodes(Xd)).to(dev)
        U = torch.from_numpy(synth.to_soa_nodes(Ud)).to(dev)
        t = torch.from_numpy(np.ascontiguousarray(Td[:, 0])).to(dev)
        plan = lto.DirectPlan(ctx, 6, n, 1, 10, lto.MU, lto.DU, lto.TU, 2000.0)
        if kernel:
            plan.set_kernel(kernel)
        defect = torch.zeros(6, S, **f64)
        errs = torch.zeros(S, **f64)
        Jac = torch.zeros(108, S, **f64)
        dtf = torch.zeros(6, S, **f64)

        def sweep(dbuf):
            plan.jacobian(X, n, U, n, t, 1, Jac, S, dtf, dbuf, S, errs, stream=st)
        desc = "C3: direct 6-dim, RKF7(8) nsteps=10 per half, on-device Jacobian blocks 6x18 + tf column + defect + errors"
        gather_rows = 6
        XC, T, prm, integ, Phi = Xd, Td, None, None, None
        w.extra.update(U=Ud, Jac=Jac, dtf=dtf, errs=errs)
    w.S, w.n, w.XC, w.T, w.prm, w.integ, w.plan, w.defect, w.Phi = S, n, XC, T, prm, integ, plan, defect, Phi
    w.sweep, w.desc, w.gather_rows = sweep, desc, gather_rows
    # the compact line's name of the workload: BASELINE config, system, batch, integrator
    w.token = {"c2": "c2 indirect %d-dim+STM S=%d %s" % (nd, S, method or "rk4x64"), "c2_defect": "c2_defect indirect %d-dim S=%d %s" % (nd, S, method or "rk4x64"),
               "hbm": "hbm indirect %d-dim+STM S=%d rk4x1" % (nd, S), "c3": "c3 direct 6-dim+Jac6x18+tf S=%d rkf78 nsteps=10" % S,
               "c4": "c4 homotopy %dx1024 12-dim+STM rk4x64" % w.levels, "c5": "c5 indirect 12-dim defect S=%d dop853 1e-13" % S,
               "c5_stm": "c5_stm indirect 12-dim+STM S=%d dop853 1e-13%s" % (S, " blocks" if w.blocks else "")}[wl]
    return w



def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=["c2", "c2_defect", "c3", "c4", "c5", "c5_stm", "hbm", "newton"])
    ap.add_argument("--no-rebalance", action="store_true",
                    help="c5: keep the natural segment order (default: lanes ordered by the warm-up sweep's step counts)")
    ap.add_argument("--segments", type=int, default=0, help="segments per GPU (default: the workload's)")
    ap.add_argument("--cols", type=int, default=0, help="STM columns per lane (0 = auto)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 per-lane, 2 wave-specialised (cooperative), 3 three-role pipeline (RK4)")
    ap.add_argument("--ndim", type=int, default=0, choices=[0, 12, 14],
                    help="14 = state + mass + costates (BASELINE configs[1]; default of c2 / c2_defect / hbm); 12 = the "
                         "reference's own constant-mass system (parity path; default of the other workloads)")
    ap.add_argument("--method", default="", choices=["", "rk4", "rkf78", "dop853"], help="override the workload's integrator")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="default line without the compact legs of the other BASELINE configs (`configs`)")
    ap.add_argument("--verbose", action="store_true",
                    help="print the long form of the line (every leg's full roofline object, the prose notes on how each figure was obtained); "
                         "default: the compact line of compact_line() -- numbers and short tokens only, under LINE_BUDGET bytes")
    ap.add_argument("--detail", default="", help="also write the long form to this file (the compact line's `detail` then names it)")
    ap.add_argument("--time-budget", type=float, default=240.0,
                    help="seconds after which the OPTIONAL legs of the line (12-dim legs, host API, Newton iteration, config legs, counter passes) "
                         "are skipped instead of started: the contract leg and its roofline / cpu_baseline / parity always run")
    ap.add_argument("--strict", action="store_true", help="exit 1 when a leg failed or a parity figure exceeds its tolerance (the line's `ok` says so either way)")
    ap.add_argument("--live-traffic", default="auto", choices=["auto", "on", "off"],
                    help="roofline.traffic from counter passes of THIS run: rank 0 at N = 1 starts `rocprofv3 --pmc FETCH_SIZE` and "
                         "`--pmc WRITE_SIZE` (separate passes) on a child that runs the workload's sweep only; auto = when rocprofv3 is on "
                         "PATH and the baseline legs run too (not with --no-cpu-baseline); on failure the stored profile's figure stays")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # the child of --live-traffic: W + K sweeps, no output
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--device-warmup-ms", type=float, default=30.0,
                    help="untimed sweeps before the W + K region so that it runs at ramped device clocks (0: off; the cold run is reported either way)")
    return ap.parse_args()


def cpu_baseline(workload, seconds, threads=1, ndim=12, reference_algorithm=False):
    """Oracle (CPU restatement of the reference algorithm) on a bounded sample of the same workload.  threads = 1 is
    the reference's own execution model (serial loop over segments); threads > 1 parallelises that loop with OpenMP.
    reference_algorithm: the indirect sweep as the reference executes it -- adaptive order-8 pair at
    reltol = abstol = 1e-13 with the Jacobian by dual numbers pushed through the solver (indirect.jl:79,107-110,121),
    12-dim -- instead of the fixed RK4 x 64 discrete map the GPU line integrates."""
    from oracle import oracle as O
    import lowthrustopt_amd as lto
    from lowthrustopt_amd import synth
    prm = [lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0]
    nseg = 64 if threads == 1 else 64 * threads
    if workload == "c3":
        X, U, T = synth.direct_problem(nseg + 1, seed=0)
        X, U, t = X[:, :, 0], U[:, :, 0], T[:, 0]

        def run():
            d, e = O.direct_defect(X, U, t, 10, lto.MU, lto.DU, lto.TU, 2000.0)
            O.direct_jacobian_fd(X, U, t, d, 10, lto.MU, lto.DU, lto.TU, 2000.0)   # the reference's FD Jacobian
            O.direct_dtf_fd(X, U, t, 10, lto.MU, lto.DU, lto.TU, 2000.0)
        per_call = nseg
        what = "direct defect + 18-column forward-difference Jacobian + tf partial (the reference's method), RKF7(8) nsteps=10"
    else:
        XC, T = synth.indirect_problem(nseg + 1, seed=0)
        XC, t = XC[:, :, 0], T[:, 0]
        if reference_algorithm:
            def run():
                O.indirect_jacobian(XC, t, prm, O.DOP853_ADAPTIVE)
            what = ("indirect 12-dim + 12x12 STM as the reference computes it: adaptive order-8 pair (DOP853 standing in for Vern8) at "
                    "rtol = atol = 1e-13, Jacobian by dual numbers through the solver (indirect.jl:79,107-110,121)")
        elif ndim == 14:
            XC = to14(XC)
            prm14 = [lto.MU, lto.DU, lto.TU, 0.05, 2000.0, 1.0, 1.0, 1.0]

            def run():
                O.indirect14(XC, t, prm14, O.RK4, 64)
            what = "indirect 14-dim + 14x14 STM by dual numbers through RK4 x 64 (same discrete map as the GPU run)"
        else:
            def run():
                O.indirect_jacobian(XC, t, prm, O.RK4, 64)
            what = "indirect 12-dim + 12x12 STM by dual numbers through RK4 x 64 (same discrete map as the GPU run)"
        per_call = nseg
    O.lib()
    used = O.set_threads(threads)
    run()
    t0 = time.perf_counter()
    calls = 0
    while True:
        run()
        calls += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    O.set_threads(1)
    return {"value": per_call * calls / el, "unit": "segment-integrations/s", "cores": used, "kind": "port",
            "sample_token": "%d seg x %d sweeps / %.1f s, oracle %s, %d host cores" % (
                per_call, calls, el, "c3 FD-Jacobian" if workload == "c3" else ("dop853 duals 12-dim" if reference_algorithm else "rk4x64 duals %d-dim" % ndim), os.cpu_count() or 0),
            "sample": "%d segments x %d sweeps in %.1f s; %s; host has %d cores" % (per_call, calls, el, what, os.cpu_count() or 0)}


def sample_launches(torch, sweep, n=N_SAMPLE):
    """(burst_ms, burst_n, isolated): average launch period (ms) of a burst of back-to-back launches of `sweep` inside ONE
    HIP event pair on the current stream (the stream the kernels are launched on), and the durations of n isolated
    launches with an event pair each.  Run straight after the timed region (clocks still ramped): nothing here perturbs
    `value`."""
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(); sweep(); e1.record()
    torch.cuda.synchronize()
    first = e0.elapsed_time(e1)
    burst_n = int(max(8, min(N_BURST, 200.0 / max(first, 1e-3))))     # about 0.2 s at most
    for _ in range(4):
        sweep()
    e0.record()
    for _ in range(burst_n):
        sweep()
    e1.record()
    torch.cuda.synchronize()
    burst_ms = e0.elapsed_time(e1) / burst_n
    iso = []
    for _ in range(n):
        torch.cuda.synchronize()
        e0.record(); sweep(); e1.record()
        torch.cuda.synchronize()
        iso.append(e0.elapsed_time(e1))
    return burst_ms, burst_n, iso


# DOP853 trial step (csrc/rk.hpp dop853_try): 12 RHS evaluations + tableau arithmetic 2 x (50 a_ij + 8 b + 8 e3 + 8 e5) per component
def dop853_flops(trial_steps, f_rhs, dim):
    return trial_steps * (12 * f_rhs + 148 * dim)


def pmc_key(wl, ndim, method=None, segments=0):
    """Name of the stored counter profile of a workload: profiles/pmc_<key>.json (tools/gpu_round.sh, tools/summarize_profile.py).
    A batch other than the workload's single-GPU BASELINE size is part of the key (`c2_8192`): a launch over 8 192 segments does not
    move the bytes of one over 4 096."""
    key = wl
    if wl in ("c2", "c2_defect", "hbm") and ndim == 12:
        key += "_ndim12"
    if segments and segments != default_segments(wl):
        key += "_%d" % segments
    if method:
        key += "_" + method
    return key


def live_traffic(argv_workload, timeout_s=150.0):
    """HBM bytes per launch of the workload's dominant kernel from counter passes of THIS run: two children under
    `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE do not fit one pass; the program itself right after `--`), each running
    `bench.py --pmc-child` = the workload's sweeps only.  Units and the gfx950 correction as tools/summarize_profile.py
    (MI355X_MICROARCH.md, HBM section): KiB, FETCH_SIZE x 2.  Returns (bytes_per_launch or None, how / why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this process already runs under a profiler"
    tmp = tempfile.mkdtemp(prefix="lto_pmc_")
    got = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                   "--steps", "5", "--warmup", "2"] + argv_workload
            env = dict(os.environ, TMPDIR="/tmp")
            try:
                r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s, cwd="/tmp", env=env)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s pass took longer than %.0f s" % (counter, timeout_s)
            except OSError as ex:
                return None, "rocprofv3 could not be started: %s" % ex
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s pass failed (exit %d): %s" % (counter, r.returncode, r.stderr.decode(errors="replace")[-200:])
            per_kernel = {}
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if row.get("Counter_Name") == counter and "lto::k_" in row.get("Kernel_Name", ""):
                            per_kernel.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            if not per_kernel:
                return None, "no %s rows for an lto:: kernel in the pass's output" % counter
            name = max(per_kernel, key=lambda k: len(per_kernel[k]))      # the child launches the workload's sweep kernel and little else
            got[counter] = (name, sum(per_kernel[name]) / len(per_kernel[name]), len(per_kernel[name]))
        if got["FETCH_SIZE"][0] != got["WRITE_SIZE"][0]:
            return None, "the two passes disagree on the dominant kernel"
        fetch_kib, write_kib = got["FETCH_SIZE"][1], got["WRITE_SIZE"][1]
        return (2.0 * fetch_kib + write_kib) * 1024.0, (
            "measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on a child running the same sweep, "
            "%d launches of %s: FETCH_SIZE %.0f KiB (x 2, gfx950) + WRITE_SIZE %.0f KiB" % (got["FETCH_SIZE"][2], got["FETCH_SIZE"][0][:60], fetch_kib, write_kib))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def roofline(wl, ndim, S, kern_ms, work=None, samples=None, burst_n=N_BURST, method=None):
    flops, nbytes = work if work is not None else WORK[(wl, ndim)]
    dur = kern_ms * 1e-3
    ach_tf = flops * S / dur / 1e12
    ach_gb = nbytes * S / dur / 1e9
    traffic, traffic_from = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_%s.json" % pmc_key(wl, ndim, method, S))
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            traffic = rec.get("hbm_bytes_per_launch")
            traffic_from = "stored rocprofv3 --pmc profile of this command (FETCH_SIZE x 2 + WRITE_SIZE, separate passes): profiles/%s, %s" % (
                os.path.basename(pmc), rec.get("source", "?"))
        except Exception as ex:      # noqa: BLE001
            traffic, traffic_from = None, "profiles/%s unreadable: %s" % (os.path.basename(pmc), ex)
    else:
        traffic_from = ("null: no counter profile of this workload is stored (profiles/%s; tools/gpu_round.sh collects FETCH_SIZE / WRITE_SIZE "
                        "in separate --pmc passes)" % os.path.basename(pmc))
    model = ("builder-counted on lowthrustopt_amd/csrc/dynamics.hpp with SURVEY 8d's convention (+ - x / sqrt tanh = 1, FMA = 2): "
             "flops = steps x (stages x F_rhs + C_tab x dim); F_rhs 14-dim + 14x14 variational = 1 490 (116 base + 86 coefficient build + "
             "14 x 92), 12-dim + 12x12 = 1 070 (SURVEY's figure; the same count gives 1 096).  Not all of the model's flops are executed: "
             "for the always-thrust-limited control laws (p = 0, 1) the lambda_m column of the 14x14 STM is the unit vector and nothing feeds "
             "lambda_m back, so the 14-dim kernel integrates 13 columns and 13 base components (kernels_indirect_pipe8.hip) -- about 4 % of "
             "the model's flops are skipped algorithmically; frac counts the model's flops, the executed-work fraction is ~0.96 x frac")
    if wl == "hbm":
        # the one workload of this path whose arithmetic intensity (4.4 flop/B) is below the machine balance (9.8): HBM is its roof
        return {
            "bound": "hbm", "bound_actual": "hbm", "achieved": ach_gb, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach_gb / PEAK_HBM_GBS,
            "traffic": traffic, "traffic_from": traffic_from, "kernel_ms": kern_ms,
            "kernel_ms_from": "one HIP event pair on the launch stream around %d back-to-back launches right after the timed region, "
                              "divided by their number (launch period: includes the gap between consecutive launches)" % burst_n,
            "kernel_ms_isolated": samples, "flops_per_segment": flops, "bytes_per_segment": nbytes, "flops_model": model,
            "fp64": {"achieved": ach_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / PEAK_FP64_TFLOPS},
            "note": "HBM evidence point: ONE RK4 step per segment with the full STM out; achieved = algorithmic bytes per launch / launch period",
        }
    return {
        # schema value "mfma" = the compute roof: the dense FP64 matrix peak of MI355X (78.6 TFLOP/s) is numerically the
        # FP64 vector peak, and the vector pipe is what this kernel runs on (bound_actual / compute_pipe); the HBM roof is in "hbm"
        "bound": "mfma", "bound_actual": "fp64_valu", "compute_pipe": "fp64_valu", "achieved": ach_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s",
        "frac": ach_tf / PEAK_FP64_TFLOPS, "traffic": traffic, "traffic_from": traffic_from,
        "kernel_ms": kern_ms,
        "kernel_ms_from": "one HIP event pair on the launch stream around %d back-to-back launches right after the timed region, "
                          "divided by their number (launch period: includes the gap between consecutive launches)" % burst_n,
        "kernel_ms_isolated": samples,
        "flops_per_segment": flops, "bytes_per_segment": nbytes, "flops_model": model,
        "hbm": {"achieved": ach_gb, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach_gb / PEAK_HBM_GBS},
        "note": "register-resident fp64 ODE integration: bound by FP64 vector issue (no MFMA instruction is issued; the schema's "
                "\"mfma\" names the compute roof, and the MI355X dense FP64 matrix peak equals the vector peak, so the roof is the same "
                "number), not by HBM -- see DESIGN.md 'Roofline'",
    }


def parity_vs_oracle(ndim, XC, T, defect, Phi, S, adaptive=False):
    """Defect / STM of the benchmark's own last sweep against the oracle (checker) on a 256-segment sample (adaptive: 64
    segments against the oracle's converged dual-number flow at the same tolerances)."""
    from oracle import oracle as O
    import lowthrustopt_amd as lto
    ns = min(64 if adaptive else 256, S)
    if adaptive:
        prm_o = [lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0]
        Phi_o, d_o, rc = O.indirect_jacobian(XC[:, :ns + 1, 0], T[:ns + 1, 0], prm_o, O.DOP853_ADAPTIVE, 0)
        against = "CPU oracle, adaptive order 8 @ 1e-13 on dual numbers (the converged flow; step sequences may differ)"
    elif ndim == 14:
        prm_o = [lto.MU, lto.DU, lto.TU, 0.05, 2000.0, 1.0, 1.0, 1.0]
        Phi_o, d_o, rc = O.indirect14(XC[:, :ns + 1, 0], T[:ns + 1, 0], prm_o, O.RK4, 64)
        against = ("CPU oracle of the same 14-dim model, same RK4 x 64 discrete map, dual-number STM (build extension: the "
                   "reference has no 14-dim CRTBP system, so this is implementation parity, not reference parity)")
    else:
        prm_o = [lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0]
        Phi_o, d_o, rc = O.indirect_jacobian(XC[:, :ns + 1, 0], T[:ns + 1, 0], prm_o, O.RK4, 64)
        against = "CPU oracle, same RK4 x 64 discrete map, dual-number STM"
    d_g = defect[:, :ns].cpu().numpy()
    P_g = Phi[:, :ns].cpu().numpy().reshape(ndim, ndim, ns).transpose(1, 0, 2)
    xn = np.linalg.norm(d_o + XC[:, 1:ns + 1, 0])
    return {"defect_rel_l2": float(np.linalg.norm(d_g - d_o) / xn),
            "stm_rel_max": float(np.abs(P_g - Phi_o).max() / np.abs(Phi_o).max()),
            "sample_segments": ns, "oracle_rc": int(rc), "tolerance": 1e-10, "against": against}


def leg_host_api(lto, ctx, XC, T, prm, integ, ndim, S, calls=30):
    """The host-pointer ABI a Julia `ccall` binds (lto_indirect_jacobian): column-major host arrays in, Phi + defect out;
    per call: plan looked up in the context's cache, H2D, sweep, D2H, one synchronise.  Wall time per call with
    preallocated pageable outputs and with page-locked buffers from lto_host_alloc.  PCIe-inclusive -- never `value`."""
    X = np.asfortranarray(XC[:, :, 0]); t = np.ascontiguousarray(T[:, 0])
    n = S + 1
    out_page = (np.zeros((ndim, ndim, S, 1), order="F"), np.zeros((ndim, S, 1), order="F"))
    out_pin = (ctx.pinned_empty((ndim, ndim, S, 1)), ctx.pinned_empty((ndim, S, 1)))
    X_pin = ctx.pinned_empty((ndim, n)); X_pin[:] = X
    t_pin = ctx.pinned_empty((n,)); t_pin[:] = t

    def per_call(fn):
        fn(); fn()
        ts, inside = [], []
        for _ in range(calls):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
            inside.append(ctx.last_call_ms())
        return float(np.median(ts)) * 1e3, float(np.median(inside))
    ms_page, in_page = per_call(lambda: lto.indirect_stm(X, t, prm, integ, ctx=ctx, out=out_page))
    ms_pin, in_pin = per_call(lambda: lto.indirect_stm(X_pin, t_pin, prm, integ, ctx=ctx, out=out_pin))
    assert np.array_equal(out_page[0], out_pin[0])
    return {"entry_point": "lto_indirect_jacobian (Phi + defect), %d-dim, %d segments, through ctypes" % (ndim, S),
            "ms_per_call_pageable": ms_page, "ms_per_call_page_locked": ms_pin, "calls": calls, "statistic": "median",
            "ms_in_library_pageable": in_page, "ms_in_library_page_locked": in_pin,
            "in_library": "entry to return of the C function (lto_last_call_ms): what a C or Julia caller waits for; the "
                          "difference to ms_per_call is the Python binding",
            "segments_per_s_page_locked": S / (ms_pin * 1e-3), "bytes_out": 8 * (ndim * ndim + ndim) * S, "bytes_in": 8 * (ndim + 1) * n,
            "note": "PCIe-inclusive host-buffer path (what a Julia ccall takes); `value` is the device-resident rate"}


def leg_12dim(lto, synth, ctx, st, torch, a, reference_integrator=False):
    """C2 on the reference's own 12-dim system: same segments and timing method.  Default: the contract's integrator (RK4 x 64).
    reference_integrator: the setting multiShoot_CRTBP_indirect really runs (adaptive order 8, reltol = abstol = 1e-13,
    indirect.jl:79,110 -- DOP853 stands in for Vern8), defect + STM, roofline flops from the sweep's own step
    counts.  Returns the result object and a closure that adds the oracle parity figures (run after all GPU timing is done)."""
    S = 4096
    n = S + 1
    XC, T = synth.indirect_problem(n, seed=0)
    X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    integ = lto.integrator(lto.DOP853_ADAPTIVE, rtol=1e-13, atol=1e-13) if reference_integrator else lto.integrator(lto.RK4, steps=64)
    plan = lto.IndirectPlan(ctx, n, 1, prm, integ)
    defect = torch.zeros(12, S, dtype=torch.float64, device="cuda")
    Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
    # as for the contract leg: device_warmup_ms of untimed sweeps of THIS workload, then W warm-up + K timed steps
    tw = time.perf_counter()
    while (time.perf_counter() - tw) * 1e3 < a.device_warmup_ms:
        for _ in range(10):
            plan.jacobian(X, n, t, 1, Phi, S, defect, S, stream=st)
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        plan.jacobian(X, n, t, 1, Phi, S, defect, S, stream=st)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        plan.jacobian(X, n, t, 1, Phi, S, defect, S, stream=st)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    kern_ms, burst_n, samples = sample_launches(torch, lambda: plan.jacobian(X, n, t, 1, Phi, S, defect, S, stream=st))
    if reference_integrator:
        acc, rej = plan.step_counts(stream=st)
        trial = float((acc + rej).sum())
        roof = roofline("c2", 12, S, kern_ms, work=(dop853_flops(trial, 1070, 156) / S, 1456), samples=samples, burst_n=burst_n, method="dop853")
        roof["flops_from"] = ("measured step counts of this sweep: %.2f accepted + %.2f rejected trial steps per segment (max %d) x (12 x 1070 + 148 x 156)"
                              % (acc.mean(), rej.mean(), int((acc + rej).max())))
        # the defect-only sweep of the same setting: what the line search runs 20 times per Newton iteration (indirect.jl:221-246)
        for _ in range(a.warmup):
            plan.defect(X, n, t, 1, defect, S, stream=st)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(a.steps):
            plan.defect(X, n, t, 1, defect, S, stream=st)
        torch.cuda.synchronize()
        el_d = time.perf_counter() - t1
        # the same two sweeps with the controller's warm start (lto_indirect_plan_set_warm_start): what consecutive Newton iterations see
        plan.set_warm_start(True)
        warm = {}
        for name, run in (("stm", lambda: plan.jacobian(X, n, t, 1, Phi, S, defect, S, stream=st)), ("defect_only", lambda: plan.defect(X, n, t, 1, defect, S, stream=st))):
            for _ in range(max(a.warmup, 2)):
                run()
            torch.cuda.synchronize()
            tw = time.perf_counter()
            for k in range(a.steps):
                run()
            torch.cuda.synchronize()
            warm[name] = (time.perf_counter() - tw) / a.steps * 1e3
        plan.set_warm_start(False)
        plan.jacobian(X, n, t, 1, Phi, S, defect, S, stream=st)     # leave the (cold) STM sweep's outputs for the parity check
        torch.cuda.synchronize()
        out = {"value": S * a.steps / el, "unit": "segment-integrations/s", "ms_per_step": el / a.steps * 1e3,
               "workload": "C2 segments on the reference's 12-dim system with the reference's integrator setting: adaptive order 8 (DOP853 for "
                           "Vern8), reltol = abstol = 1e-13, defect + 12x12 STM (what jacobianCalc, indirect.jl:93-146, runs)",
               "stm_kernel": plan.last_kernel(), "roofline": roof,
               "defect_only": {"ms_per_step": el_d / a.steps * 1e3, "value": S * a.steps / el_d,
                               "workload": "defectCalc (indirect.jl:63-90) with the same setting"},
               "warm_start": {"ms_per_step": warm["stm"], "defect_only_ms_per_step": warm["defect_only"],
                              "note": "lto_indirect_plan_set_warm_start: every segment starts from its first accepted step size of the plan's previous "
                                      "sweep (what consecutive Newton iterations / line-search trials see); off by default, `value` is the cold sweep"}}
    else:
        out = {"value": S * a.steps / el, "unit": "segment-integrations/s", "ms_per_step": el / a.steps * 1e3,
               "workload": "C2 on the reference's CRTBP_stateCostate_deriv! system: 12-dim state+costate + 12x12 STM, 4 096 "
                           "segments, RK4 x 64, fp64", "stm_kernel": plan.last_kernel(), "roofline": roofline("c2", 12, S, kern_ms, samples=samples, burst_n=burst_n)}
    plan.close()

    def add_parity():
        if not reference_integrator:
            out["parity"] = parity_vs_oracle(12, XC, T, defect, Phi, S)
        else:
            out["parity"] = parity_vs_oracle(12, XC, T, defect, Phi, S, adaptive=True)
    return out, add_parity


NEWTON_ALPHAS = 20      # LinRange(0.1, 1, 20), indirect.jl:227


def station_keeping_problem(lto, synth, ctx, S, per_rev=16, pert=1e-7, seed=7):
    """A CONVERGED shooting problem of S segments for the Newton-iteration leg: S / per_rev revolutions on the reference's first
    L2 halo orbit (L2_Anderson_1 table), p = 2 with the thrust limit left open at 10 N as in the reference's demo
    (CRTBP_Multishoot_indirect_demo.jl:178-179), costates ~ 0 -- station keeping.  One revolution's nodes come from the library's own
    flow from the table's first column (a batch of two-node problems with spans j T / per_rev), repeated period after period (the table
    closes to 1e-10); the library's Newton loop then converges it, and the interior nodes are perturbed by `pert` so that the measured
    iteration is a real one (update ~ pert, second-order correction on, tame trial points).  The synthetic stacked-halo guess of the
    sweep benchmarks is NOT used here: 4 096 segments of it are a 650 TU arc that no Newton iteration converges, its update has norm
    ~10 and the sweeps at such trial points measure the integrator's reaction to garbage (round 3's 661 us did)."""
    T1 = 99 * synth.HALO_DT[0]
    dt = T1 / per_rev
    x0 = np.concatenate([synth.halo_orbits()[0][:, 0], np.zeros(6)])
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 10.0, 1000.0, 1.0, 2.0, 1.0)
    XCb = np.zeros((12, 2, per_rev - 1), order="F")
    XCb[:, 0, :] = x0[:, None]
    Tb = np.zeros((2, per_rev - 1), order="F")
    Tb[1] = dt * np.arange(1, per_rev)
    d, _ = lto.indirect_defectCalc(XCb, Tb, prm, lto.integrator(), ctx=ctx)
    rev = np.concatenate([x0[:, None], d[:, 0, :] + XCb[:, 1, :]], axis=1)           # [12][per_rev]: x(j dt)
    n = S + 1
    XC = np.asfortranarray(rev[:, np.arange(n) % per_rev])
    t = dt * np.arange(n)
    rng = np.random.default_rng(seed)
    XC[6:, 1:-1] += 1e-8 * rng.standard_normal((6, n - 2))
    Xs, ds, status, iters, hist = lto.indirect_solve(XC, t, prm, None, False, 8, ctx=ctx)
    if status != 0:
        raise RuntimeError("station-keeping problem of %d segments did not converge: status %d, history %s" % (S, status, hist[:, 0]))
    Xp = np.array(Xs, order="F")
    Xp[:, 1:-1] += pert * rng.standard_normal((12, n - 2))
    return Xp, t, prm, {"segments": S, "revolutions": S / per_rev, "dt_TU": dt, "solve_iterations": int(iters),
                        "max_defect_converged": float(np.abs(ds).max()), "perturbation": pert}


class NewtonIteration:
    """One iteration of multiShoot_CRTBP_indirect's loop (src/multiShoot_CRTBP_indirect.jl:280-337) as the library's device-resident
    entry points run it once the line search is on (iteration > 3) -- what a caller of the reference's driver waits for per iteration:
      STM sweep (jacobianCalc, :290)  ->  block-bidiagonal least-squares step (:149-186)  ->  second-order correction: defect sweep at
      x + dx and a re-solve with the stored factorisation (:190-214)  ->  line search: the 20 trial trajectories as ONE batched defect
      sweep + sum(defect.^2) per trial (:221-246)  ->  update (:304)  ->  the defect check at the new point (:328-331), which is the
      chosen trial point bit for bit: its defect block and max |defect| are taken from the line search's sweep, not swept again.
    The reference's integrator setting (adaptive order 8, rtol = atol = 1e-13), 12-dim, about a converged station-keeping trajectory
    perturbed by 1e-7 (station_keeping_problem).  The point of linearisation stays the same
    in every repetition (the update goes to a second array), so every repetition does the same work.  `sync` = the host reads back
    once per iteration as the library's own loop (lto_indirect_solve) does since round 4 (step length, max |defect|, max |dx| through
    lto_read_scalars_dev; the correction mask and the line search's minimiser are taken on the device)."""

    def __init__(self, lto, synth, ctx, st, torch, S):
        self.lto, self.ctx, self.st, self.torch, self.S = lto, ctx, st, torch, S
        n = self.n = S + 1
        NA = NEWTON_ALPHAS
        XC, tt, prm, self.problem = station_keeping_problem(lto, synth, ctx, S)
        self.prm = prm
        f64 = dict(dtype=torch.float64, device="cuda")
        self.X = torch.from_numpy(np.ascontiguousarray(XC)).cuda()          # [12][n]: SoA, node-indexed
        self.t = torch.from_numpy(np.ascontiguousarray(tt)).cuda()
        self.plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator())
        self.plan_ls = lto.IndirectPlan(ctx, n, NA, prm, lto.integrator())
        self.Phi = torch.zeros(144, S, **f64)
        self.d = torch.zeros(12, S, **f64); self.d2 = torch.zeros(12, S, **f64); self.d3 = torch.zeros(12, S, **f64)
        self.dt = torch.zeros(12, S * NA, **f64)
        self.delta = torch.zeros(12, n, **f64); self.delta2 = torch.zeros(12, n, **f64)
        self.X2 = torch.zeros(12, n, **f64); self.Xn = torch.zeros(12, n, **f64)
        self.Xt = torch.zeros(12, n * NA, **f64)
        self.alphas = torch.linspace(0.1, 1.0, NA, **f64)
        self.ss = torch.zeros(NA, **f64); self.mxt = torch.zeros(NA, **f64)
        self.scal = torch.zeros(3, **f64)                                       # [step | max |defect| | max |dx|]: one read-back
        self.step, self.mx, self.mx2 = self.scal[0:1], self.scal[1:2], self.scal[2:3]
        self.host = np.zeros(3)
        self.count = 0

    def ops(self):
        """(name, closure) in the order of one iteration; every closure enqueues on the bench's stream."""
        lto, ctx, st, S, n, NA = self.lto, self.ctx, self.st, self.S, self.n, NEWTON_ALPHAS
        p, pl = self.plan, self.plan_ls
        axpy = lambda x, d, a, y: ctx.check(ctx.lib.lto_axpy_dev(ctx.handle, st, x.data_ptr(), d.data_ptr(), a, y.data_ptr(), 12 * n))   # noqa: E731
        return [
            ("stm_sweep", lambda: p.jacobian(self.X, n, self.t, 1, self.Phi, S, self.d, S, stream=st)),
            ("factor_solve", lambda: p.newton_solve(self.Phi, S, self.d, S, self.delta, n, stream=st)),
            ("max_dx", lambda: lto.defect_norms(ctx, self.delta, n, 12, n, 1, None, self.mx2, stream=st)),
            ("soc_point", lambda: axpy(self.X, self.delta, 1.0, self.X2)),
            ("soc_defect_sweep", lambda: p.defect(self.X2, n, self.t, 1, self.d2, S, stream=st)),
            ("soc_resolve", lambda: p.newton_solve(None, 0, self.d2, S, self.delta2, n, stream=st)),
            ("soc_add", lambda: axpy(self.delta, self.delta2, 1.0, self.delta)),
            ("trial_points", lambda: lto.trial_points(ctx, self.X, self.delta, n, 12, n, 1, self.alphas, self.Xt, n * NA, stream=st)),
            ("line_search_sweep", lambda: pl.defect(self.Xt, n * NA, self.t, 1, self.dt, S * NA, stream=st)),
            ("line_search_norms", lambda: lto.defect_norms(ctx, self.dt, S * NA, 12, S, NA, self.ss, self.mxt, stream=st)),
            # lineSearch's minimiser and the check of :328-331 without another sweep: the updated trajectory IS the chosen trial point
            ("pick_and_take", lambda: lto.line_search_pick(ctx, self.ss, self.mxt, self.alphas, self.dt, S * NA, 12, S, 1, self.step, self.mx,
                                                           self.d3, S, stream=st)),
            ("update", lambda: axpy(self.X, self.delta, 1.0, self.Xn)),
        ]

    def iteration(self, sync=True):
        for name, op in self.ops():
            op()
            if name == "line_search_sweep" and self.S * NEWTON_ALPHAS >= 16384:
                # as lto_indirect_solve: trial sweeps of this size run with the lanes ordered by an earlier sweep's step counts, the
                # order renewed every fourth iteration
                if self.count % 4 == 0:
                    self.plan_ls.rebalance(stream=self.st)
                self.count += 1
            if sync and name == "update":          # the library's loop reads back once per iteration (step length, max |defect|, max |dx|)
                self.lto.read_scalars(self.ctx, self.scal, 3, None, 0, self.host, stream=self.st)

    def measure(self, reps=40, split_reps=20):
        torch = self.torch
        for _ in range(5):
            self.iteration()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            self.iteration()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        t0 = time.perf_counter()
        for _ in range(reps):
            self.iteration(sync=False)
        torch.cuda.synchronize()
        us_nosync = (time.perf_counter() - t0) / reps * 1e6
        split = {}
        for name, op in self.ops():          # every operation alone: a burst of back-to-back calls inside one event pair on the launch stream
            for _ in range(3):
                op()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(split_reps):
                op()
            e1.record()
            e1.synchronize()
            split[name] = e0.elapsed_time(e1) / split_reps * 1e3
        self.plan.jacobian(self.X, self.n, self.t, 1, self.Phi, self.S, self.d, self.S, stream=self.st)      # the counters hold the LAST sweep's steps
        acc, rej = self.plan.step_counts(stream=self.st)
        self.plan.defect(self.X, self.n, self.t, 1, self.d2, self.S, stream=self.st)
        acc_d, rej_d = self.plan.step_counts(stream=self.st)
        finite = bool(torch.isfinite(self.d3).all() and torch.isfinite(self.dt).all() and torch.isfinite(self.delta).all())
        return {"segments": self.S, "us_per_iteration": us, "us_per_iteration_without_host_reads": us_nosync,
                "split_us": {k: round(v, 2) for k, v in split.items()}, "split_sum_us": round(sum(split.values()), 1),
                "stm_kernel": self.plan.last_kernel(), "finite": finite, "problem": self.problem,
                "trial_steps_per_segment_stm_sweep": {"mean": float((acc + rej).mean()), "max": int((acc + rej).max())},
                "trial_steps_per_segment_defect_sweep": {"mean": float((acc_d + rej_d).mean()), "max": int((acc_d + rej_d).max())},
                "max_dx": float(self.mx2.item()), "max_defect_after": float(self.mx.item())}

    def close(self):
        self.plan_ls.close(); self.plan.close()


class _OracleOps:
    """Propagation back end of the CPU baseline: the oracle's restatement of the reference's closures (checker code, used here only
    to time the reference's algorithm on the host)."""

    def __init__(self, O):
        self.O = O

    def _prm(self, params):
        return [params.MU, params.DU, params.TU, params.thrustLimit, params.mass, params.time_direction, params.p, params.rho]

    def defect(self, XC, t, params):
        return self.O.indirect_defect(XC, t, self._prm(params), self.O.DOP853_ADAPTIVE)[0]

    def stm(self, XC, t, params):
        Phi, d, _ = self.O.indirect_jacobian(XC, t, self._prm(params), self.O.DOP853_ADAPTIVE)
        return Phi, d

    def defect_batch_sumsq(self, XC_batch, t, params):
        return np.array([np.sum(self.defect(np.asfortranarray(XC_batch[:, :, b]), t, params) ** 2) for b in range(XC_batch.shape[2])])


def cpu_newton_baseline(S, seconds, problem=None):
    """The same iteration the way the reference executes it, on the host: the oracle's closures (adaptive order 8 @ 1e-13, Jacobian by
    dual numbers) behind the Python mirror of the reference's loop body (drivers.optimizeTraj_OLS: sparse `\\` twice, drivers.lineSearch:
    20 defect sweeps), one core.  Repeated until `seconds` have passed, at least once."""
    from oracle import oracle as O
    import lowthrustopt_amd as lto
    from lowthrustopt_amd import synth, drivers
    O.lib(); O.set_threads(1)
    n = S + 1
    if problem is not None:                 # the GPU leg's own point of linearisation
        XC, t, params = problem
    else:
        XC, T = synth.indirect_problem(n)
        XC, t = np.asfortranarray(XC[:, :, 0]), np.ascontiguousarray(T[:, 0])
        params = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    ops = _OracleOps(O)

    def iteration():
        Phi, d = ops.stm(XC, t, params)
        upd = drivers.optimizeTraj_OLS(XC, t, d, Phi, 6, n, params, False, ops)
        alpha = drivers.lineSearch(XC, upd, t, params, ops)
        return ops.defect(XC + alpha * upd, t, params)
    t0 = time.perf_counter()
    calls = 0
    while True:
        iteration()
        calls += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return {"value": el / calls * 1e6, "unit": "us per iteration", "cores": 1, "kind": "port",
            "sample": "%d iteration(s) of the %d-segment problem in %.1f s: oracle closures (adaptive order 8 @ 1e-13, dual-number Jacobian) + "
                      "scipy sparse solves + 20 line-search sweeps, as indirect.jl:280-337 does them" % (calls, S, el)}


def leg_newton(lto, synth, ctx, st, torch, sizes, cpu_seconds):
    out = []
    for S in sizes:
        it = NewtonIteration(lto, synth, ctx, st, torch, S)
        r = it.measure()
        problem = (np.asfortranarray(it.X.cpu().numpy()), it.t.cpu().numpy(), it.prm)
        it.close()
        if cpu_seconds > 0:
            r["cpu_baseline"] = cpu_newton_baseline(S, cpu_seconds, problem)
        out.append(r)
    return {"what": "one iteration of multiShoot_CRTBP_indirect's loop (indirect.jl:280-337) with the reference's integrator setting, device-resident: "
            "STM sweep, block-bidiagonal step, second-order correction (defect sweep + re-solve), 20-trial line search as one batched sweep, "
            "the minimiser's defect block taken from that sweep as the check of :328-331 (the updated trajectory is that trial point), update; wall time per iteration launch by launch "
            "incl. the host's read-back of the iteration's scalars (once per iteration, lto_read_scalars_dev, as lto_indirect_solve does); split_us: every operation "
            "alone, burst of back-to-back calls inside one HIP event pair on the launch stream",
            "sizes": out}


# ---- the other BASELINE configs as compact legs of the default line -------------------------------------------
# key: (workload, timed steps, warm-up steps).  Each leg: W warm-up + K timed steps of the workload at its BASELINE size on this one
# GPU, the dominant kernel's launch period from a burst inside one HIP event pair, the roofline object, a 64-segment oracle sample of
# the leg's own last sweep, and the oracle timed on one host core on a bounded sample of the same workload.
CONFIG_LEGS = (("c3", "c3", 20, 3), ("c4", "c4", 8, 2), ("c5", "c5", 20, 3), ("c5_stm", "c5_stm", 8, 2), ("hbm", "hbm", 20, 3))
PARITY_SAMPLE = 64


def _time_oracle(run, per_call, seconds, what, token=""):
    """One host core on a bounded sample: `run` repeated until `seconds` have passed (at least once after one untimed call)."""
    from oracle import oracle as O
    O.lib(); O.set_threads(1)
    run()
    t0 = time.perf_counter()
    calls = 0
    while True:
        run(); calls += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return {"value": per_call * calls / el, "unit": "segment-integrations/s", "cores": 1, "kind": "port",
            "sample_token": "%d seg x %d sweeps / %.1f s, oracle %s, %d host cores" % (per_call, calls, el, token, os.cpu_count() or 0),
            "sample": "%d segments x %d sweeps in %.1f s; %s; host has %d cores" % (per_call, calls, el, what, os.cpu_count() or 0)}


def phi_sample(w, off, cnt, nd=12):
    """Phi of segments off .. off + cnt of a workload's last sweep as [row][col][segment], whatever layout the plan writes."""
    if w.blocks:             # [S][col * nd + row]
        return w.Phi.reshape(-1)[off * nd * nd:(off + cnt) * nd * nd].cpu().numpy().reshape(cnt, nd, nd).transpose(2, 1, 0)
    return w.Phi[:, off:off + cnt].cpu().numpy().reshape(nd, nd, cnt).transpose(1, 0, 2)


def defect_sample(w, off, cnt, nd=12):
    if w.blocks:             # [S][nd]
        return w.defect.reshape(-1)[off * nd:(off + cnt) * nd].cpu().numpy().reshape(cnt, nd).T
    return w.defect[:, off:off + cnt].cpu().numpy()


def config_parity_and_cpu(w, lto, seconds):
    """(parity, cpu_baseline) of a compact leg: PARITY_SAMPLE segments of the leg's own last sweep against the oracle (checker), and
    the oracle timed on the same sample (one core; the reference's algorithm for the adaptive configs, the same discrete map for the
    fixed-step ones)."""
    from oracle import oracle as O
    wl, S, ns = w.wl, w.S, PARITY_SAMPLE
    if wl == "c3":
        Xh, Uh, th = w.XC[:, :ns + 1, 0], w.extra["U"][:, :ns + 1, 0], w.T[:ns + 1, 0]
        d_o, e_o = O.direct_defect(Xh, Uh, th, 10, lto.MU, lto.DU, lto.TU, 2000.0)
        Jd, dh, _ = O.direct_jacobian_dual(Xh, Uh, th, 10, lto.MU, lto.DU, lto.TU, 2000.0)
        d_g = w.defect[:, :ns].cpu().numpy()
        J_g = w.extra["Jac"][:, :ns].cpu().numpy().reshape(18, 6, ns).transpose(1, 0, 2)
        span = w.T[-1, 0] - w.T[0, 0]
        dtf_o = dh * (np.diff(th) / span)[None, :]
        parity = {"defect_max_abs": float(np.abs(d_g - d_o).max()), "jacobian_rel_max": float(np.abs(J_g - Jd).max() / np.abs(Jd).max()),
                  "tf_column_max_abs": float(np.abs(w.extra["dtf"][:, :ns].cpu().numpy() - dtf_o).max()),
                  "errors_rel_max": float(np.abs(w.extra["errs"][:ns].cpu().numpy() - e_o).max() / e_o.max()),
                  "sample_segments": ns, "tolerance": {"defect_max_abs": 1e-12, "jacobian_rel_max": 1e-11},
                  "against": "CPU oracle: two-sided RKF7(8) shooting (direct.jl:66-109), dual-number derivative of the same discrete map, d/dh for the tf column"}

        def run():
            d, e = O.direct_defect(Xh, Uh, th, 10, lto.MU, lto.DU, lto.TU, 2000.0)
            O.direct_jacobian_fd(Xh, Uh, th, d, 10, lto.MU, lto.DU, lto.TU, 2000.0)
            O.direct_dtf_fd(Xh, Uh, th, 10, lto.MU, lto.DU, lto.TU, 2000.0)
        return parity, _time_oracle(run, ns, seconds, "direct defect + 18-column forward-difference Jacobian + tf partial (the reference's method, "
                                                       "direct.jl:111-166,503-516), RKF7(8) nsteps=10", "c3 FD-Jacobian")
    adaptive = wl in ("c5", "c5_stm")
    method, steps = (O.DOP853_ADAPTIVE, 0) if adaptive else (O.RK4, 1 if wl == "hbm" else 64)
    if wl == "c4":
        # half the sample from the first level of this GPU's block (rho = 1), half from its last (rho = 1e-4 at 256 levels)
        spt, half = 1024, ns // 2
        picks = [(0, 0), (w.levels - 1, spt - half)]
        rhos = w.extra["rhos"]
        blocks = [(b, i0, [lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, float(rhos[b])]) for b, i0 in picks]
        offs = [b * spt + i0 for b, i0, _ in blocks]
        cnt = half
    else:
        rho = 1e-3 if adaptive else 1.0
        blocks = [(0, 0, [lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, rho])]
        offs, cnt = [0], min(ns, S)
    num = den = 0.0
    stm_err = stm_max = 0.0
    want_stm = wl != "c5"
    for (b, i0, prm_o), off in zip(blocks, offs):
        Xh, th = w.XC[:, i0:i0 + cnt + 1, b], w.T[i0:i0 + cnt + 1, b]
        if want_stm:
            P_o, d_o, rc = O.indirect_jacobian(Xh, th, prm_o, method, steps)
            P_g = phi_sample(w, off, cnt)
            stm_err = max(stm_err, float(np.abs(P_g - P_o).max())); stm_max = max(stm_max, float(np.abs(P_o).max()))
        else:
            d_o = O.indirect_defect(Xh, th, prm_o, method, steps)[0]
        d_g = defect_sample(w, off, cnt)
        num += float(np.sum((d_g - d_o) ** 2)); den += float(np.sum((d_o + Xh[:, 1:]) ** 2))
    parity = {"defect_rel_l2": float(np.sqrt(num / den)), "stm_rel_max": (stm_err / stm_max) if want_stm else None,
              "sample_segments": cnt * len(blocks), "tolerance": 1e-10,
              "against": ("CPU oracle, adaptive order 8 @ 1e-13%s (the converged flow; step sequences may differ)" % (" on dual numbers" if want_stm else "")
                          if adaptive else "CPU oracle, same RK4 x %d discrete map, dual-number STM" % steps)}
    b, i0, prm_o = blocks[-1]
    Xh, th = w.XC[:, i0:i0 + cnt + 1, b], w.T[i0:i0 + cnt + 1, b]
    if want_stm:
        def run():
            O.indirect_jacobian(Xh, th, prm_o, method, steps)
    else:
        def run():
            O.indirect_defect(Xh, th, prm_o, method, steps)
    what = {"c4": "indirect 12-dim + 12x12 STM by dual numbers through RK4 x 64 at the block's last rho level (same discrete map as the GPU run)",
            "c5": "defectCalc as the reference computes it (indirect.jl:63-90): adaptive order-8 pair (DOP853 for Vern8) at rtol = atol = 1e-13, C5's segment lengths and rho",
            "c5_stm": "jacobianCalc as the reference computes it (indirect.jl:93-146): the same adaptive solve on 12-partial dual numbers, C5's segment lengths and rho",
            "hbm": "indirect 12-dim + 12x12 STM by dual numbers through ONE RK4 step (same discrete map as the GPU run)"}[wl]
    return parity, _time_oracle(run, cnt, seconds, what, {"c4": "rk4x64 duals", "c5": "dop853", "c5_stm": "dop853 duals", "hbm": "rk4x1 duals"}[wl])


def leg_c1(lto, synth, torch, ctx, st, cpu_seconds):
    """BASELINE configs[0] -- the reference demo's size (CRTBP_Multishoot_indirect_demo.jl: 30 nodes, 29 segments; there on the CPU) --
    on the device: what one sweep costs when nothing fills the chip ("report time per sweep at S = 29").  jacobianCalc and
    defectCalc with the reference's integrator setting, the RK4 x 64 STM sweep, all 29 segments against the oracle, the oracle timed."""
    from oracle import oracle as O
    n, S = 30, 29
    XC, T = synth.indirect_problem(n, seed=0)
    X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    f64 = dict(dtype=torch.float64, device="cuda")
    Phi = torch.zeros(144, S, **f64); d = torch.zeros(12, S, **f64); d0 = torch.zeros(12, S, **f64)
    pa = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator())
    pr = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=64))
    try:
        def burst(fn, reps=200):
            for _ in range(20):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); e1.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        rk4_us = burst(lambda: pr.jacobian(X, n, t, 1, Phi, S, d, S, stream=st))
        stm_us = burst(lambda: pa.jacobian(X, n, t, 1, Phi, S, d, S, stream=st))
        def_us = burst(lambda: pa.defect(X, n, t, 1, d0, S, stream=st))
        torch.cuda.synchronize()
        out = {"segments": S, "stm_us": stm_us, "defect_us": def_us, "rk4_stm_us": rk4_us, "kernel": pa.last_kernel(),
               "ms_per_step": stm_us * 1e-3, "value": S / (stm_us * 1e-6)}
        if cpu_seconds > 0:
            prm_o = [lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0]
            P_o, d_o, rc = O.indirect_jacobian(XC[:, :, 0], T[:, 0], prm_o, O.DOP853_ADAPTIVE, 0)
            P_g = Phi.cpu().numpy().reshape(12, 12, S).transpose(1, 0, 2)
            xn = np.linalg.norm(d_o + XC[:, 1:, 0])
            out["parity"] = {"defect_rel_l2": float(max(np.linalg.norm(d.cpu().numpy() - d_o), np.linalg.norm(d0.cpu().numpy() - d_o)) / xn),
                             "stm_rel_max": float(np.abs(P_g - P_o).max() / np.abs(P_o).max()), "sample_segments": S, "oracle_rc": int(rc), "tolerance": 1e-10,
                             "against": "CPU oracle, adaptive order 8 @ 1e-13 on dual numbers, all 29 segments"}
            out["cpu_baseline"] = _time_oracle(lambda: O.indirect_jacobian(XC[:, :, 0], T[:, 0], prm_o, O.DOP853_ADAPTIVE, 0), S, cpu_seconds,
                                               "jacobianCalc of the demo's 29 segments as the reference computes it", "dop853 duals")
        return out
    finally:
        pa.close(); pr.close()


def leg_config(key, wl, steps, warmup, lto, synth, torch, ctx, st, dev, device_warmup_ms, cpu_seconds):
    """One BASELINE config as a compact leg (see CONFIG_LEGS)."""
    w = make_workload(wl, lto, synth, torch, ctx, st, dev)
    try:
        S, plan, sweep = w.S, w.plan, w.sweep
        c5 = wl in ("c5", "c5_stm")
        tw = time.perf_counter()
        while (time.perf_counter() - tw) * 1e3 < device_warmup_ms:          # ramped clocks, as for the contract leg
            for _ in range(2 if (c5 or wl == "c4") else 10):
                sweep(w.defect)
            torch.cuda.synchronize()
        for _ in range(warmup):
            sweep(w.defect)
        rebalanced = False
        if c5:
            plan.rebalance(stream=st)             # lanes ordered by the warm-up sweep's step counts (on device), as `--workload c5`
            rebalanced = True
            sweep(w.defect)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sweep(w.defect)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        kern_ms, burst_n, samples = sample_launches(torch, lambda: sweep(w.defect), n=3)
        out = {"workload": w.desc, "segments": S, "steps": steps, "warmup": warmup, "ms_per_step": el / steps * 1e3, "value": S * steps / el,
               "unit": "segment-integrations/s", "dtype": "f64"}
        if hasattr(plan, "last_kernel") and wl in ("c4", "c5_stm", "hbm"):       # the family of the STM sweep (defect-only sweeps have no such report)
            out["kernel"] = plan.last_kernel()
        else:
            out["kernel"] = {"c3": "direct-jacobian", "c5": "defect-lanes"}.get(wl)
        if c5:
            acc, rej = plan.step_counts(stream=st)
            tot = (acc + rej).astype(np.float64)
            f_rhs, dim = (1070, 156) if wl == "c5_stm" else (95, 12)
            work = (dop853_flops(float(tot.sum()), f_rhs, dim) / S, 1456 if wl == "c5_stm" else 304)
            out["roofline"] = roofline(wl, 12, S, kern_ms, work=work, samples=samples, burst_n=burst_n)
            out["roofline"]["flops_from"] = "measured step counts of this sweep: %.2f trial steps per segment (max %d) x (12 x %d + 148 x %d)" % (
                tot.mean(), int(tot.max()), f_rhs, dim)
            out["adaptive"] = {"trial_steps_mean": float(tot.mean()), "trial_steps_max": int(tot.max()), "rebalanced": rebalanced}
        else:
            out["roofline"] = roofline(wl, 12, S, kern_ms, samples=samples, burst_n=burst_n)
        for k in ("flops_model", "note", "kernel_ms_from"):          # said once, in the main line's roofline object
            out["roofline"].pop(k, None)
        assert bool(torch.isfinite(w.defect).all()), "non-finite defect in the %s leg" % key
        if cpu_seconds > 0:
            out["parity"], out["cpu_baseline"] = config_parity_and_cpu(w, lto, cpu_seconds)
        return out
    finally:
        w.plan.close()


# ---- the line the driver parses ---------------------------------------------------------------------------------------------------------
# Round 5's default line had grown to 21.9 KB (prose repeated per leg) and the driver, which keeps the last ~8 KB of stdout, could not
# parse it.  The default line is now compact_line(long form): the contract keys, numbers and short tokens only, LINE_BUDGET bytes at most
# (tests/test_bench_meta.py holds it there); how each figure is obtained is said ONCE, in DESIGN.md section 6, and `--verbose` /
# `--detail FILE` still give the long form.
LINE_BUDGET = 6000
DETAIL_DEFAULT = "profiles/r06z_bench_c2_verbose.json"      # the long form of the builder's run of this same command
TOL_REL = 1e-10                                              # north_star: relative defect error vs the reference path
# errors (direct.jl:104): ~1e-17, a cancellation at the rounding level -- a one-ulp change of an input moves it by 3e-4 of its maximum
# (tests/test_oracle_golden.py::test_direct_errors_output_is_conditioned_at_the_rounding_level), so 1e-3 is what can be asked
TOL_C3 = {"defect_max_abs": 1e-12, "jacobian_rel_max": 1e-11, "tf_column_max_abs": 1e-9, "errors_rel_max": 1e-3}


def parity_ok(par):
    """A leg's parity figures against the tolerances written next to them (indirect legs: relative defect L2 and max |dPhi| / max |Phi|
    <= 1e-10; direct leg: TOL_C3)."""
    if not par:
        return None
    if "defect_max_abs" in par:
        return all(par.get(k) is not None and par[k] <= tol for k, tol in TOL_C3.items())
    vals = [par.get("defect_rel_l2"), par.get("stm_rel_max")]
    return all(v <= TOL_REL for v in vals if v is not None) and vals[0] is not None and par.get("oracle_rc", 0) == 0


def _num(x, sig=5):
    """Numbers of the compact line: `sig` significant digits; non-finite -> None (JSON has no NaN)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    x = float(x)
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (sig, x))


def _roof_compact(r, S):
    if not r:
        return None
    other = r.get("hbm") or r.get("fp64") or {}
    o = {"bound": r["bound"], "pipe": r.get("bound_actual"), "achieved": _num(r["achieved"]), "peak": r["peak"], "unit": r["unit"], "frac": _num(r["frac"], 4),
         "traffic": _num(r.get("traffic"), 4), "algorithmic": int(r["bytes_per_segment"] * S) if r.get("bytes_per_segment") else None,
         "kernel_ms": _num(r.get("kernel_ms")), "flops_per_segment": _num(r.get("flops_per_segment"), 6), "bytes_per_segment": r.get("bytes_per_segment"),
         "other_roof_frac": _num(other.get("frac"), 3)}
    src = r.get("traffic_from") or ""
    o["traffic_src"] = "live" if src.startswith("measured in this run") else ("stored" if r.get("traffic") is not None else None)
    return o


def _leg_compact(leg, S=None):
    """A leg of the long form -> {ms_per_step, value, kernel, kernel_ms, frac, bound, traffic, algorithmic, parity_defect, parity_stm,
    cpu_value, ok}."""
    if leg is None:
        return None
    if "error" in leg:       # the leg raised (ok: false) or was not started because the run's time budget was used up (ok: null)
        return {"error": str(leg["error"])[:120], "ok": None if str(leg["error"]).startswith("skipped:") else False}
    r = leg.get("roofline") or {}
    S = S or leg.get("segments")
    par = leg.get("parity") or {}
    o = {"ms_per_step": _num(leg.get("ms_per_step")), "value": _num(leg.get("value")), "kernel": leg.get("kernel") or leg.get("stm_kernel"),
         "kernel_ms": _num(r.get("kernel_ms")), "frac": _num(r.get("frac"), 4), "bound": r.get("bound"), "traffic": _num(r.get("traffic"), 4),
         "algorithmic": int(r["bytes_per_segment"] * S) if (r.get("bytes_per_segment") and S) else None,
         "parity_defect": _num(par.get("defect_rel_l2", par.get("defect_max_abs")), 3),
         "parity_stm": _num(par.get("stm_rel_max", par.get("jacobian_rel_max")), 3),
         "cpu_value": _num((leg.get("cpu_baseline") or {}).get("value"))}
    if "errors_rel_max" in par:
        o["parity_errors"] = _num(par["errors_rel_max"], 3)
    if "stm_us" in leg:          # configs[0] (29 segments): latencies of the three sweeps instead of a roofline
        o = {"segments": leg["segments"], "stm_us": _num(leg["stm_us"], 4), "defect_us": _num(leg["defect_us"], 4), "rk4_stm_us": _num(leg["rk4_stm_us"], 4),
             "kernel": leg.get("kernel"), "parity_defect": o["parity_defect"], "parity_stm": o["parity_stm"], "cpu_value": o["cpu_value"]}
    ad = leg.get("adaptive")
    if ad:
        o["trial_steps"] = [_num(ad.get("trial_steps_mean", ad.get("steps_accepted_mean", 0.0) + ad.get("steps_rejected_mean", 0.0)), 4),
                            ad.get("trial_steps_max", None)]
    if par:
        o["ok"] = parity_ok(par)
    return o


def legs_failed(out):
    """Names of the legs of the long form whose parity is out of tolerance or that raised."""
    bad = []
    if out.get("parity") is not None and not parity_ok(out["parity"]):
        bad.append("main")
    for name in ("reference_system_12dim", "reference_integrator"):
        leg = out.get(name)
        if leg is not None and leg.get("parity") is not None and not parity_ok(leg["parity"]):
            bad.append(name)
    for key, leg in (out.get("configs") or {}).items():
        if ("error" in leg and not str(leg["error"]).startswith("skipped:")) or (leg.get("parity") is not None and not parity_ok(leg["parity"])):
            bad.append(key)
    for it in ((out.get("newton_iteration") or {}).get("sizes") or []):
        if not it.get("finite", True):
            bad.append("newton_%d" % it["segments"])
    for name, why in (out.get("leg_errors") or {}).items():          # optional legs that raised (legs skipped for time are listed in `skipped`;
        if name not in bad and not why.startswith("skipped:") and name != "live_traffic":   # the counter passes are an extra: the stored profile stands)
            bad.append(name)
    return bad


def compact_line(out, detail=None):
    """The default line: the long form `out` reduced to the contract keys, numbers and short tokens."""
    if "error" in out and "metric" not in out:
        return out
    cfg = out.get("config", {})
    S = cfg.get("segments_per_gpu")
    line = {k: out.get(k) for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _num(out["value"], 6), _num(out["ms_per_step"], 6)
    c = {"workload": cfg.get("workload_token") or str(cfg.get("workload", ""))[:100], "segments_per_gpu": S, "global_segments": cfg.get("global_segments"),
         "kernel": cfg.get("stm_kernel"), "collective": cfg.get("collective_token", "none")}
    for k in ("stream", "devices_token", "slab_ok", "transports", "rccl_ranks", "policy"):
        if cfg.get(k) is not None:
            c[k] = cfg[k]
    line["config"] = c
    line["roofline"] = _roof_compact(out.get("roofline"), S)
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _num(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb.get("sample_token") or str(cb.get("sample", ""))[:60]}
        for k, short in (("cpu_baseline_reference_algorithm", "ref_alg"), ("cpu_baseline_all_cores", "all_cores")):
            if out.get(k):
                line["cpu_baseline"][short] = [_num(out[k]["value"]), out[k]["cores"]]
    par = out.get("parity")
    if par:
        line["parity"] = {"defect_rel_l2": _num(par.get("defect_rel_l2", par.get("defect_max_abs")), 3),
                          "stm_rel_max": _num(par.get("stm_rel_max", par.get("jacobian_rel_max")), 3), "tol": TOL_REL if "defect_rel_l2" in par else TOL_C3["defect_max_abs"],
                          "n": par.get("sample_segments"), "ok": parity_ok(par)}
    if out.get("cold_clocks"):
        line["cold_ms_per_step"] = _num(out["cold_clocks"]["ms_per_step"])
    if out.get("adaptive"):
        ad = out["adaptive"]
        line["adaptive"] = {k: _num(v, 4) for k, v in ad.items() if k != "note"}
    if out.get("reference_system_12dim"):
        line["ref12"] = _leg_compact(out["reference_system_12dim"], 4096)
    ri = out.get("reference_integrator")
    if ri:
        o = _leg_compact(ri, 4096)
        o["defect_only_ms"] = _num(ri["defect_only"]["ms_per_step"])
        o["warm_ms"] = [_num(ri["warm_start"]["ms_per_step"]), _num(ri["warm_start"]["defect_only_ms_per_step"])]
        line["refint"] = o
    ha = out.get("host_api")
    if ha:
        line["host_api_ms"] = {"pageable": _num(ha["ms_per_call_pageable"], 4), "pinned": _num(ha["ms_per_call_page_locked"], 4),
                               "in_lib": [_num(ha["ms_in_library_pageable"], 4), _num(ha["ms_in_library_page_locked"], 4)]}
    nw = out.get("newton_iteration")
    if nw:
        line["newton_us"] = {str(it["segments"]): [_num(it["us_per_iteration"], 4), _num(it["us_per_iteration_without_host_reads"], 4), _num(it["split_sum_us"], 4),
                                                     _num((it.get("cpu_baseline") or {}).get("value"), 4)] for it in nw["sizes"]}
    if out.get("configs"):
        line["configs"] = {k: _leg_compact(v) for k, v in out["configs"].items()}
    bad = legs_failed(out)
    line["ok"] = not bad
    if bad:
        line["failed"] = bad
    skipped = [k for k, why in (out.get("leg_errors") or {}).items() if why.startswith("skipped:")]
    if skipped:
        line["skipped"] = skipped
    line["detail"] = detail or DETAIL_DEFAULT
    return line


def emit(out, a):
    """Print the line (compact unless --verbose) as the LAST thing on stdout; `--detail FILE` keeps the long form.  Returns the exit code
    (`--strict`: 1 when a leg failed)."""
    bad = legs_failed(out) if "metric" in out else []
    out["ok"] = not bad
    if bad:
        out["failed"] = bad
        sys.stderr.write("bench.py: legs out of tolerance or failed: %s\n" % ", ".join(bad))
    detail = None
    if a.detail:
        try:
            with open(a.detail, "w") as fh:
                json.dump(out, fh)
                fh.write("\n")
            detail = os.path.relpath(a.detail, ROOT) if os.path.isabs(a.detail) else a.detail
        except OSError as ex:
            sys.stderr.write("bench.py: --detail %s: %s\n" % (a.detail, ex))
    text = json.dumps(out if a.verbose else compact_line(out, detail))
    sys.stdout.flush()
    print(text, flush=True)
    return 1 if (bad and a.strict) else 0


def self_launch(a):
    """`bench.py --gpus N` without a launcher: start the contract's launcher as a CHILD process -- before anything in this process
    has touched the GPU (replacing a process that has initialised HIP takes the node down) -- and leave with its exit code.  Its
    rank 0 prints the JSON line on the stdout this process was given."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))
    if os.environ.get("LTO_BENCH_WATCHDOG_S"):
        # diagnostic: every rank dumps its Python stacks to stderr and leaves after this many seconds (a hung collective otherwise
        # sits until the launcher's own timeout and says nothing about where)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["LTO_BENCH_WATCHDOG_S"]), exit=True)
    import torch
    import lowthrustopt_amd as lto
    from lowthrustopt_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d (or without a launcher: "
                         "bench.py then starts one itself)" % (a.gpus, world, a.gpus))
    dist = None
    # LTO_BENCH_SHARE_DEVICE=1: every local rank uses device 0 and the window transport (which is what exists when ranks share a
    # device); torch.distributed then runs over gloo -- RCCL refuses two ranks on one device.  This is how the N > 1 path of this
    # file is exercised on a one-GPU box (tests/test_comm_gpu.py); the timing it prints is of two processes sharing one GPU.
    share = os.environ.get("LTO_BENCH_SHARE_DEVICE") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    # LTO_BENCH_FORCE_COLLECTIVE=1 under torchrun exercises the RCCL path (init + all-gather) even at world size 1
    force_coll = os.environ.get("LTO_BENCH_FORCE_COLLECTIVE") == "1" and "RANK" in os.environ
    if world > 1 or force_coll:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", dev_index)
    ctl = torch.device("cpu") if share else dev           # where torch.distributed's control tensors live
    ctx = lto.Context(dev_index)
    st = lto.current_stream_ptr()

    t_start = time.perf_counter()
    leg_errors = {}

    def optional(name, fn, fallback=None):
        """An optional leg of the line: never lose the headline to it.  Past --time-budget it is not started; an exception is recorded
        (`failed` in the line names the leg) and the fallback returned."""
        if time.perf_counter() - t_start > a.time_budget:
            leg_errors[name] = "skipped: --time-budget %.0f s used up" % a.time_budget
            return fallback
        try:
            return fn()
        except Exception as ex:      # noqa: BLE001
            leg_errors[name] = "%s: %s" % (type(ex).__name__, str(ex)[:160])
            sys.stderr.write("bench.py: leg %s failed: %s\n" % (name, leg_errors[name]))
            return fallback

    wl = a.workload
    if wl == "newton":
        # measurement aid: the Newton iteration alone -- also what `rocprofv3 --kernel-trace --stats` / `--pmc` runs of
        # tools/gpu_round.sh profile (`--pmc-child`: iterations only, no output)
        sizes = [a.segments] if a.segments else [29, 4096]
        if a.pmc_child:
            it = NewtonIteration(lto, synth, ctx, st, torch, sizes[-1])
            for _ in range(a.warmup + a.steps):
                it.iteration()
            torch.cuda.synchronize()
            it.close(); ctx.close()
            return
        res = leg_newton(lto, synth, ctx, st, torch, sizes, 0.0 if a.no_cpu_baseline else a.cpu_seconds / 3)
        big = res["sizes"][-1]
        code = emit({"metric": "time per iteration of multiShoot_CRTBP_indirect's Newton loop (device-resident)", "value": big["us_per_iteration"],
                     "unit": "us", "n_gpus": 1, "steps": 40, "warmup": 5, "ms_per_step": big["us_per_iteration"] * 1e-3,
                     "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                     "config": {"workload": "newton: %d segments, 12-dim, adaptive order 8 @ 1e-13" % big["segments"],
                                "workload_token": "newton S=%d 12-dim dop853 1e-13" % big["segments"], "segments_per_gpu": big["segments"],
                                "global_segments": big["segments"]},
                     "newton_iteration": res}, a)
        ctx.close()
        if code:
            raise SystemExit(code)
        return
    if a.ndim == 0:
        a.ndim = default_ndim(wl)
    f64 = dict(dtype=torch.float64, device=dev)
    c5 = wl in ("c5", "c5_stm")
    w = make_workload(wl, lto, synth, torch, ctx, st, dev, ndim=a.ndim, segments=a.segments, method=a.method, rank=rank, world=world,
                      cols=a.cols, kernel=a.kernel)
    S, n, XC, T, prm, integ, plan, defect, Phi, sweep, desc, gather_rows = (w.S, w.n, w.XC, w.T, w.prm, w.integ, w.plan, w.defect, w.Phi,
                                                                            w.sweep, w.desc, w.gather_rows)

    if a.pmc_child:          # under rocprofv3 --pmc (live_traffic below): the workload's sweeps and nothing else
        for _ in range(a.warmup + a.steps):
            sweep(defect)
        torch.cuda.synchronize()
        plan.close(); ctx.close()
        return
    use_coll = dist is not None
    # The collective of the product: the library's own RCCL all-gather (lto_comm_allgather_dev; communicator created from an
    # id that rank 0 makes and torch.distributed hands round).  If any rank cannot set it up, every rank falls back to
    # torch.distributed's all_gather_into_tensor (also RCCL) and the JSON line says so.
    native, native_note, transport = None, None, "torch"
    comms = {}               # transport name -> communicator that every rank set up and whose test gather came back right everywhere
    # Stream of the collective (LTO_BENCH_COLLECTIVE_STREAM):
    #   main   the sweep's stream, right after the sweep: stream order is the dependency
    #   side   a second stream: the gather of step k runs beside the sweep of step k + 1 (two defect / gather buffers alternate)
    #   auto   (default for N > 1 on distinct devices) a second stream, and the wait policy -- the next sweep waits for the gather
    #          ("serial") or only buffer reuse does ("overlap") -- is MEASURED before the timed legs: ten steps of each, the slower
    #          rank's time decides on every rank.  A window communicator belongs to ONE stream, so both policies use the side stream.
    #   Ranks that share a device (the one-GPU rehearsal) keep `main`: their timing says nothing and every extra gather is a chance
    #   for four processes on one device to starve one another.
    coll_mode = os.environ.get("LTO_BENCH_COLLECTIVE_STREAM", "main" if (share or world <= 1) else "auto")
    if coll_mode not in ("main", "side", "auto"):
        raise SystemExit("LTO_BENCH_COLLECTIVE_STREAM must be main, side or auto")
    same_stream = coll_mode == "main"
    main = torch.cuda.current_stream()
    comm_stream = (main if same_stream else torch.cuda.Stream(device=dev)) if use_coll else None
    policy = "serial" if same_stream else "overlap"      # side: overlap; auto: decided below
    policy_trial = None
    transport_trial = None
    rccl_ranks = None
    if use_coll:
        # Transports of the library's collective (LTO_BENCH_TRANSPORT = auto | windows | rccl | torch; default auto):
        #   rccl     lto_comm_create + ncclAllGather: the collective north_star names (RCCL over xGMI)
        #   windows  lto_comm_window_*: every rank pushes its slab into IPC-mapped receive windows with device copies and raises a
        #            flag; no RCCL kernel, so no compute unit is taken from the sweep that holds a workgroup on every CU
        #   torch    torch.distributed.all_gather_into_tensor (also RCCL): only when neither of the library's own could be set up
        # auto: BOTH are set up, checked (a test gather of rank-stamped slabs must come back right on every rank) and TIMED before the
        # timed legs (ten steps each, the slowest rank's time); the faster one carries the timed legs, and the line reports both times
        # and the number of ranks the RCCL communicator saw (`rccl_ranks` = ncclCommCount), whichever was chosen.  Ranks that share a
        # device (the one-GPU rehearsal) have the windows only: RCCL refuses two ranks on one device.
        want = "windows" if share else os.environ.get("LTO_BENCH_TRANSPORT", "auto")
        if want not in ("auto", "windows", "rccl", "torch"):
            raise SystemExit("LTO_BENCH_TRANSPORT must be auto, windows, rccl or torch")
        notes = []

        def agreed(ok):
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ctl)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        def gather_works(comm):
            cnt = gather_rows * S
            try:
                src = torch.full((cnt,), float(rank + 1), **f64)
                dst = torch.zeros(world, cnt, **f64)
                comm_stream.wait_stream(main)
                with torch.cuda.stream(comm_stream):   # the stream the timed gathers use: a window communicator is bound to the first one it sees
                    for _ in range(3):             # both window halves, and a reuse
                        comm.allgather(src, dst, cnt, stream=C_void_p(comm_stream.cuda_stream))
                torch.cuda.synchronize()
                good = [bool(torch.all(dst[r] == float(r + 1))) for r in range(world)]
                if not all(good):
                    notes.append("test gather on rank %d: slabs of ranks %s wrong (first values %s), communicator failed = %s" % (
                        rank, [r for r in range(world) if not good[r]], [float(dst[r][0]) for r in range(world)], comm.failed()))
                return all(good)
            except Exception as ex:            # noqa: BLE001
                notes.append("test gather: %s" % ex)
                return False

        # Every torch.distributed collective below is issued by EVERY rank whatever failed locally (advisor finding, round 3: a rank
        # that skipped one left its peers hanging in it): a failed step travels as None / an empty blob, the verdict is all-reduced.
        kinds = ["windows"] if share else {"auto": ["windows", "rccl"], "windows": ["windows", "rccl"], "rccl": ["rccl"], "torch": []}[want]
        for kind in kinds:
            cand = half = None
            try:
                if kind == "windows":
                    def exchange(blob):
                        got = [None] * world
                        dist.all_gather_object(got, blob)
                        return got
                    if os.environ.get("LTO_BENCH_FAIL_RANK") == str(rank):     # test hook: this rank's set-up fails after the exchange
                        exchange(b"")
                        raise RuntimeError("LTO_BENCH_FAIL_RANK")
                    cand = lto.Comm.windows(ctx, world, rank, gather_rows * S, exchange)
                    if share:
                        # ranks on ONE device: the collect kernel's polling blocks (thousands for a multi-megabyte slab) would keep the
                        # peers' push kernels off the compute units; such payloads go by the copy engines here (lto.h)
                        cand.set_kernel_payload(1 << 20)
                        # ... and a peer that is time-sliced out by three others may take long to raise its flag: the bounded wait gets
                        # ten times the default polls before it poisons the result (the default, ~6 s, ran out once in six four-rank runs)
                        cand.set_wait_limit(40000000)
                else:
                    box = [None]
                    if rank == 0:
                        try:
                            box[0] = lto.Comm.unique_id()
                        except Exception as ex:      # noqa: BLE001
                            notes.append("rccl id: %s" % ex)
                    dist.broadcast_object_list(box, src=0, device=ctl)
                    if box[0] is None:
                        raise RuntimeError("rank 0 could not create an RCCL id")
                    cand = lto.Comm(ctx, world, rank, box[0])
            except Exception as ex:      # noqa: BLE001 -- any failure means "this transport is out"
                half = getattr(ex, "comm", None)
                notes.append("%s: %s: %s" % (kind, type(ex).__name__, ex))
            ok = agreed(cand is not None)
            if ok:
                ok = agreed(gather_works(cand))
            if ok:
                comms[kind] = cand
                continue
            dist.barrier()               # nobody closes (unmaps) anything a peer may still be touching
            for c in (cand, half):
                if c is not None:
                    c.close()
            notes.append("%s not usable on every rank" % kind)
        native_note = "; ".join(notes) if notes else None
        if "rccl" in comms:
            rccl_ranks = comms["rccl"].rccl_ranks()
        if comms:
            transport = next(k for k in kinds if k in comms)       # overwritten below when both are there and were timed
            native = comms[transport]
        if share and native is None and world > 1:
            # ranks that share a device have no other transport: every rank leaves the same way (no hang, no half-open windows)
            sys.stderr.write("rank %d: %s\n" % (rank, native_note))
            if rank == 0:
                print(json.dumps({"error": "no usable transport for ranks that share a device", "tried": native_note}), flush=True)
            dist.barrier()
            dist.destroy_process_group()
            plan.close(); ctx.close()
            raise SystemExit(3)
    # N > 1: the defect slab of step k is all-gathered (RCCL) on a side stream while step k+1 propagates: two defect /
    # gather buffers alternate, events order producer -> collective -> buffer reuse.  All collectives complete before
    # the closing barrier + synchronize, so every one of the K steps is fully inside the timed region.
    dbufs = [defect, torch.zeros_like(defect)] if use_coll else [defect]
    gathered = [torch.zeros(world * gather_rows, S, **f64) for _ in dbufs] if use_coll else None   # [rank][row][segment]
    ev_done = [torch.cuda.Event() for _ in dbufs]      # collective on buffer b finished
    ev_ready = [torch.cuda.Event() for _ in dbufs]     # sweep into buffer b finished

    def gather(b):
        with torch.cuda.stream(comm_stream):
            if native is not None:
                native.allgather(dbufs[b], gathered[b], gather_rows * S, stream=C_void_p(comm_stream.cuda_stream))
            elif share:
                raise RuntimeError("ranks share a device and the window transport is not usable: no collective left (gloo does not gather device tensors)")
            else:
                dist.all_gather_into_tensor(gathered[b], dbufs[b])
            if not same_stream:
                ev_done[b].record(comm_stream)

    def step(k, check_free):
        b = k % len(dbufs)
        if use_coll and not same_stream and policy == "overlap" and check_free:
            main.wait_event(ev_done[b])                # buffer b is free again
        sweep(dbufs[b])
        if use_coll:
            if not same_stream:                        # same stream: its order is the dependency, no event needed
                ev_ready[b].record(main)
                comm_stream.wait_event(ev_ready[b])
            gather(b)
            if not same_stream and policy == "serial":
                main.wait_event(ev_done[b])            # the next sweep starts behind the gather, as on one stream

    def trial_ms(steps=10):
        """`steps` steps of the N > 1 path as configured right now, after four untimed ones: ms per step, the slowest rank's time (the
        same number on every rank)."""
        for k in range(4):
            step(k, k >= len(dbufs))
        comm_stream.synchronize(); torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        for k in range(steps):
            step(k, True)
        comm_stream.synchronize(); torch.cuda.synchronize()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ctl)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.barrier()
        return float(tt.item()) / steps * 1e3

    if use_coll and len(comms) > 1:
        # both transports of the library carried the test gather: time each in the step loop the timed legs run (the wait policy at
        # its safe setting), the faster one carries the timed legs unless LTO_BENCH_TRANSPORT names one
        keep = policy
        policy = "serial"
        transport_trial = {}
        for kind in kinds:
            native = comms[kind]
            transport_trial[kind + "_ms"] = trial_ms()
        policy = keep
        transport = want if want in comms else min(comms, key=lambda k: transport_trial[k + "_ms"])
        transport_trial["chosen"] = transport
        native = comms[transport]
    if use_coll and coll_mode == "auto":
        policy_trial = {}
        for pol in ("serial", "overlap"):
            policy = pol
            policy_trial[pol + "_ms"] = trial_ms()
        policy = "overlap" if policy_trial["overlap_ms"] < policy_trial["serial_ms"] else "serial"
        policy_trial["chosen"] = policy

    def timed_leg():
        """The contract's timed region: W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize."""
        for k in range(a.warmup):
            step(k, k >= len(dbufs))
        if use_coll:
            comm_stream.synchronize()
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(a.steps):
            step(k, k >= len(dbufs) or a.warmup >= len(dbufs))
        if use_coll:
            comm_stream.synchronize()
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # The device raises its clocks over the first ~20 ms of load (measured: --steps 20 --warmup 5 -> 91 us per step,
    # --warmup 200 -> 81 us, 2 000 timed steps -> 80 us): a 25-launch run sits entirely inside that ramp.  So the contract
    # leg is run TWICE -- once from cold clocks (reported as `cold_clocks`), then, after the 12-dim legs and `device_warmup_ms` of
    # untimed sweeps of the contract workload, again with W warm-up + K timed steps: that second run is `value`.  --device-warmup-ms 0 reports
    # the cold run as `value`.
    cold = None
    rebalanced = False
    if a.device_warmup_ms > 0 and not c5:
        cold_elapsed = timed_leg()
        cold = {"ms_per_step": cold_elapsed / a.steps * 1e3, "value": world * S * a.steps / cold_elapsed,
                "note": "the same W + K region run first, from idle clocks (max over ranks not taken)"}

    ref12 = refint = None
    if rank == 0 and world == 1 and wl == "c2" and a.ndim == 14 and not a.method and not a.segments:
        # the reference's own system (12-dim, constant mass): the sweep reference parity is claimed for, and the reference's own
        # integrator setting on it -- same run, timed the same way (W warm-up + K timed steps), BEFORE the contract leg's warm-up
        ref12 = optional("ref12", lambda: leg_12dim(lto, synth, ctx, st, torch, a))
        refint = optional("refint", lambda: leg_12dim(lto, synth, ctx, st, torch, a, reference_integrator=True))

    if a.device_warmup_ms > 0 and not c5:
        # untimed sweeps of THIS workload straight before its timed region (the other legs leave the clocks wherever their own
        # kernels and host-side pauses put them)
        tw = time.perf_counter()
        while (time.perf_counter() - tw) * 1e3 < a.device_warmup_ms:
            for _ in range(20):
                sweep(dbufs[0])
            torch.cuda.synchronize()

    if c5 and a.warmup > 0 and not a.no_rebalance:
        for k in range(a.warmup):
            step(k, k >= len(dbufs))
        plan.rebalance(stream=st)                      # lanes ordered by the warm-up sweep's step counts (on device)
        rebalanced = True
    elapsed = timed_leg()
    kern_ms, burst_n, samples = sample_launches(torch, lambda: sweep(dbufs[0]))   # separate pass: the dominant kernel alone

    if use_coll:
        for b in range(len(dbufs)):
            assert torch.equal(gathered[b][rank * gather_rows:(rank + 1) * gather_rows], dbufs[b]), "all-gather slab mismatch"
        slab_ok = agreed(all(bool(torch.isfinite(g).all()) for g in gathered))      # every rank's gathered vector: own slab bit-equal, all slabs finite
        assert slab_ok, "a gathered defect vector holds non-finite values on some rank"
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        kt = torch.tensor([kern_ms], dtype=torch.float64, device=ctl)
        dist.all_reduce(kt, op=dist.ReduceOp.MAX)
        kern_ms = float(kt.item())

    # sanity: the sweep produced finite numbers (a failed launch would leave zeros / raise earlier)
    assert bool(torch.isfinite(defect).all()), "non-finite defect in benchmark sweep"

    exit_code = 0
    if rank == 0:
        value = world * S * a.steps / elapsed
        out = {
            "metric": "segment-integrations/sec (state+costate+STM)" if wl in ("c2", "c4", "hbm") else "segment-integrations/sec",
            "value": value, "unit": "segment-integrations/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            # c4 / c5 shard a FIXED global size (256 levels, 65 536 segments) over the ranks; the others give every rank its own batch
            "scaling": "strong" if (wl in ("c4", "c5", "c5_stm") and not a.segments) else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "workload_token": w.token, "segments_per_gpu": S, "global_segments": world * S,
                       # collective_token: none | windows | rccl | torch;  stream: main | side;  the long forms follow
                       "collective_token": (transport if use_coll else "none"), "stream": (("main" if same_stream else "side") if use_coll else None),
                       "devices_token": (("shared" if share and world > 1 else "distinct") if use_coll else None),
                       "slab_ok": (True if use_coll else None),          # asserted above on every rank: own slab bit-equal, every slab finite
                       "transports": transport_trial, "rccl_ranks": rccl_ranks, "policy": policy_trial if policy_trial else (policy if use_coll else None),
                       "slab_check": ("passed on every rank: own slab of the gathered defect vector bit-equal to the sweep's output, every slab finite"
                                      if use_coll else "no collective"),
                       "devices": ("all %d ranks share device 0 (LTO_BENCH_SHARE_DEVICE=1): a functional run of the N > 1 path, not a scaling figure" % world
                                   if share and world > 1 else "one device per rank"),
                       "collective": ("none" if not use_coll else
                                      "lto_comm_allgather_dev of the defect slabs after every sweep, on %s; transport: %s%s" % (
                                          "the sweep's stream" if same_stream else ("a side stream, %s" % ("overlapping the next sweep" if policy == "overlap" else "the next sweep waiting for it")),
                                          {"windows": "IPC receive windows + device copies + flag kernel (no RCCL kernel, no CU taken from the sweep)",
                                           "rccl": "ncclAllGather (RCCL over xGMI)"}[transport],
                                          "" if not native_note else " [set-up notes: %s]" % native_note)
                                      if native is not None else
                                      "torch.distributed all_gather_into_tensor (RCCL) after every sweep; the library's "
                                      "communicator was not used: %s" % native_note), "integrator": "see workload"},
        }
        if cold is not None:
            out["cold_clocks"] = cold
            out["device_warmup_ms"] = a.device_warmup_ms
        if wl in ("c2", "hbm", "c4", "c5_stm"):
            out["config"]["stm_kernel"] = plan.last_kernel()
        if (wl, a.ndim) in WORK and not a.method:
            out["roofline"] = roofline(wl, a.ndim, S, kern_ms, samples=samples, burst_n=burst_n)
        elif wl == "c2" and a.method == "dop853":
            # the reference's own integrator setting (adaptive order 8 @ 1e-13 + STM): flops from the step counts of this sweep
            acc, rej = plan.step_counts(stream=st)
            trial = float((acc + rej).sum())
            f_rhs, dim, nbytes = (1490, 210, 1920) if a.ndim == 14 else (1070, 156, 1456)
            out["roofline"] = roofline(wl, a.ndim, S, kern_ms, work=(dop853_flops(trial, f_rhs, dim) / S, nbytes), samples=samples, burst_n=burst_n, method="dop853")
            out["roofline"]["flops_from"] = ("measured step counts of this sweep: %.2f accepted + %.2f rejected trial steps per segment (max %d) x "
                                             "(12 x %d + 148 x %d)" % (acc.mean(), rej.mean(), int((acc + rej).max()), f_rhs, dim))
        if c5:
            # wavefront-divergence / load-balance study: a wave runs until its slowest lane has finished
            acc, rej = plan.step_counts(stream=st)
            tot = (acc + rej).astype(np.float64)
            grp = 16 if wl == "c5_stm" else 64          # cooperative STM kernel: 16 segments share a workgroup's step loop
            pad = (-len(tot)) % grp

            def eff(v):
                w = np.concatenate([v, np.zeros(pad)]).reshape(-1, grp)
                return float(v.sum() / (w.max(axis=1).sum() * grp))
            # work actually done: accepted + rejected trial steps of every segment (DOP853: 12 RHS evaluations per trial step)
            f_rhs, dim = (1070, 156) if wl == "c5_stm" else (95, 12)
            work = (dop853_flops(float(tot.sum()), f_rhs, dim) / S, 1456 if wl == "c5_stm" else 304)
            out["roofline"] = roofline(wl, 12, S, kern_ms, work=work, samples=samples, burst_n=burst_n)
            out["roofline"]["flops_from"] = "measured step counts of this sweep: %.2f trial steps per segment x (12 x %d + 148 x %d)" % (tot.mean(), f_rhs, dim)
            out["adaptive"] = {"steps_accepted_mean": float(acc.mean()), "steps_accepted_max": int(acc.max()),
                               "steps_rejected_mean": float(rej.mean()), "steps_rejected_max": int(rej.max()),
                               "wavefront_efficiency_natural_order": eff(tot),
                               "wavefront_efficiency": eff(np.sort(tot)[::-1]) if rebalanced else eff(tot),
                               "rebalanced": rebalanced,
                               "group": grp,
                               "note": "efficiency = segment-steps executed / (group x slowest segment per wavefront / workgroup); rebalanced = "
                                       "lto_indirect_plan_rebalance ordered the lanes by the warm-up sweep's step counts"}
        if world == 1 and not a.no_cpu_baseline and wl in ("c2", "c3", "c2_defect") and not a.method:
            out["cpu_baseline"] = cpu_baseline("c3" if wl == "c3" else "c2", a.cpu_seconds, ndim=a.ndim)
            if wl == "c2":
                # the metric's second half: defect L2 error of this very run against the oracle (checker), 256-segment sample
                out["parity"] = parity_vs_oracle(a.ndim, XC, T, defect, Phi, S)
            if wl == "c2":
                # the same sweep the way the reference executes it (adaptive order 8 @ 1e-13 + dual numbers, 12-dim)
                r = optional("cpu_ref_alg", lambda: cpu_baseline("c2", max(min(3.0, a.cpu_seconds), a.cpu_seconds / 2), ndim=12, reference_algorithm=True))
                if r:
                    out["cpu_baseline_reference_algorithm"] = r
            ncpu = os.cpu_count() or 1
            if ncpu > 1:   # same restatement with the segment loop spread over every host core (reported, not the target)
                r = optional("cpu_all_cores", lambda: cpu_baseline("c3" if wl == "c3" else "c2", max(min(3.0, a.cpu_seconds), a.cpu_seconds / 3), threads=ncpu, ndim=a.ndim))
                if r:
                    out["cpu_baseline_all_cores"] = r
        if world == 1 and not a.no_cpu_baseline and not a.method and (wl in ("c3", "c4", "c5", "c5_stm") or (wl == "hbm" and a.ndim == 12)):
            # the single-workload lines carry what their `configs` legs carry: an oracle sample of this run's own last sweep and the
            # oracle timed on one core on that sample
            par, cpu = config_parity_and_cpu(w, lto, max(min(1.0, a.cpu_seconds), a.cpu_seconds / 4))
            out["parity"] = par
            if wl == "c3":
                out.setdefault("cpu_baseline", cpu)         # (the leg above timed the same thing on a larger sample)
            else:
                out["cpu_baseline"] = cpu
        if world == 1 and wl == "c2" and not a.method and not a.segments:
            r = optional("host_api", lambda: leg_host_api(lto, ctx, XC, T, prm, integ, a.ndim, S))
            if r:
                out["host_api"] = r
        want_live = a.live_traffic == "on" or (a.live_traffic == "auto" and not a.no_cpu_baseline)
        if world == 1 and want_live and "roofline" in out and wl in ("c2", "c2_defect", "c3", "c4", "hbm"):   # (the C5 sweeps order their lanes first)
            argv_wl = ["--workload", wl, "--ndim", str(a.ndim)]
            if a.segments: argv_wl += ["--segments", str(a.segments)]
            if a.method: argv_wl += ["--method", a.method]
            if a.kernel: argv_wl += ["--kernel", str(a.kernel)]
            if a.cols: argv_wl += ["--cols", str(a.cols)]
            if a.no_rebalance: argv_wl += ["--no-rebalance"]
            live, how = optional("live_traffic", lambda: live_traffic(argv_wl), (None, "skipped or failed"))   # an extra: never lose the line to it
            rf = out["roofline"]
            if live is not None:
                rf["traffic_stored"], rf["traffic_stored_from"] = rf.get("traffic"), rf.get("traffic_from")
                rf["traffic"], rf["traffic_from"] = live, how
            else:
                rf["traffic_live"] = "not measured in this run (%s): the stored profile's figure stands" % how
        if world == 1 and wl == "c2" and a.ndim == 14 and not a.method and not a.segments:
            # what a user of the reference's driver waits for per iteration
            r = optional("newton", lambda: leg_newton(lto, synth, ctx, st, torch, [29, 4096], 0.0 if a.no_cpu_baseline else max(min(2.0, a.cpu_seconds), a.cpu_seconds / 4)))
            if r:
                out["newton_iteration"] = r
        if world == 1 and wl == "c2" and a.ndim == 14 and not a.method and not a.segments and not a.no_configs:
            # BASELINE configs[2..4] and the HBM evidence point, each at its full single-GPU size, in this same line
            plan.close(); plan = None
            cfgs = {}
            # configs[0]: the demo's size, where a sweep is pure latency
            cfgs["c1"] = optional("c1", lambda: leg_c1(lto, synth, torch, ctx, st, 0.0 if a.no_cpu_baseline else max(min(1.0, a.cpu_seconds), a.cpu_seconds / 12)))
            for key, cwl, ksteps, kwarm in CONFIG_LEGS:      # one leg's failure is reported in its place, the line is kept
                cfgs[key] = optional(key, lambda: leg_config(key, cwl, ksteps, kwarm, lto, synth, torch, ctx, st, dev, a.device_warmup_ms,
                                                             0.0 if a.no_cpu_baseline else max(min(1.0, a.cpu_seconds), a.cpu_seconds / 6)))
                torch.cuda.empty_cache()
            out["configs"] = {k: (v if v is not None else {"error": leg_errors.get(k, "failed")}) for k, v in cfgs.items()}
        for name, leg in (("reference_system_12dim", ref12), ("reference_integrator", refint)):
            if leg is not None:
                out[name] = leg[0]
                if not a.no_cpu_baseline:
                    optional(name + "_parity", leg[1])
        if leg_errors:
            out["leg_errors"] = leg_errors
        exit_code = emit(out, a)
    if use_coll:
        dist.barrier()              # nobody unmaps a window a peer may still be pushing into
    for c in comms.values():
        c.close()
    if use_coll:
        dist.destroy_process_group()
    if plan is not None:
        plan.close()   # plans before their context (lto_destroy frees what lto_*_plan_destroy touches)
    ctx.close()
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
