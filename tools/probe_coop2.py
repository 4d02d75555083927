#!/usr/bin/env python3
"""Cooperative kernels with the reference's integrator setting (DOP853, rtol = atol = 1e-13, + STM): one-piece lanes
(LTO_KERNEL_COOP) against two lanes per state (LTO_KERNEL_COOP2): results, step counts, time per sweep.  Round 6: the 12-dim one-piece
form is gone (its selector runs the two-lane kernel), so the comparison is made on the 14-dim system (NDIM=14, the default), where both
exist for the always-thrust-limited laws; NDIM=12 times the two-lane kernel alone."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth


def main():
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    ndim = int(os.environ.get("NDIM", "14"))
    for S in [int(x) for x in os.environ.get("SEGS", "29,4096,65536").split(",")]:
        n = S + 1
        if S == 65536:
            XC, T = synth.indirect_problem(n, seed=5, dt_range=(0.05, 0.5)); rho = 1e-3
        else:
            XC, T = synth.indirect_problem(n); rho = 1.0
        prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 2000.0 if ndim == 14 else 1000.0, 1.0, 1.0, rho)
        if ndim == 14:
            X14 = np.zeros((14,) + XC.shape[1:], order="F")
            X14[:6] = XC[:6]; X14[6] = 1000.0; X14[7:13] = XC[6:]; X14[13] = 0.1
            XC = X14
        X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
        res = {}
        for name, kern in (("coop", 2), ("coop2", 6)):
            plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(), ndim=ndim)
            plan.set_kernel(kern)
            d = torch.zeros(ndim, S, dtype=torch.float64, device="cuda")
            Phi = torch.zeros(ndim * ndim, S, dtype=torch.float64, device="cuda")
            for _ in range(3):
                plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
            if S > 8192:
                plan.rebalance(stream=st)
            reps = 200 if S <= 4096 else 20
            for _ in range(reps // 4):
                plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            acc, rej = plan.step_counts(stream=st)
            res[name] = (Phi.cpu().numpy(), d.cpu().numpy(), acc.copy(), rej.copy())
            print("ndim %d S=%6d %-6s (ran %s) %.4f ms per sweep; steps accepted %.2f (max %d), rejected %.2f; finite %s" % (
                ndim, S, name, plan.last_kernel(), ms, acc.mean(), acc.max(), rej.mean(), bool(np.isfinite(res[name][0]).all())), flush=True)
            plan.close()
        P1, d1, a1, r1 = res["coop"]; P2, d2, a2, r2 = res["coop2"]
        print("   coop2 vs coop: max |dPhi| / max |Phi| = %.2e, max |ddefect| = %.2e, step counts equal: %s" % (
            np.abs(P1 - P2).max() / np.abs(P1).max(), np.abs(d1 - d2).max(), bool(np.array_equal(a1, a2) and np.array_equal(r1, r2))), flush=True)
    if os.environ.get("LTO_HIP_LIB", "").endswith("liblto_probe.so") and ndim == 12:
        # probe build: barrier-wait and loop ticks per role (rows 16-23 of a 24-row defect buffer), 4 096 segments
        S = 4096; n = S + 1
        XC, T = synth.indirect_problem(n)
        prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
        X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
        plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator()); plan.set_kernel(6)
        d = torch.zeros(24, S, dtype=torch.float64, device="cuda")
        Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
        for _ in range(50):
            plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
        torch.cuda.synchronize()
        r = d.cpu().numpy()[:, ::16]
        trials = r[18]
        w = np.argmax(trials)
        print("probe: trials per workgroup mean %.1f max %d" % (trials.mean(), trials.max()))
        for name, r0 in (("base", 16), ("top wave 0", 20), ("bottom wave 0", 22)):
            print("  %-14s loop ticks per trial %.0f (slowest workgroup %.0f), of which waiting at barriers %.0f (%.0f)" % (
                name, (r[r0 + 1] / trials).mean(), r[r0 + 1][w] / trials[w], (r[r0] / trials).mean(), r[r0][w] / trials[w]))
        plan.close()
    ctx.close()


if __name__ == "__main__":
    main()
