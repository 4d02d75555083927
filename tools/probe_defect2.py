#!/usr/bin/env python3
"""Defect-only sweep with the reference's integrator setting (DOP853 @ 1e-13, 12-dim): one lane per segment
against two and four lanes per segment (lto_indirect_plan_set_defect_lanes), ordered lanes above 8 192 segments."""
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
    for S in [int(x) for x in os.environ.get("SEGS", "29,580,4096,16384,65536,131072,262144").split(",")]:
        n = S + 1
        big = S >= 16384
        XC, T = synth.indirect_problem(n, seed=5, dt_range=(0.05, 0.5)) if big else synth.indirect_problem(n)
        prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1e-3 if big else 1.0)
        X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
        out = {}
        for name, lanes in (("one lane", 1), ("two lanes", 2), ("four lanes", 4)):
            plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator())
            plan.set_defect_lanes(lanes)
            d = torch.zeros(12, S, dtype=torch.float64, device="cuda")
            for _ in range(3):
                plan.defect(X, n, t, 1, d, S, stream=st)
            if S > 8192:
                plan.rebalance(stream=st)
            reps = 200 if S <= 4096 else 30
            for _ in range(reps // 4):
                plan.defect(X, n, t, 1, d, S, stream=st)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                plan.defect(X, n, t, 1, d, S, stream=st)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3
            acc, rej = plan.step_counts(stream=st)
            out[name] = (d.cpu().numpy(), acc, rej)
            print("S=%7d %-9s %.4f ms per sweep (%.3e seg/s); trial steps %.2f per segment (max %d)" % (
                S, name, ms, S / ms * 1e3, (acc + rej).mean(), (acc + rej).max()), flush=True)
            plan.close()
        d1, a1, r1 = out["one lane"]
        for name in ("two lanes", "four lanes"):
            d2, a2, r2 = out[name]
            print("   %-10s vs one lane: max |ddefect| = %.2e; step counts equal: %s" % (name, np.abs(d1 - d2).max(), bool(np.array_equal(a1, a2) and np.array_equal(r1, r2))), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
