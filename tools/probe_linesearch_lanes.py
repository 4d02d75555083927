#!/usr/bin/env python3
"""Line-search sweep of the Newton iteration (20 trial trajectories x S segments in one launch, bench.py's converged station-keeping
problem) with 4 / 2 / 1 lanes per segment: in this sweep every segment takes about the same number of trial steps, so the launch
is throughput-bound, unlike the C5 study (wide spread: the slowest segment sets the time) that AUTO's thresholds were taken from."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
import bench


def main():
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    for S in (29, 1024, 4096):
        it = bench.NewtonIteration(lto, synth, ctx, st, torch, S)
        it.iteration()
        torch.cuda.synchronize()
        n, NA = it.n, bench.NEWTON_ALPHAS
        for lanes in (0, 4, 2, 1):
            it.plan_ls.set_defect_lanes(lanes)
            for _ in range(5):
                it.plan_ls.defect(it.Xt, n * NA, it.t, 1, it.dt, S * NA, stream=st)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50):
                it.plan_ls.defect(it.Xt, n * NA, it.t, 1, it.dt, S * NA, stream=st)
            torch.cuda.synchronize()
            us = (time.perf_counter() - t0) / 50 * 1e6
            acc, rej = it.plan_ls.step_counts(stream=st)
            print("S=%5d x 20 trial trajectories, lanes %d: %.1f us per sweep; trial steps mean %.2f max %d" % (S, lanes, us, (acc + rej).mean(), (acc + rej).max()), flush=True)
        it.plan_ls.set_defect_lanes(0)
        if S * NA >= 16384:                    # the same sweep with the lanes ordered by the previous sweep's step counts
            it.plan_ls.defect(it.Xt, n * NA, it.t, 1, it.dt, S * NA, stream=st)
            ref = it.dt.clone()
            it.plan_ls.rebalance(stream=st)
            for lanes in (0, 4, 2, 1):
                it.plan_ls.set_defect_lanes(lanes)
                for _ in range(5):
                    it.plan_ls.defect(it.Xt, n * NA, it.t, 1, it.dt, S * NA, stream=st)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(50):
                    it.plan_ls.defect(it.Xt, n * NA, it.t, 1, it.dt, S * NA, stream=st)
                torch.cuda.synchronize()
                us = (time.perf_counter() - t0) / 50 * 1e6
                print("S=%5d x 20 REBALANCED, lanes %d: %.1f us per sweep; max |d - natural order| %.2e" % (S, lanes, us, float((it.dt - ref).abs().max())), flush=True)
        for lanes in (0, 4, 2, 1):
            it.plan.set_defect_lanes(lanes)
            for _ in range(5):
                it.plan.defect(it.X, n, it.t, 1, it.d3, S, stream=st)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50):
                it.plan.defect(it.X, n, it.t, 1, it.d3, S, stream=st)
            torch.cuda.synchronize()
            print("S=%5d single trajectory, lanes %d: %.1f us per sweep" % (S, lanes, (time.perf_counter() - t0) / 50 * 1e6), flush=True)
        it.close()
    ctx.close()


if __name__ == "__main__":
    main()
