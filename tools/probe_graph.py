#!/usr/bin/env python3
"""One Newton iteration of the indirect method on the device -- STM sweep with the reference's integrator setting, block-bidiagonal
solve, update x <- x + delta, defect sweep at the new point -- enqueued launch by launch against replayed as ONE HIP graph captured
from the same calls (the device-resident entry points are stream-ordered and allocate nothing after their first call, so a caller may
capture them): time per iteration at the demo size and at the contract size."""
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
    s = torch.cuda.Stream()
    sp = s.cuda_stream
    for S in [int(x) for x in os.environ.get("SEGS", "29,4096").split(",")]:
        n = S + 1
        XC, T = synth.indirect_problem(n)
        prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
        X0 = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
        plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator())
        X = X0.clone()
        Xn = torch.zeros_like(X)
        d = torch.zeros(12, S, dtype=torch.float64, device="cuda")
        d2 = torch.zeros(12, S, dtype=torch.float64, device="cuda")
        Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
        delta = torch.zeros(12, n, dtype=torch.float64, device="cuda")

        def iteration():
            plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=sp)
            plan.newton_solve(Phi, S, d, S, delta, n, stream=sp)
            ctx.check(ctx.lib.lto_axpy_dev(ctx.handle, sp, X.data_ptr(), delta.data_ptr(), 1.0, Xn.data_ptr(), 12 * n))
            plan.defect(Xn, n, t, 1, d2, S, stream=sp)

        with torch.cuda.stream(s):
            for _ in range(3):
                iteration()            # first calls allocate the plan's scratch: never inside a capture
            s.synchronize()
            ref = (Xn.clone(), d2.clone())
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                iteration()
            Xn.zero_(); d2.zero_()
            g.replay(); s.synchronize()
            same = bool(torch.equal(Xn, ref[0]) and torch.equal(d2, ref[1]))
            reps = 200
            for _ in range(20):
                iteration()
            s.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                iteration()
            s.synchronize(); t_plain = (time.perf_counter() - t0) / reps * 1e6
            for _ in range(20):
                g.replay()
            s.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            s.synchronize(); t_graph = (time.perf_counter() - t0) / reps * 1e6
        print("S=%5d  Newton iteration (STM sweep + solve + update + defect sweep): %.1f us launch by launch, %.1f us as one graph; "
              "replay equals direct calls bit for bit: %s" % (S, t_plain, t_graph, same), flush=True)
        plan.close()
    ctx.close()


if __name__ == "__main__":
    main()
