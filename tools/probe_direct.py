#!/usr/bin/env python3
"""Direct Jacobian sweep (RKF7(8), nsteps = 10, 6-state): per-lane against the pipelined kernel over a range of segment counts
(where should AUTO switch?)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    for nstate in (6, 7):
        for S in [int(x) for x in os.environ.get("SEGS", "29,512,1024,2048,3072,4096,8192,16384,65536").split(",")]:
            n = S + 1
            Xd, Ud, Td = synth.direct_problem(n, nstate=nstate)
            X = torch.from_numpy(synth.to_soa_nodes(Xd)).cuda(); U = torch.from_numpy(synth.to_soa_nodes(Ud)).cuda()
            t = torch.from_numpy(np.ascontiguousarray(Td[:, 0])).cuda()
            nvar = 2 * (nstate + 3)
            defect = torch.zeros(nstate, S, dtype=torch.float64, device="cuda"); err = torch.zeros(S, dtype=torch.float64, device="cuda")
            Jac = torch.zeros(nstate * nvar, S, dtype=torch.float64, device="cuda"); dtf = torch.zeros(nstate, S, dtype=torch.float64, device="cuda")
            plan = lto.DirectPlan(ctx, nstate, n, 1, 10, lto.MU, lto.DU, lto.TU, 2000.0)
            res = []
            for kern, name in ((1, "per-lane"), (3, "pipeline")):
                plan.set_kernel(kern)
                us = timeit(lambda: plan.jacobian(X, n, U, n, t, 1, Jac, S, dtf, defect, S, err, stream=st), iters=50 if S <= 16384 else 15)
                res.append("%s %8.1f us" % (name, us))
            print("nstate=%d S=%6d  " % (nstate, S) + "  ".join(res), flush=True)
            plan.close()
    ctx.close()


if __name__ == "__main__":
    main()
