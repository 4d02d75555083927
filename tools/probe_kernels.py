#!/usr/bin/env python3
"""STM sweep, fixed-step RK4 x 64: per-lane vs cooperative vs three-role pipeline kernel, 12- and 14-dim, over a range
of segment counts (development aid; bench.py is the contract benchmark)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    sizes = [int(x) for x in sys.argv[1:]] or [29, 1024, 4096, 8192, 16384, 65536]
    for ndim in (14, 12):
        for S in sizes:
            n = S + 1
            XC, T = synth.indirect_problem(n)
            if ndim == 14:
                Xh = np.zeros((14, n, 1), order="F")
                Xh[:6] = XC[:6]; Xh[6] = 1000.0; Xh[7:13] = XC[6:]; Xh[13] = 0.2
                slot = 2000.0
            else:
                Xh, slot = XC, 1000.0
            prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, slot, 1.0, 1.0, 1.0)
            X = torch.from_numpy(synth.to_soa_nodes(Xh)).cuda()
            t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
            d = torch.zeros(ndim, S, dtype=torch.float64, device="cuda")
            Phi = torch.zeros(ndim * ndim, S, dtype=torch.float64, device="cuda")
            plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=64), ndim=ndim)
            res = []
            for kern, cols, name in ((1, 1, "lane1"), (1, 2, "lane2"), (1, 3, "lane3"), (2, 0, "coop"), (5, 0, "pipe8"), (8, 0, "pipe32"), (7, 0, "pipe48")):
                if (ndim == 14 and cols == 3) or (ndim == 12 and cols == 2):
                    continue
                plan.set_kernel(kern)
                plan.set_cols_per_lane(cols)
                ms = timeit(lambda: plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st), iters=30 if S <= 16384 else 8)
                res.append("%s %8.1f us" % (name, ms * 1e3))
            print("ndim=%d S=%6d  " % (ndim, S) + "  ".join(res), flush=True)
            plan.close()


if __name__ == "__main__":
    main()
