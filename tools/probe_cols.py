#!/usr/bin/env python3
"""Per-lane RK4 STM kernel (few steps per segment): columns per lane 1 / 2 / 3 against AUTO, 12-dim, over segment counts
(development aid: decides which column groupings the library keeps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
from probe_kernels import timeit


def main():
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    steps = int(os.environ.get("STEPS", "4"))
    ndim = int(os.environ.get("NDIM", "12"))
    sizes = [int(x) for x in sys.argv[1:]] or [1024, 4096, 8192, 12288, 16384, 24576, 32768, 45056, 65536]
    for S in sizes:
        n = S + 1
        XC, T = synth.indirect_problem(n)
        slot = 1000.0
        if ndim == 14:
            Xh = np.zeros((14, n, 1), order="F")
            Xh[:6] = XC[:6]; Xh[6] = 1000.0; Xh[7:13] = XC[6:]; Xh[13] = 0.2
            XC, slot = Xh, 2000.0
        prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, slot, 1.0, 1.0, 1.0)
        X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
        d = torch.zeros(ndim, S, dtype=torch.float64, device="cuda")
        Phi = torch.zeros(ndim * ndim, S, dtype=torch.float64, device="cuda")
        plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=steps), ndim=ndim)
        res = []
        for kern, cols, name in ((1, 1, "cols1"), (1, 2, "cols2"), (1, 3, "cols3"), (0, 0, "auto")):   # cols2: refused since the form was removed (shown as --)
            plan.set_kernel(kern)
            try:
                plan.set_cols_per_lane(cols)
                ms = timeit(lambda: plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st), iters=30)
                res.append("%s %7.1f us" % (name, ms * 1e3))
            except Exception as e:
                res.append("%s -- (%s)" % (name, str(e)[:30]))
        print("ndim=%d steps=%d S=%6d  " % (ndim, steps, S) + "  ".join(res) + "  auto ran " + str(plan.last_kernel()), flush=True)
        plan.close()


if __name__ == "__main__":
    main()
