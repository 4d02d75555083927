#!/usr/bin/env python3
"""Contract sweep (k_indirect_pipe8, 4 096 segments, RK4 x 64) by control-law class: what the base wave's control-law chain costs."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
from probe_kernels import timeit

ctx = lto.Context(0)
st = lto.current_stream_ptr()
S = 4096
n = S + 1
for ndim in (14, 12):
    XC, T = synth.indirect_problem(n)
    if ndim == 14:
        Xh = np.zeros((14, n, 1), order="F")
        Xh[:6] = XC[:6]; Xh[6] = 1000.0; Xh[7:13] = XC[6:]; Xh[13] = 0.2
        slot = 2000.0
    else:
        Xh, slot = XC, 1000.0
    X = torch.from_numpy(synth.to_soa_nodes(Xh)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
    d = torch.zeros(ndim, S, dtype=torch.float64, device="cuda")
    Phi = torch.zeros(ndim * ndim, S, dtype=torch.float64, device="cuda")
    for p, thrust in ((0.0, 0.05), (1.0, 0.05), (2.0, 10.0), (1.5, 0.05)):
        prm = lto.make_params(lto.MU, lto.DU, lto.TU, thrust, slot, 1.0, p, 1.0)
        plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=64), ndim=ndim)
        plan.set_kernel(5)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            for _ in range(8):
                plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
            torch.cuda.synchronize()
        ms = timeit(lambda: plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st), iters=200, warm=20)
        print("ndim=%d p=%.1f  pipe8 %.2f us" % (ndim, p, ms * 1e3), flush=True)
        plan.close()
