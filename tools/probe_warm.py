#!/usr/bin/env python3
"""Development aid: the reference's integrator setting (12-dim, adaptive order 8 @ 1e-13) with and without the warm start of the
step-size controller (lto_indirect_plan_set_warm_start): STM sweep and defect-only sweep, step counts and time per sweep."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
from probe_kernels import timeit

ctx = lto.Context(0)
st = lto.current_stream_ptr()
for S, kw, rho in ((4096, {}, 1.0), (29, {}, 1.0), (65536, {"dt_range": (0.05, 0.5)}, 1e-3)):
    n = S + 1
    XC, T = synth.indirect_problem(n, seed=0 if S != 65536 else 1, **kw)
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, rho)
    X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
    ref = None
    for warm in (False, True):
        plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator())
        plan.set_warm_start(warm)
        Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
        d = torch.zeros(12, S, dtype=torch.float64, device="cuda")
        d0 = torch.zeros(12, S, dtype=torch.float64, device="cuda")
        if S > 8192:
            plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st); plan.rebalance(stream=st)
        ms_j = timeit(lambda: plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st), iters=30 if S <= 8192 else 5)
        acc, rej = plan.step_counts(stream=st)
        ms_d = timeit(lambda: plan.defect(X, n, t, 1, d0, S, stream=st), iters=30 if S <= 8192 else 5)
        acc0, rej0 = plan.step_counts(stream=st)
        out = (Phi.cpu().numpy(), d.cpu().numpy(), d0.cpu().numpy())
        if ref is None:
            ref = out
        print("S=%6d warm=%d  STM %8.1f us  trials mean %.2f max %d (rej %.2f) | defect %7.1f us trials mean %.2f max %d | vs cold: dPhi %.1e dd %.1e dd0 %.1e" % (
            S, warm, ms_j * 1e3, (acc + rej).mean(), (acc + rej).max(), rej.mean(), ms_d * 1e3, (acc0 + rej0).mean(), (acc0 + rej0).max(),
            np.abs(out[0] - ref[0]).max() / np.abs(ref[0]).max(), np.abs(out[1] - ref[1]).max(), np.abs(out[2] - ref[2]).max()), flush=True)
        plan.close()
