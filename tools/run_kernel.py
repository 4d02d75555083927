#!/usr/bin/env python3
"""Development aid: launch the STM sweep (RK4 x 64, 4 096 segments, 14-dim unless NDIM=12) N times back to back with one
kernel family (for rocprofv3 runs).  Usage: python tools/run_kernel.py <LTO_KERNEL id> [launches] [role mask]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth

kern = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mask = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ndim = int(os.environ.get("NDIM", "14"))
S = int(os.environ.get("SEGS", "4096"))
ctx = lto.Context(0)
st = lto.current_stream_ptr()
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
plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=64, max_steps=(1 << 20) | mask), ndim=ndim)
plan.set_kernel(kern)
for _ in range(reps):
    plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
torch.cuda.synchronize()
plan.close()
ctx.close()
