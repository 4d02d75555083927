#!/usr/bin/env python3
"""Development aid: run the probe build of the pipeline kernel with one role mask many times (under rocprofv3 --pmc
GRBM_GUI_ACTIVE --kernel-trace the ratio cycles / duration is the shader clock during that configuration)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth

mask = int(sys.argv[1]); kern = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = lto.Context(0)
st = lto.current_stream_ptr()
S = 4096; n = S + 1; ndim = 14
XC, T = synth.indirect_problem(n)
Xh = np.zeros((14, n, 1), order="F")
Xh[:6] = XC[:6]; Xh[6] = 1000.0; Xh[7:13] = XC[6:]; Xh[13] = 0.2
prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 2000.0, 1.0, 1.0, 1.0)
X = torch.from_numpy(synth.to_soa_nodes(Xh)).cuda()
t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
d = torch.zeros(ndim, S, dtype=torch.float64, device="cuda")
Phi = torch.zeros(ndim * ndim, S, dtype=torch.float64, device="cuda")
plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=64, max_steps=(1 << 20) | mask), ndim=ndim)
plan.set_kernel(kern)
for _ in range(60):
    plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
torch.cuda.synchronize()
