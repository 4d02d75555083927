#!/usr/bin/env python3
"""Development aid: which role of the pipeline kernel bounds a phase?  Needs the -DPIPE_PROBE build of
kernels_indirect_pipe.hip (LTO_HIP_LIB=build/liblto_probe.so); max_steps carries the role mask."""
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
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, slot, 1.0, 1.0, 1.0)
    X = torch.from_numpy(synth.to_soa_nodes(Xh)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
    d = torch.zeros(20, S, dtype=torch.float64, device="cuda")   # rows 16..18: probe build diagnostics
    Phi = torch.zeros(ndim * ndim, S, dtype=torch.float64, device="cuda")
    for mask, name in [(int(m), "mask %s" % m) for m in os.environ.get("MASKS", "0,6,5,3,7,4,1").split(",")]:
        plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=64, max_steps=(1 << 20) | mask), ndim=ndim)
        plan.set_kernel(int(os.environ.get("KERNEL", "3")))
        ms = timeit(lambda: plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st), iters=int(os.environ.get('ITERS', '30')), warm=int(os.environ.get('WARM', '5')))
        dh = d.cpu().numpy()
        cyc, wall = dh[17, ::16], dh[18, ::16]
        clk = " loop %.1f us, %.0f kcycles, shader clock %.3f GHz" % (np.median(wall) / 100.0, np.median(cyc) / 1e3, np.median(cyc / wall) * 0.1) if np.median(wall) > 100 else ""
        if os.environ.get("KERNEL", "3") == "5":     # pipe8 probe build: cycles every wave waited at the phase barriers
            w = dh[16].reshape(-1, 16)[:, :8]
            clk += "  barrier wait kcycles by wave " + " ".join("%.0f" % (np.median(w[:, i]) / 1e3) for i in range(8))
        print("ndim=%d  %-14s %8.1f us%s" % (ndim, name, ms * 1e3, clk), flush=True)
        plan.close()
