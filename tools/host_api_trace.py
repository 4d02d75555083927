#!/usr/bin/env python3
"""Page-locked host-pointer Jacobian call at 4 096 segments, 60 calls: run under rocprofv3 --kernel-trace --stats to see
what the call is made of (pack kernel reading the host, sweep, unpack kernels writing the host)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth

S = int(os.environ.get("SEGS", "4096"))
ctx = lto.Context(0)
XC, T = synth.indirect_problem(S + 1)
prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
integ = lto.integrator(lto.RK4, steps=64)
X = ctx.pinned_empty((12, S + 1)); X[:] = XC[:, :, 0]
t = ctx.pinned_empty((S + 1,)); t[:] = T[:, 0]
out = (ctx.pinned_empty((12, 12, S, 1)), ctx.pinned_empty((12, S, 1)))
for _ in range(10):
    lto.indirect_stm(X, t, prm, integ, ctx=ctx, out=out)
t0 = time.perf_counter()
for _ in range(50):
    lto.indirect_stm(X, t, prm, integ, ctx=ctx, out=out)
print("ms per call %.4f" % ((time.perf_counter() - t0) / 50 * 1e3))
ctx.close()
