#!/usr/bin/env python3
"""Wall time of lto_indirect_solve (the whole multiShoot_CRTBP_indirect loop in one call) on bench.py's station-keeping problem,
perturbed enough to need several iterations: total / iterations against the launch-by-launch figure of bench.py's newton_iteration."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
import bench

ctx = lto.Context(0)
for S in (29, 4096):
    Xp, t, prm, info = bench.station_keeping_problem(lto, synth, ctx, S, pert=1e-4)
    for rep in range(3):
        t0 = time.perf_counter()
        Xs, ds, status, iters, hist = lto.indirect_solve(Xp, t, prm, None, False, 12, ctx=ctx)
        wall = time.perf_counter() - t0
    print("S=%5d  status %d  iterations %d  wall %.3f ms  = %.1f us per iteration (incl. staging in / out and the first sweep); max |defect| history %s"
          % (S, status, iters, wall * 1e3, wall * 1e6 / max(iters, 1), " ".join("%.1e" % v for v in hist[:, 0])), flush=True)
ctx.close()
