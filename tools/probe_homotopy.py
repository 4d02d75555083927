#!/usr/bin/env python3
"""Concurrent vs sequential rho continuation on the demo problem (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib.util
import numpy as np
import lowthrustopt_amd as lto
from lowthrustopt_amd import drivers
from lowthrustopt_amd.constants import MU, DU, TU
spec = importlib.util.spec_from_file_location("demo", os.path.join(os.path.dirname(__file__), "..", "examples", "halo_transfer_demo.py"))
demo = importlib.util.module_from_spec(spec); spec.loader.exec_module(demo)
n = 30
X, t = demo.stacked_guess(n)
rng = np.random.default_rng(0)
XC = np.vstack([X, 0.1 * rng.standard_normal((6, n))])
XC[:, 1:-1] += 1e-10 * rng.standard_normal((12, n - 2))
XC, _, f = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, 1e3, 10.0, False, True, 10, 2.0, 1.0, verbose=False)
XC, _, f = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, 1e3, 10.0, False, False, 50, 2.0, 1.0, verbose=False)
XC1, _, f1 = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, 1e3, 0.05, False, False, 30, 1.0, 1.0, verbose=False)
print("base solves:", f, f1)
for levels in (7, 14, 28):
    rhos = np.geomspace(0.5, 1e-4 if levels > 7 else 2.0 ** -7, levels)
    for rep in range(2):
        t0 = time.perf_counter()
        Xl, Dl, st, waves = drivers.homotopy_solve(XC1, t, MU, DU, TU, 1e3, 0.05, rhos, verbose=False)
        dt = time.perf_counter() - t0
    print("concurrent: %2d levels down to rho = %.1e: %d waves, %d converged, %.1f ms" % (levels, rhos[-1], waves, int((st == 0).sum()), dt * 1e3))
    for rep in range(2):
        t0 = time.perf_counter()
        Xs, ds, fs = drivers.reduceFuel_indirect(XC1, t, MU, DU, TU, n, 1e3, 0.05, 1.0, float(rhos[-1]), verbose=False)
        dt = time.perf_counter() - t0
    print("sequential reduceFuel_indirect to rho = %.1e: status %d, %.1f ms; |X_conc - X_seq| = %.1e" % (
        rhos[-1], fs, dt * 1e3, np.abs(Xl[:, :, -1] - Xs).max() if st[-1] == 0 else float("nan")))
