#!/usr/bin/env python3
"""Can the library's Newton loop converge (a) the synthetic stacked-halo guess the way the reference's demo does (p = 2, thrust 10 N:
adjoints only first, then all unknowns -- CRTBP_Multishoot_indirect_demo.jl:178-196) and (b) a station-keeping problem: every node on
ONE halo orbit (table interpolation, 1e-3 off the true orbit), costates ~ 0?  Prints status, iterations and the history of
max |defect|: bench.py's Newton-iteration leg linearises about such a converged point."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth


def main():
    ctx = lto.Context(0)
    for S in [int(x) for x in os.environ.get("SEGS", "29,4096").split(",")]:
        n = S + 1
        for kind, sigma in (("stacked", 0.1), ("one_halo", 1e-6), ("one_halo", 1e-2)):
            XC, T = synth.indirect_problem(n, lam_sigma=sigma)
            XC, t = np.asfortranarray(XC[:, :, 0]), np.ascontiguousarray(T[:, 0])
            if kind == "one_halo":
                XC[:6] = synth.halo_state(0, 0.75 * 99 * synth.HALO_DT[0] + t)
            prm = lto.make_params(lto.MU, lto.DU, lto.TU, 10.0, 1000.0, 1.0, 2.0, 1.0)
            t0 = time.perf_counter()
            if kind == "stacked":
                X1, d1, s1, it1, h1 = lto.indirect_solve(XC, t, prm, None, True, 10, ctx=ctx)
                XC = X1 if s1 != 2 else XC
            X2, d2, s2, it2, h2 = lto.indirect_solve(XC, t, prm, None, False, 30, ctx=ctx)
            el = time.perf_counter() - t0
            print("S=%5d %s sigma=%g: status %d after %d it; history %s; max |costate| %.2e; %.2f s" % (
                S, kind, sigma, s2, it2, " ".join("%.1e" % v for v in h2[:, 0]), np.abs(X2[6:]).max(), el), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
