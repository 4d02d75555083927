#!/usr/bin/env python3
"""Earth-Moon L2 halo -> halo low-thrust transfer by indirect multiple shooting, on the GPU.

Follows the indirect part of the reference demo (CRTBP_Multishoot_indirect_demo.jl): 30 nodes over 20 days,
trajectory-stacking initial guess (:74-115), p = 2 with adjoints-only iterations first (:178-186), then all
variables (:188-192), then p = 1 at 0.05 N (:240-247) and the rho continuation (:277-281).  The reference first runs
its direct method (JuMP/Ipopt QP, out of scope here) to smooth the stacked guess; this script goes straight to the
indirect method, so it needs a few more Newton iterations.  Every defect / Jacobian / Newton solve runs in
liblto_hip.so; random costates are seeded (the reference's are not).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lowthrustopt_amd as lto  # noqa: E402
from lowthrustopt_amd import drivers, synth  # noqa: E402
from lowthrustopt_amd.constants import MU, DU, TU, day  # noqa: E402


def stacked_guess(n_nodes=30, tof_days=20.0, tau1=0.75):
    """Nodes on halo 1 for the first half of the flight and on halo 2 afterwards (demo :74-115), using periodic
    interpolation of the orbit tables instead of ballistic propagation + cubic splines."""
    tof = tof_days * day / TU
    t = np.linspace(0.0, tof, n_nodes)
    tof1 = tof / 2
    T1, T2 = 99 * synth.HALO_DT[0], 99 * synth.HALO_DT[1]
    X = np.zeros((6, n_nodes))
    first = t < tof1
    X[:, first] = synth.halo_state(0, tau1 * T1 + t[first])
    # closest point of orbit 2 to the end of the first arc
    xe = synth.halo_state(0, tau1 * T1 + tof1)
    taus = np.linspace(0, T2, 2001)
    d = np.linalg.norm(synth.halo_state(1, taus) - xe[:, None], axis=0)
    tau2 = taus[np.argmin(d)]
    X[:, ~first] = synth.halo_state(1, tau2 + (t[~first] - tof1))
    return X, t


def main(seed=0, verbose=True, rho_target=1e-2, python_loop=False):
    ctx = lto.default_context(0)
    # default: every multiShoot_CRTBP_indirect call is ONE library call (lto_indirect_solve: Newton loop, line search
    # and end-state pinning on the device); --python-loop drives the same device operators from the Python mirror of
    # the reference loop.  Integrator: adaptive order-8 pair @1e-13 (the reference's setting).
    ops = drivers.HipOps(ctx) if python_loop else None
    n = 30
    X, t = stacked_guess(n)
    rng = np.random.default_rng(seed)
    XC = np.vstack([X, 0.1 * rng.standard_normal((6, n))])
    XC[:, 1:-1] += 1e-10 * rng.standard_normal((12, n - 2))
    mass = 1e3
    t0 = time.perf_counter()
    # p = 2 (minimum energy), thrust unconstrained: adjoints only, then everything
    XC, defect, flag = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, mass, 10.0, False, True, 10, 2.0, 1.0, ops=ops, verbose=verbose)
    XC, defect, flag = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, mass, 10.0, False, False, 50, 2.0, 1.0, ops=ops, verbose=verbose)
    print("p = 2: status %d, max defect %.2e" % (flag, np.abs(defect).max()))
    res = {"p2": (flag, float(np.abs(defect).max()))}
    if flag == 0:
        # p = 1 (minimum fuel) at 0.05 N, rho = 1, then continuation to rho = 1e-2
        XC1, defect, flag1 = drivers.multiShoot_CRTBP_indirect(XC, t, MU, DU, TU, n, mass, 0.05, False, False, 30, 1.0, 1.0, ops=ops, verbose=verbose)
        print("p = 1, rho = 1: status %d, max defect %.2e" % (flag1, np.abs(defect).max()))
        res["p1"] = (flag1, float(np.abs(defect).max()))
        if flag1 == 0:
            XC2, defect, flag2 = drivers.reduceFuel_indirect(XC1, t, MU, DU, TU, n, mass, 0.05, 1.0, rho_target, ops=ops, verbose=verbose)
            print("rho -> %g: status %d, max defect %.2e" % (rho_target, flag2, np.abs(defect).max()))
            res["rho"] = (flag2, float(np.abs(defect).max()))
            if flag2 == 0:
                XD, td = lto.densify(XC2, t, lto.make_params(MU, DU, TU, 0.05, mass, 1.0, 1.0, rho_target), 300, ctx=ctx)
                lam = np.linalg.norm(XD[9:12], axis=0)
                thr = 0.5 * (1 + np.tanh((lam - 1) / (2 * rho_target))) * 0.05
                print("thrust profile: on %.0f %% of the flight, max %.3f N" % (100 * np.mean(thr > 0.025), thr.max()))
    print("wall time %.2f s" % (time.perf_counter() - t0))
    return res


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    main(rho_target=float(args[0]) if args else 1e-2, verbose="-q" not in sys.argv, python_loop="--python-loop" in sys.argv)
