#!/usr/bin/env python3
"""Quick device-resident timing probe (development aid; bench.py is the contract benchmark)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    print(torch.cuda.get_device_name(0))
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    for S in (4096, 16384, 65536, 262144):
        n = S + 1
        XC, T = synth.indirect_problem(n)
        X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
        defect = torch.zeros(12, S, dtype=torch.float64, device="cuda")
        Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
        for method, steps, name in ((lto.RK4, 64, "rk4x64"), (lto.RKF78_FIXED, 4, "rkf78x4"), (lto.DOP853_ADAPTIVE, 0, "dop853")):
            plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(method, steps=steps))
            ms = timeit(lambda: plan.defect(X, n, t, 1, defect, S, stream=st))
            print("S=%7d %-8s defect        %9.3f ms  %10.3e seg/s" % (S, name, ms, S / ms * 1e3), flush=True)
            plan.set_kernel(2)
            ms = timeit(lambda: plan.jacobian(X, n, t, 1, Phi, S, defect, S, stream=st), iters=5 if S > 20000 else 10)
            print("S=%7d %-8s stm COOP      %9.3f ms  %10.3e seg/s" % (S, name, ms, S / ms * 1e3), flush=True)
            plan.set_kernel(1)
            if method == lto.RK4:
                for cols in (1, 3):
                    plan.set_cols_per_lane(cols)
                    ms = timeit(lambda: plan.jacobian(X, n, t, 1, Phi, S, defect, S, stream=st), iters=10)
                    print("S=%7d %-8s stm cols=%d    %9.3f ms  %10.3e seg/s  (%.2f TFLOP/s model)" % (
                        S, name, cols, ms, S / ms * 1e3, S * 414e3 / ms / 1e9), flush=True)
            elif S <= 65536:
                ms = timeit(lambda: plan.jacobian(X, n, t, 1, Phi, S, defect, S, stream=st), iters=5)
                print("S=%7d %-8s stm           %9.3f ms  %10.3e seg/s" % (S, name, ms, S / ms * 1e3), flush=True)
            plan.close()
    # direct
    for S in (16384, 131072):
        n = S + 1
        Xd, Ud, Td = synth.direct_problem(n)
        X = torch.from_numpy(synth.to_soa_nodes(Xd)).cuda(); U = torch.from_numpy(synth.to_soa_nodes(Ud)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(Td[:, 0])).cuda()
        defect = torch.zeros(6, S, dtype=torch.float64, device="cuda"); err = torch.zeros(S, dtype=torch.float64, device="cuda")
        Jac = torch.zeros(108, S, dtype=torch.float64, device="cuda"); dtf = torch.zeros(6, S, dtype=torch.float64, device="cuda")
        plan = lto.DirectPlan(ctx, 6, n, 1, 10, lto.MU, lto.DU, lto.TU, 2000.0)
        ms = timeit(lambda: plan.defect(X, n, U, n, t, 1, defect, S, err, stream=st))
        print("S=%7d direct defect           %9.3f ms  %10.3e seg/s" % (S, ms, S / ms * 1e3), flush=True)
        for kern, kname in ((1, "per-lane"), (2, "coop")):
            plan.set_kernel(kern)
            ms = timeit(lambda: plan.jacobian(X, n, U, n, t, 1, Jac, S, dtf, defect, S, err, stream=st), iters=10)
            print("S=%7d direct jacobian %-8s %9.3f ms  %10.3e seg/s  (%.2f TFLOP/s model)" % (S, kname, ms, S / ms * 1e3, S * 197e3 / ms / 1e9), flush=True)
        plan.close()


if __name__ == "__main__" and not any(f in sys.argv for f in ("--host", "--newton", "--small", "--solve")):
    main()


def host_api_rate():
    """PCIe-inclusive rate of the host-pointer ABI (what a Julia ccall sees): numpy in, numpy out."""
    import time
    ctx = lto.Context(0)
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    for S in (29, 4096):
        XC, T = synth.indirect_problem(S + 1)
        XC, t = XC[:, :, 0], T[:, 0]
        integ = lto.integrator(lto.RK4, steps=64)
        for name, fn in (("stm", lambda: lto.indirect_stm(XC, t, prm, integ, ctx=ctx)),
                         ("defect", lambda: lto.indirect_defectCalc(XC, t, prm, integ, ctx=ctx))):
            fn(); fn()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            dt = (time.perf_counter() - t0) / 20
            print("host-pointer ABI S=%5d rk4x64 %-6s %9.3f ms per call  %10.3e seg/s (PCIe + pack/unpack + plan inclusive)" % (S, name, dt * 1e3, S / dt), flush=True)


if __name__ == "__main__" and "--host" in sys.argv:
    host_api_rate()


def newton_rate():
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    for S in (29, 4096, 65536):
        n = S + 1
        XC, T = synth.indirect_problem(n)
        X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
        d = torch.zeros(12, S, dtype=torch.float64, device="cuda"); Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
        delta = torch.zeros(12, n, dtype=torch.float64, device="cuda")
        plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=64))
        plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
        ms = timeit(lambda: plan.newton_solve(Phi, S, d, S, delta, n, stream=st), iters=10)
        ms2 = timeit(lambda: plan.newton_solve(None, 0, d, S, delta, n, stream=st), iters=10)
        print("device Newton solve S=%6d: factor+solve %8.3f ms   re-solve (SOC) %8.3f ms" % (S, ms, ms2), flush=True)


if __name__ == "__main__" and "--newton" in sys.argv:
    newton_rate()


def small_sweeps():
    """Time per sweep at the demo size (29 segments) and at 4 096, device-resident."""
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    for S in (29, 4096):
        n = S + 1
        XC, T = synth.indirect_problem(n)
        X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
        t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
        d = torch.zeros(12, S, dtype=torch.float64, device="cuda"); Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
        for method, steps, name in ((lto.RK4, 64, "rk4x64"), (lto.RKF78_FIXED, 4, "rkf78x4"), (lto.DOP853_ADAPTIVE, 0, "dop853@1e-13")):
            plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(method, steps=steps))
            a = timeit(lambda: plan.defect(X, n, t, 1, d, S, stream=st), iters=50) * 1e3
            b = timeit(lambda: plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st), iters=50) * 1e3
            print("S=%5d %-13s defect sweep %7.1f us   STM sweep %7.1f us" % (S, name, a, b), flush=True)
        Xd, Ud, Td = synth.direct_problem(n)
        Xs = torch.from_numpy(synth.to_soa_nodes(Xd)).cuda(); Us = torch.from_numpy(synth.to_soa_nodes(Ud)).cuda()
        td = torch.from_numpy(np.ascontiguousarray(Td[:, 0])).cuda()
        dd = torch.zeros(6, S, dtype=torch.float64, device="cuda"); e = torch.zeros(S, dtype=torch.float64, device="cuda")
        J = torch.zeros(108, S, dtype=torch.float64, device="cuda"); dtf = torch.zeros(6, S, dtype=torch.float64, device="cuda")
        dp = lto.DirectPlan(ctx, 6, n, 1, 10, lto.MU, lto.DU, lto.TU, 2000.0)
        a = timeit(lambda: dp.defect(Xs, n, Us, n, td, 1, dd, S, e, stream=st), iters=50) * 1e3
        b = timeit(lambda: dp.jacobian(Xs, n, Us, n, td, 1, J, S, dtf, dd, S, e, stream=st), iters=50) * 1e3
        print("S=%5d direct rkf78 n=10 defect sweep %7.1f us   Jacobian sweep %7.1f us" % (S, a, b), flush=True)


if __name__ == "__main__" and "--small" in sys.argv:
    small_sweeps()


if __name__ == "__main__" and "--solve" in sys.argv:
    # whole Newton loop in one call (lto_indirect_solve) vs the Python mirror of the loop on the same device operators
    from lowthrustopt_amd import drivers
    from lowthrustopt_amd.constants import MU, DU, TU
    ctx = lto.default_context(0)
    for n in (30, 300, 3000):
        XC, T = synth.indirect_problem(n, seed=3, dt_seg=0.05, lam_sigma=0.05)
        XC, t = XC[:, :, 0], T[:, 0]
        prm = lto.make_params(MU, DU, TU, 10.0, 1000.0, 1.0, 2.0, 1.0)
        # make the guess consistent first so that the timed solves converge in a few iterations from a small perturbation
        X0, d0, st0, it0, h0 = lto.indirect_solve(XC, t, prm, None, False, 50, ctx=ctx)
        rng = np.random.default_rng(0)
        Xp = X0.copy(); Xp[:, 1:-1] += 1e-4 * rng.standard_normal((12, n - 2))
        for name, fn in (("lto_indirect_solve", lambda: lto.indirect_solve(Xp, t, prm, None, False, 20, ctx=ctx)),
                         ("python loop", lambda: drivers.multiShoot_CRTBP_indirect(Xp, t, MU, DU, TU, n, 1000.0, 10.0, False, False, 20,
                                                                                   2.0, 1.0, ops=drivers.HipOps(ctx), verbose=False))):
            out = fn()
            t0 = time.perf_counter()
            for _ in range(10):
                out = fn()
            dt = (time.perf_counter() - t0) / 10
            its = out[3] if len(out) > 3 else -1
            print("n_nodes=%5d %-20s status %d  %s iterations  %.3f ms per solve%s (base guess: status %d after %d)" % (
                n, name, out[2], its if its >= 0 else "?", dt * 1e3, ("  = %.0f us per Newton iteration" % (dt * 1e6 / its)) if its > 0 else "", st0, it0),
                flush=True)
