#!/usr/bin/env python3
"""Host-pointer ABI (what a Julia ccall takes): wall time per call of lto_indirect_jacobian / lto_indirect_defect with
numpy arrays in and out (plan lookup, H2D, sweep, D2H, synchronise), against the device-resident sweep alone."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth


def per_call(fn, reps):
    """median wall time per call, ms (a first launch of a kernel family costs milliseconds once)"""
    fn(); fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def main():
    ctx = lto.Context(0)
    for S in (29, 4096):
        n = S + 1
        XC, T = synth.indirect_problem(n)
        XC, t = XC[:, :, 0], T[:, 0]
        prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
        out_page = (np.zeros((12, 12, S, 1), order="F"), np.zeros((12, S, 1), order="F"))
        out_pin = (ctx.pinned_empty((12, 12, S, 1)), ctx.pinned_empty((12, S, 1)))
        XC_pin = ctx.pinned_empty((12, n)); XC_pin[:] = XC
        t_pin = ctx.pinned_empty((n,)); t_pin[:] = t
        for name, integ in (("RK4x64", lto.integrator(lto.RK4, steps=64)), ("DOP853@1e-13", lto.integrator())):
            ms_j = per_call(lambda: lto.indirect_stm(XC, t, prm, integ, ctx=ctx), 50)
            ms_jo = per_call(lambda: lto.indirect_stm(XC, t, prm, integ, ctx=ctx, out=out_page), 50)
            ms_jp = per_call(lambda: lto.indirect_stm(XC_pin, t_pin, prm, integ, ctx=ctx, out=out_pin), 50)
            in_lib = ctx.last_call_ms()
            ms_d = per_call(lambda: lto.indirect_defectCalc(XC, t, prm, integ, ctx=ctx), 50)
            assert np.array_equal(out_page[0], out_pin[0]) and np.array_equal(out_page[1], out_pin[1])
            print("S=%5d %-13s host-pointer API, ms per call: jacobian (Phi + defect) fresh outputs %.3f, preallocated pageable %.3f, "
                  "page-locked (lto_host_alloc) %.3f (inside the library, last call: %.3f); defect only %.3f" % (S, name, ms_j, ms_jo, ms_jp, in_lib, ms_d), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
