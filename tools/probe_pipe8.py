#!/usr/bin/env python3
"""Development aid: STM sweep RK4 x 64 at the contract size, pipeline kernel forms side by side (isolated launches timed
with an event pair each, and back-to-back launches)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
from probe_kernels import timeit


def main():
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    sizes = [int(x) for x in sys.argv[1:]] or [4096]
    for ndim in (14, 12):
        for S in sizes:
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
            res = []
            ref = None
            for kern, name in ((5, "pipe8"), (7, "pipe48")):
                d = torch.zeros(ndim, S, dtype=torch.float64, device="cuda")
                Phi = torch.zeros(ndim * ndim, S, dtype=torch.float64, device="cuda")
                plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=64), ndim=ndim)
                plan.set_kernel(kern)
                run = lambda: plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
                b2b = timeit(run, iters=50)
                iso = []
                for _ in range(10):
                    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record(); run(); e1.record(); torch.cuda.synchronize()
                    iso.append(e0.elapsed_time(e1))
                out = (Phi.cpu().numpy(), d.cpu().numpy())
                if ref is None:
                    ref = out
                dp = np.abs(out[0] - ref[0]).max() / np.abs(ref[0]).max()
                dd = np.abs(out[1] - ref[1]).max()
                res.append("%s b2b %6.1f iso %6.1f us (dPhi %.1e dd %.1e)" % (name, b2b * 1e3, np.median(iso) * 1e3, dp, dd))
                plan.close()
            print("ndim=%d S=%6d  " % (ndim, S) + "  ".join(res), flush=True)


if __name__ == "__main__":
    main()
