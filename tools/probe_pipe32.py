import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
from probe_kernels import timeit
ctx = lto.Context(0); st = lto.current_stream_ptr()
for ndim in (14, 12):
    for S in (8192, 262144):
        n = S + 1
        XC, T = synth.indirect_problem(n)
        if ndim == 14:
            Xh = np.zeros((14, n, 1), order="F"); Xh[:6] = XC[:6]; Xh[6] = 1000.0; Xh[7:13] = XC[6:]; Xh[13] = 0.2; slot = 2000.0
        else:
            Xh, slot = XC, 1000.0
        prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, slot, 1.0, 1.0, 1.0)
        X = torch.from_numpy(synth.to_soa_nodes(Xh)).cuda(); t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
        d = torch.zeros(ndim, S, dtype=torch.float64, device="cuda"); Phi = torch.zeros(ndim * ndim, S, dtype=torch.float64, device="cuda")
        plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=64), ndim=ndim)
        for kern, name in ((5, "pipe8"), (8, "pipe32"), (7, "pipe48"), (0, "auto")):
            plan.set_kernel(kern)
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.1:
                for _ in range(4): plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
                torch.cuda.synchronize()
            ms = timeit(lambda: plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st), iters=20 if S < 100000 else 5)
            print("ndim=%d S=%6d %-7s %9.1f us (%s)" % (ndim, S, name, ms * 1e3, plan.last_kernel()), flush=True)
        plan.close()
