#!/usr/bin/env python3
"""Writes the pipelined direct Jacobian kernel's outputs (7- and 6-state, ragged size) to a file: run once per library build
(LTO_HIP_LIB), compare the bytes (development aid for changes that must not move a bit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth
ctx = lto.Context(0)
outs = []
for nstate in (7, 6):
    n = 1000
    X, U, T = synth.direct_problem(n, seed=5, nstate=nstate)
    S = n - 1
    Xs = torch.from_numpy(synth.to_soa_nodes(X)).cuda(); Us = torch.from_numpy(synth.to_soa_nodes(U)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
    plan = lto.DirectPlan(ctx, nstate, n, 1, 10, lto.MU, lto.DU, lto.TU, 2000.0)
    nvar = 2 * (nstate + 3)
    plan.set_kernel(3)
    Jac = torch.zeros(nstate * nvar, S, dtype=torch.float64, device="cuda"); dtf = torch.zeros(nstate, S, dtype=torch.float64, device="cuda")
    d = torch.zeros(nstate, S, dtype=torch.float64, device="cuda"); e = torch.zeros(S, dtype=torch.float64, device="cuda")
    plan.jacobian(Xs, n, Us, n, t, 1, Jac, S, dtf, d, S, e)
    torch.cuda.synchronize()
    outs += [v.cpu().numpy().ravel() for v in (Jac, dtf, d, e)]
np.concatenate(outs).tofile(sys.argv[1])
print("wrote", sys.argv[1], float(np.abs(np.concatenate(outs)).max()))
