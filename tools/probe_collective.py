#!/usr/bin/env python3
"""Development aid: host enqueue time vs GPU time of the bench's N > 1 step (sweep + RCCL all-gather on a side stream).
Launch under torch.distributed.run with --nproc-per-node 1 (world size 1 exercises the same code path)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
ctx = lto.Context(0)
st = lto.current_stream_ptr()
S = 4096; n = S + 1
XC, T = synth.indirect_problem(n)
Xh = np.zeros((14, n, 1), order="F")
Xh[:6] = XC[:6]; Xh[6] = 1000.0; Xh[7:13] = XC[6:]; Xh[13] = 0.2
prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 2000.0, 1.0, 1.0, 1.0)
X = torch.from_numpy(synth.to_soa_nodes(Xh)).cuda()
t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator(lto.RK4, steps=64), ndim=14)
Phi = torch.zeros(196, S, dtype=torch.float64, device="cuda")
d = [torch.zeros(14, S, dtype=torch.float64, device="cuda") for _ in range(2)]
g = [torch.zeros(14, S, dtype=torch.float64, device="cuda") for _ in range(2)]
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
K = 200


def run(mode):
    ev_ready = [torch.cuda.Event() for _ in range(2)]
    ev_done = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        b = k & 1
        if mode != "sweep" and k >= 2:
            main.wait_event(ev_done[b])
        plan.jacobian(X, n, t, 1, Phi, S, d[b], S, stream=st)
        if mode == "side":
            ev_ready[b].record(main)
            side.wait_event(ev_ready[b])
            with torch.cuda.stream(side):
                dist.all_gather_into_tensor(g[b], d[b])
                ev_done[b].record(side)
        elif mode == "same":
            dist.all_gather_into_tensor(g[b], d[b])
            ev_done[b].record(main)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-6s host enqueue %.1f us per step, total %.1f us per step" % (mode, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6), flush=True)


for mode in ("sweep", "side", "same", "sweep", "side"):
    run(mode)
dist.destroy_process_group()
