#!/usr/bin/env python3
"""Where a C5 + STM sweep's time goes outside its trial steps (round 6; VERDICT round 5, item 3: 1.49 ms measured against 1.30 ms of
perfectly packed trial steps).  Probe build only (make -C lowthrustopt_amd/csrc probe; LTO_HIP_LIB=build/liblto_probe.so): every
workgroup of k_indirect_coop2 stamps the 100 MHz wall clock at entry, at the start and the end of its trial loop and at its exit, and
notes the compute unit it ran on (hook::Stamps, rows 24-28 of the probe's defect buffer).  Per compute unit the workgroups are put in
order: time inside trial loops, prologues (entry -> first trial step: operands, Hairer's first step size), epilogues (stores), and
the gaps between one workgroup's exit and the next one's entry."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth


def main():
    assert os.environ.get("LTO_HIP_LIB", "").endswith("liblto_probe.so"), "run with the probe build: LTO_HIP_LIB=build/liblto_probe.so"
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    S = int(os.environ.get("SEGS", "65536")); n = S + 1
    XC, T = synth.indirect_problem(n, seed=1, dt_range=(0.05, 0.5))
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1e-3)
    X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
    plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator())
    d = torch.zeros(29, S, dtype=torch.float64, device="cuda")
    Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
    for _ in range(3):
        plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
    for ordered in (False, True):
        if ordered:
            plan.rebalance(stream=st)
        for _ in range(5):
            plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st); e1.record(); torch.cuda.synchronize()
        r = d.cpu().numpy()[:, ::16]
        trials, t_in, t_l0, t_l1, t_out, cu = r[18], r[24], r[25], r[26], r[27], r[28].astype(int)
        us = 1e-2                                               # 100 MHz ticks -> microseconds
        span = (t_out.max() - t_in.min()) * us
        print("%s order: %d workgroups on %d compute units; event time of the sweep %.1f us, first entry to last exit %.1f us" % (
            "lane" if ordered else "natural", len(trials), len(np.unique(cu)), e0.elapsed_time(e1) * 1e3, span))
        print("  trial steps per workgroup: mean %.2f, max %d; loop time per trial step: mean %.2f us" % (
            trials.mean(), trials.max(), ((t_l1 - t_l0) / np.maximum(trials, 1)).mean() * us))
        print("  per workgroup: prologue (entry -> loop) %.2f us, loop %.2f us, epilogue (loop end -> exit) %.2f us" % (
            (t_l0 - t_in).mean() * us, (t_l1 - t_l0).mean() * us, (t_out - t_l1).mean() * us))
        loops = gaps = lead = tail = pro = epi = 0.0
        ngap = 0
        for c in np.unique(cu):
            idx = np.where(cu == c)[0]
            idx = idx[np.argsort(t_in[idx])]
            loops += (t_l1[idx] - t_l0[idx]).sum(); pro += (t_l0[idx] - t_in[idx]).sum(); epi += (t_out[idx] - t_l1[idx]).sum()
            g = t_in[idx][1:] - t_out[idx][:-1]
            gaps += g.sum(); ngap += len(g)
            lead += t_in[idx][0] - t_in.min(); tail += t_out.max() - t_out[idx][-1]
        ncu = len(np.unique(cu))
        tot = span * ncu
        print("  of the %d compute units x %.1f us: trial loops %.1f %%, prologues %.1f %%, epilogues %.1f %%, gaps between workgroups %.1f %% "
              "(mean gap %.2f us), before a unit's first workgroup %.1f %%, after its last %.1f %%" % (
                  ncu, span, 100 * loops * us / tot, 100 * pro * us / tot, 100 * epi * us / tot, 100 * gaps * us / tot, gaps / max(ngap, 1) * us,
                  100 * lead * us / tot, 100 * tail * us / tot))
    plan.close(); ctx.close()


if __name__ == "__main__":
    main()
