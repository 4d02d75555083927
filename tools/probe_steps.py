#!/usr/bin/env python3
"""Trial-step counts of the adaptive STM sweep (DOP853 @ 1e-13, 12-dim, the contract's 4 096 segments): histogram, the
workgroups' maxima (a workgroup of 16 segments runs until its slowest member is done), and what the slowest segments look like."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import lowthrustopt_amd as lto
from lowthrustopt_amd import synth


def main():
    ctx = lto.Context(0)
    st = lto.current_stream_ptr()
    S = int(os.environ.get("SEGS", "4096")); n = S + 1
    XC, T = synth.indirect_problem(n)
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    X = torch.from_numpy(synth.to_soa_nodes(XC)).cuda()
    t = torch.from_numpy(np.ascontiguousarray(T[:, 0])).cuda()
    plan = lto.IndirectPlan(ctx, n, 1, prm, lto.integrator())
    d = torch.zeros(12, S, dtype=torch.float64, device="cuda")
    Phi = torch.zeros(144, S, dtype=torch.float64, device="cuda")
    plan.jacobian(X, n, t, 1, Phi, S, d, S, stream=st)
    torch.cuda.synchronize()
    acc, rej = plan.step_counts(stream=st)
    tr = acc + rej
    print("trial steps per segment: mean %.2f, max %d; accepted mean %.2f max %d; rejected mean %.2f max %d" % (
        tr.mean(), tr.max(), acc.mean(), acc.max(), rej.mean(), rej.max()))
    print("histogram of trial steps:", dict(zip(*np.unique(tr, return_counts=True))))
    wg = tr[: (S // 16) * 16].reshape(-1, 16).max(axis=1)
    print("per-workgroup maxima: mean %.2f, histogram:" % wg.mean(), dict(zip(*np.unique(wg, return_counts=True))))
    XCn = XC[:, :, 0]
    rm = np.sqrt((XCn[0] - 1 + lto.MU) ** 2 + XCn[1] ** 2 + XCn[2] ** 2)
    Ph = Phi.cpu().numpy()
    for s in np.argsort(-tr)[:12]:
        print("  seg %5d trials %3d (acc %d rej %d) r_moon %.4f |lv| %.3f |lr| %.3f min|y| %.2e max|Phi| %.2e" % (
            s, tr[s], acc[s], rej[s], rm[s], np.linalg.norm(XCn[9:12, s]), np.linalg.norm(XCn[6:9, s]),
            np.abs(XCn[:, s]).min(), np.abs(Ph[:, s]).max()))
    q = np.argsort(tr)[S // 2]
    print("  median seg %d trials %d r_moon %.4f min|y| %.2e max|Phi| %.2e" % (q, tr[q], rm[q], np.abs(XCn[:, q]).min(), np.abs(Ph[:, q]).max()))
    print("corr(trials, -log r_moon) = %.3f, corr(trials, log max|Phi|) = %.3f, corr(trials, -log min|y|) = %.3f" % (
        np.corrcoef(tr, -np.log(rm[:S]))[0, 1], np.corrcoef(tr, np.log(np.abs(Ph).max(axis=0)))[0, 1],
        np.corrcoef(tr, -np.log(np.abs(XCn[:, :S]).min(axis=0) + 1e-300))[0, 1]))
    plan.close(); ctx.close()


if __name__ == "__main__":
    main()
