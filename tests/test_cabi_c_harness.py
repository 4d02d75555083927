"""The boundary as a C host sees it: tests/cabi/harness.c includes include/lto.h as strict C99 and links liblto_hip.so
directly (no Python, no torch in that process).  CPU: it builds, and without a device lto_create refuses (no fallback).
GPU: its outputs through pageable and page-locked buffers equal the ctypes path's bit for bit and match the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import lowthrustopt_amd as lto
from lowthrustopt_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MU, DU, TU = lto.MU, lto.DU, lto.TU


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cabi") / "harness")
    libdir = os.path.dirname(lto.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cabi", "harness.c"), "-o", exe, "-L", libdir, "-llto_hip", "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_c99_and_library_refuses_without_a_device(harness):
    import torch
    r = subprocess.run([harness], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "LTO_KERNEL_AUTO resolves as documented" in r.stdout          # the pure host logic of the ABI, called from C without a device
    if torch.cuda.device_count() == 0:
        assert "LTO_ENODEVICE" in r.stdout
    else:
        assert "misuse paths" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("ndim,method,steps", [(12, lto.RK4, 24), (12, lto.DOP853_ADAPTIVE, 0), (14, lto.RK4, 16)])
def test_c_host_gets_the_ctypes_results(harness, gpu_ctx, oracle, tmp_path, ndim, method, steps):
    n = 50
    X12, T = synth.indirect_problem(n, seed=8)
    X12 = X12[:, :, 0]
    t = np.ascontiguousarray(T[:, 0])
    if ndim == 12:
        XC = np.asfortranarray(X12)
        prm_l = [MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 0.5]
    else:                                                       # (r, v, m, lambda_r, lambda_v, lambda_m)
        XC = np.zeros((14, n), order="F")
        XC[:6] = X12[:6]; XC[6] = 1000.0 - 0.05 * np.arange(n); XC[7:13] = X12[6:]; XC[13] = 0.3
        prm_l = [MU, DU, TU, 0.05, 2000.0, 1.0, 1.0, 0.5]      # 14-dim: the mass field carries Isp
    inp, outp = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(inp, "wb") as f:
        f.write(struct.pack("4i", ndim, n, method, steps))
        f.write(struct.pack("2d", 1e-13, 1e-13))
        f.write(struct.pack("8d", *prm_l))
        f.write(XC.tobytes(order="F"))
        f.write(t.tobytes())
    r = subprocess.run([harness, inp, outp], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    S = n - 1
    raw = np.fromfile(outp, dtype=np.float64)
    assert raw.size == ndim * S + S + ndim * ndim * S + ndim * S
    d_c = raw[:ndim * S].reshape((ndim, S), order="F")
    e_c = raw[ndim * S:ndim * S + S]
    Phi_c = raw[ndim * S + S:ndim * S + S + ndim * ndim * S].reshape((ndim, ndim, S), order="F")
    d2_c = raw[-ndim * S:].reshape((ndim, S), order="F")
    integ = lto.integrator(method, steps=steps)
    prm = lto.make_params(*prm_l)
    d_p, e_p = lto.indirect_defectCalc(XC, t, prm, integ, ctx=gpu_ctx)
    Phi_p, d2_p = lto.indirect_stm(XC, t, prm, integ, ctx=gpu_ctx)
    assert np.array_equal(d_c, d_p) and np.array_equal(e_c, e_p)
    assert np.array_equal(Phi_c, Phi_p) and np.array_equal(d2_c, d2_p)
    if ndim == 12:
        if method == lto.RK4:
            P_o, d_o, rc = oracle.indirect_jacobian(XC, t, prm_l, oracle.RK4, steps)
        else:
            P_o, d_o, rc = oracle.indirect_jacobian(XC, t, prm_l, oracle.DOP853_ADAPTIVE, 0, 1e-13, 1e-13)
        assert rc == 0
        assert np.abs(Phi_c - P_o).max() < 1e-10 * np.abs(P_o).max()
        assert np.linalg.norm(d2_c - d_o) < 1e-10 * np.linalg.norm(XC[:, 1:] + d_o)
