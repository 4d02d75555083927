"""Caller-supplied output arrays of the host mirror (hotpath.indirect_defectCalc / indirect_stm / direct_jacobian_blocks):
the library -- for page-locked memory the GPU itself -- writes 8 bytes per element in place, so anything that is not a
writeable Fortran-ordered float64 array of the exact shape must be refused before a pointer is taken."""
import numpy as np
import pytest

from lowthrustopt_amd import hotpath
from lowthrustopt_amd._lib import LtoError


def _good(shape):
    return np.zeros(shape, order="F")


@pytest.mark.parametrize("bad", ["float32", "int64", "c_order", "readonly", "shape", "not_array"])
def test_check_out_refuses(bad):
    shapes = ((12, 12, 7, 1), (12, 7, 1))
    a, b = _good(shapes[0]), _good(shapes[1])
    hotpath._check_out((a, b), shapes, "ok")          # the good pair passes
    if bad == "float32":
        a = np.zeros(shapes[0], dtype=np.float32, order="F")
    elif bad == "int64":
        b = np.zeros(shapes[1], dtype=np.int64, order="F")
    elif bad == "c_order":
        a = np.zeros(shapes[0], order="C")
    elif bad == "readonly":
        b.flags.writeable = False
    elif bad == "shape":
        a = _good((12, 12, 6, 1))
    else:
        a = [[0.0]]
    with pytest.raises(LtoError):
        hotpath._check_out((a, b), shapes, "x")


@pytest.mark.gpu
def test_out_arrays_of_the_wrong_type_are_refused_untouched(gpu_ctx):
    import lowthrustopt_amd as lto
    from lowthrustopt_amd import synth
    XC, T = synth.indirect_problem(6)
    XC, t = np.asfortranarray(XC[:, :, 0]), T[:, 0].copy()
    prm = lto.make_params(lto.MU, lto.DU, lto.TU, 0.05, 1000.0, 1.0, 1.0, 1.0)
    integ = lto.integrator(lto.RK4, steps=4)
    Phi32 = np.full((12, 12, 5, 1), 7.0, dtype=np.float32, order="F")
    d = np.full((12, 5, 1), 7.0, order="F")
    with pytest.raises(LtoError):
        lto.indirect_stm(XC, t, prm, integ, ctx=gpu_ctx, out=(Phi32, d))
    assert (Phi32 == 7.0).all() and (d == 7.0).all()
    e_ro = np.full((5, 1), 7.0, order="F")
    e_ro.flags.writeable = False
    with pytest.raises(LtoError):
        lto.indirect_defectCalc(XC, t, prm, integ, ctx=gpu_ctx, out=(d, e_ro))
    assert (d == 7.0).all()
    # and the good case still works in place
    Phi = np.zeros((12, 12, 5, 1), order="F")
    P2, d2 = lto.indirect_stm(XC, t, prm, integ, ctx=gpu_ctx, out=(Phi, d))
    assert np.isfinite(Phi).all() and not (d == 7.0).all()
