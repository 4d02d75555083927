"""ctypes wrapper around oracle/liblto_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (lowthrustopt_amd) never does.  Arrays follow the Julia (column-major) layouts
of the reference: XC_all is [12 x n_nodes], i.e. numpy arrays of shape (n_nodes, 12) C-order hold
the same bytes; to keep call sites readable the wrappers take/return Fortran-ordered
(ndim, n_nodes) arrays exactly like the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

RK4, RKF78_FIXED, RKF78_ADAPTIVE, DOP853_ADAPTIVE = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build():
    """Compile the oracle with g++ (no GPU involved)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblto_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.lto_o_flow_prop_ep.restype = C.c_double
        _LIB.lto_o_set_threads(C.c_int(1))
    return _LIB


def set_threads(n):
    """OpenMP threads used by the annotated sweeps (all-cores CPU baseline); default is 1 = the reference's serial loop."""
    return int(lib().lto_o_set_threads(C.c_int(int(n))))


def _f(a):
    return np.asfortranarray(np.array(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(_dp)


def make_params(MU, DU, TU, thrustLimit, mass, time_direction, p, rho):
    """The reference's params 8-tuple (src/multiShoot_CRTBP_indirect.jl:260)."""
    return np.array([MU, DU, TU, thrustLimit, mass, time_direction, p, rho], dtype=np.float64)


def rhs_state_costate(y, prm):
    y = _f(y); prm = _f(prm); dy = np.zeros(12)
    rc = lib().lto_o_rhs_state_costate(_p(y), _p(prm), _p(dy))
    if rc:
        raise ValueError("Invalid value of p!")
    return dy


def rhs_state_costate_q(y, prm):
    y = _f(y); prm = _f(prm); hi = np.zeros(12); lo = np.zeros(12)
    lib().lto_o_rhs_state_costate_q(_p(y), _p(prm), _p(hi), _p(lo))
    return hi, lo


def rhs_state_costate_jac(y, prm):
    y = _f(y); prm = _f(prm); J = np.zeros((12, 12))
    lib().lto_o_rhs_state_costate_jac(_p(y), _p(prm), _p(J))
    return J


def rhs_prop_ep(s, MU, DU, TU, Isp, control, td):
    s = _f(s); control = _f(control); ds = np.zeros(len(s))
    lib().lto_o_rhs_prop_ep(_p(s), C.c_int(len(s)), C.c_double(MU), C.c_double(DU), C.c_double(TU), C.c_double(Isp),
                            _p(control), C.c_double(td), _p(ds))
    return ds


def flow_state_costate(y, prm, span, method, steps=0, rtol=1e-13, atol=1e-13):
    y = _f(y).copy(); prm = _f(prm); na = C.c_int(0); nr = C.c_int(0)
    rc = lib().lto_o_flow_state_costate(_p(y), _p(prm), C.c_double(span), C.c_int(method), C.c_int(steps),
                                        C.c_double(rtol), C.c_double(atol), C.byref(na), C.byref(nr))
    return y, rc, na.value, nr.value


def flow_state_costate_q(y, prm, span, method, steps):
    y = _f(y).copy(); prm = _f(prm); lo = np.zeros(12)
    lib().lto_o_flow_state_costate_q(_p(y), _p(prm), C.c_double(span), C.c_int(method), C.c_int(steps), _p(lo))
    return y, lo


def flow_stm_state_costate(y, prm, span, method, steps=0, rtol=1e-13, atol=1e-13):
    y = _f(y).copy(); prm = _f(prm); Phi = np.zeros((12, 12), order="F"); na = C.c_int(0); nr = C.c_int(0)
    rc = lib().lto_o_flow_stm_state_costate(_p(y), _p(prm), C.c_double(span), C.c_int(method), C.c_int(steps),
                                            C.c_double(rtol), C.c_double(atol), _p(Phi), C.byref(na), C.byref(nr))
    return y, Phi, rc, na.value, nr.value


def indirect_defect(XC, t, prm, method, steps=0, rtol=1e-13, atol=1e-13):
    XC = _f(XC); t = _f(t); prm = _f(prm); n = XC.shape[1]
    defect = np.zeros((12, n - 1), order="F"); errors = np.zeros(n - 1)
    rc = lib().lto_o_indirect_defect(_p(XC), _p(t), C.c_int(n), _p(prm), C.c_int(method), C.c_int(steps),
                                     C.c_double(rtol), C.c_double(atol), _p(defect), _p(errors))
    return defect, errors, rc


def indirect_jacobian(XC, t, prm, method, steps=0, rtol=1e-13, atol=1e-13):
    """Returns (Phi[12,12,S] with Phi[:,:,i] = d x(t_{i+1}) / d x_i, defect[12,S], rc)."""
    XC = _f(XC); t = _f(t); prm = _f(prm); n = XC.shape[1]
    Phi = np.zeros((12, 12, n - 1), order="F"); defect = np.zeros((12, n - 1), order="F")
    rc = lib().lto_o_indirect_jacobian(_p(XC), _p(t), C.c_int(n), _p(prm), C.c_int(method), C.c_int(steps),
                                       C.c_double(rtol), C.c_double(atol), _p(Phi), _p(defect))
    return Phi, defect, rc


def indirect_scatter_dense(Phi):
    Phi = _f(Phi); S = Phi.shape[2]; n = S + 1
    J = np.zeros((12 * S, 12 * n), order="F")
    lib().lto_o_indirect_scatter_dense(_p(Phi), C.c_int(n), _p(J))
    return J


def direct_defect(X, U, t, nsteps, MU, DU, TU, Isp):
    X = _f(X); U = _f(U); t = _f(t); nstate, n = X.shape
    defect = np.zeros((nstate, n - 1), order="F"); errors = np.zeros(n - 1)
    lib().lto_o_direct_defect(_p(X), _p(U), _p(t), C.c_int(nstate), C.c_int(n), C.c_int(nsteps), C.c_double(MU),
                              C.c_double(DU), C.c_double(TU), C.c_double(Isp), _p(defect), _p(errors))
    return defect, errors


def direct_jacobian_fd(X, U, t, defect, nsteps, MU, DU, TU, Isp, pert=1e-8):
    """Jac_temp[nstate, nvar, S] (block i = d defect_i / d [x_i; x_{i+1}; u_i; u_{i+1}])."""
    X = _f(X); U = _f(U); t = _f(t); defect = _f(defect); nstate, n = X.shape; nvar = 2 * (nstate + 3)
    J = np.zeros((nstate, nvar, n - 1), order="F")
    lib().lto_o_direct_jacobian_fd(_p(X), _p(U), _p(t), _p(defect), C.c_int(nstate), C.c_int(n), C.c_int(nsteps),
                                   C.c_double(MU), C.c_double(DU), C.c_double(TU), C.c_double(Isp), C.c_double(pert), _p(J))
    return J


def direct_jacobian_dual(X, U, t, nsteps, MU, DU, TU, Isp):
    X = _f(X); U = _f(U); t = _f(t); nstate, n = X.shape; nvar = 2 * (nstate + 3)
    J = np.zeros((nstate, nvar, n - 1), order="F"); dh = np.zeros((nstate, n - 1), order="F")
    d = np.zeros((nstate, n - 1), order="F")
    lib().lto_o_direct_jacobian_dual(_p(X), _p(U), _p(t), C.c_int(nstate), C.c_int(n), C.c_int(nsteps), C.c_double(MU),
                                     C.c_double(DU), C.c_double(TU), C.c_double(Isp), _p(J), _p(dh), _p(d))
    return J, dh, d


def direct_dtf_fd(X, U, t, nsteps, MU, DU, TU, Isp, pert_tf=1e-3):
    X = _f(X); U = _f(U); t = _f(t); nstate, n = X.shape
    out = np.zeros((nstate, n - 1), order="F")
    lib().lto_o_direct_dtf_fd(_p(X), _p(U), _p(t), C.c_int(nstate), C.c_int(n), C.c_int(nsteps), C.c_double(MU),
                              C.c_double(DU), C.c_double(TU), C.c_double(Isp), C.c_double(pert_tf), _p(out))
    return out


def direct_scatter_dense(Jac_temp, ddefect_dt):
    Jt = _f(Jac_temp); nstate, nvar, S = Jt.shape; n = S + 1
    dd = _f(ddefect_dt)
    J = np.zeros((nstate * S, n * (nstate + 3) + 1), order="F")
    lib().lto_o_direct_scatter_dense(_p(Jt), _p(dd), C.c_int(nstate), C.c_int(n), _p(J))
    return J


def flow_prop_ep(x, control, td, span, method, steps, MU, DU, TU, Isp, rtol=1e-13, atol=1e-13):
    x = _f(x).copy(); control = _f(control)
    err = lib().lto_o_flow_prop_ep(_p(x), C.c_int(len(x)), _p(control), C.c_double(td), C.c_double(span), C.c_int(method),
                                   C.c_int(steps), C.c_double(rtol), C.c_double(atol), C.c_double(MU), C.c_double(DU),
                                   C.c_double(TU), C.c_double(Isp))
    return x, err


# ---- 14-dim extension (no reference output exists; see lto_oracle.cpp)
def rhs_state_costate_mass(y, prm):
    y = _f(y); prm = _f(prm); dy = np.zeros(14)
    rc = lib().lto_o_rhs_state_costate_mass(_p(y), _p(prm), _p(dy))
    if rc:
        raise ValueError("Invalid value of p!")
    return dy


def indirect14(XC, t, prm, method, steps=0, rtol=1e-13, atol=1e-13, want_stm=True):
    """Returns (Phi[14,14,S] or None, defect[14,S], rc)."""
    XC = _f(XC); t = _f(t); prm = _f(prm); n = XC.shape[1]
    Phi = np.zeros((14, 14, n - 1), order="F") if want_stm else None
    defect = np.zeros((14, n - 1), order="F")
    rc = lib().lto_o_indirect14(_p(XC), _p(t), C.c_int(n), _p(prm), C.c_int(method), C.c_int(steps), C.c_double(rtol),
                                C.c_double(atol), _p(Phi) if want_stm else None, _p(defect))
    return Phi, defect, rc
