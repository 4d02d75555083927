"""Deterministic synthetic Earth-Moon L2 halo->halo shooting problems.

Nodes follow the reference demo's "trajectory stacking" guess (CRTBP_Multishoot_indirect_demo.jl:74-115):
the first half of the nodes sits on halo orbit 1, the second half on halo orbit 2, at phases advancing with
the segment length; costates are 0.1 N(0,1) (demo :166) and interior nodes carry 1e-10 N(0,1) jitter
(demo :176).  Orbit samples come from the reference's two data files (re-emitted under data/).
Pure numpy; used by bench.py, the tests and the golden-fixture generators.
"""
import os

import numpy as np

from .constants import MU, DU, TU, day  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))

# equal-time sample spacing of the two data files (TU), periods = 99 * spacing
HALO_DT = (0.0293768997000, 0.0316124695172)
DT_SEG_DEMO = 20.0 * day / TU / 29.0   # demo: tof = 20 days over 29 segments = 0.15860057 TU

_HALO = None


def halo_orbits():
    """The two 6 x 100 halo-orbit tables (closed: column 100 == column 1)."""
    global _HALO
    if _HALO is None:
        _HALO = tuple(np.loadtxt(os.path.join(_HERE, "data", "halo_L2_%d.txt" % k)) for k in (1, 2))
    return _HALO


def halo_state(which, tau):
    """State on halo `which` (0/1) at time tau (TU) from the first sample; periodic linear interpolation."""
    tab = halo_orbits()[which]
    dt = HALO_DT[which]
    n = tab.shape[1] - 1          # 99 distinct intervals
    u = (np.asarray(tau, dtype=np.float64) / dt) % n
    k = np.floor(u).astype(int)
    w = u - k
    return tab[:, k] * (1.0 - w) + tab[:, k + 1] * w


def indirect_problem(n_nodes, n_batch=1, seed=0, dt_seg=DT_SEG_DEMO, lam_sigma=0.1, jitter=1e-10, dt_range=None):
    """XC_all [12 x n_nodes x n_batch] and t_TU [n_nodes x n_batch].

    dt_range = (lo, hi): per-segment lengths ~ U[lo, hi] (the adaptive / load-balance configuration);
    otherwise every segment has length dt_seg."""
    rng = np.random.default_rng(seed)
    XC = np.zeros((12, n_nodes, n_batch), order="F")
    T = np.zeros((n_nodes, n_batch), order="F")
    for b in range(n_batch):
        if dt_range is None:
            t = np.arange(n_nodes) * dt_seg
        else:
            t = np.concatenate([[0.0], np.cumsum(rng.uniform(dt_range[0], dt_range[1], n_nodes - 1))])
        phase = rng.uniform(0.0, 99 * HALO_DT[0]) if n_batch > 1 else 0.75 * 99 * HALO_DT[0]
        half = n_nodes // 2
        XC[:6, :half, b] = halo_state(0, phase + t[:half])
        XC[:6, half:, b] = halo_state(1, phase + t[half:])
        XC[6:, :, b] = lam_sigma * rng.standard_normal((6, n_nodes))
        if n_nodes > 2:
            XC[:, 1:-1, b] += jitter * rng.standard_normal((12, n_nodes - 2))
        T[:, b] = t
    return XC, T


def direct_problem(n_nodes, n_batch=1, seed=0, dt_seg=DT_SEG_DEMO, thrust_sigma=0.05 / np.sqrt(3.0), nstate=6, mass0=1000.0):
    """X_all [nstate x n_nodes x n_batch], u_all [3 x n_nodes x n_batch] (N), t_TU [n_nodes x n_batch]."""
    rng = np.random.default_rng(seed + 1000)
    X = np.zeros((nstate, n_nodes, n_batch), order="F")
    U = np.zeros((3, n_nodes, n_batch), order="F")
    T = np.zeros((n_nodes, n_batch), order="F")
    for b in range(n_batch):
        t = np.arange(n_nodes) * dt_seg
        phase = rng.uniform(0.0, 99 * HALO_DT[0]) if n_batch > 1 else 0.75 * 99 * HALO_DT[0]
        half = n_nodes // 2
        X[:6, :half, b] = halo_state(0, phase + t[:half])
        X[:6, half:, b] = halo_state(1, phase + t[half:])
        if n_nodes > 2:
            X[:6, 1:-1, b] += 1e-4 * rng.standard_normal((6, n_nodes - 2))
        if nstate == 7:
            X[6, :, b] = mass0 - 0.01 * np.arange(n_nodes)
        U[:, :, b] = thrust_sigma * rng.standard_normal((3, n_nodes))
        T[:, b] = t
    return X, U, T


def homotopy_rhos(levels, rho_hi=1.0, rho_lo=1e-4):
    """rho_l = 10^(-4 l / (levels-1)): the range walked by reduceFuel_indirect
    (CRTBP_Multishoot_indirect_demo.jl:277-278; src/HelperFunctions.jl:105-193)."""
    if levels == 1:
        return np.array([rho_hi])
    return rho_hi * (rho_lo / rho_hi) ** (np.arange(levels) / (levels - 1.0))


def to_soa_nodes(A):
    """[ndim x n_nodes x n_batch] (Julia layout) -> SoA [ndim x (n_nodes*n_batch)], node j = b*n_nodes + k."""
    A = np.asarray(A)
    if A.ndim == 2:
        A = A[:, :, None]
    return np.ascontiguousarray(A.transpose(0, 2, 1).reshape(A.shape[0], -1))


def from_soa_segments(D, seg_per_traj, n_batch):
    """SoA [ndim x S] -> [ndim x seg_per_traj x n_batch]."""
    D = np.asarray(D)
    return np.asfortranarray(D[:, :seg_per_traj * n_batch].reshape(D.shape[0], n_batch, seg_per_traj).transpose(0, 2, 1))
