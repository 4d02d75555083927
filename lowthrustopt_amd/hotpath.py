"""Host-side mirror of the reference's hot-path closures, calling the HIP library through its C ABI.

Reference seam (there is no FFI in the reference; the seam is created at these four closures):
  indirect defectCalc    src/multiShoot_CRTBP_indirect.jl:63-90
  indirect jacobianCalc  src/multiShoot_CRTBP_indirect.jl:93-146
  direct   defectCalc    src/multiShoot_CRTBP_direct.jl:66-109
  direct   jacobianCalc  src/multiShoot_CRTBP_direct.jl:111-166  (+ tf partial :503-516)

Names, argument meaning and return shapes follow the Julia closures; arrays are numpy in the reference's
(column-major) shapes, e.g. XC_all is (12, n_nodes).  The Julia glue that binds the same C ABI is
julia/LowThrustOptHIP.jl.  Nothing here computes on the CPU: without the HIP
library and a GPU every call raises.
"""
import ctypes as C

import weakref

import numpy as np

from . import _lib
from .constants import RK4, RKF78_FIXED, RKF78_ADAPTIVE, DOP853_ADAPTIVE  # noqa: F401
from ._lib import LtoError, LtoIntegrator, LtoParams, LtoDirectParams, LTO_EINVAL


def integrator(method=DOP853_ADAPTIVE, steps=0, rtol=1e-13, atol=1e-13, max_steps=0):
    """lto_integrator.  Default = adaptive order-8 pair at reltol = abstol = 1e-13, the reference's
    Vern8 setting (src/multiShoot_CRTBP_indirect.jl:79)."""
    return LtoIntegrator(int(method), int(steps), float(rtol), float(atol), int(max_steps))


def make_params(MU, DU, TU, thrustLimit, mass, time_direction, p, rho):
    """The reference's `params` tuple (src/multiShoot_CRTBP_indirect.jl:260)."""
    return LtoParams(float(MU), float(DU), float(TU), float(thrustLimit), float(mass), float(time_direction), float(p),
                     float(rho))


def _params_array(params):
    if isinstance(params, LtoParams):
        params = [params]
    params = [p if isinstance(p, LtoParams) else make_params(*p) for p in params]
    arr = (LtoParams * len(params))(*params)
    return arr, len(params)


def _f64(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """One lto_ctx = one GPU (one process per GPU)."""

    def __init__(self, device=0):
        self.lib = _lib.load_library()
        h = C.c_void_p()
        rc = self.lib.lto_create(C.byref(h), int(device))
        if rc != 0:
            raise LtoError(rc, "lto_create failed (no gfx950 device visible?)")
        self.handle = h
        self.device = int(device)
        self._plans = weakref.WeakSet()      # device-resident plans created on this context

    def close(self):
        """Plans first, then the context.  (The C library tolerates the other order too -- lto_destroy defers while
        plans are alive -- but a closed Context should not leave live handles behind.)"""
        if getattr(self, "handle", None):
            for pl in list(self._plans):
                pl.close()
            # page-locked blocks stay with their numpy arrays (pinned_empty): each is freed when its last view dies, and
            # the library completes this destroy with the last of them (lto.h: lifetime)
            self.lib.lto_destroy(self.handle)
            self.handle = None

    def pinned_empty(self, shape, order="F"):
        """numpy float64 array in page-locked host memory (lto_host_alloc).  The host-pointer API reads and writes such
        arrays (and contiguous views into them) in place from the GPU: no copy-engine operation per operand.  The
        memory lives as long as the array or any view of it does, also beyond Context.close(): the block is freed when the
        last of them is collected (lto_host_free finds the owning context by itself)."""
        n = int(np.prod(shape))
        ptr = C.c_void_p()
        self.check(self.lib.lto_host_alloc(self.handle, max(n, 1) * 8, C.byref(ptr)))
        buf = (C.c_double * max(n, 1)).from_address(ptr.value)
        weakref.finalize(buf, self.lib.lto_host_free, None, C.c_void_p(ptr.value))
        return np.frombuffer(buf, dtype=np.float64, count=n).reshape(shape, order=order)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            msg = self.lib.lto_last_error(self.handle)
            raise LtoError(rc, msg.decode() if msg else "")

    def fn(self, name):
        """Entry point lto_<name> of this handle type (a Group substitutes lto_group_<name>)."""
        return getattr(self.lib, "lto_" + name)

    def set_timing(self, on):
        self.check(self.lib.lto_set_timing(self.handle, 1 if on else 0))

    def last_kernel_ms(self):
        return float(self.lib.lto_last_kernel_ms(self.handle))

    def last_call_ms(self):
        """Wall time of the last host-pointer call as measured inside the library (entry to return)."""
        return float(self.lib.lto_last_call_ms(self.handle))

    def last_call_order(self):
        """Lane order of the last host-pointer indirect call: 0 natural, 1 global, 2 windowed (lto_last_call_order)."""
        return int(self.lib.lto_last_call_order(self.handle))

    def calibrate_kernels(self):
        """Measure AUTO's cost table (us per round of every RK4 STM family) on this context's device (lto_calibrate_kernels)."""
        self.check(self.lib.lto_calibrate_kernels(self.handle))
        return {nd: self.kernel_round_costs(nd)[0] for nd in (12, 14)}

    def kernel_lane_round_us(self):
        """us per round of 256 x CUs segments (64 steps) of the whole-segment RK4 kernel (LTO_KERNEL_LANE): default or calibrated."""
        return float(self.lib.lto_kernel_lane_round_us(self.handle))

    def kernel_round_costs(self, ndim):
        """([pipeline8, pipeline48 (48 segments per workgroup), per-lane, pipeline48 (44 segments), pipeline32] us per round at 64
        steps, calibrated?) -- what LTO_KERNEL_AUTO chooses by."""
        out = (C.c_double * 5)()
        cal = C.c_int(0)
        self.check(self.lib.lto_kernel_round_costs(self.handle, int(ndim), out, C.byref(cal)))
        return [float(v) for v in out], bool(cal.value)


class Group:
    """Several GPUs behind this one process (lto_group_*): pass as `ctx=` to indirect_defectCalc, indirect_stm,
    direct_defectCalc and direct_jacobian_blocks (and the functions built on them).  The sweep is split into
    contiguous shards, one host thread and one context per entry of `devices` (ids may repeat)."""

    SWEEPS = ("indirect_defect", "indirect_jacobian", "direct_defect", "direct_jacobian")

    def __init__(self, devices):
        self.lib = _lib.load_library()
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self.lib.lto_group_create(len(devices), ids, C.byref(h))
        if rc != 0:
            raise LtoError(rc, "lto_group_create failed")
        self.handle = h
        self.devices = [int(d) for d in devices]

    def fn(self, name):
        if name not in self.SWEEPS:
            raise LtoError(-3, "lto_%s has no group form: use a Context" % name)
        return getattr(self.lib, "lto_group_" + name)

    def check(self, rc):
        if rc != 0:
            msg = self.lib.lto_group_last_error(self.handle)
            raise LtoError(rc, msg.decode() if msg else "")

    def __len__(self):
        return int(self.lib.lto_group_size(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lto_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def auto_kernel(ndim, method, steps, p, n_segments, n_cus=256, ordered=False):
    """lto_indirect_auto_kernel: the family LTO_KERNEL_AUTO resolves to for an STM sweep of this shape (MI355X cost table), by name.
    A pure function of the library: needs no GPU."""
    k = _lib.load_library().lto_indirect_auto_kernel(int(ndim), int(method), int(steps), float(p), int(n_segments), int(n_cus), 1 if ordered else 0)
    if k < 0:
        raise LtoError(k, "lto_indirect_auto_kernel: invalid shape")
    return IndirectPlan.KERNEL_NAMES[k]


class Comm:
    """lto_comm: RCCL communicator of one rank (one process per GPU).  `uid` = 128 bytes from Comm.unique_id() on one rank,
    handed to every rank by the launcher (torch.distributed broadcast, MPI, a file).  Operands are device pointers
    (torch tensors or ints); the collectives are asynchronous on `stream`."""

    @staticmethod
    def available():
        return bool(_lib.load_library().lto_comm_available())

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = _lib.load_library().lto_comm_unique_id(buf)
        if rc != 0:
            raise LtoError(rc, "lto_comm_unique_id failed (no RCCL in the process?)")
        return buf.raw

    def __init__(self, ctx, world, rank, uid):
        self.ctx, self.lib = ctx, ctx.lib
        h = C.c_void_p()
        rc = self.lib.lto_comm_create(ctx.handle, int(world), int(rank), C.create_string_buffer(bytes(uid), 128), C.byref(h))
        if rc != 0:
            raise LtoError(rc, "lto_comm_create failed")
        self.handle = h
        self.world, self.rank = int(world), int(rank)

    WINDOW_BYTES = 128

    @classmethod
    def windows(cls, ctx, world, rank, max_count, exchange):
        """The window transport (lto_comm_window_*): device copies into IPC-mapped receive windows, no RCCL, no compute units for
        the payload; also works for ranks that share a device.  `exchange(handle: bytes) -> list of world handles in rank order`
        is the launcher's all-gather of 128-byte blobs (torch.distributed.all_gather_object, MPI, queues)."""
        # Every rank calls exchange() exactly once, whatever happened before it (a rank that skipped the launcher's collective would
        # leave its peers hanging in it): a failed export travels as an empty blob, and then EVERY rank raises.
        self = cls.__new__(cls)
        self.ctx, self.lib = ctx, ctx.lib
        self.handle = None
        self.world, self.rank = int(world), int(rank)
        h = C.c_void_p()
        blob = C.create_string_buffer(cls.WINDOW_BYTES)
        rc = self.lib.lto_comm_window_export(ctx.handle, int(world), int(rank), int(max_count), blob, C.byref(h))
        if rc == 0:
            self.handle = h
        handles = exchange(blob.raw if rc == 0 else b"")
        if rc != 0:
            raise LtoError(rc, "lto_comm_window_export failed")
        if len(handles) != self.world or any(len(b) != cls.WINDOW_BYTES for b in handles):
            self.close()                 # nobody has opened anything yet: every rank sees the same list and raises here
            raise LtoError(-1, "window export failed on a peer (or exchange() did not return the world handles in rank order)")
        try:
            self.check(self.lib.lto_comm_window_open(self.handle, C.create_string_buffer(b"".join(handles), cls.WINDOW_BYTES * self.world)))
        except LtoError as e:
            e.comm = self                # peers may have mapped this rank's window already: the caller closes it after a barrier
            raise
        return self

    def uses_windows(self):
        return bool(self.lib.lto_comm_uses_windows(self.handle))

    def rccl_ranks(self):
        """Ranks RCCL reports for this communicator (ncclCommCount); 0 for a window communicator."""
        n = self.lib.lto_comm_rccl_ranks(self.handle)
        if n < 0:
            raise LtoError(n, "lto_comm_rccl_ranks failed")
        return int(n)

    def check(self, rc):
        if rc != 0:
            msg = self.lib.lto_comm_last_error(self.handle)
            raise LtoError(rc, msg.decode() if msg else "")

    def failed(self, stream=None):
        """True once a wait of this (window) communicator has run out or the ranks have lost step: results are NaN from then on."""
        f = C.c_int(0)
        self.check(self.lib.lto_comm_status(self.handle, stream, C.byref(f)))
        return bool(f.value)

    def set_wait_limit(self, polls):
        self.check(self.lib.lto_comm_set_wait_limit(self.handle, int(polls)))

    def set_kernel_payload(self, nbytes):
        """Window transport: payloads up to `nbytes` per rank go by the push / collect kernels, larger ones by the copy engines."""
        self.check(self.lib.lto_comm_set_kernel_payload(self.handle, int(nbytes)))

    def allgather(self, send, recv, count, stream=None):
        """recv [world][count] <- send [count] of every rank."""
        self.check(self.lib.lto_comm_allgather_dev(self.handle, stream, _dptr(send), _dptr(recv), int(count)))

    def allreduce(self, buf, count, op="sum", stream=None):
        self.check(self.lib.lto_comm_allreduce_dev(self.handle, stream, _dptr(buf), int(count), 0 if op == "sum" else 1))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lto_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _MemberContext(Context):
    """A group member's context as a Context object (owned by the group: close() only forgets it)."""

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle
        self.device = int(lib.lto_ctx_device(handle))
        self._plans = weakref.WeakSet()

    def close(self):
        for pl in list(self._plans):
            pl.close()
        self.handle = None


class GroupComm:
    """lto_group_comm: the collectives of an lto_group (one host process, several GPUs).  member(k) is member k's Context
    for device-resident plans; allgather / allreduce take one device pointer per member and run on the members' streams."""

    def __init__(self, group):
        self.group, self.lib = group, group.lib
        h = C.c_void_p()
        rc = self.lib.lto_group_comm_create(group.handle, C.byref(h))
        if rc != 0:
            raise LtoError(rc, "lto_group_comm_create failed")
        self.handle = h
        self.n = len(group)
        self.members = [_MemberContext(self.lib, C.c_void_p(self.lib.lto_group_ctx(group.handle, k))) for k in range(self.n)]

    def uses_rccl(self):
        return bool(self.lib.lto_group_comm_uses_rccl(self.handle))

    def member(self, k):
        return self.members[k]

    def stream(self, k):
        return C.c_void_p(self.lib.lto_ctx_stream(self.members[k].handle))

    def check(self, rc):
        if rc != 0:
            msg = self.lib.lto_group_comm_last_error(self.handle)
            raise LtoError(rc, msg.decode() if msg else "")

    def _ptrs(self, bufs):
        return (C.c_void_p * self.n)(*[_dptr(b) for b in bufs])

    def allgather(self, send, recv, count):
        self.check(self.lib.lto_group_comm_allgather_dev(self.handle, self._ptrs(send), self._ptrs(recv), int(count)))

    def allreduce(self, bufs, count, op="sum"):
        self.check(self.lib.lto_group_comm_allreduce_dev(self.handle, self._ptrs(bufs), int(count), 0 if op == "sum" else 1))

    def synchronize(self):
        import torch
        for k in range(self.n):
            torch.cuda.synchronize(self.members[k].device)

    def close(self):
        if getattr(self, "handle", None):
            for m in self.members:
                m.close()
            self.lib.lto_group_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_CTX = {}


def default_context(device=0):
    if device not in _DEFAULT_CTX:
        _DEFAULT_CTX[device] = Context(device)
    return _DEFAULT_CTX[device]


# ------------------------------------------------------------------------------------------------
# Host-pointer operators (numpy in / numpy out; data crosses PCIe inside the call)
# ------------------------------------------------------------------------------------------------

def _batch_dims(XC):
    if XC.ndim == 2:
        return XC.shape[0], XC.shape[1], 1, False
    if XC.ndim == 3:
        return XC.shape[0], XC.shape[1], XC.shape[2], True
    raise ValueError("expected [ndim x n_nodes] or [ndim x n_nodes x n_batch]")


def _check_out(arrays, shapes, what):
    """Caller-supplied output arrays are written in place by the library (by the GPU itself when page-locked): exact shape,
    Fortran order, float64 and writeable, or nothing is touched."""
    for a, sh in zip(arrays, shapes):
        if not isinstance(a, np.ndarray) or a.shape != sh or not a.flags.f_contiguous or a.dtype != np.float64 or not a.flags.writeable:
            raise LtoError(-1, "out arrays must be writeable Fortran-ordered float64 arrays of shapes " + what)


def _tgrids(t, n_nodes, n_batch):
    t = _f64(t)
    if t.ndim == 1:
        if t.shape[0] != n_nodes:
            raise ValueError("t_TU must have n_nodes entries")
        return t, 1
    if t.shape != (n_nodes, n_batch):
        raise ValueError("t_TU must be [n_nodes] or [n_nodes x n_batch]")
    return t, n_batch


def indirect_defectCalc(XC_all, t_TU, params, integ=None, ctx=None, out=None):
    """defectCalc of multiShoot_CRTBP_indirect (:63-90): returns (defect[12 x (n-1)], errors[n-1]).
    A trailing batch axis on XC_all sweeps several trajectories (line-search trial points, homotopy levels)
    in one launch; params may then be one tuple or one per trajectory.  out = (defect, errors): Fortran-ordered float64
    arrays of shapes (ndim, n-1, B) and (n-1, B) written in place."""
    ctx = ctx or default_context()
    integ = integ or integrator()
    XC = _f64(XC_all)
    ndim, n, B, batched = _batch_dims(XC)
    t, ntg = _tgrids(t_TU, n, B)
    prm, nprm = _params_array(params)
    if out is not None:
        defect, errors = out
        _check_out((defect, errors), ((ndim, n - 1, B), (n - 1, B)), "(ndim, n-1, B) and (n-1, B)")
    else:
        defect = np.zeros((ndim, n - 1, B), order="F")
        errors = np.zeros((n - 1, B), order="F")
    ctx.check(ctx.fn("indirect_defect")(ctx.handle, ndim, n, B, _ptr(XC), _ptr(t), ntg, prm, nprm, C.byref(integ),
                                          _ptr(defect), _ptr(errors)))
    if not batched:
        return defect[:, :, 0], errors[:, 0]
    return defect, errors


def indirect_stm(XC_all, t_TU, params, integ=None, ctx=None, out=None):
    """Compact Jacobian blocks: Phi[12 x 12 x (n-1)] with Phi[:,:,i] = d x(t_{i+1}) / d XC_all[:,i]
    (the ForwardDiff.jacobian(f, x0) of :121), plus the defect.  out = (Phi, defect): Fortran-ordered float64 arrays of
    shapes (ndim, ndim, n-1, B) and (ndim, n-1, B) written in place (e.g. from Context.pinned_empty)."""
    ctx = ctx or default_context()
    integ = integ or integrator()
    XC = _f64(XC_all)
    ndim, n, B, batched = _batch_dims(XC)
    t, ntg = _tgrids(t_TU, n, B)
    prm, nprm = _params_array(params)
    if out is not None:
        Phi, defect = out
        _check_out((Phi, defect), ((ndim, ndim, n - 1, B), (ndim, n - 1, B)), "(ndim, ndim, n-1, B) and (ndim, n-1, B)")
    else:
        Phi = np.empty((ndim, ndim, n - 1, B), order="F")
        defect = np.empty((ndim, n - 1, B), order="F")
    ctx.check(ctx.fn("indirect_jacobian")(ctx.handle, ndim, n, B, _ptr(XC), _ptr(t), ntg, prm, nprm, C.byref(integ),
                                            _ptr(Phi), _ptr(defect)))
    if not batched:
        return Phi[:, :, :, 0], defect[:, :, 0]
    return Phi, defect


def indirect_scatter(Phi, sparse=False):
    """Band scatter of jacobianCalc (:123-142): row block i = [Phi_i | -I] at columns 12(i-1)+(1:24)
    (1-based), then columns 1:6 and (end-11):(end-6) zeroed (fixed end states)."""
    nd, _, S = Phi.shape
    ns = nd // 2
    n = S + 1
    if sparse:
        import scipy.sparse as sp
        rows = (np.arange(S)[:, None, None] * nd + np.arange(nd)[None, :, None] + np.zeros((1, 1, nd), int)).ravel()
        cols = (np.arange(S)[:, None, None] * nd + np.zeros((1, nd, 1), int) + np.arange(nd)[None, None, :]).ravel()
        vals = np.transpose(Phi, (2, 0, 1)).ravel()
        ir = (np.arange(S)[:, None] * nd + np.arange(nd)[None, :]).ravel()
        ic = ir + nd
        J = sp.coo_matrix((np.concatenate([vals, -np.ones(S * nd)]),
                           (np.concatenate([rows, ir]), np.concatenate([cols, ic]))), shape=(nd * S, nd * n)).tolil()
        J[:, 0:ns] = 0.0
        J[:, nd * n - nd:nd * n - ns] = 0.0
        return J.tocsc()
    J = np.zeros((nd * S, nd * n))
    for i in range(S):
        J[nd * i:nd * (i + 1), nd * i:nd * (i + 1)] = Phi[:, :, i]
        J[nd * i:nd * (i + 1), nd * (i + 1):nd * (i + 2)] = -np.eye(nd)
    J[:, 0:ns] = 0.0
    J[:, nd * n - nd:nd * n - ns] = 0.0
    return J


def indirect_jacobianCalc(XC_all, t_TU, params, integ=None, ctx=None, sparse=False):
    """jacobianCalc of multiShoot_CRTBP_indirect (:93-146): Jac_full [12(n-1) x 12n]."""
    Phi, _ = indirect_stm(XC_all, t_TU, params, integ, ctx)
    return indirect_scatter(Phi, sparse=sparse)


def indirect_newton_step(XC_all, t_TU, params, integ=None, ctx=None, soc_threshold=1e-1, flag_adjointsOnly=False):
    """One Newton iteration on the device (jacobianCalc + least-squares step of optimizeTraj_OLS incl. the
    adjoints-only column mask + second-order correction, indirect.jl:290-296): returns (xc_update, defect)."""
    ctx = ctx or default_context()
    integ = integ or integrator()
    XC = _f64(XC_all)
    ndim, n, B, batched = _batch_dims(XC)
    t, ntg = _tgrids(t_TU, n, B)
    prm, nprm = _params_array(params)
    upd = np.zeros((ndim, n, B), order="F")
    defect = np.zeros((ndim, n - 1, B), order="F")
    ctx.check(ctx.fn("indirect_newton_step")(ctx.handle, ndim, n, B, _ptr(XC), _ptr(t), ntg, prm, nprm, C.byref(integ),
                                               1 if flag_adjointsOnly else 0, float(soc_threshold), _ptr(upd), _ptr(defect)))
    if not batched:
        return upd[:, :, 0], defect[:, :, 0]
    return upd, defect


def indirect_solve(XC_all, t_TU, params, integ=None, flag_adjointsOnly=False, maxIter=10, ctx=None):
    """The Newton loop of multiShoot_CRTBP_indirect (indirect.jl:254-345) as ONE library call, trajectory resident on
    the device: returns (XC_all, defect, status_flag, iterCount, history[k] = (max|defect|, alpha) of iteration k+1)."""
    ctx = ctx or default_context()
    integ = integ or integrator()
    XC = _f64(XC_all)
    ndim, n = XC.shape
    t = np.ascontiguousarray(t_TU, dtype=np.float64)
    prm, _ = _params_array(params)
    XC_out = np.zeros((ndim, n), order="F")
    defect = np.zeros((ndim, n - 1), order="F")
    hist = np.full((2, max(int(maxIter), 1)), np.nan, order="F")
    status = C.c_int(0)
    iters = C.c_int(0)
    ctx.check(ctx.fn("indirect_solve")(ctx.handle, ndim, n, _ptr(XC), _ptr(t), prm, C.byref(integ), 1 if flag_adjointsOnly else 0,
                                       int(maxIter), _ptr(XC_out), _ptr(defect), C.byref(status), C.byref(iters), _ptr(hist)))
    done = int(np.count_nonzero(~np.isnan(hist[1])))          # alpha is written for every completed iteration
    return XC_out, defect, status.value, iters.value, hist[:, :done].T.copy()


def indirect_solve_batch(XC_all, t_TU, params, integ=None, flag_adjointsOnly=False, maxIter=10, ctx=None):
    """n_batch independent Newton loops side by side (lto_indirect_solve_batch): XC_all [12 x n x B], t_TU [n] or
    [n x B], params one tuple or B.  Returns (XC_all, defect, status_flag[B], iterCount[B], history) with
    history[b] = array of (max|defect|, alpha) per completed iteration of trajectory b."""
    ctx = ctx or default_context()
    integ = integ or integrator()
    XC = _f64(XC_all)
    ndim, n, B, batched = _batch_dims(XC)
    t, ntg = _tgrids(t_TU, n, B)
    prm, nprm = _params_array(params)
    XC_out = np.zeros((ndim, n, B), order="F")
    defect = np.zeros((ndim, n - 1, B), order="F")
    mi = max(int(maxIter), 1)
    hist = np.full((2, mi, B), np.nan, order="F")
    status = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    ctx.check(ctx.fn("indirect_solve_batch")(ctx.handle, ndim, n, B, _ptr(XC), _ptr(t), ntg, prm, nprm, C.byref(integ),
                                             1 if flag_adjointsOnly else 0, int(maxIter), _ptr(XC_out), _ptr(defect),
                                             _ptr(status), _ptr(iters), _ptr(hist) if maxIter > 0 else None))
    history = [hist[:, ~np.isnan(hist[1, :, b]), b].T.copy() for b in range(B)]
    return XC_out, defect, status, iters, history


def densify(XC_all, t_TU, params, n_desired, integ=None, ctx=None):
    """densify (src/HelperFunctions.jl:51-101): (XC_dense[ndim x n_desired], t_dense[n_desired]); every segment is
    re-propagated on the GPU and sampled at the uniformly spaced t_dense points that fall inside it."""
    ctx = ctx or default_context()
    integ = integ or integrator()
    XC = _f64(XC_all)
    t = _f64(t_TU)
    ndim, n = XC.shape
    prm, _ = _params_array(params)
    XC_dense = np.zeros((ndim, int(n_desired)), order="F")
    t_dense = np.zeros(int(n_desired))
    ctx.check(ctx.fn("indirect_densify")(ctx.handle, ndim, n, _ptr(XC), _ptr(t), prm, C.byref(integ), int(n_desired),
                                           _ptr(XC_dense), _ptr(t_dense)))
    return XC_dense, t_dense


def direct_defectCalc(X_all, u_all, t_TU, nsteps, MU, DU, TU, Isp, ctx=None):
    """defectCalc of multiShoot_CRTBP_direct (:66-109): returns (defect[nstate x (n-1)], errors[n-1])."""
    ctx = ctx or default_context()
    X = _f64(X_all)
    U = _f64(u_all)
    ns, n, B, batched = _batch_dims(X)
    t, ntg = _tgrids(t_TU, n, B)
    prm = LtoDirectParams(float(MU), float(DU), float(TU), float(Isp))
    defect = np.zeros((ns, n - 1, B), order="F")
    errors = np.zeros((n - 1, B), order="F")
    ctx.check(ctx.fn("direct_defect")(ctx.handle, ns, n, B, _ptr(X), _ptr(U), _ptr(t), ntg, int(nsteps), C.byref(prm),
                                        _ptr(defect), _ptr(errors)))
    if not batched:
        return defect[:, :, 0], errors[:, 0]
    return defect, errors


def direct_midpoints(X_all, u_all, t_TU, nsteps, MU, DU, TU, Isp, ctx=None):
    """The propagation of meshRefine_direct (direct.jl:645-656) for every segment at once: returns
    (x_mid[nstate x (n-1)], defect, errors) with x_mid[:, i] = state at the middle of segment i propagated from node i
    with u_i.  nsteps = 2 is the reference's single `ode7` step."""
    ctx = ctx or default_context()
    X = _f64(X_all)
    U = _f64(u_all)
    ns, n, B, batched = _batch_dims(X)
    t, ntg = _tgrids(t_TU, n, B)
    prm = LtoDirectParams(float(MU), float(DU), float(TU), float(Isp))
    x_mid = np.zeros((ns, n - 1, B), order="F")
    defect = np.zeros((ns, n - 1, B), order="F")
    errors = np.zeros((n - 1, B), order="F")
    ctx.check(ctx.fn("direct_midpoints")(ctx.handle, ns, n, B, _ptr(X), _ptr(U), _ptr(t), ntg, int(nsteps), C.byref(prm),
                                           _ptr(x_mid), _ptr(defect), _ptr(errors)))
    if not batched:
        return x_mid[:, :, 0], defect[:, :, 0], errors[:, 0]
    return x_mid, defect, errors


def direct_jacobian_blocks(X_all, u_all, t_TU, nsteps, MU, DU, TU, Isp, ctx=None, out=None):
    """Compact direct Jacobian: (Jac_temp[nstate x nvar x (n-1)], ddefect_dtf[nstate x (n-1)], defect, errors);
    nvar = 2(nstate+3), variable order [x_i; x_{i+1}; u_i; u_{i+1}] (:125).  out = (Jac_temp, ddefect_dtf, defect, errors):
    Fortran-ordered float64 arrays with the trailing batch axis, written in place (e.g. from Context.pinned_empty)."""
    ctx = ctx or default_context()
    X = _f64(X_all)
    U = _f64(u_all)
    ns, n, B, batched = _batch_dims(X)
    nvar = 2 * (ns + 3)
    t, ntg = _tgrids(t_TU, n, B)
    prm = LtoDirectParams(float(MU), float(DU), float(TU), float(Isp))
    if out is not None:
        Jt, dtf, defect, errors = out
        shapes = ((ns, nvar, n - 1, B), (ns, n - 1, B), (ns, n - 1, B), (n - 1, B))
        _check_out(out, shapes, "%s" % (shapes,))
    else:
        Jt = np.zeros((ns, nvar, n - 1, B), order="F")
        dtf = np.zeros((ns, n - 1, B), order="F")
        defect = np.zeros((ns, n - 1, B), order="F")
        errors = np.zeros((n - 1, B), order="F")
    ctx.check(ctx.fn("direct_jacobian")(ctx.handle, ns, n, B, _ptr(X), _ptr(U), _ptr(t), ntg, int(nsteps), C.byref(prm),
                                          _ptr(Jt), _ptr(dtf), _ptr(defect), _ptr(errors)))
    if not batched:
        return Jt[:, :, :, 0], dtf[:, :, 0], defect[:, :, 0], errors[:, 0]
    return Jt, dtf, defect, errors


def direct_scatter(Jac_temp, ddefect_dtf=None):
    """Band scatter of the direct jacobianCalc (:146-162) and the tf column (:516):
    Jac_full [nstate(n-1) x n(nstate+3) (+1)]; state columns node-major, then control columns."""
    ns, nvar, S = Jac_temp.shape
    n = S + 1
    ncol = n * (ns + 3) + (1 if ddefect_dtf is not None else 0)
    J = np.zeros((ns * S, ncol))
    for i in range(S):
        r = slice(ns * i, ns * (i + 1))
        J[r, ns * i:ns * i + 2 * ns] = Jac_temp[:, :2 * ns, i]
        J[r, ns * n + 3 * i:ns * n + 3 * i + 6] = Jac_temp[:, 2 * ns:, i]
    if ddefect_dtf is not None:
        J[:, -1] = np.asarray(ddefect_dtf).reshape(-1, order="F")
    return J


def direct_endpoint_partials(Jac_temp, ddefect_dtf):
    """The three finite-difference blocks endpointPartials assembles (multiShoot_CRTBP_direct.jl:168-246), read off the
    analytic Jacobian blocks instead of re-propagating:
      ddefect_dt     [nstate(n-1)]  partial of all defects wrt tf               (:176-190)  = the tf column
      d_defect1_dV1  [nstate x 3]   partial of defect 1 wrt an impulse at node 1 (:193-214)  = d defect_1 / d v_1
      d_defectN_dV2  [nstate x 3]   partial of defect N wrt an impulse at node n (:216-222)  = d defect_N / d v_n
    (the reference's tau1/tau2 columns use variables that are never defined, :232,:235; they are not reproduced)."""
    ns = Jac_temp.shape[0]
    return (np.asarray(ddefect_dtf).reshape(-1, order="F").copy(), Jac_temp[:, 3:6, 0].copy(),
            Jac_temp[:, ns + 3:ns + 6, -1].copy())


def direct_jacobianCalc(X_all, u_all, t_TU, nsteps, MU, DU, TU, Isp, ctx=None, with_tf=True):
    """jacobianCalc (+ tf partial) of multiShoot_CRTBP_direct: Jac_full [nstate(n-1) x n(nstate+3)+1]."""
    Jt, dtf, _, _ = direct_jacobian_blocks(X_all, u_all, t_TU, nsteps, MU, DU, TU, Isp, ctx)
    return direct_scatter(Jt, dtf if with_tf else None)


# ------------------------------------------------------------------------------------------------
# Device-resident plans (operands stay in HBM; SoA layouts of include/lto.h)
# ------------------------------------------------------------------------------------------------

def _dptr(x):
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


def current_stream_ptr():
    """hipStream_t of torch's current stream (so torch events / collectives order against our launches)."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class IndirectPlan:
    """lto_indirect_plan: per-trajectory parameters uploaded once, then repeated asynchronous sweeps."""

    def __init__(self, ctx, n_nodes, n_batch, params, integ, ndim=12):
        self.ctx = ctx
        self.n_nodes, self.n_batch, self.ndim = int(n_nodes), int(n_batch), int(ndim)
        self.S = (self.n_nodes - 1) * self.n_batch
        prm, nprm = _params_array(params)
        h = C.c_void_p()
        ctx.check(ctx.lib.lto_indirect_plan_create(ctx.handle, self.ndim, self.n_nodes, self.n_batch, prm, nprm,
                                                   C.byref(integ), C.byref(h)))
        self.handle = h
        ctx._plans.add(self)

    KERNEL_AUTO, KERNEL_PER_LANE, KERNEL_COOP, KERNEL_PIPE8, KERNEL_COOP2, KERNEL_PIPE48, KERNEL_PIPE32, KERNEL_LANE = 0, 1, 2, 5, 6, 7, 8, 9   # 3: LTO_KERNEL_DIRECT_PIPE (direct plans)
    KERNEL_NAMES = {0: "none yet", 1: "per-lane", 2: "cooperative", 5: "pipeline8", 6: "cooperative2", 7: "pipeline48", 8: "pipeline32", 9: "segment-lane"}

    def set_kernel(self, kernel):
        self.ctx.check(self.ctx.lib.lto_indirect_plan_set_kernel(self.handle, int(kernel)))

    LAYOUT_SOA, LAYOUT_BLOCKS = 0, 1

    def set_output_layout(self, layout):
        """LAYOUT_BLOCKS: the sweeps write defect [S][ndim] and Phi [S][ndim*ndim] (one column-major block per segment: Julia's
        layout) instead of struct-of-arrays; 12-dim DOP853 plans only (lto_indirect_plan_set_output_layout)."""
        self.ctx.check(self.ctx.lib.lto_indirect_plan_set_output_layout(self.handle, int(layout)))

    def set_defect_lanes(self, lanes=0):
        """Lanes per segment of the defect-only sweep of a 12-dim DOP853 plan: 0 = choose, 1, 2 or 4
        (lto_indirect_plan_set_defect_lanes)."""
        self.ctx.check(self.ctx.lib.lto_indirect_plan_set_defect_lanes(self.handle, int(lanes)))

    def set_warm_start(self, on=True):
        """Adaptive sweeps of this plan start every segment from its first accepted step size of the plan's previous sweep of
        the same kind (lto_indirect_plan_set_warm_start; 12-dim DOP853 plans)."""
        self.ctx.check(self.ctx.lib.lto_indirect_plan_set_warm_start(self.handle, 1 if on else 0))

    def last_kernel(self):
        """Name of the kernel family the last STM sweep ran (what AUTO resolved to)."""
        return self.KERNEL_NAMES[self.ctx.lib.lto_indirect_plan_last_kernel(self.handle)]

    def staging(self):
        """Record staging of the plan's ordered sweeps (lto_indirect_plan_staging): bit 1 node / defect records in place, bit 2 Phi records
        too (plans that run STM sweeps), bit 4 an allocation failed and staging is off."""
        return int(self.ctx.lib.lto_indirect_plan_staging(self.handle))

    def set_cols_per_lane(self, cols):
        self.ctx.check(self.ctx.lib.lto_indirect_plan_set_cols_per_lane(self.handle, int(cols)))

    def defect(self, X, ldx, t, n_tgrids, defect, ldd, errors=None, stream=None):
        self.ctx.check(self.ctx.lib.lto_indirect_defect_dev(self.handle, stream, _dptr(X), int(ldx), _dptr(t), int(n_tgrids),
                                                            _dptr(defect), int(ldd), _dptr(errors)))

    def jacobian(self, X, ldx, t, n_tgrids, Phi, ldp, defect=None, ldd=0, stream=None):
        self.ctx.check(self.ctx.lib.lto_indirect_jacobian_dev(self.handle, stream, _dptr(X), int(ldx), _dptr(t),
                                                              int(n_tgrids), _dptr(Phi), int(ldp), _dptr(defect), int(ldd)))

    def newton_solve(self, Phi, ldp, defect, ldd, delta, ldx, stream=None, adjoints_only=False):
        """delta = -J \\ defect on the device; Phi=None re-uses the stored factorisation (SOC re-solve)."""
        self.ctx.check(self.ctx.lib.lto_indirect_newton_solve_dev(self.handle, stream, _dptr(Phi), int(ldp), _dptr(defect),
                                                                  int(ldd), 1 if adjoints_only else 0, _dptr(delta), int(ldx)))

    def rebalance(self, stream=None):
        """Order the lanes of subsequent adaptive sweeps by the last sweep's step counts (heaviest first)."""
        self.ctx.check(self.ctx.lib.lto_indirect_plan_rebalance(self.handle, stream))

    def reset_order(self):
        self.ctx.check(self.ctx.lib.lto_indirect_plan_reset_order(self.handle))

    def step_counts(self, stream=None):
        """(accepted[S], rejected[S]) of the last adaptive sweep (numpy int32)."""
        acc = np.zeros(self.S, dtype=np.int32)
        rej = np.zeros(self.S, dtype=np.int32)
        self.ctx.check(self.ctx.lib.lto_indirect_plan_copy_steps(self.handle, stream, _ptr(acc), _ptr(rej)))
        return acc, rej

    def steps_accepted_ptr(self):
        return self.ctx.lib.lto_indirect_plan_steps_accepted(self.handle)

    def steps_rejected_ptr(self):
        return self.ctx.lib.lto_indirect_plan_steps_rejected(self.handle)

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.lto_indirect_plan_destroy(self.handle)
            self.handle = None
            self.ctx._plans.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DirectPlan:
    def __init__(self, ctx, nstate, n_nodes, n_batch, nsteps, MU, DU, TU, Isp):
        self.ctx = ctx
        self.nstate, self.n_nodes, self.n_batch, self.nsteps = int(nstate), int(n_nodes), int(n_batch), int(nsteps)
        self.S = (self.n_nodes - 1) * self.n_batch
        prm = LtoDirectParams(float(MU), float(DU), float(TU), float(Isp))
        h = C.c_void_p()
        ctx.check(ctx.lib.lto_direct_plan_create(ctx.handle, self.nstate, self.n_nodes, self.n_batch, self.nsteps,
                                                 C.byref(prm), C.byref(h)))
        self.handle = h
        ctx._plans.add(self)

    def set_kernel(self, kernel):
        self.ctx.check(self.ctx.lib.lto_direct_plan_set_kernel(self.handle, int(kernel)))

    def defect(self, X, ldx, U, ldu, t, n_tgrids, defect, ldd, errors=None, stream=None):
        self.ctx.check(self.ctx.lib.lto_direct_defect_dev(self.handle, stream, _dptr(X), int(ldx), _dptr(U), int(ldu),
                                                          _dptr(t), int(n_tgrids), _dptr(defect), int(ldd), _dptr(errors)))

    def midpoints(self, X, ldx, U, ldu, t, n_tgrids, x_mid, ldm, defect=None, ldd=0, errors=None, stream=None):
        self.ctx.check(self.ctx.lib.lto_direct_midpoints_dev(self.handle, stream, _dptr(X), int(ldx), _dptr(U), int(ldu),
                                                             _dptr(t), int(n_tgrids), _dptr(x_mid), int(ldm), _dptr(defect),
                                                             int(ldd), _dptr(errors)))

    def jacobian(self, X, ldx, U, ldu, t, n_tgrids, Jac, ldj, dtf=None, defect=None, ldd=0, errors=None, stream=None):
        self.ctx.check(self.ctx.lib.lto_direct_jacobian_dev(self.handle, stream, _dptr(X), int(ldx), _dptr(U), int(ldu),
                                                            _dptr(t), int(n_tgrids), _dptr(Jac), int(ldj), _dptr(dtf),
                                                            _dptr(defect), int(ldd), _dptr(errors)))

    def close(self):
        if getattr(self, "handle", None):
            self.ctx.lib.lto_direct_plan_destroy(self.handle)
            self.handle = None
            self.ctx._plans.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def pack_soa(ctx, aos, ndim, count, soa, ld, stream=None):
    ctx.check(ctx.lib.lto_pack_soa_dev(ctx.handle, stream, _dptr(aos), int(ndim), int(count), _dptr(soa), int(ld)))


def unpack_soa(ctx, soa, ld, ndim, count, aos, stream=None):
    ctx.check(ctx.lib.lto_unpack_soa_dev(ctx.handle, stream, _dptr(soa), int(ld), int(ndim), int(count), _dptr(aos)))


def defect_norms(ctx, defect, ldd, ndim, seg_per_traj, n_batch, sumsq, maxabs, stream=None):
    ctx.check(ctx.lib.lto_defect_norms_dev(ctx.handle, stream, _dptr(defect), int(ldd), int(ndim), int(seg_per_traj),
                                           int(n_batch), _dptr(sumsq), _dptr(maxabs)))


def trial_points(ctx, X, delta, ld, ndim, n_nodes, n_batch, alphas, Xt, ldt, stream=None):
    """Xt = the len(alphas) trial trajectories X + alpha * delta of every trajectory of the batch (device arrays; lineSearch, indirect.jl:227-233)."""
    ctx.check(ctx.lib.lto_trial_points_dev(ctx.handle, stream, _dptr(X), _dptr(delta), int(ld), int(ndim), int(n_nodes), int(n_batch),
                                           int(alphas.numel() if hasattr(alphas, "numel") else len(alphas)), _dptr(alphas), _dptr(Xt), int(ldt)))


def line_search_pick(ctx, sumsq, maxabs, alphas, trial_defect, ldt, ndim, seg_per_traj, n_batch, step, maxabs_out, defect, ldd, stream=None):
    """lineSearch's first minimiser per trajectory (indirect.jl:244-245) taken on the device: step <- alpha, maxabs_out <- the chosen trial's
    max |defect|, defect <- its defect block (the check of :328-331 without another sweep).  Device arrays."""
    na = int(alphas.numel() if hasattr(alphas, "numel") else len(alphas))
    ctx.check(ctx.lib.lto_line_search_pick_dev(ctx.handle, stream, _dptr(sumsq), _dptr(maxabs), _dptr(alphas), na, _dptr(trial_defect), int(ldt),
                                               int(ndim), int(seg_per_traj), int(n_batch), _dptr(step), _dptr(maxabs_out), _dptr(defect), int(ldd)))


def read_scalars(ctx, a, na, b, nb, out, stream=None):
    """out[:na] <- device a, out[na:na+nb] <- device b (or None), back when they have arrived (lto_read_scalars_dev): the per-iteration
    read-back of a Newton loop.  `out`: a C-contiguous float64 numpy array of at least na + nb elements -- checked here, because the
    library copies na + nb doubles to the address it is given (a short or wrong-typed array would be a heap overwrite)."""
    na, nb = int(na), int(nb if b is not None else 0)
    if (not isinstance(out, np.ndarray) or out.dtype != np.float64 or not out.flags["C_CONTIGUOUS"] or not out.flags["WRITEABLE"]
            or na < 0 or nb < 0 or out.size < na + nb):
        raise LtoError(LTO_EINVAL, "read_scalars: `out` must be a writeable C-contiguous float64 array of at least na + nb = %d elements" % (na + nb))
    ctx.check(ctx.lib.lto_read_scalars_dev(ctx.handle, stream, _dptr(a), int(na), _dptr(b), int(nb), out.ctypes.data_as(C.c_void_p)))
    return out
