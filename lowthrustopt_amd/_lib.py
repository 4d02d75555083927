"""ctypes binding of liblto_hip.so (the C ABI declared in include/lto.h).

There is deliberately no CPU fallback: if the HIP library is missing or no GPU is visible the calls
raise.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LTO_HIP_LIB") or os.path.join(_HERE, "liblto_hip.so")

LTO_OK, LTO_EINVAL, LTO_ENULL, LTO_EUNSUPPORTED = 0, -1, -2, -3
LTO_EHIP, LTO_EBADP, LTO_ENODEVICE, LTO_ENOMEM = 1, 2, 3, 4


class LtoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("lto error %d: %s" % (code, msg))
        self.code = code


class LtoIntegrator(C.Structure):
    _fields_ = [("method", C.c_int), ("steps", C.c_int), ("rtol", C.c_double), ("atol", C.c_double),
                ("max_steps", C.c_int)]


class LtoParams(C.Structure):
    """The reference's params tuple (src/multiShoot_CRTBP_indirect.jl:260)."""
    _fields_ = [(n, C.c_double) for n in ("MU", "DU", "TU", "thrustLimit", "mass", "time_direction", "p", "rho")]


class LtoDirectParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("MU", "DU", "TU", "Isp")]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); one entry per symbol declared in include/lto.h
SIGNATURES = {
    "lto_version": (C.c_int, []),
    "lto_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "lto_destroy": (None, [_vp]),
    "lto_host_alloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "lto_host_free": (C.c_int, [_vp, _vp]),
    "lto_last_error": (C.c_char_p, [_vp]),
    "lto_ctx_stream": (_vp, [_vp]),
    "lto_ctx_device": (C.c_int, [_vp]),
    "lto_set_timing": (C.c_int, [_vp, C.c_int]),
    "lto_last_kernel_ms": (C.c_double, [_vp]),
    "lto_last_call_ms": (C.c_double, [_vp]),
    "lto_indirect_defect": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.POINTER(LtoParams), C.c_int,
                                      C.POINTER(LtoIntegrator), _vp, _vp]),
    "lto_indirect_jacobian": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.POINTER(LtoParams), C.c_int,
                                        C.POINTER(LtoIntegrator), _vp, _vp]),
    "lto_indirect_newton_step": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.POINTER(LtoParams), C.c_int,
                                           C.POINTER(LtoIntegrator), C.c_int, C.c_double, _vp, _vp]),
    "lto_indirect_solve": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.POINTER(LtoParams), C.POINTER(LtoIntegrator), C.c_int,
                                     C.c_int, _vp, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _vp]),
    "lto_indirect_solve_batch": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.POINTER(LtoParams), C.c_int,
                                           C.POINTER(LtoIntegrator), C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "lto_indirect_densify": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, C.POINTER(LtoParams), C.POINTER(LtoIntegrator), C.c_int,
                                       _vp, _vp]),
    "lto_direct_defect": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int,
                                    C.POINTER(LtoDirectParams), _vp, _vp]),
    "lto_direct_jacobian": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int,
                                      C.POINTER(LtoDirectParams), _vp, _vp, _vp, _vp]),
    "lto_direct_midpoints": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int,
                                       C.POINTER(LtoDirectParams), _vp, _vp, _vp]),
    "lto_group_create": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(_vp)]),
    "lto_group_destroy": (None, [_vp]),
    "lto_group_last_error": (C.c_char_p, [_vp]),
    "lto_group_size": (C.c_int, [_vp]),
    "lto_group_indirect_defect": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.POINTER(LtoParams), C.c_int,
                                            C.POINTER(LtoIntegrator), _vp, _vp]),
    "lto_group_indirect_jacobian": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.POINTER(LtoParams), C.c_int,
                                              C.POINTER(LtoIntegrator), _vp, _vp]),
    "lto_group_direct_defect": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int,
                                          C.POINTER(LtoDirectParams), _vp, _vp]),
    "lto_group_direct_jacobian": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int,
                                            C.POINTER(LtoDirectParams), _vp, _vp, _vp, _vp]),
    "lto_indirect_plan_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(LtoParams), C.c_int,
                                           C.POINTER(LtoIntegrator), C.POINTER(_vp)]),
    "lto_indirect_plan_destroy": (None, [_vp]),
    "lto_indirect_defect_dev": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp, C.c_int, _vp, C.c_long, _vp]),
    "lto_indirect_jacobian_dev": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp, C.c_int, _vp, C.c_long, _vp, C.c_long]),
    "lto_indirect_newton_solve_dev": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp, C.c_long, C.c_int, _vp, C.c_long]),
    "lto_axpy_dev": (C.c_int, [_vp, _vp, _vp, _vp, C.c_double, _vp, C.c_long]),
    "lto_trial_points_dev": (C.c_int, [_vp, _vp, _vp, _vp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_long]),
    "lto_calibrate_kernels": (C.c_int, [_vp]),
    "lto_kernel_round_costs": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "lto_kernel_lane_round_us": (C.c_double, [_vp]),
    "lto_read_scalars_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp]),
    "lto_line_search_pick_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_long, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_long]),
    "lto_indirect_dense_dev": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp, C.c_int, _vp, _vp, _vp, C.c_long, _vp]),
    "lto_indirect_plan_steps_accepted": (_vp, [_vp]),
    "lto_indirect_plan_steps_rejected": (_vp, [_vp]),
    "lto_indirect_plan_copy_steps": (C.c_int, [_vp, _vp, _vp, _vp]),
    "lto_indirect_plan_set_cols_per_lane": (C.c_int, [_vp, C.c_int]),
    "lto_indirect_plan_last_kernel": (C.c_int, [_vp]),
    "lto_indirect_plan_staging": (C.c_int, [_vp]),
    "lto_indirect_plan_rebalance": (C.c_int, [_vp, _vp]),
    "lto_indirect_plan_reset_order": (C.c_int, [_vp]),
    "lto_indirect_plan_set_warm_start": (C.c_int, [_vp, C.c_int]),
    "lto_indirect_plan_set_defect_lanes": (C.c_int, [_vp, C.c_int]),
    "lto_indirect_plan_set_kernel": (C.c_int, [_vp, C.c_int]),
    "lto_direct_plan_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(LtoDirectParams),
                                         C.POINTER(_vp)]),
    "lto_direct_plan_destroy": (None, [_vp]),
    "lto_direct_plan_set_kernel": (C.c_int, [_vp, C.c_int]),
    "lto_direct_defect_dev": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp, C.c_long, _vp, C.c_int, _vp, C.c_long, _vp]),
    "lto_direct_midpoints_dev": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp, C.c_long, _vp, C.c_int, _vp, C.c_long, _vp, C.c_long,
                                           _vp]),
    "lto_direct_jacobian_dev": (C.c_int, [_vp, _vp, _vp, C.c_long, _vp, C.c_long, _vp, C.c_int, _vp, C.c_long, _vp, _vp,
                                          C.c_long, _vp]),
    "lto_pack_soa_dev": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_long, _vp, C.c_long]),
    "lto_unpack_soa_dev": (C.c_int, [_vp, _vp, _vp, C.c_long, C.c_int, C.c_long, _vp]),
    "lto_defect_norms_dev": (C.c_int, [_vp, _vp, _vp, C.c_long, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    # collectives (RCCL over xGMI)
    "lto_comm_available": (C.c_int, []),
    "lto_comm_unique_id": (C.c_int, [_vp]),
    "lto_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "lto_comm_destroy": (None, [_vp]),
    "lto_comm_last_error": (C.c_char_p, [_vp]),
    "lto_comm_size": (C.c_int, [_vp]),
    "lto_comm_rank": (C.c_int, [_vp]),
    "lto_comm_rccl_ranks": (C.c_int, [_vp]),
    "lto_last_call_order": (C.c_int, [_vp]),
    "lto_indirect_plan_set_output_layout": (C.c_int, [_vp, C.c_int]),
    "lto_indirect_auto_kernel": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_double, C.c_long, C.c_int, C.c_int]),
    "lto_comm_allgather_dev": (C.c_int, [_vp, _vp, _vp, _vp, C.c_long]),
    "lto_comm_allreduce_dev": (C.c_int, [_vp, _vp, _vp, C.c_long, C.c_int]),
    "lto_comm_window_export": (C.c_int, [_vp, C.c_int, C.c_int, C.c_long, _vp, C.POINTER(_vp)]),
    "lto_comm_window_open": (C.c_int, [_vp, _vp]),
    "lto_comm_uses_windows": (C.c_int, [_vp]),
    "lto_comm_status": (C.c_int, [_vp, _vp, C.POINTER(C.c_int)]),
    "lto_comm_set_wait_limit": (C.c_int, [_vp, C.c_long]),
    "lto_comm_set_kernel_payload": (C.c_int, [_vp, C.c_long]),
    "lto_group_ctx": (_vp, [_vp, C.c_int]),
    "lto_group_comm_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "lto_group_comm_destroy": (None, [_vp]),
    "lto_group_comm_last_error": (C.c_char_p, [_vp]),
    "lto_group_comm_uses_rccl": (C.c_int, [_vp]),
    "lto_group_comm_allgather_dev": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.c_long]),
    "lto_group_comm_allreduce_dev": (C.c_int, [_vp, C.POINTER(_vp), C.c_long, C.c_int]),
}

_LIB = None


def load_library():
    """dlopen liblto_hip.so and attach prototypes.  Raises if the library has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise LtoError(LTO_ENODEVICE, "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (there is no CPU fallback)" % LIB_PATH)
        # PyTorch wheels bundle their own libamdhip64.so.7; the dynamic loader keeps whichever copy is loaded
        # first.  If torch is going to share the process (bench.py, tests, torch.distributed) its runtime must
        # come first, otherwise torch later finds "No HIP GPUs".  Without torch (Julia, C) the system ROCm is used.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB
