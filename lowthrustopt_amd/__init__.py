"""lowthrustopt_amd -- MI355X-native multiple-shooting segment propagator for low-thrust CRTBP
trajectories: the defectCalc / jacobianCalc hot path of travelingspaceman/LowThrustOpt behind a C ABI
(include/lto.h, liblto_hip.so) with a host-side mirror of the reference's closures."""
from .constants import MU, DU, TU, day, RK4, RKF78_FIXED, RKF78_ADAPTIVE, DOP853_ADAPTIVE  # noqa: F401
from ._lib import LtoError, LtoParams, LtoIntegrator, LtoDirectParams, load_library, LIB_PATH  # noqa: F401
from .hotpath import (Context, Group, Comm, GroupComm, auto_kernel, auto_kernel, default_context, integrator, make_params, indirect_defectCalc, indirect_stm,  # noqa: F401
                      indirect_scatter, indirect_jacobianCalc, direct_defectCalc, direct_jacobian_blocks,
                      direct_scatter, direct_jacobianCalc, direct_endpoint_partials, direct_midpoints, densify, indirect_newton_step, indirect_solve, indirect_solve_batch, IndirectPlan, DirectPlan, pack_soa, unpack_soa,
                      defect_norms, trial_points, line_search_pick, read_scalars, current_stream_ptr)
from . import synth  # noqa: F401

__version__ = "0.1.0"
