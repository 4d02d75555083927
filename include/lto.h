/*
 * lto.h -- C ABI of liblto_hip.so: the MI355X (gfx950) multiple-shooting segment propagator.
 *
 * This is the drop-in boundary for the one data-parallel hot path of
 * travelingspaceman/LowThrustOpt: the `defectCalc` / `jacobianCalc` closures nested inside
 *   multiShoot_CRTBP_indirect  (src/multiShoot_CRTBP_indirect.jl:63-90, :93-146)
 *   multiShoot_CRTBP_direct    (src/multiShoot_CRTBP_direct.jl:66-109, :111-166, tf partial :503-516)
 * The reference has no FFI of its own (pure Julia); these entry points are what a `ccall` from the
 * two Julia drivers binds instead of running the closures' serial `for i = 1:n_nodes-1` loops
 * (INTEGRATION.md shows the Julia side).  Plain C: pointers, ints and doubles only.
 *
 * Conventions
 *   - All floating point data is binary64.  Host arrays use the reference's Julia layouts
 *     (column-major): XC_all is [ndim x n_nodes], defect is [ndim x (n_nodes-1)], ...
 *   - `n_batch` independent trajectories (line-search trial points, homotopy levels) can be swept by
 *     one call; host arrays then carry a trailing batch dimension.
 *   - Return value: 0 = ok; < 0 = API misuse; > 0 = runtime failure (see LTO_E*).  Non-finite
 *     results are not errors: NaN/Inf propagate into the outputs so the caller's driver reproduces the
 *     reference's status_flag = 2 path (src/multiShoot_CRTBP_indirect.jl:339-341).
 *   - The library never throws across the ABI: it is built without exception support, and what it allocates on the host
 *     inside a call (work arrays, its bookkeeping lists, the threads of a lto_group call) is checked -- running out of
 *     host memory or of threads there returns LTO_ENOMEM, it does not end the caller's process (a Julia session).  It
 *     installs no signal handlers and keeps
 *     no host pointer after a call returns.  One thread per context at a time; distinct contexts are
 *     independent.
 */
#ifndef LTO_H
#define LTO_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LTO_VERSION 102 /* 0.1.2: round 6 added lto_comm_rccl_ranks, lto_last_call_order, lto_indirect_auto_kernel, lto_indirect_plan_set_output_layout; LTO_KERNEL_PIPE is now LTO_KERNEL_DIRECT_PIPE (same value), LTO_KERNEL_PIPE6_REMOVED is gone; lto_indirect_plan_set_cols_per_lane: 14 names the whole-segment one-step form of 14-dim plans, 2 is refused for 12-dim plans (0.1.1, round 5: LTO_ENOMEM, LTO_KERNEL_LANE, lto_indirect_plan_staging, lto_comm_set_kernel_payload, lto_kernel_lane_round_us) */

/* error codes */
#define LTO_OK 0
#define LTO_EINVAL (-1)       /* bad dimension / count / enum                                   */
#define LTO_ENULL (-2)        /* required pointer is NULL                                        */
#define LTO_EUNSUPPORTED (-3) /* valid request this build does not implement                     */
#define LTO_EHIP 1            /* HIP runtime error (message in lto_last_error)                   */
#define LTO_EBADP 2           /* reference: error("Invalid value of p!") stateCostate_deriv.jl:52 */
#define LTO_ENODEVICE 3       /* no usable gfx950 device                                         */
#define LTO_ENOMEM 4          /* host memory (or a host thread) could not be had inside a call   */

/* Integrators.  RK4 = GeneralCode/ode.jl:21-73; RKF78_FIXED = ode7_8, ode.jl:773-953 (the direct
 * path's integrator); RKF78_ADAPTIVE = ode78, ode.jl:364-544; DOP853_ADAPTIVE = order-8 adaptive pair
 * standing in for OrdinaryDiffEq's Vern8() at reltol=abstol=1e-13 (multiShoot_CRTBP_indirect.jl:79). */
#define LTO_RK4 0
#define LTO_RKF78_FIXED 1
#define LTO_RKF78_ADAPTIVE 2
#define LTO_DOP853_ADAPTIVE 3

typedef struct lto_ctx lto_ctx;
typedef struct lto_indirect_plan lto_indirect_plan;
typedef struct lto_direct_plan lto_direct_plan;

typedef struct lto_integrator {
  int method;    /* LTO_RK4 ...                                                        */
  int steps;     /* fixed-step methods: number of equal steps per segment              */
  double rtol;   /* adaptive methods (RKF78_ADAPTIVE uses rtol as ode78's `tol`)        */
  double atol;
  int max_steps; /* adaptive methods: per-segment cap on accepted+rejected steps (0 = 100000) */
} lto_integrator;

/* The reference's `params` tuple, src/multiShoot_CRTBP_indirect.jl:260 /
 * src/CRTBP_stateCostate_deriv.jl:13, field for field. */
typedef struct lto_params {
  double MU, DU, TU, thrustLimit, mass, time_direction, p, rho;
} lto_params;

/* Arguments the direct closures forward to ode7_8 / CRTBP_prop_EP_deriv
 * (src/multiShoot_CRTBP_direct.jl:86: MU, DU, TU, Isp). */
typedef struct lto_direct_params {
  double MU, DU, TU, Isp;
} lto_direct_params;

/* ------------------------------------------------------------------------------- context */
/* One context per GPU (one process per GPU under torch.distributed / one Julia task).  device_id
 * is the HIP ordinal.  Owns a stream, grow-only device staging buffers and the plans of the host-pointer
 * API (kept between calls: a Newton iteration calls with the same shapes and parameters every time).
 * Lifetime: plans from lto_*_plan_create keep their context alive -- lto_destroy with such plans
 * outstanding only marks the context, and the last lto_*_plan_destroy frees it (a garbage collector may
 * run the finalizers of a context and of its plans in any order). */
int lto_create(lto_ctx** out, int device_id);
void lto_destroy(lto_ctx* ctx);
/* Page-locked host memory for the arrays of the host-pointer API.  An operand that lies inside a block from here (the
 * whole block or any contiguous part of it) is read / written by the GPU in place: the layout kernels of the call are the
 * transfer and no copy operation is queued (Jacobian call at 4 096 segments: 0.20 ms, of which 0.09 ms are the 4.7 MB of
 * Phi crossing the link).  Other buffers work too: pageable ones are staged by the HIP runtime (0.31 ms for the same call).
 * Julia: unsafe_wrap the pointer as an Array and free it in a finalizer (julia/LowThrustOptHIP.jl: pinned_array). */
int lto_host_alloc(lto_ctx* ctx, size_t bytes, void** out);
/* Lifetime: a block keeps its context alive the way a plan does (lto_destroy only marks a context that still has blocks or
 * plans; whoever releases the last of them -- from any thread, in any order: finalizers -- frees it, exactly once).
 * lto_host_free(NULL, ptr) is allowed: the owner of a block is looked up, so a finalizer need not keep the handle.  Freeing a
 * pointer that is not a live block returns LTO_EINVAL. */
int lto_host_free(lto_ctx* ctx, void* ptr);
const char* lto_last_error(const lto_ctx* ctx);
int lto_version(void);
void* lto_ctx_stream(lto_ctx* ctx); /* the context's own non-blocking hipStream_t */
int lto_ctx_device(const lto_ctx* ctx); /* HIP ordinal the context was created on */
/* When enabled, every sweep brackets its dominant kernel with HIP events on the launch stream;
 * lto_last_kernel_ms blocks on the stop event and returns that kernel's duration. */
int lto_set_timing(lto_ctx* ctx, int enabled);
double lto_last_kernel_ms(lto_ctx* ctx);
/* Wall time [ms] of the last host-pointer call on this context (lto_indirect_defect, lto_indirect_jacobian, lto_direct_*),
 * from entry to return as measured inside the library: what a C or Julia caller waits for, without a binding's overhead. */
double lto_last_call_ms(const lto_ctx* ctx);
/* Lane order the last host-pointer indirect call of this context swept with: 0 = natural, 1 = the global order, 2 = the windowed
 * order (both made from an earlier call's step counts and kept in the context, one slot per kind: defect sweeps take the windowed
 * order, STM sweeps and Newton steps the global one, so a loop that alternates defectCalc and jacobianCalc keeps both). */
int lto_last_call_order(const lto_ctx* ctx);

/* Kernel choice of the RK4 STM sweeps above one round of workgroups (lto_indirect_plan_set_kernel, LTO_KERNEL_AUTO): the family
 * whose rounds are cheapest for the segment count, from a table of microseconds per round at 64 steps -- us_per_round[0]:
 * eight-wave pipeline, rounds of 16 x CUs segments; [1]: large-batch pipeline with 48 segments per workgroup, 48 x CUs; [2]:
 * per-lane kernel with three columns, 64 x CUs; [3]: large-batch pipeline with 44 segments per workgroup, 44 x CUs; [4]:
 * 32-segment / twelve-wave pipeline (LTO_KERNEL_PIPE32), 32 x CUs ([2], [3]: 12-dim only, reported as -1 for 14;
 * LTO_KERNEL_PIPE48 stands for both of its forms and the cheaper one runs).  A new context holds the figures measured on MI355X
 * (profiles/r04z).  lto_calibrate_kernels measures them on the context's own device (about 50 ms: 30 ms of warm-up sweeps, then
 * the median of five launches of one full round per family and dimension) and AUTO uses those from then on;
 * lto_kernel_round_costs reads the table (us_per_round[5]; *calibrated = 1 after a calibration).  Results never depend on the
 * choice. */
int lto_calibrate_kernels(lto_ctx* ctx);
int lto_kernel_round_costs(const lto_ctx* ctx, int ndim, double* us_per_round, int* calibrated);
/* The sixth family AUTO weighs for ndim = 12 (round 5; kept out of the five-entry table so that its callers' arrays stay valid):
 * microseconds per round of 256 x CUs segments at 64 steps of the whole-segment lanes (LTO_KERNEL_LANE) -- 505 on MI355X by
 * default, this device's figure after lto_calibrate_kernels (which then also sweeps one such round: ~5 ms more, and the context's
 * work arena grows to ~110 MB). */
double lto_kernel_lane_round_us(const lto_ctx* ctx);

/* --------------------------------------------------------- host-pointer API (what Julia ccalls)
 * Each call: plan looked up in the context's cache by (shape, integrator, parameter values) -> H2D ->
 * sweep -> D2H -> one stream synchronise.  The caller's buffers are only touched inside the call. */

/* Replaces defectCalc of multiShoot_CRTBP_indirect (src/multiShoot_CRTBP_indirect.jl:63-90).
 *   XC      [ndim x n_nodes x n_batch]   ndim = 12: the reference's state+costate system.
 *                                        ndim = 14: (r, v, m, lambda_r, lambda_v, lambda_m), an EXTENSION with no
 *                                        reference counterpart (BASELINE configs[1]; model after
 *                                        GeneralCode/twoBody_stateCostate_mass_deriv.jl:11-78 in CRTBP units); the
 *                                        `mass` field of lto_params then carries Isp [s] (mass is state[7]).
 *   t       [n_nodes x n_tgrids]         n_tgrids = 1 (shared grid) or n_batch
 *   prm     [n_prm]                      n_prm = 1 or n_batch
 *   defect  [ndim x (n_nodes-1) x n_batch]  = x(t_{i+1}; XC[:,i]) - XC[:,i+1]          (:82)
 *   errors  [(n_nodes-1) x n_batch] or NULL: 0 for RK4 / adaptive (reference: always 0, :85),
 *           RKF7(8) 8th-order estimate for RKF78_FIXED. */
int lto_indirect_defect(lto_ctx* ctx, int ndim, int n_nodes, int n_batch, const double* XC, const double* t,
                        int n_tgrids, const lto_params* prm, int n_prm, const lto_integrator* integ, double* defect,
                        double* errors);

/* Replaces jacobianCalc of multiShoot_CRTBP_indirect (:93-146), compact form:
 *   Phi     [ndim x ndim x (n_nodes-1) x n_batch], Phi[:,:,i] = d x(t_{i+1}) / d XC[:,i]
 *           (= ForwardDiff.jacobian(f, x0), :121).  The caller forms [Phi_i | -I] (:123), the band
 *           scatter (:128-138) and the fixed-endpoint column mask (:141-142).
 *   defect  as above, or NULL. */
int lto_indirect_jacobian(lto_ctx* ctx, int ndim, int n_nodes, int n_batch, const double* XC, const double* t,
                          int n_tgrids, const lto_params* prm, int n_prm, const lto_integrator* integ, double* Phi,
                          double* defect);

/* One whole Newton iteration of multiShoot_CRTBP_indirect (indirect.jl:290-296): jacobianCalc (:93-146), the
 * least-squares step of optimizeTraj_OLS (:149-218) incl. the flag_adjointsOnly column mask (:169-178) and the
 * second-order correction (:190-214, applied when norm(xc_update, Inf) < soc_threshold; the reference uses 1e-1).
 * Only XC and t are uploaded and xc_update [ndim x n_nodes x n_batch] and (optionally) the nominal defect are
 * downloaded; Phi stays in HBM. */
int lto_indirect_newton_step(lto_ctx* ctx, int ndim, int n_nodes, int n_batch, const double* XC, const double* t,
                             int n_tgrids, const lto_params* prm, int n_prm, const lto_integrator* integ,
                             int flag_adjointsOnly, double soc_threshold, double* xc_update, double* defect);

/* The whole Newton loop of multiShoot_CRTBP_indirect (src/multiShoot_CRTBP_indirect.jl:254-345) in one call, trajectory
 * resident in HBM: while max|defect| > 1e-10 { jacobianCalc; optimizeTraj_OLS incl. adjoints-only mask and second-order
 * correction (:149-218); after iteration 3 the 20-point lineSearch (:221-246) as ONE batched sweep; XC += alpha *
 * xc_update; end states re-pinned (:324-325); defectCalc }.  Only scalars cross PCIe inside the loop.
 *   XC_in, XC_out [12 x n_nodes] (may alias), defect [12 x (n_nodes-1)] or NULL,
 *   *status_flag: 0 converged, 1 maxIter reached (also after "Not likely to converge", :333-336), 2 NaN (:339-341),
 *   *iterations (or NULL): the reference's iterCount on exit,
 *   history (or NULL): [2 x maxIter] column k = (max|defect|, alpha) after iteration k+1 -- the progress line of :332. */
int lto_indirect_solve(lto_ctx* ctx, int ndim, int n_nodes, const double* XC_in, const double* t, const lto_params* prm,
                       const lto_integrator* integ, int flag_adjointsOnly, int maxIter, double* XC_out, double* defect,
                       int* status_flag, int* iterations, double* history);

/* n_batch independent problems through the same loop, side by side (homotopy / thrust levels, several initial guesses:
 * the concurrent form of the continuation of src/HelperFunctions.jl:105-193).  Every device operation covers the
 * whole batch; a trajectory that has left the reference loop (converged, NaN, iteration limit) is frozen by a zero
 * step length.  Arrays carry a trailing batch dimension: XC [12 x n_nodes x n_batch], t [n_nodes x n_tgrids],
 * prm [n_prm] (n_tgrids, n_prm = 1 or n_batch), defect [12 x (n_nodes-1) x n_batch], status_flag / iterations
 * [n_batch], history [2 x maxIter x n_batch]. */
int lto_indirect_solve_batch(lto_ctx* ctx, int ndim, int n_nodes, int n_batch, const double* XC_in, const double* t,
                             int n_tgrids, const lto_params* prm, int n_prm, const lto_integrator* integ,
                             int flag_adjointsOnly, int maxIter, double* XC_out, double* defect, int* status_flag,
                             int* iterations, double* history);

/* Replaces densify (src/HelperFunctions.jl:51-101) for one trajectory: t_dense = LinRange(t[1], t[end], n_desired),
 * every segment re-propagated from its node and sampled at the t_dense points inside [t_i, t_{i+1}), final propagated
 * state appended.  XC_dense [ndim x n_desired], t_dense [n_desired]. */
int lto_indirect_densify(lto_ctx* ctx, int ndim, int n_nodes, const double* XC, const double* t, const lto_params* prm,
                         const lto_integrator* integ, int n_desired, double* XC_dense, double* t_dense);

/* Replaces defectCalc of multiShoot_CRTBP_direct (src/multiShoot_CRTBP_direct.jl:66-109).
 *   X [nstate x n_nodes x n_batch] (nstate = 6 or 7), U [3 x n_nodes x n_batch] thrust in N,
 *   nsteps = points of the half-segment grid, i.e. nsteps-1 RKF7(8) steps per half (:84).
 *   defect [nstate x (n_nodes-1) x n_batch], errors [(n_nodes-1) x n_batch] (:104). */
int lto_direct_defect(lto_ctx* ctx, int nstate, int n_nodes, int n_batch, const double* X, const double* U,
                      const double* t, int n_tgrids, int nsteps, const lto_direct_params* prm, double* defect,
                      double* errors);

/* The propagation inside meshRefine_direct (src/multiShoot_CRTBP_direct.jl:645-656): x_mid[:, i] = state at
 * t_i + (t_{i+1} - t_i)/2 propagated forward from node i with control u_i, for EVERY segment in one sweep (the
 * reference propagates one segment per refinement pass with ode7 = one RKF7(8) step, i.e. nsteps = 2).
 *   x_mid [nstate x (n_nodes-1) x n_batch]; defect, errors as lto_direct_defect on the same grid, or NULL. */
int lto_direct_midpoints(lto_ctx* ctx, int nstate, int n_nodes, int n_batch, const double* X, const double* U,
                         const double* t, int n_tgrids, int nsteps, const lto_direct_params* prm, double* x_mid,
                         double* defect, double* errors);

/* Replaces jacobianCalc of multiShoot_CRTBP_direct (:111-143) and the tf partial (:503-516).
 *   Jac_temp    [nstate x nvar x (n_nodes-1) x n_batch], nvar = 2(nstate+3); block i is
 *               d defect_i / d [x_i; x_{i+1}; u_i; u_{i+1}] (variable order of :125), computed from
 *               the variational equations instead of the reference's forward differences.
 *   ddefect_dtf [nstate x (n_nodes-1) x n_batch] or NULL (last column of Jac_full, :516)
 *   defect, errors as lto_direct_defect, or NULL. */
int lto_direct_jacobian(lto_ctx* ctx, int nstate, int n_nodes, int n_batch, const double* X, const double* U,
                        const double* t, int n_tgrids, int nsteps, const lto_direct_params* prm, double* Jac_temp,
                        double* ddefect_dtf, double* defect, double* errors);

/* ------------------------------------------------- device-resident API (operands already in HBM)
 * Struct-of-arrays, segment/node index fastest, so that a wavefront's 64 lanes read 512
 * contiguous bytes per component.  With J = n_nodes*n_batch nodes and S = (n_nodes-1)*n_batch
 * segments (node j = b*n_nodes + k, segment s = b*(n_nodes-1) + i):
 *   X[c*ldx + j]  t[g*n_nodes + k]  defect[c*ldd + s]  Phi[(col*ndim+row)*ldp + s]  errors[s]
 *   U[c*ldu + j]  Jac[(col*nstate+row)*ldj + s]  dtf[c*ldd + s]
 * Launches are asynchronous on `stream`, a hipStream_t taken literally (NULL = HIP's default stream,
 * which is also PyTorch's default current stream); lto_ctx_stream() returns the context's own stream.
 * One exception to "asynchronous": lto_indirect_defect_dev on an ndim = 12 DOP853_ADAPTIVE plan under LTO_KERNEL_AUTO with at
 * least 64 x CUs segments may WAIT ON THE HOST for an event recorded behind an EARLIER defect sweep of the same plan (the trial-step
 * statistics its lanes-per-segment choice reads: after the plan's first two sweeps, then every sixteenth) -- i.e. for work the
 * caller enqueued before, never for the sweep being enqueued; lto_indirect_plan_set_defect_lanes(plan, 1 | 2 | 4) fixes the
 * choice and removes the wait, and inside a stream capture nothing is waited for. */
int lto_indirect_plan_create(lto_ctx* ctx, int ndim, int n_nodes, int n_batch, const lto_params* prm, int n_prm,
                             const lto_integrator* integ, lto_indirect_plan** out);
void lto_indirect_plan_destroy(lto_indirect_plan* plan);
int lto_indirect_defect_dev(lto_indirect_plan* plan, void* stream, const double* X, long ldx, const double* t,
                            int n_tgrids, double* defect, long ldd, double* errors);
int lto_indirect_jacobian_dev(lto_indirect_plan* plan, void* stream, const double* X, long ldx, const double* t,
                              int n_tgrids, double* Phi, long ldp, double* defect, long ldd);
/* Per-segment accepted / rejected step counts of the last adaptive sweep (device pointers owned by the
 * plan, S ints each; NULL for fixed-step plans). */
const int* lto_indirect_plan_steps_accepted(const lto_indirect_plan* plan);
const int* lto_indirect_plan_steps_rejected(const lto_indirect_plan* plan);
/* Copies the counters of the last adaptive sweep launched on `stream` to host arrays of S ints (either may be NULL). */
int lto_indirect_plan_copy_steps(lto_indirect_plan* plan, void* stream, int* accepted, int* rejected);
/* Load balancing of adaptive sweeps.  Segments that need many steps (long or sharply switching arcs) hold their
 * whole wavefront / workgroup until they finish.  lto_indirect_plan_rebalance orders the lanes of all SUBSEQUENT sweeps
 * of this plan by the step counts of the LAST sweep, heaviest first, so that neighbouring lanes do similar work
 * (counting sort on the device, asynchronous on `stream`).  Results do not change: every segment still takes its own
 * step sequence and is stored at its own index.  Call it again when the trajectory has moved enough to change the
 * step counts; lto_indirect_plan_reset_order returns to the natural order.  Fixed-step plans, and plans that have not
 * swept yet: LTO_EINVAL.
 * Two kinds of order (round 5), chosen by what the plan has run so far.  A plan that has run an STM sweep gets the GLOBAL order
 * (heaviest segment first over the whole batch) and, for ndim = 12 DOP853 plans, record staging: nodes in and results out travel
 * as per-segment records with coalesced transposes either side of the sweep (lto_indirect_plan_staging) -- the shortest sweep,
 * 2.7-2.9 x the algorithmic HBM bytes.  A plan that has only run defect sweeps gets the WINDOWED order: segments ordered inside
 * windows of 1 024 consecutive segments, the windows ranked by their slowest segment, dealt to the XCDs in turn and interleaved
 * there 16 segments at a time, the sweep's workgroups mapped to contiguous ranges per XCD -- the wavefronts that share a window's
 * cache lines then share an L2, the sweep gathers from and scatters to the caller's arrays directly (no records, no extra
 * passes): the same sweep time and ~1.0 x the algorithmic bytes (65 536 segments of BASELINE configs[4]: 0.29 ms and 17.7 MB
 * against 0.30 ms and 54 MB). */
int lto_indirect_plan_rebalance(lto_indirect_plan* plan, void* stream);
int lto_indirect_plan_reset_order(lto_indirect_plan* plan);
/* Warm start of the adaptive step-size controller (ndim = 12, DOP853_ADAPTIVE -- the reference's integrator setting, whose
 * sweeps last as long as their slowest segment; other plans: LTO_EINVAL).  When on, every STM sweep / defect-only sweep of
 * this plan that runs the two-lane kernels starts each segment from the step size that segment's first accepted step had in
 * the plan's PREVIOUS sweep of the same kind, instead of Hairer's start rule (an extra right-hand-side evaluation and a start a
 * decade or two low).  Consecutive Newton iterations and line-search trials sweep nearly the same trajectory, and any positive
 * start is valid: the controller corrects it.  Results then depend on the plan's history at the level of the tolerance
 * (1e-13), which is why it is off by default: with it off, equal inputs give equal bits.  Turning it off forgets the stored sizes. */
int lto_indirect_plan_set_warm_start(lto_indirect_plan* plan, int on);

/* Tuning knobs for the STM sweep.  Kernel: LTO_KERNEL_AUTO picks
 *   - fixed-step RK4 with >= 6 steps per segment: the three-role pipeline kernels -- the eight-wave form (LTO_KERNEL_PIPE8) while
 *     the batch is one round of it (16 segments per CU: 4 096 on MI355X); above that the family whose rounds are cheapest for the
 *     segment count, by the context's cost table (lto_kernel_round_costs / lto_calibrate_kernels above): eight-wave form in
 *     rounds of 16 x CUs segments, 32-segment form (LTO_KERNEL_PIPE32, where it is built) in rounds of 32 x CUs, large-batch form
 *     (LTO_KERNEL_PIPE48) in rounds of 48 x CUs -- 12-dim also 44 x CUs --, and for ndim = 12 the whole-segment lanes
 *     (LTO_KERNEL_LANE) in rounds of 256 x CUs (lto_indirect_auto_kernel below is this rule as a pure function).  On
 *     MI355X (256 CUs, default table): 4 097 ... 8 192 segments -> PIPE32, 8 193 ... 12 288 -> PIPE48, 65 536 and 262 144 -> LANE
 *     (12-dim) / PIPE32 (14-dim);
 *   - RK4 with fewer steps: the per-lane kernel (each lane re-integrates the base state with 1-3 columns); for ndim = 12 on a
 *     full chip the whole-segment forms instead: ONE step per segment and >= 65 536 segments the one-step sweep
 *     (lto_indirect_plan_set_cols_per_lane, 12), 2 ... 5 steps LTO_KERNEL_LANE when its rounds are the cheaper ones;
 *   - the 13-stage integrators: the wave-specialised kernel (LTO_KERNEL_COOP: base wave + column waves per 16 segments,
 *     coefficients handed over through LDS at every RK stage) -- for ndim = 12 with DOP853_ADAPTIVE, the reference's setting, its
 *     form with two lanes per state (LTO_KERNEL_COOP2).
 * Results never depend on the choice beyond round-off; lto_indirect_plan_last_kernel reports what ran.
 * Round 6 removed three dominated forms; their selectors stay valid and resolve to the family that took over: LTO_KERNEL_PER_LANE on
 * a 13-stage plan selects the one-lane DEFECT sweep only (its STM sweep runs the cooperative kernels -- the one-column-per-lane form
 * with memory-resident slopes is gone), LTO_KERNEL_COOP on an RK4 plan runs the pipeline AUTO would take, LTO_KERNEL_COOP on a
 * 12-dim DOP853 plan runs LTO_KERNEL_COOP2. */
#define LTO_KERNEL_AUTO 0
#define LTO_KERNEL_PER_LANE 1
#define LTO_KERNEL_COOP 2
/* Direct plans only (lto_direct_plan_set_kernel): the pipelined Jacobian kernel (base wave + one wave per sensitivity column per 32
 * segments, skewed by one RKF7(8) step); AUTO takes it from 3 072 segments.  On an indirect plan: LTO_EINVAL. */
#define LTO_KERNEL_DIRECT_PIPE 3
/* RK4 plans only (other integrators: LTO_EINVAL): base wave, coefficient wave and column waves per 16 segments run as a software
 * pipeline skewed by one RK4 step.  Four column waves, one STM column per lane, a DPP row = one segment and the
 * coefficients broadcast inside the FMA (v_fmac_f64_dpp row_newbcast), TWO RK4 steps per phase and eight waves -- a fourth of
 * the column work alternates between two SIMDs so that all four SIMDs of a CU carry the same load -- and a base wave that
 * evaluates RK4 stages 1|2 and then 3|4 side by side in neighbouring lanes (one workgroup per CU: 91 KB of LDS). */
#define LTO_KERNEL_PIPE8 5
/* ndim = 12, DOP853_ADAPTIVE plans only: the cooperative kernel with every 12-component state split over two lanes (top /
 * bottom halves of a column in different waves, the two halves of the base state in neighbouring DPP banks): six components
 * per lane keep all slopes of the 13-stage method in addressable registers (round 3: the base state takes a DPP quad, three
 * components per lane).  Other plans: LTO_EINVAL.  The defect-only sweep of such a plan comes with one, two or four lanes per
 * segment (see lto_indirect_plan_set_defect_lanes for what AUTO takes): LTO_KERNEL_PER_LANE and LTO_KERNEL_COOP2 select the
 * first two, lto_indirect_plan_set_defect_lanes any of them. */
#define LTO_KERNEL_COOP2 6
/* RK4 plans only: the pipeline for large batches -- 48 segments and 16 wavefronts per workgroup, the base wave's lanes are 48
 * different segments, twelve column waves with one segment per DPP row; 12-dim also with 44 segments and eleven column waves (the
 * cheaper of the two forms for the segment count runs). */
#define LTO_KERNEL_PIPE48 7
/* 32 segments and twelve wavefronts per workgroup: the eight-wave form's roles (paired-stage base role, four lanes per segment) with one
 * barrier per step.  For batches between one round of LTO_KERNEL_PIPE8 and a few (4 097 ... 8 192 segments on MI355X: one round
 * instead of two).  RK4; 12-dim, and 14-dim with p = 0 or p = 1; anything else: LTO_EINVAL from lto_indirect_plan_set_kernel
 * (AUTO does not consider it there).  Results equal LTO_KERNEL_PIPE8's bit for bit. */
#define LTO_KERNEL_PIPE32 8
/* RK4, ndim = 12 plans only (others: LTO_EINVAL): a lane owns a whole segment -- its base trajectory, the four stage matrices of
 * every step and all twelve STM columns, which it sends through those matrices one after the other (eight columns parked in
 * accumulation registers, four in LDS).  No DPP row with idle lanes, no barrier, no hand-over: ~61 wave-instructions per segment
 * and RK4 step against ~118 of LTO_KERNEL_PIPE48 -- but one wavefront of 64 segments per SIMD, so it only pays once the batch
 * fills the chip: AUTO compares its rounds of 256 x CUs segments (505 us at 64 steps on MI355X) with the pipelines' rounds and
 * takes it from 36 865 segments on MI355X (not for the sizes just above a multiple of a pipeline's smaller round: 65 537 ...
 * 78 848).  The defect equals the pipeline kernels' bit for bit; Phi agrees with theirs to round-off (~1e-15 of max |Phi|: since
 * round 6 the stage matrices come from the base evaluations' by-products, not from a second evaluation of the control law). */
#define LTO_KERNEL_LANE 9
int lto_indirect_plan_set_kernel(lto_indirect_plan* plan, int kernel);
/* What LTO_KERNEL_AUTO resolves to for the STM sweep of a plan of this shape on a device with `n_cus` compute units, by the MI355X
 * cost table (a context's own table after lto_calibrate_kernels may differ): ndim 12 | 14, method LTO_RK4 ..., steps per segment,
 * p the control-law exponent (0, 1, 2 or > 1), n_segments = (n_nodes - 1) x n_batch, ordered = the plan sweeps with a lane order.
 * A pure function: no context, no device -- callable on a host without a GPU (sizing an N-GPU run, tests).  Returns LTO_KERNEL_* or
 * LTO_EINVAL. */
int lto_indirect_auto_kernel(int ndim, int method, int steps, double p, long n_segments, int n_cus, int ordered);
/* Lanes per segment of the DEFECT-ONLY sweep of an ndim = 12 DOP853_ADAPTIVE plan (the reference's setting, indirect.jl:63-90):
 * 1, 2 or 4 (a DPP quad per segment: r, v, lambda_v, lambda_r), or 0 = choose (default).  The choice: by size -- four lanes up to
 * eight wavefronts of 16 segments per SIMD, i.e. 512 x CUs segments (131 072 on MI355X), two lanes up to 262 144 segments, one
 * beyond -- and, from 64 x CUs segments, by the trial-step statistics of an EARLIER defect sweep of the same plan once they are
 * in (taken after the plan's first two sweeps, then every sixteenth): if no segment took more than three times the mean number
 * of trial steps (a line search's trial trajectories) there is no tail worth shortening and fewer lanes issue fewer
 * instructions -- two lanes up to 160 x CUs segments, one above.  The statistics are consumed behind an event, so the choice is
 * a function of the plan's call sequence, not of timing (inside a graph capture the last verdict stands).  All forms take the
 * same step controller (rk.hpp dp8_decide) but sum the error norm in different orders: results agree to round-off of the
 * converged flow (~1e-15), and bit for bit only between sweeps with the same number of lanes.  Two and four lanes on other
 * plans: LTO_EINVAL. */
int lto_indirect_plan_set_defect_lanes(lto_indirect_plan* plan, int lanes);
/* Record staging of a plan's ordered (rebalanced) sweeps, a bit mask: 1 = node and defect records are in place, 2 = Phi records
 * too (only plans that run STM sweeps get them), 4 = an allocation for them failed and staging is off for this plan -- the sweeps
 * then read and write the caller's arrays directly (same results, 3-8 x the HBM traffic); lto_last_error() holds the note. */
int lto_indirect_plan_staging(const lto_indirect_plan* plan);
/* Output layout of a plan's sweeps (round 6).  LTO_LAYOUT_SOA (default): defect [ndim][ldd], Phi [(col*ndim+row)][ldp] -- struct of
 * arrays, segment index fastest.  LTO_LAYOUT_BLOCKS: one block per segment, defect [S][ndim] and Phi [S][ndim*ndim] with the block
 * column-major -- i.e. exactly the reference's (Julia's, column-major) defect[ndim x S] and the Phi_i blocks of jacobianCalc
 * (indirect.jl:121-123), what the host-pointer entry points return; ldd / ldp are then ignored.  Built for ndim = 12
 * DOP853_ADAPTIVE plans (the reference's integrator setting; others: LTO_EUNSUPPORTED): their kernels write a segment's results
 * as one record, so with a lane order (lto_indirect_plan_rebalance) the sweep needs no record arrays of its own and no transposes
 * behind it -- C5 + STM (65 536 segments): 104 instead of 280 MB of HBM traffic per sweep (algorithmic 95), same bits.  The
 * defect-only sweep of such a plan runs with two or four lanes per segment (the one-lane kernel writes struct-of-arrays only);
 * lto_indirect_newton_solve_dev reads struct-of-arrays and refuses such a plan. */
#define LTO_LAYOUT_SOA 0
#define LTO_LAYOUT_BLOCKS 1
int lto_indirect_plan_set_output_layout(lto_indirect_plan* plan, int layout);
/* LTO_KERNEL_* family the last STM sweep of this plan ran (what AUTO resolved to); LTO_KERNEL_AUTO before any sweep. */
int lto_indirect_plan_last_kernel(const lto_indirect_plan* plan);
/* Per-lane kernel only: STM columns integrated per lane (12-dim: 1 or 3, 14-dim: 1 or 2 -- LTO_EUNSUPPORTED for the other
 * grouping; every lane re-integrates the base state with its columns); 0 = choose from S.  cols = the plan's dimension (12 or 14) = the
 * whole STM in the segment's own lane (kernels_indirect_stream.hip): built for RK4 plans with ONE step per segment (LTO_EINVAL
 * otherwise, and for the other dimension's value) -- the HBM-bound corner of the sweep, where the lane of a segment runs the four
 * stage evaluations once and sends the columns through the four stage matrices; 0 chooses it for such plans from 65 536 segments.
 * One kernel per dimension whatever the batch's control laws (the law is chosen per trajectory at run time; round 6). */
int lto_indirect_plan_set_cols_per_lane(lto_indirect_plan* plan, int cols);

/* Newton step of the indirect method solved on the device: delta = -Jac_full \ defect for the block-bidiagonal
 * [Phi_i | -I] system with both end states fixed (src/multiShoot_CRTBP_indirect.jl:123-142, :181-182), by structured
 * orthogonal cyclic reduction.  adjoints_only = 0: the square system of the regular iterations.  adjoints_only = 1:
 * the state columns of every node are masked out (:169-178) and the over-determined system is solved in the
 * least-squares sense, as `\` does.  Phi != NULL factors and solves; Phi == NULL re-uses the stored factorisation
 * of the same variant for a new right-hand side (the second-order-correction re-solve, :190-214).
 * delta is SoA [12][ldx], node-indexed. */
int lto_indirect_newton_solve_dev(lto_indirect_plan* plan, void* stream, const double* Phi, long ldp,
                                  const double* defect, long ldd, int adjoints_only, double* delta, long ldx);
/* y[i] = x[i] + alpha d[i], i < count (trial points X + alpha dX, update accumulation) */
int lto_axpy_dev(lto_ctx* ctx, void* stream, const double* x, const double* d, double alpha, double* y, long count);
/* The n_alpha trial trajectories of lineSearch (src/multiShoot_CRTBP_indirect.jl:227-233) of every trajectory of a batch, one
 * launch: Xt[c*ldt + (b*n_alpha + a)*n_nodes + k] = X[c*ld + b*n_nodes + k] + alphas[a] * delta[c*ld + b*n_nodes + k] for
 * c < ndim, b < n_batch, a < n_alpha, k < n_nodes (SoA, node-indexed; alphas is a DEVICE array).  Together with a plan of
 * n_batch*n_alpha trajectories and lto_defect_norms_dev this is the batched line search for callers that keep their
 * own loop around the device-resident entry points. */
int lto_trial_points_dev(lto_ctx* ctx, void* stream, const double* X, const double* delta, long ld, int ndim, int n_nodes,
                         int n_batch, int n_alpha, const double* alphas, double* Xt, long ldt);
/* The per-iteration read-back of a Newton loop kept around the device-resident entry points: out[0..na) = a[..], out[na..na+nb) =
 * b[..] (device arrays; b may be NULL with nb = 0), returning when the values have arrived.  One small kernel writes them into a
 * page-locked block of the context behind everything queued on `stream`, and the host polls a sequence word -- no copy-engine
 * operation and no stream synchronisation; falls back to copies + synchronisation when the block cannot be mapped.  Not
 * thread-safe per context; `stream` must not be capturing. */
int lto_read_scalars_dev(lto_ctx* ctx, void* stream, const double* a, int na, const double* b, int nb, double* out);
/* lineSearch's decision (indirect.jl:244-245) and the defect check that follows the update (:328-331), without another sweep.  For
 * every trajectory b < n_batch: a* = first minimiser of sumsq[b*n_alpha .. ) (NaN trials never win), step[b] = alphas[a*];
 * maxabs_out[b] = maxabs[b*n_alpha + a*]; defect[c*ldd + b*seg + i] = trial_defect[c*ldt + (b*n_alpha + a*)*seg + i].  The updated
 * trajectory XC_all + xc_update*alpha with its end states pinned (:304, :324-325) is the chosen trial point bit for bit (one fma
 * each, the update's end-state rows are zero), so `defectCalc` at it is the part of the line search's own sweep that integrated it.
 * sumsq / maxabs as lto_defect_norms_dev leaves them for the n_batch*n_alpha trial trajectories; alphas is a DEVICE array.
 * maxabs with maxabs_out and trial_defect with defect may be NULL in pairs. */
int lto_line_search_pick_dev(lto_ctx* ctx, void* stream, const double* sumsq, const double* maxabs, const double* alphas, int n_alpha,
                             const double* trial_defect, long ldt, int ndim, int seg_per_traj, int n_batch, double* step,
                             double* maxabs_out, double* defect, long ldd);

/* Dense output (device): segment s is sampled at t_samples[first[s] .. first[s+1]) (sorted, inside the segment);
 * Y[c*ldy + j] = x_c(t_samples[j]); final_state[c*n_batch + b] (or NULL) = x(t_n) of trajectory b.  Built for what densify
 * (src/HelperFunctions.jl:51-101) needs -- ndim = 12 with LTO_DOP853_ADAPTIVE (for its Vern8) -- and for LTO_RK4; other plans:
 * LTO_EUNSUPPORTED (round 6 removed the 14-dim and RKF7(8) instantiations, which nothing ran). */
int lto_indirect_dense_dev(lto_indirect_plan* plan, void* stream, const double* X, long ldx, const double* t,
                           int n_tgrids, const int* first, const double* t_samples, double* Y, long ldy,
                           double* final_state);

int lto_direct_plan_create(lto_ctx* ctx, int nstate, int n_nodes, int n_batch, int nsteps,
                           const lto_direct_params* prm, lto_direct_plan** out);
void lto_direct_plan_destroy(lto_direct_plan* plan);
/* Jacobian kernel: LTO_KERNEL_PER_LANE (each lane re-integrates the half-arc with one sensitivity column) or
 * LTO_KERNEL_PIPE (base wave + one wave per sensitivity column for 32 segments, skewed by one RKF7(8) step: one barrier
 * per step).  AUTO = PIPE from 3 072 segments, PER_LANE below.  LTO_KERNEL_COOP (the wave-specialised form of rounds 1-2,
 * one barrier per RK stage, never the fastest) was removed in round 3: LTO_EINVAL. */
int lto_direct_plan_set_kernel(lto_direct_plan* plan, int kernel);
int lto_direct_defect_dev(lto_direct_plan* plan, void* stream, const double* X, long ldx, const double* U, long ldu,
                          const double* t, int n_tgrids, double* defect, long ldd, double* errors);
/* x_mid[c*ldm + s] = forward half-arc end state of segment s (see lto_direct_midpoints); defect/errors optional. */
int lto_direct_midpoints_dev(lto_direct_plan* plan, void* stream, const double* X, long ldx, const double* U, long ldu,
                             const double* t, int n_tgrids, double* x_mid, long ldm, double* defect, long ldd,
                             double* errors);
int lto_direct_jacobian_dev(lto_direct_plan* plan, void* stream, const double* X, long ldx, const double* U,
                            long ldu, const double* t, int n_tgrids, double* Jac, long ldj, double* dtf,
                            double* defect, long ldd, double* errors);

/* Layout kernels: Julia column-major [ndim x count] (node-contiguous) <-> SoA [ndim][ld]. */
int lto_pack_soa_dev(lto_ctx* ctx, void* stream, const double* aos, int ndim, long count, double* soa, long ld);
int lto_unpack_soa_dev(lto_ctx* ctx, void* stream, const double* soa, long ld, int ndim, long count, double* aos);

/* Per-trajectory reductions the drivers take of a defect array (line search cost sum(defect.^2),
 * src/multiShoot_CRTBP_indirect.jl:240; convergence test norm(defect[:], Inf), :331):
 *   sumsq[b], maxabs[b] for b < n_batch over the ndim x seg_per_traj block of trajectory b. */
int lto_defect_norms_dev(lto_ctx* ctx, void* stream, const double* defect, long ldd, int ndim, int seg_per_traj,
                         int n_batch, double* sumsq, double* maxabs);

/* ------------------------------------------------- several GPUs behind one host process
 * For a single-process host (the Julia drivers) that owns more than one GPU.  A group holds one context per entry of
 * device_ids (ids may repeat).  Each call is split into contiguous shards -- whole trajectories when n_batch > 1,
 * otherwise segment blocks of the one trajectory with a one-node halo (segment i reads nodes i and i+1 only:
 * multiShoot_CRTBP_indirect.jl:71-86, multiShoot_CRTBP_direct.jl:77-105) -- and every shard runs the single-device entry
 * point of the same name on its own host thread.  Arguments, layouts, results and error codes are those of
 * lto_indirect_defect / lto_indirect_jacobian / lto_direct_defect / lto_direct_jacobian; each shard writes its own
 * contiguous slab of the caller's column-major outputs, so there is no collective.  (Multi-process hosts shard the
 * same way with one lto_ctx per process: bench.py, lowthrustopt_amd/sharding.py.) */
typedef struct lto_group lto_group;
int lto_group_create(int n_devices, const int* device_ids, lto_group** out);
void lto_group_destroy(lto_group* group);
const char* lto_group_last_error(const lto_group* group);
int lto_group_size(const lto_group* group);
lto_ctx* lto_group_ctx(lto_group* group, int k); /* member k's context (owned by the group): device-resident plans and sweeps per GPU */
int lto_group_indirect_defect(lto_group* group, int ndim, int n_nodes, int n_batch, const double* XC, const double* t,
                              int n_tgrids, const lto_params* prm, int n_prm, const lto_integrator* integ, double* defect,
                              double* errors);
int lto_group_indirect_jacobian(lto_group* group, int ndim, int n_nodes, int n_batch, const double* XC, const double* t,
                                int n_tgrids, const lto_params* prm, int n_prm, const lto_integrator* integ, double* Phi,
                                double* defect);
int lto_group_direct_defect(lto_group* group, int nstate, int n_nodes, int n_batch, const double* X, const double* U,
                            const double* t, int n_tgrids, int nsteps, const lto_direct_params* prm, double* defect,
                            double* errors);
int lto_group_direct_jacobian(lto_group* group, int nstate, int n_nodes, int n_batch, const double* X, const double* U,
                              const double* t, int n_tgrids, int nsteps, const lto_direct_params* prm, double* Jac_temp,
                              double* ddefect_dtf, double* defect, double* errors);

/* ------------------------------------------------------------------------------- collectives (RCCL over xGMI)
 * The one exchange step of the path.  A sweep shards with no data-path collective (segment i needs nodes i, i+1 only:
 * multiShoot_CRTBP_indirect.jl:71-86); what every rank needs afterwards is the full defect vector or its norms for the
 * convergence test and the line-search decision (indirect.jl:240 sum(defect.^2), :331 norm(defect, Inf)).  These entry
 * points do that device to device -- nothing returns to the host between sweep and decision:
 *   sweep (lto_*_dev) -> lto_defect_norms_dev on the local slab -> lto_comm_allreduce_dev(SUM / MAX), or
 *   sweep -> lto_comm_allgather_dev of the slabs -> lto_defect_norms_dev on the gathered vector.
 * RCCL is bound at run time (dlopen): without it these calls return LTO_EUNSUPPORTED and everything else works. */
#define LTO_COMM_ID_BYTES 128
#define LTO_COMM_SUM 0
#define LTO_COMM_MAX 1
typedef struct lto_comm lto_comm;
int lto_comm_available(void);
/* One process per GPU: one rank calls lto_comm_unique_id, the launcher (torch.distributed, MPI, a file) hands the 128
 * bytes to every rank, every rank calls lto_comm_create (collective: ncclCommInitRank on the context's device). */
int lto_comm_unique_id(void* id128);
int lto_comm_create(lto_ctx* ctx, int world, int rank, const void* id128, lto_comm** out);
void lto_comm_destroy(lto_comm* comm);
const char* lto_comm_last_error(const lto_comm* comm);
int lto_comm_size(const lto_comm* comm);
int lto_comm_rank(const lto_comm* comm);
/* Number of ranks RCCL itself reports for this communicator (ncclCommCount): world for an lto_comm_create communicator, 0 for a
 * window communicator (no RCCL behind it), LTO_ENULL / LTO_EUNSUPPORTED (< 0) when it cannot be asked.  What a scaling run quotes as "RCCL saw N ranks". */
int lto_comm_rccl_ranks(const lto_comm* comm);
/* recv [world][count] <- send [count] of every rank (equal counts); asynchronous on `stream` (hipStream_t). */
int lto_comm_allgather_dev(lto_comm* comm, void* stream, const double* send, double* recv, long count);
/* buf [count] <- LTO_COMM_SUM / LTO_COMM_MAX over ranks, in place; asynchronous on `stream`.  Both propagate NaN (the max
 * carries a NaN indicator per element), so a NaN on one rank reaches every rank: norm(defect, Inf) of indirect.jl:330. */
int lto_comm_allreduce_dev(lto_comm* comm, void* stream, double* buf, long count, int op);
/* Second transport for the same two collectives, without RCCL and without compute units for the payload ("windows"): every
 * rank owns a receive window in device memory, its peers map it through HIP IPC, a rank pushes its slab into every window
 * with device copies and raises a sequence flag there; the consumer's stream waits on the flags with a one-wavefront kernel
 * (bounded: a rank that never arrives turns the result into NaN instead of hanging the stream).  It is also the transport for
 * two ranks that share one device, which RCCL refuses.  Set-up, every rank:
 *   lto_comm_window_export(ctx, world, rank, max_count, handle, &comm)   handle: LTO_COMM_WINDOW_BYTES bytes
 *   the launcher gathers the world handles in rank order (torch.distributed all_gather, MPI_Allgather, a file)
 *   lto_comm_window_open(comm, all_handles)
 * then lto_comm_allgather_dev / lto_comm_allreduce_dev with count <= max_count, lto_comm_destroy at the end (after a
 * barrier of the launcher: a peer may still be pushing).  One process per rank.
 * ONE STREAM: the collectives of a window communicator must all be enqueued on the same stream (the first one it is used on; a
 * different one returns LTO_EINVAL) -- the reuse of a window half, the push counters and the sequence numbers are ordered by it.
 * A driver that gathers on a side stream and reduces norms on its main stream uses two communicators.
 * FAILURE: a wait that runs out (lto_comm_set_wait_limit polls of ~1.5 us each, default 4e6) or a peer found two or more
 * collectives ahead (the ranks have lost step) sets the communicator's fail word: that collective and every later one of this
 * rank return NaN -- never a slab of another iteration -- and lto_comm_status reports it to the host. */
#define LTO_COMM_WINDOW_BYTES 128
int lto_comm_status(lto_comm* comm, void* stream, int* failed);
int lto_comm_set_wait_limit(lto_comm* comm, long polls);
/* Window transport: payloads of up to `bytes` per rank travel by the push / collect KERNELS (default 4 MiB: a kernel after a
 * kernel costs ~2 us of queue hand-over, a copy-engine operation between kernels ~10 us), larger ones by the copy engines with a
 * one-wavefront wait kernel; 0 = always the copy engines.  Every block of the collect kernel polls for its peers' flags, which is
 * free when each rank owns its GPU (the deployment) and starves the peers' push kernels when several ranks SHARE one device
 * and the payload needs thousands of blocks: rehearsals on a shared device lower this (bench.py, LTO_BENCH_SHARE_DEVICE). */
int lto_comm_set_kernel_payload(lto_comm* comm, long bytes);
int lto_comm_window_export(lto_ctx* ctx, int world, int rank, long max_count, void* handle_out, lto_comm** out);
int lto_comm_window_open(lto_comm* comm, const void* all_handles);
int lto_comm_uses_windows(const lto_comm* comm);
/* One host process, several GPUs: the communicators of an lto_group (ncclCommInitAll over its devices).  send[k] /
 * recv[k] / buf[k] live on member k's device; the work is enqueued on member k's context stream (lto_ctx_stream), after
 * the sweep that produced send[k] there.  A group that repeats one device (1-GPU boxes) uses device copies instead. */
typedef struct lto_group_comm lto_group_comm;
/* all-gather payloads up to this many bytes per member go by peer copies ordered by events (copy engines over xGMI, no
 * compute unit taken from the sweeps) even when the group has an RCCL clique; larger ones through RCCL */
#define LTO_GROUP_PEER_COPY_BYTES (1 << 20)
int lto_group_comm_create(lto_group* group, lto_group_comm** out);
void lto_group_comm_destroy(lto_group_comm* comm);
const char* lto_group_comm_last_error(const lto_group_comm* comm);
int lto_group_comm_uses_rccl(const lto_group_comm* comm);
int lto_group_comm_allgather_dev(lto_group_comm* comm, const double* const* send, double* const* recv, long count);
int lto_group_comm_allreduce_dev(lto_group_comm* comm, double* const* buf, long count, int op);

#ifdef __cplusplus
}
#endif
#endif /* LTO_H */
