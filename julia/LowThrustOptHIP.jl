# LowThrustOptHIP.jl -- Julia binding of liblto_hip.so (include/lto.h) for LowThrustOpt's shooting drivers.
#
# What it replaces: the bodies of the four closures nested in the reference drivers
#   multiShoot_CRTBP_indirect:  defectCalc (src/multiShoot_CRTBP_indirect.jl:63-90), jacobianCalc (:93-146)
#   multiShoot_CRTBP_direct:    defectCalc (src/multiShoot_CRTBP_direct.jl:66-109),  jacobianCalc (:111-166),
#                               tf partial (:503-516)
# The drivers' signatures, return tuples and inner array shapes are unchanged (INTEGRATION.md shows the patch).
#
# STATUS: there is no `julia` binary in the build image or on the GPU box, so this file has never been
# executed.  It is kept deliberately thin: every ccall below has a line-for-line twin in
# lowthrustopt_amd/_lib.py + hotpath.py (ctypes), and THAT twin is what the test-suite drives on the GPU.
module LowThrustOptHIP

using SparseArrays, LinearAlgebra, Libdl

export LtoIndirectPlan, LtoDirectPlan, LtoComm, LtoCommWindows, pinned_array, pack_soa!, unpack_soa!, defect_norms!, indirect_defect_dev!,
       indirect_jacobian_dev!, newton_solve_dev!, axpy_dev!, direct_defect_dev!, direct_jacobian_dev!, rebalance!, set_kernel!, set_warm_start!, set_defect_lanes!,
       comm_unique_id, allgather_dev!, allreduce_dev!, ctx_stream, last_call_ms
export LtoContext, LtoGroup, indirect_defectCalc, indirect_jacobianCalc, indirect_stm, indirect_newton_step, indirect_solve, indirect_solve_batch, densify,
       direct_defectCalc, direct_jacobianCalc, direct_midpoints, LTO_RK4, LTO_RKF78_FIXED, LTO_RKF78_ADAPTIVE, LTO_DOP853_ADAPTIVE

const liblto = get(ENV, "LTO_HIP_LIB", joinpath(@__DIR__, "..", "lowthrustopt_amd", "liblto_hip.so"))

const LTO_RK4 = Cint(0)
const LTO_RKF78_FIXED = Cint(1)
const LTO_RKF78_ADAPTIVE = Cint(2)
const LTO_DOP853_ADAPTIVE = Cint(3)
# error codes (include/lto.h): < 0 = misuse, > 0 = run time
const LTO_EINVAL, LTO_ENULL, LTO_EUNSUPPORTED = Cint(-1), Cint(-2), Cint(-3)
const LTO_EHIP, LTO_EBADP, LTO_ENODEVICE, LTO_ENOMEM = Cint(1), Cint(2), Cint(3), Cint(4)
# kernel families of an indirect plan's STM sweep (lto_indirect_plan_set_kernel; 0 = let the library choose)
const LTO_KERNEL_AUTO, LTO_KERNEL_PER_LANE, LTO_KERNEL_COOP, LTO_KERNEL_DIRECT_PIPE = Cint(0), Cint(1), Cint(2), Cint(3)
const LTO_KERNEL_PIPE8, LTO_KERNEL_COOP2, LTO_KERNEL_PIPE48, LTO_KERNEL_PIPE32, LTO_KERNEL_LANE = Cint(5), Cint(6), Cint(7), Cint(8), Cint(9)
const LTO_LAYOUT_SOA, LTO_LAYOUT_BLOCKS = Cint(0), Cint(1)

# isbits mirrors of the C structs (include/lto.h)
struct LtoIntegrator
    method::Cint
    steps::Cint
    rtol::Cdouble
    atol::Cdouble
    max_steps::Cint
end
# default = the reference's Vern8() setting: adaptive order-8 pair, reltol = abstol = 1e-13 (indirect.jl:79)
LtoIntegrator() = LtoIntegrator(LTO_DOP853_ADAPTIVE, 0, 1e-13, 1e-13, 0)

struct LtoParams          # the `params` tuple of indirect.jl:260, field for field
    MU::Cdouble; DU::Cdouble; TU::Cdouble; thrustLimit::Cdouble
    mass::Cdouble; time_direction::Cdouble; p::Cdouble; rho::Cdouble
end
LtoParams(t::Tuple) = LtoParams(map(Float64, t)...)

struct LtoDirectParams
    MU::Cdouble; DU::Cdouble; TU::Cdouble; Isp::Cdouble
end

mutable struct LtoContext
    handle::Ptr{Cvoid}
    function LtoContext(device::Integer = 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:lto_create, liblto), Cint, (Ref{Ptr{Cvoid}}, Cint), h, device)
        rc == 0 || error("lto_create failed with code $rc (no gfx950 device?)")
        ctx = new(h[])
        finalizer(c -> (c.handle == C_NULL || ccall((:lto_destroy, liblto), Cvoid, (Ptr{Cvoid},), c.handle); c.handle = C_NULL), ctx)
        ctx
    end
end

"""Several GPUs behind this one Julia process (lto_group_*, include/lto.h): `LtoGroup([0, 1, 2, 3])`.  Accepted wherever
the four sweep closures take an `LtoContext`; each sweep is split into contiguous shards (segment blocks with a
one-node halo, or whole trajectories of a batch), one host thread and one device context per entry."""
mutable struct LtoGroup
    handle::Ptr{Cvoid}
    function LtoGroup(devices::Vector{<:Integer})
        h = Ref{Ptr{Cvoid}}(C_NULL)
        ids = Cint.(devices)
        rc = ccall((:lto_group_create, liblto), Cint, (Cint, Ptr{Cint}, Ref{Ptr{Cvoid}}), length(ids), ids, h)
        rc == 0 || error("lto_group_create failed with code $rc")
        g = new(h[])
        finalizer(c -> (c.handle == C_NULL || ccall((:lto_group_destroy, liblto), Cvoid, (Ptr{Cvoid},), c.handle); c.handle = C_NULL), g)
        g
    end
end

const LtoHandle = Union{LtoContext, LtoGroup}
const _libhandle = Ref{Ptr{Cvoid}}(C_NULL)
libhandle() = (_libhandle[] == C_NULL && (_libhandle[] = Libdl.dlopen(liblto)); _libhandle[])
# entry point of a sweep: lto_<name> for one context, lto_group_<name> for a group (identical argument lists)
entry(::LtoContext, name::Symbol) = Libdl.dlsym(libhandle(), Symbol("lto_", name))
entry(::LtoGroup, name::Symbol) = Libdl.dlsym(libhandle(), Symbol("lto_group_", name))
last_error(ctx::LtoContext) = unsafe_string(ccall((:lto_last_error, liblto), Cstring, (Ptr{Cvoid},), ctx.handle))
last_error(g::LtoGroup) = unsafe_string(ccall((:lto_group_last_error, liblto), Cstring, (Ptr{Cvoid},), g.handle))

function check(ctx::LtoHandle, rc::Cint)
    rc == 0 && return
    msg = last_error(ctx)
    # code 2 is the reference's own error("Invalid value of p!") (CRTBP_stateCostate_deriv.jl:52); code 4 = host memory / threads ran out inside the call
    rc == LTO_ENOMEM && throw(OutOfMemoryError())
    error(rc == LTO_EBADP ? msg : "lto error $rc: $msg")
end

# ---------------------------------------------------------------------------------------------- indirect
"defectCalc of multiShoot_CRTBP_indirect: returns (defect1[2nstate x (n_nodes-1)], errors[n_nodes-1])."
function indirect_defectCalc(ctx::LtoHandle, XC_all::Matrix{Float64}, t_TU::Vector{Float64}, params;
                             integ::LtoIntegrator = LtoIntegrator())
    ndim, n_nodes = size(XC_all)
    defect1 = zeros(ndim, n_nodes - 1)
    errors = zeros(n_nodes - 1)
    prm = Ref(LtoParams(params))
    rc = ccall(entry(ctx, :indirect_defect), Cint,
               (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ref{LtoParams}, Cint, Ref{LtoIntegrator},
                Ptr{Cdouble}, Ptr{Cdouble}),
               ctx.handle, ndim, n_nodes, 1, XC_all, t_TU, 1, prm, 1, Ref(integ), defect1, errors)
    check(ctx, rc)
    (defect1, errors)
end

"Compact Jacobian blocks Phi[2nstate x 2nstate x (n_nodes-1)] (= ForwardDiff.jacobian(f, x0) of indirect.jl:121) and the defect."
function indirect_stm(ctx::LtoHandle, XC_all::Matrix{Float64}, t_TU::Vector{Float64}, params;
                      integ::LtoIntegrator = LtoIntegrator())
    ndim, n_nodes = size(XC_all)
    Phi = zeros(ndim, ndim, n_nodes - 1)
    defect1 = zeros(ndim, n_nodes - 1)
    prm = Ref(LtoParams(params))
    rc = ccall(entry(ctx, :indirect_jacobian), Cint,
               (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ref{LtoParams}, Cint, Ref{LtoIntegrator},
                Ptr{Cdouble}, Ptr{Cdouble}),
               ctx.handle, ndim, n_nodes, 1, XC_all, t_TU, 1, prm, 1, Ref(integ), Phi, defect1)
    check(ctx, rc)
    (Phi, defect1)
end

"""jacobianCalc of multiShoot_CRTBP_indirect: Jac_full [2nstate(n_nodes-1) x 2nstate*n_nodes], row block i =
[Phi_i | -I] at columns 2nstate(i-1)+(1:4nstate), with the columns of the two fixed end states zeroed
(indirect.jl:123-142).  Returned sparse: the driver's least-squares step sparsifies it anyway (:181)."""
function indirect_jacobianCalc(ctx::LtoHandle, XC_all, t_TU, nstate, n_nodes, params; integ = LtoIntegrator())
    (Phi, _) = indirect_stm(ctx, XC_all, t_TU, params; integ = integ)
    nd = 2 * nstate
    S = n_nodes - 1
    I_idx = Int[]; J_idx = Int[]; V = Float64[]
    for i = 1:S, c = 1:nd, r = 1:nd
        push!(I_idx, (i - 1) * nd + r); push!(J_idx, (i - 1) * nd + c); push!(V, Phi[r, c, i])
    end
    for i = 1:S, r = 1:nd
        push!(I_idx, (i - 1) * nd + r); push!(J_idx, i * nd + r); push!(V, -1.0)
    end
    Jac_full = sparse(I_idx, J_idx, V, nd * S, nd * n_nodes)
    Jac_full[:, 1:nstate] .= 0
    Jac_full[:, (nd * n_nodes - 2 * nstate + 1):(nd * n_nodes - nstate)] .= 0
    dropzeros!(Jac_full)
end

"""One Newton iteration on the device (indirect.jl:290-296): jacobianCalc, the least-squares step of
optimizeTraj_OLS (incl. the flag_adjointsOnly column mask) and its second-order correction; returns (xc_update, defect)."""
function indirect_newton_step(ctx::LtoContext, XC_all::Matrix{Float64}, t_TU::Vector{Float64}, params;
                              integ::LtoIntegrator = LtoIntegrator(), flag_adjointsOnly::Bool = false,
                              soc_threshold::Float64 = 1e-1)
    ndim, n_nodes = size(XC_all)
    xc_update = zeros(ndim, n_nodes)
    defect1 = zeros(ndim, n_nodes - 1)
    rc = ccall((:lto_indirect_newton_step, liblto), Cint,
               (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ref{LtoParams}, Cint, Ref{LtoIntegrator},
                Cint, Cdouble, Ptr{Cdouble}, Ptr{Cdouble}),
               ctx.handle, ndim, n_nodes, 1, XC_all, t_TU, 1, Ref(LtoParams(params)), 1, Ref(integ),
               flag_adjointsOnly ? 1 : 0, soc_threshold, xc_update, defect1)
    check(ctx, rc)
    (xc_update, defect1)
end

"densify (src/HelperFunctions.jl:51-101): (XC_dense[ndim x n_desired], t_dense)."
function densify(ctx::LtoContext, XC_all::Matrix{Float64}, t_TU::Vector{Float64}, params, n_desired::Integer;
                 integ::LtoIntegrator = LtoIntegrator())
    ndim, n_nodes = size(XC_all)
    XC_dense = zeros(ndim, n_desired)
    t_dense = zeros(n_desired)
    rc = ccall((:lto_indirect_densify, liblto), Cint,
               (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ref{LtoParams}, Ref{LtoIntegrator}, Cint, Ptr{Cdouble},
                Ptr{Cdouble}),
               ctx.handle, ndim, n_nodes, XC_all, t_TU, Ref(LtoParams(params)), Ref(integ), n_desired, XC_dense, t_dense)
    check(ctx, rc)
    (XC_dense, t_dense)
end

"""The whole Newton loop of multiShoot_CRTBP_indirect (indirect.jl:254-345) in one library call, trajectory resident on
the GPU: returns (XC_all, defect, status_flag) exactly as the reference driver does, so
`multiShoot_CRTBP_indirect(XC_all, t_TU, MU, DU, TU, n_nodes, mass0, thrustLimit, plot_yn, flag_adjointsOnly, maxIter, p, rho)`
can forward to `indirect_solve(LTO, XC_all, t_TU, (MU, DU, TU, thrustLimit, mass0, 1.0, p, rho), flag_adjointsOnly, maxIter)`
when `plot_yn` is false."""
function indirect_solve(ctx::LtoContext, XC_all::Matrix{Float64}, t_TU::Vector{Float64}, params, flag_adjointsOnly::Bool,
                        maxIter::Integer; integ::LtoIntegrator = LtoIntegrator(), verbose::Bool = true)
    ndim, n_nodes = size(XC_all)
    XC_new = zeros(ndim, n_nodes)
    defect1 = zeros(ndim, n_nodes - 1)
    history = fill(NaN, 2, max(maxIter, 1))
    status = Ref{Cint}(0); iters = Ref{Cint}(0)
    rc = ccall((:lto_indirect_solve, liblto), Cint,
               (Ptr{Cvoid}, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ref{LtoParams}, Ref{LtoIntegrator}, Cint, Cint,
                Ptr{Cdouble}, Ptr{Cdouble}, Ref{Cint}, Ref{Cint}, Ptr{Cdouble}),
               ctx.handle, ndim, n_nodes, XC_all, t_TU, Ref(LtoParams(params)), Ref(integ), flag_adjointsOnly ? 1 : 0, maxIter,
               XC_new, defect1, status, iters, history)
    check(ctx, rc)
    if verbose
        for k = 1:size(history, 2)
            isnan(history[2, k]) && break
            println("Iter $k. Max defect = $(history[1, k]). α = $(history[2, k]).")
        end
        status[] == 1 && println("Reached max iteration count at $(iters[]) iterations")
    end
    (XC_new, defect1, Int(status[]))
end

"""n_batch independent Newton loops side by side (`lto_indirect_solve_batch`): `XC_all` [12 x n_nodes x n_batch], `t_TU`
[n_nodes] (shared grid), `params` a vector of n_batch parameter tuples (e.g. one rho per level of a continuation ladder).
Returns (XC_all, defect, status_flags, iterCounts)."""
function indirect_solve_batch(ctx::LtoContext, XC_all::Array{Float64,3}, t_TU::Vector{Float64}, params::Vector,
                              flag_adjointsOnly::Bool, maxIter::Integer; integ::LtoIntegrator = LtoIntegrator())
    ndim, n_nodes, n_batch = size(XC_all)
    prm = [LtoParams(q) for q in params]
    XC_new = zeros(ndim, n_nodes, n_batch)
    defect1 = zeros(ndim, n_nodes - 1, n_batch)
    status = zeros(Cint, n_batch); iters = zeros(Cint, n_batch)
    rc = ccall((:lto_indirect_solve_batch, liblto), Cint,
               (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Ptr{LtoParams}, Cint, Ref{LtoIntegrator}, Cint, Cint,
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cint}, Ptr{Cdouble}),
               ctx.handle, ndim, n_nodes, n_batch, XC_all, t_TU, 1, prm, length(prm), Ref(integ), flag_adjointsOnly ? 1 : 0, maxIter,
               XC_new, defect1, status, iters, C_NULL)
    check(ctx, rc)
    (XC_new, defect1, Int.(status), Int.(iters))
end

# ---------------------------------------------------------------------------------------------- direct
"defectCalc of multiShoot_CRTBP_direct: returns (defect1[nstate x (n_nodes-1)], errors[n_nodes-1])."
function direct_defectCalc(ctx::LtoHandle, X_all::Matrix{Float64}, u_all::Matrix{Float64}, t_TU::Vector{Float64},
                           nstate, n_nodes, nsteps, Isp, MU, DU, TU)
    defect1 = zeros(nstate, n_nodes - 1)
    errors = zeros(n_nodes - 1)
    prm = Ref(LtoDirectParams(MU, DU, TU, Isp))
    rc = ccall(entry(ctx, :direct_defect), Cint,
               (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Ref{LtoDirectParams},
                Ptr{Cdouble}, Ptr{Cdouble}),
               ctx.handle, nstate, n_nodes, 1, X_all, u_all, t_TU, 1, nsteps, prm, defect1, errors)
    check(ctx, rc)
    (defect1, errors)
end

"""Mid-point propagation of meshRefine_direct (direct.jl:645-656) for every segment in one sweep: returns
(x_mid[nstate x (n_nodes-1)], defect1, errors).  `nsteps = 2` reproduces the reference's single `ode7` step."""
function direct_midpoints(ctx::LtoContext, X_all::Matrix{Float64}, u_all::Matrix{Float64}, t_TU::Vector{Float64},
                          nstate, n_nodes, nsteps, Isp, MU, DU, TU)
    x_mid = zeros(nstate, n_nodes - 1)
    defect1 = zeros(nstate, n_nodes - 1)
    errors = zeros(n_nodes - 1)
    prm = Ref(LtoDirectParams(MU, DU, TU, Isp))
    rc = ccall((:lto_direct_midpoints, liblto), Cint,
               (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Ref{LtoDirectParams},
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
               ctx.handle, nstate, n_nodes, 1, X_all, u_all, t_TU, 1, nsteps, prm, x_mid, defect1, errors)
    check(ctx, rc)
    (x_mid, defect1, errors)
end

"""jacobianCalc + tf partial of multiShoot_CRTBP_direct: Jac_full [nstate(n_nodes-1) x n_nodes(nstate+3)+1]
(state columns node-major, then control columns, then tf: direct.jl:146-162, :516).  The blocks come from the
variational equations integrated on the GPU, not from 18 perturbed re-propagations per segment."""
function direct_jacobianCalc(ctx::LtoHandle, X_all::Matrix{Float64}, u_all::Matrix{Float64}, t_TU::Vector{Float64},
                             nstate, n_nodes, nsteps, Isp, MU, DU, TU)
    nvar = 2 * (nstate + 3)
    S = n_nodes - 1
    Jac_temp = zeros(nstate, nvar, S)
    ddefect_dt = zeros(nstate, S)
    defect1 = zeros(nstate, S)
    errors = zeros(S)
    prm = Ref(LtoDirectParams(MU, DU, TU, Isp))
    rc = ccall(entry(ctx, :direct_jacobian), Cint,
               (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Cint, Cint, Ref{LtoDirectParams},
                Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}, Ptr{Cdouble}),
               ctx.handle, nstate, n_nodes, 1, X_all, u_all, t_TU, 1, nsteps, prm, Jac_temp, ddefect_dt, defect1, errors)
    check(ctx, rc)
    Jac_full = zeros(nstate * S, n_nodes * (nstate + 3) + 1)
    for i = 1:S
        rows = (i - 1) * nstate + 1 : i * nstate
        Jac_full[rows, (i - 1) * nstate + 1 : (i + 1) * nstate] = Jac_temp[:, 1:2 * nstate, i]
        Jac_full[rows, nstate * n_nodes + 3 * (i - 1) + 1 : nstate * n_nodes + 3 * (i - 1) + 6] = Jac_temp[:, 2 * nstate + 1 : end, i]
    end
    Jac_full[:, end] = ddefect_dt[:]
    (Jac_full, defect1, errors)
end

# ------------------------------------------------------------------------------------------- device-resident API
# The operands stay in HBM between calls (struct-of-arrays, see include/lto.h "device-resident API"); every function
# below takes RAW DEVICE POINTERS (`Ptr{Cvoid}`) and a `hipStream_t` (`Ptr{Cvoid}`, C_NULL = HIP's default stream) and
# returns as soon as the work is enqueued.  With AMDGPU.jl: `p = Ptr{Cvoid}(UInt(pointer(A)))` for a `ROCArray{Float64}`
# `A`, `AMDGPU.stream()` for the stream; without it, device memory can come from any HIP allocation in the process.
# This is how a Julia Newton loop keeps the trajectory on the GPU: pack once (`pack_soa!`), then per iteration
# `indirect_jacobian_dev!` -> `newton_solve_dev!` -> `axpy_dev!` -> `indirect_defect_dev!` -> `defect_norms!`, copying two
# scalars per trajectory back.  ctypes twins (executed by the GPU tests): hotpath.IndirectPlan / DirectPlan / pack_soa /
# unpack_soa / defect_norms / Comm / Context.pinned_empty.
const DevPtr = Ptr{Cvoid}
devptr(p::Ptr) = convert(Ptr{Cvoid}, p)
devptr(p::Integer) = Ptr{Cvoid}(UInt(p))
devptr(::Nothing) = C_NULL

"hipStream_t owned by the context (non-blocking); pass it as `stream` to keep library work off the default stream."
ctx_stream(ctx::LtoContext) = ccall((:lto_ctx_stream, liblto), Ptr{Cvoid}, (Ptr{Cvoid},), ctx.handle)
"Wall time [ms] of the last host-pointer call on `ctx`, entry to return, as measured inside the library."
last_call_ms(ctx::LtoContext) = ccall((:lto_last_call_ms, liblto), Cdouble, (Ptr{Cvoid},), ctx.handle)
"Lane order of the last host-pointer indirect call: 0 natural, 1 global, 2 windowed."
last_call_order(ctx::LtoContext) = Int(ccall((:lto_last_call_order, liblto), Cint, (Ptr{Cvoid},), ctx.handle))

"Measure the cost table LTO_KERNEL_AUTO chooses the RK4 STM kernel family by (microseconds per round) on this context's device."
calibrate_kernels!(ctx::LtoContext) = check(ctx, ccall((:lto_calibrate_kernels, liblto), Cint, (Ptr{Cvoid},), ctx.handle))

"Microseconds per round (256 x CUs segments, 64 steps) of the whole-segment lanes (LTO_KERNEL_LANE, 12-dim): the default or this device's (calibrate_kernels!)."
kernel_lane_round_us(ctx::LtoContext) = ccall((:lto_kernel_lane_round_us, liblto), Cdouble, (Ptr{Cvoid},), ctx.handle)

"(`[pipeline8, pipeline48 with 48 segments per workgroup, per-lane, pipeline48 with 44, pipeline32]` microseconds per round at 64 steps, calibrated?) for `ndim` = 12 or 14."
function kernel_round_costs(ctx::LtoContext, ndim::Integer)
    us = zeros(Float64, 5)
    cal = Ref{Cint}(0)
    check(ctx, ccall((:lto_kernel_round_costs, liblto), Cint, (Ptr{Cvoid}, Cint, Ptr{Cdouble}, Ptr{Cint}), ctx.handle, ndim, us, cal))
    return us, cal[] != 0
end

"""`Array{Float64}` of the given size in page-locked host memory (lto_host_alloc): the GPU reads and writes such arrays (and
contiguous views into them) in place during a host-pointer call -- no copy is queued (Jacobian call at 4 096 segments: 0.20 ms
instead of 0.29 ms with ordinary arrays).  Freed by a finalizer, which may run before or after the context's: the library finds
the block's owner by itself (ctx = NULL) and a context whose destroy was deferred goes with its last block (lto.h, lifetime)."""
function pinned_array(ctx::LtoContext, dims::Integer...)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    check(ctx, ccall((:lto_host_alloc, liblto), Cint, (Ptr{Cvoid}, Csize_t, Ref{Ptr{Cvoid}}), ctx.handle, 8 * prod(dims), p))
    A = unsafe_wrap(Array, convert(Ptr{Float64}, p[]), dims; own = false)
    blk = p[]
    finalizer(_ -> ccall((:lto_host_free, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), C_NULL, blk), A)
    A
end

"""lto_indirect_plan: per-trajectory parameters uploaded once, then repeated asynchronous sweeps.  `params` = one tuple or a
vector of n_batch tuples (homotopy levels, line-search trial points).  The plan keeps its context alive (lto.h, lifetime)."""
mutable struct LtoIndirectPlan
    handle::Ptr{Cvoid}
    ctx::LtoContext
    ndim::Int; n_nodes::Int; n_batch::Int
    function LtoIndirectPlan(ctx::LtoContext, ndim::Integer, n_nodes::Integer, n_batch::Integer, params; integ::LtoIntegrator = LtoIntegrator())
        prm = params isa Tuple ? [LtoParams(params)] : [LtoParams(q) for q in params]
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ctx, ccall((:lto_indirect_plan_create, liblto), Cint,
                         (Ptr{Cvoid}, Cint, Cint, Cint, Ptr{LtoParams}, Cint, Ref{LtoIntegrator}, Ref{Ptr{Cvoid}}),
                         ctx.handle, ndim, n_nodes, n_batch, prm, length(prm), Ref(integ), h))
        pl = new(h[], ctx, ndim, n_nodes, n_batch)
        finalizer(q -> (q.handle == C_NULL || ccall((:lto_indirect_plan_destroy, liblto), Cvoid, (Ptr{Cvoid},), q.handle); q.handle = C_NULL), pl)
        pl
    end
end
nseg(pl::LtoIndirectPlan) = (pl.n_nodes - 1) * pl.n_batch

"defect[c*ldd + s] (SoA) of every segment; X[c*ldx + j], t[g*n_nodes + k] on the device.  Replaces indirect.jl:63-90."
function indirect_defect_dev!(pl::LtoIndirectPlan, stream, X, ldx::Integer, t, n_tgrids::Integer, defect, ldd::Integer; errors = nothing)
    check(pl.ctx, ccall((:lto_indirect_defect_dev, liblto), Cint,
                        (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Clong, DevPtr, Cint, DevPtr, Clong, DevPtr),
                        pl.handle, devptr(stream), devptr(X), ldx, devptr(t), n_tgrids, devptr(defect), ldd, devptr(errors)))
end

"Phi[(col*ndim+row)*ldp + s] and (optionally) the defect.  Replaces indirect.jl:93-146 (compact blocks)."
function indirect_jacobian_dev!(pl::LtoIndirectPlan, stream, X, ldx::Integer, t, n_tgrids::Integer, Phi, ldp::Integer; defect = nothing, ldd::Integer = 0)
    check(pl.ctx, ccall((:lto_indirect_jacobian_dev, liblto), Cint,
                        (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Clong, DevPtr, Cint, DevPtr, Clong, DevPtr, Clong),
                        pl.handle, devptr(stream), devptr(X), ldx, devptr(t), n_tgrids, devptr(Phi), ldp, devptr(defect), ldd))
end

"delta = -Jac_full \\ defect on the device (indirect.jl:181-182; `Phi = nothing` re-uses the factorisation: the SOC re-solve :190-214)."
function newton_solve_dev!(pl::LtoIndirectPlan, stream, Phi, ldp::Integer, defect, ldd::Integer, delta, ldx::Integer; adjoints_only::Bool = false)
    check(pl.ctx, ccall((:lto_indirect_newton_solve_dev, liblto), Cint,
                        (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Clong, DevPtr, Clong, Cint, DevPtr, Clong),
                        pl.handle, devptr(stream), devptr(Phi), ldp, devptr(defect), ldd, adjoints_only ? 1 : 0, devptr(delta), ldx))
end

"y = x + alpha d over `count` doubles on the device (trial points, update accumulation)."
axpy_dev!(ctx::LtoContext, stream, x, d, alpha::Real, y, count::Integer) =
    check(ctx, ccall((:lto_axpy_dev, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, DevPtr, Cdouble, DevPtr, Clong),
                     ctx.handle, devptr(stream), devptr(x), devptr(d), alpha, devptr(y), count))

"The `n_alpha` trial trajectories X + alphas[a] * delta of every trajectory of a batch in one launch (lineSearch, indirect.jl:227-233); `alphas` is a device array."
trial_points_dev!(ctx::LtoContext, stream, X, delta, ld::Integer, ndim::Integer, n_nodes::Integer, n_batch::Integer, n_alpha::Integer, alphas, Xt, ldt::Integer) =
    check(ctx, ccall((:lto_trial_points_dev, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, DevPtr, Clong, Cint, Cint, Cint, Cint, DevPtr, DevPtr, Clong),
                     ctx.handle, devptr(stream), devptr(X), devptr(delta), ld, ndim, n_nodes, n_batch, n_alpha, devptr(alphas), devptr(Xt), ldt))

"A few device scalars to a host `Vector{Float64}` (`out[1:na] <- a`, `out[na+1:na+nb] <- b`), back when they have arrived: the per-iteration read-back of a Newton loop."
function read_scalars_dev!(ctx::LtoContext, stream, a, na::Integer, b, nb::Integer, out::Vector{Float64})
    # the library copies na + nb doubles to `out`: a short vector would be a heap overwrite
    (na >= 0 && nb >= 0 && length(out) >= na + nb) || throw(ArgumentError("read_scalars_dev!: out needs at least na + nb = $(na + nb) elements"))
    check(ctx, ccall((:lto_read_scalars_dev, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Cint, DevPtr, Cint, Ptr{Cdouble}),
                     ctx.handle, devptr(stream), devptr(a), na, devptr(b), nb, out))
end

"lineSearch's first minimiser per trajectory on the device (indirect.jl:244-245): step <- alpha, maxabs_out <- the chosen trial's max |defect|, defect <- its defect block (the check of :328-331 without another sweep)."
line_search_pick_dev!(ctx::LtoContext, stream, sumsq, maxabs, alphas, n_alpha::Integer, trial_defect, ldt::Integer, ndim::Integer, seg_per_traj::Integer, n_batch::Integer, step, maxabs_out, defect, ldd::Integer) =
    check(ctx, ccall((:lto_line_search_pick_dev, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, DevPtr, DevPtr, Cint, DevPtr, Clong, Cint, Cint, Cint, DevPtr, DevPtr, DevPtr, Clong),
                     ctx.handle, devptr(stream), devptr(sumsq), devptr(maxabs), devptr(alphas), n_alpha, devptr(trial_defect), ldt, ndim, seg_per_traj, n_batch, devptr(step), devptr(maxabs_out), devptr(defect), ldd))

"Order the lanes of the following adaptive sweeps by the last sweep's step counts (results unchanged)."
rebalance!(pl::LtoIndirectPlan, stream) = check(pl.ctx, ccall((:lto_indirect_plan_rebalance, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), pl.handle, devptr(stream)))
"LTO_KERNEL_*: 0 auto, 1 per-lane, 2 cooperative, 5 eight-wave pipeline (RK4 plans), 6 cooperative with two lanes per state (12-dim DOP853 plans), 7 pipeline for large batches, 8 32-segment pipeline, 9 whole-segment lanes (RK4 plans); 3 = the direct plans' pipelined Jacobian kernel."
set_kernel!(pl::LtoIndirectPlan, kernel::Integer) = check(pl.ctx, ccall((:lto_indirect_plan_set_kernel, liblto), Cint, (Ptr{Cvoid}, Cint), pl.handle, kernel))
"Family LTO_KERNEL_AUTO resolves to for an STM sweep of this shape on a device with `n_cus` compute units (MI355X cost table): LTO_KERNEL_* (no GPU needed)."
function auto_kernel(ndim::Integer, method::Integer, steps::Integer, p::Real, n_segments::Integer; n_cus::Integer = 256, ordered::Bool = false)
    k = ccall((:lto_indirect_auto_kernel, liblto), Cint, (Cint, Cint, Cint, Cdouble, Clong, Cint, Cint), ndim, method, steps, p, n_segments, n_cus, ordered ? 1 : 0)
    k >= 0 || error("lto_indirect_auto_kernel: invalid shape (code $k)")
    Int(k)
end
"Adaptive sweeps start every segment from its first accepted step size of the plan's previous sweep of the same kind (12-dim DOP853 plans; lto.h)."
set_warm_start!(pl::LtoIndirectPlan, on::Bool = true) = check(pl.ctx, ccall((:lto_indirect_plan_set_warm_start, liblto), Cint, (Ptr{Cvoid}, Cint), pl.handle, on ? 1 : 0))
"Output layout of the plan's sweeps: 0 = struct of arrays, 1 = one block per segment (defect [S][ndim], Phi [S][ndim*ndim] column-major: Julia's own layout; 12-dim DOP853 plans)."
set_output_layout!(pl::LtoIndirectPlan, layout::Integer) = check(pl.ctx, ccall((:lto_indirect_plan_set_output_layout, liblto), Cint, (Ptr{Cvoid}, Cint), pl.handle, layout))
"Record staging of the plan's ordered sweeps, a bit mask: 1 node / defect records in place, 2 Phi records too, 4 an allocation failed and staging is off (lto.h)."
plan_staging(pl::LtoIndirectPlan) = Int(ccall((:lto_indirect_plan_staging, liblto), Cint, (Ptr{Cvoid},), pl.handle))
"STM columns per lane of the per-lane RK4 kernel: 0 auto, 1 or 3 (12-dim) / 1 or 2 (14-dim), or the plan's dimension (12 / 14) = the whole STM in the segment's lane (one-step plans)."
set_cols_per_lane!(pl::LtoIndirectPlan, cols::Integer) = check(pl.ctx, ccall((:lto_indirect_plan_set_cols_per_lane, liblto), Cint, (Ptr{Cvoid}, Cint), pl.handle, cols))
"Lanes per segment of the defect-only sweep of a 12-dim DOP853 plan: 0 = choose (four up to 131 072 segments, then two, then one), 1, 2 or 4."
set_defect_lanes!(pl::LtoIndirectPlan, lanes::Integer = 0) = check(pl.ctx, ccall((:lto_indirect_plan_set_defect_lanes, liblto), Cint, (Ptr{Cvoid}, Cint), pl.handle, lanes))

"Julia column-major [ndim x count] on the device -> SoA [ndim][ld] (and back)."
pack_soa!(ctx::LtoContext, stream, aos, ndim::Integer, count::Integer, soa, ld::Integer) =
    check(ctx, ccall((:lto_pack_soa_dev, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Cint, Clong, DevPtr, Clong),
                     ctx.handle, devptr(stream), devptr(aos), ndim, count, devptr(soa), ld))
unpack_soa!(ctx::LtoContext, stream, soa, ld::Integer, ndim::Integer, count::Integer, aos) =
    check(ctx, ccall((:lto_unpack_soa_dev, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Clong, Cint, Clong, DevPtr),
                     ctx.handle, devptr(stream), devptr(soa), ld, ndim, count, devptr(aos)))

"sumsq[b] = sum(defect_b.^2) (indirect.jl:240), maxabs[b] = norm(defect_b[:], Inf) (:331) on the device."
defect_norms!(ctx::LtoContext, stream, defect, ldd::Integer, ndim::Integer, seg_per_traj::Integer, n_batch::Integer, sumsq, maxabs) =
    check(ctx, ccall((:lto_defect_norms_dev, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Clong, Cint, Cint, Cint, DevPtr, DevPtr),
                     ctx.handle, devptr(stream), devptr(defect), ldd, ndim, seg_per_traj, n_batch, devptr(sumsq), devptr(maxabs)))

mutable struct LtoDirectPlan
    handle::Ptr{Cvoid}
    ctx::LtoContext
    function LtoDirectPlan(ctx::LtoContext, nstate::Integer, n_nodes::Integer, n_batch::Integer, nsteps::Integer, MU, DU, TU, Isp)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(ctx, ccall((:lto_direct_plan_create, liblto), Cint, (Ptr{Cvoid}, Cint, Cint, Cint, Cint, Ref{LtoDirectParams}, Ref{Ptr{Cvoid}}),
                         ctx.handle, nstate, n_nodes, n_batch, nsteps, Ref(LtoDirectParams(MU, DU, TU, Isp)), h))
        pl = new(h[], ctx)
        finalizer(q -> (q.handle == C_NULL || ccall((:lto_direct_plan_destroy, liblto), Cvoid, (Ptr{Cvoid},), q.handle); q.handle = C_NULL), pl)
        pl
    end
end

"Replaces direct.jl:66-109 with operands in HBM: X[c*ldx + j], U[c*ldu + j] (N), t; defect[c*ldd + s], errors[s]."
function direct_defect_dev!(pl::LtoDirectPlan, stream, X, ldx::Integer, U, ldu::Integer, t, n_tgrids::Integer, defect, ldd::Integer; errors = nothing)
    check(pl.ctx, ccall((:lto_direct_defect_dev, liblto), Cint,
                        (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Clong, DevPtr, Clong, DevPtr, Cint, DevPtr, Clong, DevPtr),
                        pl.handle, devptr(stream), devptr(X), ldx, devptr(U), ldu, devptr(t), n_tgrids, devptr(defect), ldd, devptr(errors)))
end

"Replaces direct.jl:111-166 + :503-516: Jac[(col*nstate+row)*ldj + s], dtf[c*ldd + s], defect, errors."
function direct_jacobian_dev!(pl::LtoDirectPlan, stream, X, ldx::Integer, U, ldu::Integer, t, n_tgrids::Integer, Jac, ldj::Integer;
                              dtf = nothing, defect = nothing, ldd::Integer = 0, errors = nothing)
    check(pl.ctx, ccall((:lto_direct_jacobian_dev, liblto), Cint,
                        (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Clong, DevPtr, Clong, DevPtr, Cint, DevPtr, Clong, DevPtr, DevPtr, Clong, DevPtr),
                        pl.handle, devptr(stream), devptr(X), ldx, devptr(U), ldu, devptr(t), n_tgrids, devptr(Jac), ldj, devptr(dtf),
                        devptr(defect), ldd, devptr(errors)))
end

# ------------------------------------------------------------------------------------------- collectives (RCCL)
# One Julia process (Distributed worker / MPI rank) per GPU: rank 0 calls comm_unique_id(), ships the 128 bytes to every
# rank (MPI.Bcast!, a RemoteChannel, a file), every rank builds LtoComm(ctx, world, rank, id).  allgather_dev! gives every
# rank the full defect vector, allreduce_dev! its norms -- device to device, nothing returns to the host in between.
function comm_unique_id()
    id = zeros(UInt8, 128)
    rc = ccall((:lto_comm_unique_id, liblto), Cint, (Ptr{UInt8},), id)
    rc == 0 || error("lto_comm_unique_id failed with code $rc (no RCCL in the process?)")
    id
end

mutable struct LtoComm
    handle::Ptr{Cvoid}
    ctx::LtoContext
    world::Int; rank::Int
    function LtoComm(ctx::LtoContext, world::Integer, rank::Integer, id::Vector{UInt8})
        h = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:lto_comm_create, liblto), Cint, (Ptr{Cvoid}, Cint, Cint, Ptr{UInt8}, Ref{Ptr{Cvoid}}), ctx.handle, world, rank, id, h)
        rc == 0 || error("lto_comm_create failed with code $rc")
        LtoComm(h[], ctx, world, rank)
    end
    function LtoComm(handle::Ptr{Cvoid}, ctx::LtoContext, world::Integer, rank::Integer)
        c = new(handle, ctx, world, rank)
        finalizer(q -> (q.handle == C_NULL || ccall((:lto_comm_destroy, liblto), Cvoid, (Ptr{Cvoid},), q.handle); q.handle = C_NULL), c)
        c
    end
end
"""The window transport (lto_comm_window_*): device copies into IPC-mapped receive windows -- no RCCL, no compute units for the
payload, and the one that works for ranks sharing a device.  `exchange(handle::Vector{UInt8}) -> Vector{UInt8}` of world x 128 bytes in
rank order is the launcher's all-gather (MPI.Allgather, a RemoteChannel, files)."""
function LtoCommWindows(ctx::LtoContext, world::Integer, rank::Integer, max_count::Integer, exchange)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    blob = zeros(UInt8, 128)
    rc = ccall((:lto_comm_window_export, liblto), Cint, (Ptr{Cvoid}, Cint, Cint, Clong, Ptr{UInt8}, Ref{Ptr{Cvoid}}), ctx.handle, world, rank, max_count, blob, h)
    rc == 0 || error("lto_comm_window_export failed with code $rc")
    c = LtoComm(h[], ctx, world, rank)
    all = exchange(blob)
    length(all) == 128 * world || error("exchange must return world x 128 bytes in rank order")
    comm_check(c, ccall((:lto_comm_window_open, liblto), Cint, (Ptr{Cvoid}, Ptr{UInt8}), c.handle, all))
    c
end
comm_check(c::LtoComm, rc::Cint) = rc == 0 || error("lto_comm error $rc: " * unsafe_string(ccall((:lto_comm_last_error, liblto), Cstring, (Ptr{Cvoid},), c.handle)))

"recv [world][count] <- send [count] of every rank (device pointers), asynchronous on `stream`."
allgather_dev!(c::LtoComm, stream, send, recv, count::Integer) =
    comm_check(c, ccall((:lto_comm_allgather_dev, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, DevPtr, Clong), c.handle, devptr(stream), devptr(send), devptr(recv), count))
"buf [count] <- sum (op = 0) or max (op = 1) over ranks, in place."
allreduce_dev!(c::LtoComm, stream, buf, count::Integer, op::Integer) =
    comm_check(c, ccall((:lto_comm_allreduce_dev, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, DevPtr, Clong, Cint), c.handle, devptr(stream), devptr(buf), count, op))

"True once a wait of a window communicator has run out or the ranks have lost step (every collective since returned NaN); waits for `stream`."
function comm_failed(c::LtoComm, stream)
    f = Ref{Cint}(0)
    comm_check(c, ccall((:lto_comm_status, liblto), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cint}), c.handle, devptr(stream), f))
    f[] != 0
end
"Polls (~1.5 us each) before a wait for a peer gives up and poisons the result."
set_wait_limit!(c::LtoComm, polls::Integer) = comm_check(c, ccall((:lto_comm_set_wait_limit, liblto), Cint, (Ptr{Cvoid}, Clong), c.handle, polls))
"Window transport: payloads of up to `bytes` per rank travel by the push / collect kernels, larger ones by the copy engines (lto.h)."
set_kernel_payload!(c::LtoComm, bytes::Integer) = comm_check(c, ccall((:lto_comm_set_kernel_payload, liblto), Cint, (Ptr{Cvoid}, Clong), c.handle, bytes))
"Ranks RCCL itself reports for this communicator (ncclCommCount); 0 for a window communicator."
function rccl_ranks(c::LtoComm)
    n = ccall((:lto_comm_rccl_ranks, liblto), Cint, (Ptr{Cvoid},), c.handle)
    n >= 0 || error("lto_comm_rccl_ranks failed with code $n")
    Int(n)
end

end # module
