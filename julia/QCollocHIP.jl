# QCollocHIP.jl — thin `ccall` layer over libqcolloc_hip.so (include/qcolloc.h).
#
# UNTESTED IN THIS REPOSITORY'S CI: the build image has no Julia.  It is the binding a maintainer of
# QuantumCollocation.jl would add; it builds an object with the fields the MOI evaluator of
# QuantumCollocationCore consumes — `F`, `∂F`, `∂F_structure`, `μ∂²F`, `μ∂²F_structure`, `dim` —
# exactly as they are used in test/scripts/integrator_test_1qubit.jl:41-52, so that
#
#     prob.dynamics = QCollocHIP.dynamics(prob.integrators, prob.trajectory, prob.system)
#
# swaps the CPU per-knot loop for the MI355X kernels while Ipopt keeps driving the solve unchanged.
module QCollocHIP

using Libdl

const LIB = Ref{String}(get(ENV, "QCOLLOC_HIP_LIB", "libqcolloc_hip.so"))
const QC_MAX_DERIV = 8

# mirror of `qc_desc` (include/qcolloc.h); field order and types must match the header
struct QCDesc
    N::Int32; m::Int32; T::Int64; zdim::Int32; global_dim::Int64
    off_U::Int32; off_a::Int32; off_dt::Int32; dt_fixed::Float64
    integrator::Int32; pade_order::Int32; n_deriv::Int32
    deriv_x_off::NTuple{QC_MAX_DERIV,Int32}; deriv_dx_off::NTuple{QC_MAX_DERIV,Int32}; deriv_dim::NTuple{QC_MAX_DERIV,Int32}
    G_drift::Ptr{Float64}; G_drives::Ptr{Float64}
    device::Int32; kernel::Int32; t_begin::Int64; t_end::Int64
    state_cols::Int32; reserved0::Int32     # 0 = unitary iso-vec; K = K ket integrators stored back to back
    rows_per_interval::Int64; row_offset::Int64; jac_per_interval::Int64; jac_offset::Int64   # composition (all 0 =
    hess_per_interval::Int64; hess_offset::Int64                                               # this handle is the whole dynamics)
end

struct QCDims
    n_rows::Int64; n_cols::Int64; ddim::Int64; jac_nnz_interval::Int64; hess_nnz_interval::Int64
    n_intervals::Int64; F_len::Int64; jac_nnz::Int64; hess_nnz::Int64; Z_len::Int64; kernel::Int32; reserved::Int32
end

function check(rc::Cint, h::Ptr{Cvoid}=C_NULL)
    rc == 0 && return
    msg = unsafe_string(ccall((:qc_last_error, LIB[]), Cstring, (Ptr{Cvoid},), h))
    error("libqcolloc_hip error $rc: $msg")
end

pad8(v) = ntuple(i -> i <= length(v) ? Int32(v[i]) : Int32(0), QC_MAX_DERIV)

"""
Field-compatible stand-in for `QuantumDynamics` (QuantumCollocationCore.Dynamics).
"""
mutable struct HIPDynamics
    handle::Ptr{Cvoid}
    dims::QCDims
    F::Function
    ∂F::Function
    ∂F_structure::Vector{Tuple{Int,Int}}
    μ∂²F::Union{Function,Nothing}
    μ∂²F_structure::Union{Vector{Tuple{Int,Int}},Nothing}
    dim::Int
end

"""
    dynamics(integrators, traj, system; device=0, eval_hessian=true)

`integrators[1]` must be the `UnitaryPadeIntegrator` / `UnitaryExponentialIntegrator`, followed by
`DerivativeIntegrator`s (the order of unitary_smooth_pulse_problem.jl:175-179).  `component_offset(traj, name)`
is `first(traj.components[name]) - 1`.
"""
function dynamics(integrators, traj, system; device::Int=0, eval_hessian::Bool=true,
                  state_name=:Ũ⃗, control_name=:a, pade_order::Int=4, exponential::Bool=false,
                  derivative_pairs=[(:a, :da), (:da, :dda)], n_kets::Int=0)
    off(name) = first(traj.components[name]) - 1
    n = 2 * system.levels
    G0 = Matrix{Float64}(system.G_drift)                       # column-major n x n
    Gd = reduce(hcat, [vec(Matrix{Float64}(G)) for G in system.G_drives])   # n^2 x m, column j = vec(G_j)
    free_time = traj.timestep isa Symbol
    xs = [off(p[1]) for p in derivative_pairs]; dxs = [off(p[2]) for p in derivative_pairs]
    dms = [length(traj.components[p[1]]) for p in derivative_pairs]
    h = Ref{Ptr{Cvoid}}(C_NULL)
    dims = Ref{QCDims}()
    GC.@preserve G0 Gd begin
        desc = Ref(QCDesc(system.levels, length(system.G_drives), traj.T, traj.dim, traj.global_dim,
                          off(state_name), off(control_name), free_time ? off(traj.timestep) : -1,
                          free_time ? 0.0 : Float64(traj.timestep),
                          exponential ? 1 : 0, exponential ? 0 : pade_order, length(derivative_pairs),
                          pad8(xs), pad8(dxs), pad8(dms), pointer(G0), pointer(Gd), device, 0, 0, 0, n_kets, 0, 0, 0, 0, 0, 0, 0))
        check(ccall((:qc_create, LIB[]), Cint, (Ref{QCDesc}, Ref{Ptr{Cvoid}}), desc, h))
    end
    check(ccall((:qc_dims, LIB[]), Cint, (Ptr{Cvoid}, Ref{QCDims}), h[], dims), h[])
    d = dims[]
    rows = Vector{Int64}(undef, d.jac_nnz); cols = similar(rows)
    check(ccall((:qc_jac_structure, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint), h[], rows, cols, 1), h[])
    ∂F_structure = collect(zip(Int.(rows), Int.(cols)))          # 1-based (row, col) tuples, value order
    Fbuf = Vector{Float64}(undef, d.F_len); Jbuf = Vector{Float64}(undef, d.jac_nnz)
    F = function (Z⃗::AbstractVector{Float64})
        GC.@preserve Z⃗ Fbuf check(ccall((:qc_eval_F, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h[], Z⃗, Fbuf), h[])
        return Fbuf
    end
    ∂F = function (Z⃗::AbstractVector{Float64})
        GC.@preserve Z⃗ Jbuf check(ccall((:qc_eval_jac, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h[], Z⃗, Jbuf), h[])
        return Jbuf
    end
    μ∂²F = nothing; μ∂²F_structure = nothing
    if eval_hessian && d.hess_nnz > 0
        hr = Vector{Int64}(undef, d.hess_nnz); hc = similar(hr)
        check(ccall((:qc_hess_structure, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint), h[], hr, hc, 1), h[])
        μ∂²F_structure = collect(zip(Int.(hr), Int.(hc)))       # upper triangle, as test/test_utils.jl:14-27 expects
        Hbuf = Vector{Float64}(undef, d.hess_nnz)
        μ∂²F = function (Z⃗::AbstractVector{Float64}, μ⃗::AbstractVector{Float64})
            GC.@preserve Z⃗ μ⃗ Hbuf check(ccall((:qc_eval_hess, LIB[]), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h[], Z⃗, μ⃗, Hbuf), h[])
            return Hbuf
        end
    end
    dyn = HIPDynamics(h[], d, F, ∂F, ∂F_structure, μ∂²F, μ∂²F_structure, Int(d.ddim))
    finalizer(x -> ccall((:qc_destroy, LIB[]), Cvoid, (Ptr{Cvoid},), x.handle), dyn)
    return dyn
end

end # module
