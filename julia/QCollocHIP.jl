# QCollocHIP.jl — thin `ccall` layer over libqcolloc_hip.so (include/qcolloc.h).
#
# UNTESTED IN THIS REPOSITORY'S CI: the build image has no Julia.  It is the binding a maintainer of
# QuantumCollocation.jl would add; it builds an object with the fields the MOI evaluator of
# QuantumCollocationCore consumes — `F`, `∂F`, `∂F_structure`, `μ∂²F`, `μ∂²F_structure`, `dim` —
# exactly as they are used in test/scripts/integrator_test_1qubit.jl:41-52, so that replacing the constructor call of that script,
#
#     dynamics = QuantumDynamics(f, Z)        ->        dynamics = QCollocHIP.QuantumDynamics(f, Z)
#
# (line 41; the same positional arguments) swaps the CPU per-knot loop for the MI355X kernels while Ipopt keeps driving the solve unchanged.
module QCollocHIP

using Libdl

const LIB = Ref{String}(get(ENV, "QCOLLOC_HIP_LIB", "libqcolloc_hip.so"))
const QC_MAX_DERIV = 8
const QC_ABI_VERSION = 6      # QC_VERSION_MAJOR * 1000 + QC_VERSION_MINOR of the include/qcolloc.h these mirrors were written against

# mirror of `qc_desc` (include/qcolloc.h); field order and types must match the header -- `__init__` checks the sizes against
# the library's own `sizeof` (qc_sizeof_desc / qc_sizeof_dims / qc_sizeof_terms_desc) when the module is loaded
struct QCDesc
    N::Int32; m::Int32; T::Int64; zdim::Int32; global_dim::Int64
    off_U::Int32; off_a::Int32; off_dt::Int32; dt_fixed::Float64
    integrator::Int32; pade_order::Int32; n_deriv::Int32
    deriv_x_off::NTuple{QC_MAX_DERIV,Int32}; deriv_dx_off::NTuple{QC_MAX_DERIV,Int32}; deriv_dim::NTuple{QC_MAX_DERIV,Int32}
    G_drift::Ptr{Float64}; G_drives::Ptr{Float64}
    device::Int32; kernel::Int32; t_begin::Int64; t_end::Int64
    state_cols::Int32                       # 0 = unitary iso-vec; K = K ket integrators stored back to back
    hess_align::Int32                       # 0 / 1 = exactly the structural entries (default); 16 = interval blocks padded to whole 128-byte lines
    rows_per_interval::Int64; row_offset::Int64; jac_per_interval::Int64; jac_offset::Int64   # composition (all 0 =
    hess_per_interval::Int64; hess_offset::Int64                                               # this handle is the whole dynamics)
    row_placement::Int32                    # 0 = rows stacked in integrator order, 1 = rows at the state components' positions
    hess_tail_zeros::Int32
    deriv_row_off::NTuple{QC_MAX_DERIV,Int32}
    jac_block_order::NTuple{5,Int32}        # order of the value blocks inside an interval (ABI 0.6): permutations of QC_JB_* / QC_HB_*,
    hess_block_order::NTuple{8,Int32}       # all zeros = the library's default order (what `julia/reconcile.jl` prints is Core's)
end

struct QCDims
    n_rows::Int64; n_cols::Int64; ddim::Int64; jac_nnz_interval::Int64; hess_nnz_interval::Int64
    n_intervals::Int64; F_len::Int64; jac_nnz::Int64; hess_nnz::Int64; Z_len::Int64; kernel::Int32; reserved::Int32
end

# mirror of `qc_terms_desc`
struct QCTermsDesc
    T::Int64; zdim::Int32; off_dt::Int32; global_dim::Int64; dt_fixed::Float64
    n_reg::Int32; weighting::Int32; reg_index::Ptr{Int32}; reg_R::Ptr{Float64}; reg_baseline::Ptr{Float64}
    min_time_D::Float64; min_time_knots::Int64; device::Int32; reserved0::Int32
end

# mirror of `qc_fidelity_desc`
struct QCFidelityDesc
    kind::Int32; N::Int32; goal_iso::Ptr{Float64}; subspace::Ptr{Int32}; n_sub::Int32; form::Int32; n_phases::Int32; device::Int32
    phase_dims::Ptr{Int32}; phase_ops::Ptr{Float64}
end

function __init__()
    # constants were renumbered between ABI 0.1 / 0.2 / 0.3 and retired in 0.4 (QC_REG_*): the struct sizes do not show that, the version does
    abi = ccall(dlsym(dlopen(LIB[]), :qc_abi_version), Int32, ())
    abi == QC_ABI_VERSION || error("QCollocHIP: $(LIB[]) has ABI version $abi, this binding mirrors $QC_ABI_VERSION")
    # a stale mirror would corrupt memory silently: compare with the structs the library was compiled with
    for (sym, T) in ((:qc_sizeof_desc, QCDesc), (:qc_sizeof_dims, QCDims), (:qc_sizeof_terms_desc, QCTermsDesc))
        lib = ccall(dlsym(dlopen(LIB[]), sym), Int64, ())
        lib == sizeof(T) || error("QCollocHIP: $(T) has $(sizeof(T)) bytes, $(LIB[]) expects $lib (header / binding version mismatch)")
    end
end

function check(rc::Cint, h::Ptr{Cvoid}=C_NULL)
    rc == 0 && return
    msg = unsafe_string(ccall((:qc_last_error, LIB[]), Cstring, (Ptr{Cvoid},), h))
    error("libqcolloc_hip error $rc: $msg")
end

pad8(v) = ntuple(i -> i <= length(v) ? Int32(v[i]) : Int32(0), QC_MAX_DERIV)

# Pinned host memory of the library (`qc_host_alloc`) for the optional result rings.  The blocks belong to ONE owner object that the
# `HIPDynamics` references and that frees them in its own finalizer, after `qc_destroy` (ADVICE round 4): `reshape` / `vec` of a
# result on Julia >= 1.11 share the memory without keeping the wrapping Array alive, so a per-array finalizer could free a block under
# a live alias.  RING RESULTS MUST NOT OUTLIVE THE DYNAMICS OBJECT; `fresh = true` (and the default `result_ring = 0`) return ordinary
# Julia vectors with no such condition.
mutable struct PinnedOwner
    blocks::Vector{Ptr{Cvoid}}
end
free_pinned!(o::PinnedOwner) = (foreach(p -> ccall((:qc_host_free, LIB[]), Cint, (Ptr{Cvoid},), p), o.blocks); empty!(o.blocks); nothing)
function pinned_zeros!(o::PinnedOwner, len::Integer)
    len * 8 >= 65536 || return zeros(Float64, len)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    ccall((:qc_host_alloc, LIB[]), Cint, (Int64, Ptr{Ptr{Cvoid}}), len * 8, p) == 0 || return zeros(Float64, len)
    push!(o.blocks, p[])
    return fill!(unsafe_wrap(Array, Ptr{Float64}(p[]), len; own=false), 0.0)
end
# Result vectors of one closure, handed out in turn (opt-in): allocated (pinned) and written once when the ring is built, so that no call
# pays for page faults or per-call pinning.
struct ResultRing
    bufs::Vector{Vector{Float64}}
    next::Base.RefValue{Int}
end
ResultRing(o::PinnedOwner, len::Integer, n::Integer) = ResultRing(Vector{Float64}[pinned_zeros!(o, len) for _ in 1:(len > 0 ? n : 0)], Ref(1))
function next!(r::ResultRing, len::Integer, fresh::Bool)
    (fresh || isempty(r.bufs)) && return Vector{Float64}(undef, len)
    v = r.bufs[r.next[]]
    r.next[] = r.next[] % length(r.bufs) + 1
    return v
end

"""
Field-compatible stand-in for `QuantumDynamics` (QuantumCollocationCore.Dynamics).

`F(Z⃗)`, `∂F(Z⃗)`, `μ∂²F(Z⃗, μ⃗)` -- the only shapes QuantumCollocationCore's evaluator uses (test/scripts/integrator_test_1qubit.jl:45-52)
-- return a newly allocated vector that is the caller's for good, as the reference's closures do (`result_ring = 0`, the default).
A fresh `Vector{Float64}(undef, 5_034_960)` per `∂F` call costs 2.7 - 4 ms of first-touch page faults at BASELINE config 3, ten times
the evaluation: an evaluator that copies each result into Ipopt's buffer at once (Core's does) opts in to `result_ring = 3` -- the next
vector of a ring of three per closure, pinned and written once when the ring is built; such a result stays intact until the third-next
call OF THE SAME CLOSURE and must not outlive the dynamics object; `∂F(Z⃗; fresh=true)` still allocates.
`F!`, `∂F!`, `μ∂²F!` write into a caller-owned vector (what an evaluator that owns the MOI callbacks does with Ipopt's buffers).
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
    F!::Function
    ∂F!::Function
    μ∂²F!::Union{Function,Nothing}
    handles::Vector{Ptr{Cvoid}}      # every qc_handle this object owns (one; one per state integrator for `dynamics_list`)
    pinned::PinnedOwner              # the result rings' memory
end
# handles first, then the pinned blocks (nothing of the library refers to them once the handles are gone)
function release!(dyn::HIPDynamics)
    foreach(h -> ccall((:qc_destroy, LIB[]), Cvoid, (Ptr{Cvoid},), h), dyn.handles)
    empty!(dyn.handles)
    dyn.handle = C_NULL
    free_pinned!(dyn.pinned)
end

# first row of state component `name` among the trajectory's state components (controls carry no dynamics rows)
function state_row_offset(traj, name)
    r = 0
    for other in traj.names
        other == name && return r
        other in traj.control_names || (r += length(traj.components[other]))
    end
    error("no state component $name")
end

# qc_desc.jac_block_order / hess_block_order: `nothing` = the library's default order (all zeros), else a permutation of 0:n-1
block_order(order, n::Int) = isnothing(order) ? ntuple(_ -> Int32(0), n) :
    (sort(collect(order)) == collect(0:n-1) || error("block order must be a permutation of 0:$(n-1)"); ntuple(i -> Int32(order[i]), n))

"""
    dynamics(integrators, traj, system; device=0, devices=nothing, eval_hessian=true, rows=:stacked, padded=false, result_ring=0)

`integrators[1]` must be the `UnitaryPadeIntegrator` / `UnitaryExponentialIntegrator`, followed by
`DerivativeIntegrator`s (the order of unitary_smooth_pulse_problem.jl:175-179).  `component_offset(traj, name)`
is `first(traj.components[name]) - 1`.

`devices = 0:7` builds ONE evaluator over several GPUs (`qc_create_multi`): the knots are sharded inside the library, each
GPU copies its own contiguous slice of `∂F` into the caller's vector over its own PCIe link; everything else is unchanged.
`rows = :by_component` places every integrator's rows at its state component's position (`Z.dims.states` rows per interval).
`μ∂²F_structure` holds exactly the structural entries (config 3: 1 832 per interval), so `length` and `==` comparisons with
QuantumCollocationCore's own structure are meaningful (julia/reconcile.jl).  `padded = true` (= `hess_align = 16`) is the layout of
device-resident consumers: every interval's value block padded to whole cache lines with explicit zero duplicates, which a COO
consumer sums away -- a host-buffer call is bound by PCIe and gains nothing from it.
`set_new_x!(dyn, false)` is Ipopt's `new_x = false`: the following calls reuse the knots already on the device.
`result_ring`: 0 (default) = `F` / `∂F` / `μ∂²F` return fresh vectors; n >= 3 = rings of n pinned result vectors per closure (see `HIPDynamics`).
`jac_block_order` / `hess_block_order`: the order of the value blocks inside an interval (`qc_desc.jac_block_order`, ABI 0.6) -- a
permutation of `0:4` (−F copies, B copies / identity, ∂a, ∂Δt, derivative integrators) / `0:7` ((Ũ_t,a), (a,Ũ_t+1), (Ũ_t,Δt),
(Δt,Ũ_t+1), (a,a), (a,Δt), (Δt,Δt), (dx,Δt)); `julia/reconcile.jl` prints the order QuantumCollocationCore's structures have.
"""
function dynamics(integrators, traj, system; device::Int=0, devices=nothing, eval_hessian::Bool=true,
                  state_name=:Ũ⃗, control_name=:a, pade_order::Int=4, exponential::Bool=false,
                  derivative_pairs=[(:a, :da), (:da, :dda)], n_kets::Int=0, rows::Symbol=:stacked, hess_align::Int=0,
                  padded::Bool=false, result_ring::Int=0, offsets=nothing, generators=nothing,
                  jac_block_order=nothing, hess_block_order=nothing)
    (result_ring == 0 || result_ring >= 3) || error("result_ring must be 0 (fresh vectors) or at least 3")
    padded && (hess_align = 16)
    # `offsets` (0-based positions inside a knot, keyed by name) and `generators` = (G_drift, [G_1 .. G_m]) override what is read from
    # `traj.components` and `system`: what `QuantumDynamics(integrators, traj)` below passes after reading the integrator objects
    off(name) = (!isnothing(offsets) && haskey(offsets, name)) ? Int(offsets[name]) : first(traj.components[name]) - 1
    G0 = Matrix{Float64}(isnothing(generators) ? system.G_drift : generators[1])                       # column-major n x n
    Gs = isnothing(generators) ? system.G_drives : generators[2]
    Gd = reduce(hcat, [vec(Matrix{Float64}(G)) for G in Gs])   # n^2 x m, column j = vec(G_j)
    levels = size(G0, 1) ÷ 2
    free_time = traj.timestep isa Symbol
    xs = [off(p[1]) for p in derivative_pairs]; dxs = [off(p[2]) for p in derivative_pairs]
    dms = [length(traj.components[p[1]]) for p in derivative_pairs]
    bycomp = rows == :by_component
    h = Ref{Ptr{Cvoid}}(C_NULL)
    dims = Ref{QCDims}()
    devs = isnothing(devices) ? Int32[] : Int32.(collect(devices))
    GC.@preserve G0 Gd devs begin
        desc = Ref(QCDesc(levels, length(Gs), traj.T, traj.dim, traj.global_dim,
                          off(state_name), off(control_name), free_time ? off(traj.timestep) : -1,
                          free_time ? 0.0 : Float64(traj.timestep),
                          exponential ? 1 : 0, exponential ? 0 : pade_order, length(derivative_pairs),
                          pad8(xs), pad8(dxs), pad8(dms), pointer(G0), pointer(Gd),
                          isempty(devs) ? device : devs[1], 0, 0, 0, n_kets, hess_align,
                          bycomp ? traj.dims.states : 0, bycomp ? state_row_offset(traj, state_name) : 0, 0, 0, 0, 0,
                          bycomp ? 1 : 0, 0, bycomp ? pad8([state_row_offset(traj, p[1]) for p in derivative_pairs]) : pad8(Int[]),
                          block_order(jac_block_order, 5), block_order(hess_block_order, 8)))
        if isempty(devs)
            check(ccall((:qc_create, LIB[]), Cint, (Ref{QCDesc}, Ref{Ptr{Cvoid}}), desc, h))
        else
            check(ccall((:qc_create_multi, LIB[]), Cint, (Ref{QCDesc}, Int32, Ptr{Int32}, Ref{Ptr{Cvoid}}), desc, length(devs), devs, h))
        end
    end
    check(ccall((:qc_dims, LIB[]), Cint, (Ptr{Cvoid}, Ref{QCDims}), h[], dims), h[])
    d = dims[]
    rows_ = Vector{Int64}(undef, d.jac_nnz); cols = similar(rows_)
    check(ccall((:qc_jac_structure, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint), h[], rows_, cols, 1), h[])
    ∂F_structure = collect(zip(Int.(rows_), Int.(cols)))          # 1-based (row, col) tuples, value order
    F! = function (out::AbstractVector{Float64}, Z⃗::AbstractVector{Float64})
        length(out) == d.F_len || error("F!: output has length $(length(out)), expected $(d.F_len)")
        GC.@preserve Z⃗ out check(ccall((:qc_eval_F, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h[], Z⃗, out), h[])
        return out
    end
    ∂F! = function (out::AbstractVector{Float64}, Z⃗::AbstractVector{Float64})
        length(out) == d.jac_nnz || error("∂F!: output has length $(length(out)), expected $(d.jac_nnz)")
        GC.@preserve Z⃗ out check(ccall((:qc_eval_jac, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), h[], Z⃗, out), h[])
        return out
    end
    pinned = PinnedOwner(Ptr{Cvoid}[])
    ringF = ResultRing(pinned, d.F_len, result_ring); ring∂F = ResultRing(pinned, d.jac_nnz, result_ring)
    F = (Z⃗; fresh::Bool=false) -> F!(next!(ringF, d.F_len, fresh), Z⃗)
    ∂F = (Z⃗; fresh::Bool=false) -> ∂F!(next!(ring∂F, d.jac_nnz, fresh), Z⃗)
    μ∂²F = nothing; μ∂²F! = nothing; μ∂²F_structure = nothing
    if eval_hessian && d.hess_nnz > 0
        hr = Vector{Int64}(undef, d.hess_nnz); hc = similar(hr)
        check(ccall((:qc_hess_structure, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint), h[], hr, hc, 1), h[])
        μ∂²F_structure = collect(zip(Int.(hr), Int.(hc)))       # upper triangle, as test/test_utils.jl:14-27 expects; with
                                                                # `padded` the padding repeats an entry with value 0 (duplicates are summed)
        μ∂²F! = function (out::AbstractVector{Float64}, Z⃗::AbstractVector{Float64}, μ⃗::AbstractVector{Float64})
            length(out) == d.hess_nnz || error("μ∂²F!: output has length $(length(out)), expected $(d.hess_nnz)")
            length(μ⃗) == d.n_rows || error("μ∂²F!: μ has length $(length(μ⃗)), expected $(d.n_rows)")
            GC.@preserve Z⃗ μ⃗ out check(ccall((:qc_eval_hess, LIB[]), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h[], Z⃗, μ⃗, out), h[])
            return out
        end
        ringH = ResultRing(pinned, d.hess_nnz, result_ring)
        μ∂²F = (Z⃗, μ⃗; fresh::Bool=false) -> μ∂²F!(next!(ringH, d.hess_nnz, fresh), Z⃗, μ⃗)
    end
    dyn = HIPDynamics(h[], d, F, ∂F, ∂F_structure, μ∂²F, μ∂²F_structure, Int(bycomp ? traj.dims.states : d.ddim), F!, ∂F!, μ∂²F!,
                      Ptr{Cvoid}[h[]], pinned)
    finalizer(release!, dyn)
    return dyn
end

# ---------------------------------------------------------------------------------------------------------------
#  The reference's own constructor call: QuantumDynamics(integrators, traj)        (test/scripts/integrator_test_1qubit.jl:41)
# ---------------------------------------------------------------------------------------------------------------

# first of `names` that is a property of `x` (nothing when none is): the integrator structs of QuantumCollocationCore are not vendored
# with the reference, so every field is looked up under the names recalled for 0.3.x (SURVEY Appendix B) AND has a keyword fallback
function field_of(x, names...)
    for nm in names
        hasproperty(x, nm) && return getproperty(x, nm)
    end
    return nothing
end
type_name(x) = String(nameof(typeof(x)))
is_derivative_integrator(x) = occursin("DerivativeIntegrator", type_name(x))
is_state_integrator(x) = !is_derivative_integrator(x) && (occursin("Pade", type_name(x)) || occursin("Exponential", type_name(x)))
# trajectory component whose index range starts at `first_index` (1-based position inside a knot)
function component_at(traj, first_index::Integer)
    for nm in traj.names
        first(traj.components[nm]) == first_index && return nm
    end
    error("QCollocHIP.QuantumDynamics: no trajectory component starts at knot position $first_index")
end
# (G_drift, [G_1 .. G_m]) from the integrator's generator closure G(a) = G_drift + sum_j a_j G_j (affine in a: m + 1 evaluations),
# and ONE more evaluation at a generic point that checks the assumption: a closure that is not affine in a (a nonlinear drive, a
# time-dependent frame) would otherwise be replaced by its linearisation without a word (ADVICE round 5)
function generators_from_closure(G, m::Integer)
    G0 = Matrix{Float64}(G(zeros(m)))
    Gs = [Matrix{Float64}(G([j == k ? 1.0 : 0.0 for k in 1:m])) - G0 for j in 1:m]
    a = [0.37 + 0.21 * k - 0.05 * k^2 for k in 1:m]
    model = m == 0 ? G0 : G0 + sum(a[k] * Gs[k] for k in 1:m)
    err = maximum(abs, Matrix{Float64}(G(a)) - model; init=0.0)
    err <= 1e-12 * max(1.0, maximum(abs, model; init=0.0)) ||
        error("QCollocHIP.QuantumDynamics: the integrator's generator closure G(a) is not affine in a (deviation $err at a test point): " *
              "this library evaluates G(a) = G_drift + sum_j a_j G_j only; pass `system = ...` if the closure is not the system's")
    return G0, Gs
end

"""
    QuantumDynamics(integrators, traj; system=nothing, state_name=nothing, control_name=nothing, pade_order=nothing,
                    derivative_pairs=nothing, kwargs...)

The positional arguments of the reference's own constructor call, `dynamics = QuantumDynamics(f, Z)` with `f = [P, D]`
(test/scripts/integrator_test_1qubit.jl:39-41; `QuantumControlProblem` makes the same call with `prob.integrators`,
unitary_smooth_pulse_problem.jl:175-190): qualifying that one call with `QCollocHIP.` is the whole change to the script.
Everything `dynamics(integrators, traj, system; ...)` needs is read from the integrator objects:

| needed                         | read from (first that exists)                                                  | keyword fallback     |
|--------------------------------|--------------------------------------------------------------------------------|----------------------|
| state component                | `P.unitary_components` / `P.state_components` (index range inside a knot), `P.unitary_name` / `P.state_name` / `P.state_symb` | `state_name`  |
| control component              | `P.drive_components`, `P.control_name` / `P.drive_name` / `P.drive_symb`      | `control_name`       |
| generators `G_drift, G_drives` | `P.system` (`.G_drift`, `.G_drives`), else `P.G_drift` / `P.G_drives`, else the closure `P.G` evaluated at `0` and the unit vectors | `system` |
| Padé order                     | `P.order` (an `...ExponentialIntegrator` type name selects the exponential)    | `pade_order`         |
| derivative integrators         | `D.variable_components` / `D.derivative_components`, or `D.variable` / `D.derivative` names, of every element whose type name contains `DerivativeIntegrator` | `derivative_pairs` |

NOT YET A VERIFIED DROP-IN: this constructor has never run against QuantumCollocationCore (no Julia in the build image; ADVICE
round 5) -- `julia/reconcile.jl` is the first thing to run on a machine that has both.
The field names are those recalled for QuantumCollocationCore 0.3.x (not vendored with the reference, SURVEY Appendix B): a field that
is absent under every listed name raises an error naming the keyword that supplies it.  Lists with several state integrators (sampling
and direct-sum problems) go to `dynamics_list`.  Remaining keywords (`device`, `devices`, `eval_hessian`, `rows`, `padded`,
`result_ring`) are `dynamics`' own.
"""
function QuantumDynamics(integrators::AbstractVector, traj; system=nothing, state_name=nothing, control_name=nothing,
                         pade_order=nothing, derivative_pairs=nothing, kwargs...)
    states = [I for I in integrators if is_state_integrator(I)]
    derivs = [I for I in integrators if is_derivative_integrator(I)]
    length(states) + length(derivs) == length(integrators) || error("QCollocHIP.QuantumDynamics: an integrator is neither a Padé / exponential state integrator nor a DerivativeIntegrator")
    isempty(states) && error("QCollocHIP.QuantumDynamics: the list holds no state integrator")
    need(x, what, kw) = isnothing(x) ? error("QCollocHIP.QuantumDynamics: cannot read $what from $(type_name(states[1])); pass `$kw = ...`") : x
    component(I, range_fields, name_fields, given, what, kw) = begin
        !isnothing(given) && return given
        r = field_of(I, range_fields...)
        !isnothing(r) && return component_at(traj, first(r))
        need(field_of(I, name_fields...), what, kw)
    end
    part_of(P) = begin
        sname = component(P, (:unitary_components, :state_components, :ket_components), (:unitary_name, :state_name, :state_symb, :unitary_symb),
                          length(states) == 1 ? state_name : nothing, "the state component", "state_name")
        cname = component(P, (:drive_components, :control_components), (:control_name, :drive_name, :drive_symb),
                          control_name, "the control component", "control_name")
        expo = occursin("Exponential", type_name(P))
        ord = expo ? 0 : Int(need(isnothing(pade_order) ? field_of(P, :order) : pade_order, "the Padé order", "pade_order"))
        m = length(traj.components[cname])
        gens = begin
            sys = isnothing(system) ? field_of(P, :system, :sys) : system
            if !isnothing(sys)
                (sys.G_drift, collect(sys.G_drives))
            elseif !isnothing(field_of(P, :G_drift)) && !isnothing(field_of(P, :G_drives))
                (P.G_drift, collect(P.G_drives))
            else
                generators_from_closure(need(field_of(P, :G), "the generators (no `system`, `G_drift` / `G_drives` or `G` field)", "system"), m)
            end
        end
        (sname, cname, expo, ord, gens)
    end
    dpairs = if !isnothing(derivative_pairs)
        collect(derivative_pairs)
    else
        map(derivs) do D
            x = field_of(D, :variable_components); dx = field_of(D, :derivative_components)
            if !isnothing(x) && !isnothing(dx)
                (component_at(traj, first(x)), component_at(traj, first(dx)))
            else
                v = field_of(D, :variable, :variable_name, :x); dv = field_of(D, :derivative, :derivative_name, :dx)
                (isnothing(v) || isnothing(dv)) && error("QCollocHIP.QuantumDynamics: cannot read the components of $(type_name(D)); pass `derivative_pairs = [(:a, :da), ...]`")
                (Symbol(v), Symbol(dv))
            end
        end
    end
    any(I -> occursin("DensityOperator", type_name(I)), states) &&
        error("QCollocHIP.QuantumDynamics: density-operator integrators need the Lindblad generators: call `dynamics(...; n_kets = 1, generators = ...)`")
    kets_of(I) = occursin("QuantumState", type_name(I)) ? 1 : 0          # a QuantumState...Integrator propagates one ket (2N x 1 iso state)
    if length(states) == 1
        sname, cname, expo, ord, gens = part_of(states[1])
        return dynamics(integrators, traj, nothing; state_name=sname, control_name=cname, pade_order=expo ? 4 : ord, exponential=expo,
                        derivative_pairs=dpairs, generators=gens, n_kets=kets_of(states[1]), kwargs...)
    end
    # several state integrators: each with the derivative integrators that follow it in the list (the order QuantumDynamics stacks rows in)
    parts = []
    for (i, I) in enumerate(integrators)
        if is_state_integrator(I)
            sname, cname, expo, ord, gens = part_of(I)
            push!(parts, (system=(G_drift=gens[1], G_drives=gens[2], levels=size(gens[1], 1) ÷ 2), state_name=sname, control_name=cname,
                          derivative_pairs=Tuple{Symbol,Symbol}[], pade_order=expo ? 4 : ord, exponential=expo, n_kets=kets_of(I)))
        else
            isempty(parts) && error("QCollocHIP.QuantumDynamics: a DerivativeIntegrator comes before every state integrator of the list; " *
                                    "rows are stacked in list order behind the state integrator a derivative integrator follows " *
                                    "(unitary_smooth_pulse_problem.jl:175-179): put the Padé / exponential integrator first")
            k = count(is_derivative_integrator, integrators[1:i])
            push!(parts[end].derivative_pairs, dpairs[k])
        end
    end
    allowed = (:device, :devices, :eval_hessian, :padded, :result_ring)
    return dynamics_list(parts, traj; (k => v for (k, v) in kwargs if k in allowed)...)
end

"""
    dynamics_list(parts, traj; device=0, devices=nothing, eval_hessian=true, padded=false, result_ring=0)

The integrator lists with SEVERAL state integrators -- `UnitarySamplingProblem` ([U_1 .. U_K, D, D], shared controls:
unitary_sampling_problem.jl:134-155), `UnitaryDirectSumProblem` ([U_1, D, D, U_2, D, D, ...], own controls per member:
unitary_direct_sum_problem.jl:127-130), `QuantumStateSamplingProblem` (quantum_state_sampling_problem.jl:98-122) -- as ONE
`HIPDynamics` with the fields and call shapes of `dynamics`.  `parts` holds one named tuple per state integrator in list order,
each with the derivative integrators that FOLLOW it in the list:

    (system = sys, state_name = :Ũ⃗_system_1, control_name = :a, derivative_pairs = [], pade_order = 4, exponential = false, n_kets = 0)

(only `system` and `state_name` are required).  One composed handle per part; rows and values come out interval-major, in integrator
order inside an interval (the order `QuantumDynamics` stacks them in).  The evaluations go through `qc_eval_*_list`: one upload
of `Z⃗`, one batched launch where the parts' shapes allow it, results copied straight into the result vectors.
`devices = 0:7`: every part is created over the same device list (`qc_create_multi` on its composed descriptor), so shard s of every part
covers the same intervals on the same GPU, and the list is evaluated shard by shard -- each GPU lands its slice of the result vectors
over its own PCIe link (a K-system robust-control problem is the workload that wants eight GPUs).
"""
function dynamics_list(parts, traj; device::Int=0, devices=nothing, eval_hessian::Bool=true, padded::Bool=false, result_ring::Int=0)
    (result_ring == 0 || result_ring >= 3) || error("result_ring must be 0 (fresh vectors) or at least 3")
    length(parts) >= 1 || error("dynamics_list: no state integrators")
    devs = isnothing(devices) ? Int32[] : Int32.(collect(devices))
    isempty(devs) || (device = Int(devs[1]))
    off(name) = first(traj.components[name]) - 1
    free_time = traj.timestep isa Symbol
    opt(p, k, default) = haskey(p, k) ? p[k] : default
    keep = Any[]                                               # generator matrices the descriptors point at
    function desc_of(p, place)
        sys = p.system
        dpairs = opt(p, :derivative_pairs, [])
        expo = opt(p, :exponential, false)
        G0 = Matrix{Float64}(sys.G_drift)
        Gd = reduce(hcat, [vec(Matrix{Float64}(G)) for G in sys.G_drives])
        push!(keep, G0); push!(keep, Gd)
        xs = [off(q[1]) for q in dpairs]; dxs = [off(q[2]) for q in dpairs]; dms = [length(traj.components[q[1]]) for q in dpairs]
        return QCDesc(size(G0, 1) ÷ 2, length(sys.G_drives), traj.T, traj.dim, traj.global_dim,
                      off(p.state_name), off(opt(p, :control_name, :a)), free_time ? off(traj.timestep) : -1,
                      free_time ? 0.0 : Float64(traj.timestep),
                      expo ? 1 : 0, expo ? 0 : opt(p, :pade_order, 4), length(dpairs),
                      pad8(xs), pad8(dxs), pad8(dms), pointer(G0), pointer(Gd),
                      device, 0, 0, 0, opt(p, :n_kets, 0), 1,
                      place[1], place[2], place[3], place[4], place[5], place[6], 0, place[7], pad8(Int[]),
                      block_order(nothing, 5), block_order(nothing, 8))
    end
    # pass 1: every part's own sizes (no device work), hence its slot in the shared per-interval blocks
    own = QCDims[]
    for p in parts
        d = Ref{QCDims}()
        GC.@preserve keep check(ccall((:qc_desc_dims, LIB[]), Cint, (Ref{QCDesc}, Ref{QCDims}), Ref(desc_of(p, (0, 0, 0, 0, 0, 0, 0))), d))
        push!(own, d[])
    end
    rows = sum(x.ddim for x in own); jac = sum(x.jac_nnz_interval for x in own)
    with_hess = eval_hessian && all(x.hess_nnz_interval > 0 for x in own)
    hess_own = with_hess ? sum(x.hess_nnz_interval for x in own) : 0
    al = padded ? 16 : 1                                       # `padded`: the shared Hessian block is padded as a whole, through its last handle
    hess = cld(hess_own, al) * al
    handles = Ptr{Cvoid}[]
    ro = jo = ho = 0
    for (i, (p, x)) in enumerate(zip(parts, own))
        tail = (with_hess && i == length(parts)) ? hess - hess_own : 0
        h = Ref{Ptr{Cvoid}}(C_NULL)
        place = (rows, ro, jac, jo, with_hess ? hess : 0, with_hess ? ho : 0, tail)
        if isempty(devs)
            GC.@preserve keep check(ccall((:qc_create, LIB[]), Cint, (Ref{QCDesc}, Ref{Ptr{Cvoid}}), Ref(desc_of(p, place)), h))
        else
            GC.@preserve keep devs check(ccall((:qc_create_multi, LIB[]), Cint, (Ref{QCDesc}, Int32, Ptr{Int32}, Ref{Ptr{Cvoid}}),
                                               Ref(desc_of(p, place)), length(devs), devs, h))
        end
        push!(handles, h[])
        ro += x.ddim; jo += x.jac_nnz_interval; ho += x.hess_nnz_interval
    end
    h0 = handles[1]
    n_int = Int(own[1].n_intervals)
    F_len = rows * n_int; jac_nnz = jac * n_int; hess_nnz = hess * n_int
    # structures: every handle reports its own entries with the problem's row / column numbers; interleave them per interval
    function structure(hessian::Bool, per)
        rs = Matrix{Int64}[]; cs = Matrix{Int64}[]
        for (h, n) in zip(handles, per)
            r = Vector{Int64}(undef, n * n_int); c = similar(r)
            if hessian
                check(ccall((:qc_hess_structure, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint), h, r, c, 1), h)
            else
                check(ccall((:qc_jac_structure, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint), h, r, c, 1), h)
            end
            push!(rs, reshape(r, n, n_int)); push!(cs, reshape(c, n, n_int))
        end
        return collect(zip(Int.(vec(reduce(vcat, rs))), Int.(vec(reduce(vcat, cs)))))
    end
    ∂F_structure = structure(false, [Int(x.jac_nnz_interval) for x in own])
    nh = length(handles)
    F! = function (out::AbstractVector{Float64}, Z⃗::AbstractVector{Float64})
        length(out) == F_len || error("F!: output has length $(length(out)), expected $F_len")
        GC.@preserve Z⃗ out handles check(ccall((:qc_eval_F_list, LIB[]), Cint, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Float64}, Ptr{Float64}), handles, nh, Z⃗, out), h0)
        return out
    end
    ∂F! = function (out::AbstractVector{Float64}, Z⃗::AbstractVector{Float64})
        length(out) == jac_nnz || error("∂F!: output has length $(length(out)), expected $jac_nnz")
        GC.@preserve Z⃗ out handles check(ccall((:qc_eval_jac_list, LIB[]), Cint, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Float64}, Ptr{Float64}), handles, nh, Z⃗, out), h0)
        return out
    end
    pinned = PinnedOwner(Ptr{Cvoid}[])
    ringF = ResultRing(pinned, F_len, result_ring); ring∂F = ResultRing(pinned, jac_nnz, result_ring)
    F = (Z⃗; fresh::Bool=false) -> F!(next!(ringF, F_len, fresh), Z⃗)
    ∂F = (Z⃗; fresh::Bool=false) -> ∂F!(next!(ring∂F, jac_nnz, fresh), Z⃗)
    μ∂²F = nothing; μ∂²F! = nothing; μ∂²F_structure = nothing
    if with_hess
        per = [Int(x.hess_nnz_interval) for x in own]; per[end] += hess - hess_own
        μ∂²F_structure = structure(true, per)
        μ∂²F! = function (out::AbstractVector{Float64}, Z⃗::AbstractVector{Float64}, μ⃗::AbstractVector{Float64})
            length(out) == hess_nnz || error("μ∂²F!: output has length $(length(out)), expected $hess_nnz")
            length(μ⃗) == F_len || error("μ∂²F!: μ has length $(length(μ⃗)), expected $F_len")
            GC.@preserve Z⃗ μ⃗ out handles check(ccall((:qc_eval_hess_list, LIB[]), Cint,
                (Ptr{Ptr{Cvoid}}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), handles, nh, Z⃗, μ⃗, out), h0)
            return out
        end
        ringH = ResultRing(pinned, hess_nnz, result_ring)
        μ∂²F = (Z⃗, μ⃗; fresh::Bool=false) -> μ∂²F!(next!(ringH, hess_nnz, fresh), Z⃗, μ⃗)
    end
    d0 = own[1]
    d = QCDims(F_len, d0.n_cols, rows, jac, hess, n_int, F_len, jac_nnz, hess_nnz, d0.Z_len, d0.kernel, 0)
    dyn = HIPDynamics(h0, d, F, ∂F, ∂F_structure, μ∂²F, μ∂²F_structure, Int(rows), F!, ∂F!, μ∂²F!, handles, pinned)
    finalizer(release!, dyn)
    return dyn
end

"Ipopt's `new_x`: `false` declares that the next host-buffer calls receive the x of the previous one (`qc_set_new_x`)."
set_new_x!(dyn::HIPDynamics, new_x::Bool) = check(ccall((:qc_set_new_x, LIB[]), Cint, (Ptr{Cvoid}, Cint), dyn.handle, new_x ? 1 : 0), dyn.handle)

"""
Uploads of a trajectory vector this handle has done so far (`qc_knot_generation`).  An evaluator that elides uploads remembers it
after the call that put ITS x on the device and passes `new_x = false` only while it is unchanged -- any other call on the same
handle in between moves it -- and goes back to `set_new_x!(dyn, true)` right after the elided call.
"""
knot_generation(dyn::HIPDynamics) = ccall((:qc_knot_generation, LIB[]), Int64, (Ptr{Cvoid},), dyn.handle)

# ---------------------------------------------------------------------------------------------------------------
#  Objective terms and rollouts (SURVEY.md 8f): the same `ccall` pattern over qc_terms_* / qc_fidelity_* / qc_rollout
# ---------------------------------------------------------------------------------------------------------------


"""
    regularizers(traj, names_and_R; D=0.0, device=0, dt_scaled=true)

`names_and_R = [(:a, R_a), (:da, R_da), (:dda, R_dda)]` (scalars or vectors, unitary_smooth_pulse_problem.jl:151-153);
`D` adds `MinimumTimeObjective(traj; D)` (unitary_minimum_time_problem.jl:67-69).  Returns `(L, ∇L, ∂²L, ∂²L_structure)`
closures over one device handle.
"""
function regularizers(traj, names_and_R; D::Float64=0.0, device::Int=0, dt_scaled::Bool=true)
    idx = Int32[]; R = Float64[]
    for (name, r) in names_and_R
        comps = collect(traj.components[name]) .- 1
        append!(idx, Int32.(comps)); append!(R, r isa Number ? fill(Float64(r), length(comps)) : Float64.(r))
    end
    p = sortperm(idx); idx = idx[p]; R = R[p]
    free_time = traj.timestep isa Symbol
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve idx R begin
        desc = Ref(QCTermsDesc(traj.T, traj.dim, free_time ? first(traj.components[traj.timestep]) - 1 : -1, traj.global_dim,
                               free_time ? 0.0 : Float64(traj.timestep), length(idx), dt_scaled ? 2 : 3,   # QC_REG_DT_SCALED = 2 (templates pass timestep_name=), QC_REG_PLAIN = 3 (docstring form); 0 / 1 are retired
                               pointer(idx), pointer(R), C_NULL, D, D == 0.0 ? 0 : traj.T - 1, device, 0))
        rc = ccall((:qc_terms_create, LIB[]), Cint, (Ref{QCTermsDesc}, Ref{Ptr{Cvoid}}), desc, h)
        rc == 0 || error("qc_terms_create: " * unsafe_string(ccall((:qc_terms_last_error, LIB[]), Cstring, (Ptr{Cvoid},), C_NULL)))
    end
    nnz = Ref{Int64}(0)
    ccall((:qc_terms_hess_nnz, LIB[]), Cint, (Ptr{Cvoid}, Ref{Int64}), h[], nnz)
    hr = Vector{Int64}(undef, nnz[]); hc = similar(hr)
    ccall((:qc_terms_hess_structure, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int64}, Cint), h[], hr, hc, 1)
    Zlen = traj.dim * traj.T + traj.global_dim
    ev(Z⃗, J, g, H) = ccall((:qc_terms_eval, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), h[], Z⃗, J, g, H)
    L(Z⃗) = (J = Ref(0.0); GC.@preserve Z⃗ ev(Z⃗, J, C_NULL, C_NULL); J[])
    ∇L(Z⃗) = (g = Vector{Float64}(undef, Zlen); GC.@preserve Z⃗ g ev(Z⃗, C_NULL, g, C_NULL); g)          # fresh vectors: results never alias
    ∂²L(Z⃗) = (H = Vector{Float64}(undef, nnz[]); GC.@preserve Z⃗ H ev(Z⃗, C_NULL, C_NULL, H); H)
    return L, ∇L, ∂²L, collect(zip(Int.(hr), Int.(hc)))
end

"""
    unitary_rollout(dyn::HIPDynamics, Z⃗, Ũ⃗_init)  ->  Ũ⃗ (2N² × T)

`unitary_rollout` / `rollout` / `open_rollout` (trajectory_initialization.jl:426,493,547) with the controls and timesteps of `Z⃗`.
"""
function unitary_rollout(dyn::HIPDynamics, Z⃗::AbstractVector{Float64}, init::AbstractVector{Float64}, T::Int)
    out = Matrix{Float64}(undef, length(init), T)
    GC.@preserve Z⃗ init out check(ccall((:qc_rollout, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                                        dyn.handle, Z⃗, init, out), dyn.handle)
    return out
end

"""
    iso_vec_unitary_fidelity(Ũ⃗, Ũ⃗_goal; subspace=nothing, device=0, squared=false)          (unitary_minimum_time_problem.jl:77)
    iso_vec_unitary_free_phase_fidelity(Ũ⃗, Ũ⃗_goal, phases, phase_operators; subspace=nothing)  (unitary_minimum_time_problem.jl:86-90)

`squared = true` selects |tr|²/n² instead of the docstring's |tr|/n (INTEGRATION.md, table of unverifiable choices).
"""
function iso_vec_unitary_free_phase_fidelity(Ũ⃗::AbstractVector{Float64}, Ũ⃗_goal::AbstractVector{Float64}, phases, phase_operators;
                                             subspace=nothing, device::Int=0, squared::Bool=false)
    N = isqrt(length(Ũ⃗_goal) ÷ 2)
    sub = isnothing(subspace) ? Int32[] : Int32.(collect(subspace) .- 1)
    dims = Int32[size(Op, 1) for Op in phase_operators]
    planes = isempty(dims) ? Float64[] : reduce(vcat, [vcat(vec(Float64.(real.(Op))), vec(Float64.(imag.(Op)))) for Op in phase_operators])
    x = vcat(Ũ⃗, Float64.(collect(phases)))
    h = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve Ũ⃗_goal sub dims planes begin
        desc = Ref(QCFidelityDesc(0, N, pointer(Ũ⃗_goal), isempty(sub) ? C_NULL : pointer(sub), length(sub), squared ? 1 : 0,
                                  length(dims), device, isempty(dims) ? C_NULL : pointer(dims), isempty(planes) ? C_NULL : pointer(planes)))
        rc = ccall((:qc_fidelity_create_desc, LIB[]), Cint, (Ref{QCFidelityDesc}, Ref{Ptr{Cvoid}}), desc, h)
        rc == 0 || error("qc_fidelity_create_desc: " * unsafe_string(ccall((:qc_fidelity_last_error, LIB[]), Cstring, (Ptr{Cvoid},), C_NULL)))
    end
    F = Ref(0.0)
    GC.@preserve x ccall((:qc_fidelity_eval, LIB[]), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ref{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                         h[], x, F, C_NULL, C_NULL, C_NULL)
    ccall((:qc_fidelity_destroy, LIB[]), Cvoid, (Ptr{Cvoid},), h[])
    return F[]
end

iso_vec_unitary_fidelity(Ũ⃗::AbstractVector{Float64}, Ũ⃗_goal::AbstractVector{Float64}; kw...) =
    iso_vec_unitary_free_phase_fidelity(Ũ⃗, Ũ⃗_goal, Float64[], Matrix{ComplexF64}[]; kw...)

end # module
