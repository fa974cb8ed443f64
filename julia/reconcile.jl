# reconcile.jl -- turns "parity unpinned" into one command for anyone who has Julia and the reference's packages:
#
#     julia --project=<QuantumCollocation.jl checkout> julia/reconcile.jl [out_dir = tests/golden]
#
# It evaluates QuantumCollocationCore 0.3's OWN `QuantumDynamics` (the object the reference's harness times,
# test/scripts/integrator_test_1qubit.jl:22-52, built as unitary_smooth_pulse_problem.jl:163-179 builds it) on
#   fixture : the reference's data fixture (test/test_utils.jl:54-70, free time) with the system of test_utils.jl:123
#   config1 : 1-qubit Hadamard, T = 50, dt = 0.2, X / Y drives        (BASELINE config 1)
#   config2 : 2-qubit CNOT, T = 200, order-4 Pade                      (BASELINE config 2)
#   toffoli3: 3-qubit Toffoli, T = 12                                   (BASELINE configs 3 / 4 in small: the 2N = 16 MFMA kernels,
#                                                                        `mfma16-pade4`, `mfma16-pade4-hess2`, `mfma16-pade4-fused`)
#   qft4    : 4-qubit QFT, T = 6                                        (BASELINE config 5 in small: the 2N = 32 sparse-drive kernels,
#                                                                        `mfma32-pade4-ell`, `-hess-ell`, `-fused-ell`)
#   order6  : 2-qubit CNOT, T = 10, order-6 Pade                        (the any-order kernels, `mfma16-padeP`, `mfma16-padeP-hess`)
#   sampling2 / directsum2 / bangbang: the integrator lists of UnitarySamplingProblem (two systems, shared controls),
#             UnitaryDirectSumProblem (two members, own controls) and UnitaryBangBangProblem (one derivative integrator, L1 slack
#             components, Pade order 12) exactly as the templates build them (`prob.integrators`, `prob.trajectory`): they pin the ROW
#             ORDER of lists with several integrators, which this repository takes to be the integrators' order
# and writes tests/golden/ref_<case>.json: Z, mu, F, dF / mu_d2F values AND structures, plus the scalar definitions this
# repository could only recall (INTEGRATION.md "Choices this repository cannot verify").  tests/test_reference_golden.py
# picks the files up (values as (row, col) -> summed value at rtol 1e-10; structures by `length` and, sorted, `==`: the default layout of
# the bindings is exactly the structural entries) and the verdict of
# every row of that table is printed below.  Nothing here runs in the build container (no Julia): keep it in step with the
# harness script when Core's constructors change.
using QuantumCollocation, NamedTrajectories, LinearAlgebra, Random

out_dir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")
include(joinpath(pkgdir(QuantumCollocation), "test", "test_utils.jl"))      # named_trajectory_type_1 (test_utils.jl:54-118)

json(x::AbstractFloat) = isfinite(x) ? repr(Float64(x)) : "null"
json(x::Integer) = string(x)
json(x::Bool) = x ? "true" : "false"
json(x::AbstractString) = repr(String(x))
json(x::Symbol) = repr(String(x))
json(x::Tuple) = "[" * join(json.(x), ",") * "]"
json(x::AbstractArray) = "[" * join(json.(vec(collect(x))), ",") * "]"
json(x::AbstractDict) = "{" * join(("$(repr(String(k))):$(json(v))" for (k, v) in x), ",") * "}"
json(::Nothing) = "null"

# With QCOLLOC_HIP_LIB set (a machine that has Julia, Core AND an MI355X), every record is also checked on the spot against the GPU
# library through the one-line change INTEGRATION.md describes -- `QCollocHIP.QuantumDynamics(integrators, traj)` with exactly the
# arguments Core's constructor got: values at rtol 1e-10, structures by `length` and sorted `==`.
const HIP = haskey(ENV, "QCOLLOC_HIP_LIB")
HIP && include(joinpath(@__DIR__, "QCollocHIP.jl"))
hip_verdicts = Pair{String,Any}[]
function hip_check(name, integrators, traj, dynamics, Z⃗, μ)
    HIP || return
    hip = QCollocHIP.QuantumDynamics(integrators, traj)
    close(a, b) = maximum(abs.(a .- b)) <= 1e-10 * max(1.0, maximum(abs.(b)))
    coo(vals, st) = (d = Dict{Tuple{Int,Int},Float64}(); foreach(((v, k),) -> d[k] = get(d, k, 0.0) + v, zip(vals, st)); d)
    same_coo(a, b) = all(abs(get(a, k, 0.0) - get(b, k, 0.0)) <= 1e-10 * max(1.0, maximum(abs.(values(b)))) for k in union(keys(a), keys(b)))
    push!(hip_verdicts, "$name: F" => close(hip.F(Z⃗), dynamics.F(Z⃗)))
    push!(hip_verdicts, "$name: length(∂F_structure)" => (length(hip.∂F_structure), length(dynamics.∂F_structure)))
    push!(hip_verdicts, "$name: sort(∂F_structure) ==" => (sort(hip.∂F_structure) == sort(collect(dynamics.∂F_structure))))
    push!(hip_verdicts, "$name: ∂F values" => same_coo(coo(hip.∂F(Z⃗), hip.∂F_structure), coo(dynamics.∂F(Z⃗), dynamics.∂F_structure)))
    if dynamics.μ∂²F !== nothing && hip.μ∂²F !== nothing
        push!(hip_verdicts, "$name: length(μ∂²F_structure)" => (length(hip.μ∂²F_structure), length(dynamics.μ∂²F_structure)))
        push!(hip_verdicts, "$name: sort(μ∂²F_structure) ==" => (sort(hip.μ∂²F_structure) == sort(collect(dynamics.μ∂²F_structure))))
        push!(hip_verdicts, "$name: μ∂²F values" => same_coo(coo(hip.μ∂²F(Z⃗, μ), hip.μ∂²F_structure), coo(dynamics.μ∂²F(Z⃗, μ), dynamics.μ∂²F_structure)))
    end
end

"The harness of integrator_test_1qubit.jl:41-52 on (system, traj): Core's own closures, nothing of this repository."
function reference_record(system, traj; order=4, seed=1, integrators=nothing, described=nothing, systems=nothing, name="record")
    if isnothing(integrators)
        P = UnitaryPadeIntegrator(:Ũ⃗, :a, system, traj; order=order)                   # unitary_smooth_pulse_problem.jl:165-167
        integrators = [P, DerivativeIntegrator(:a, :da, traj), DerivativeIntegrator(:da, :dda, traj)]   # :175-179
    end
    dynamics = QuantumDynamics(integrators, traj)
    Z⃗ = traj.datavec
    F = dynamics.F(Z⃗)
    μ = randn(MersenneTwister(seed), length(F))
    ∂F = dynamics.∂F(Z⃗)
    rec = Dict{String,Any}(
        "T" => traj.T, "dim" => traj.dim, "global_dim" => traj.global_dim, "names" => collect(traj.names),
        "components" => Dict(String(k) => collect(v) for (k, v) in pairs(traj.components)),
        "timestep" => traj.timestep isa Symbol ? String(traj.timestep) : Float64(traj.timestep), "pade_order" => order,
        "H_drift_re" => real.(system.H_drift), "H_drift_im" => imag.(system.H_drift),                  # column-major, as `vec` gives them
        "H_drives_re" => [real.(H) for H in system.H_drives], "H_drives_im" => [imag.(H) for H in system.H_drives],
        "levels" => size(system.H_drift, 1), "Z" => Z⃗, "mu" => μ, "F" => F,
        "dF" => ∂F, "dF_rows" => first.(dynamics.∂F_structure), "dF_cols" => last.(dynamics.∂F_structure),   # 1-based, Core's order
        "rows_declared" => traj.dims.states * (traj.T - 1), "control_names" => collect(traj.control_names))
    if !isnothing(described)      # an integrator list written down by hand from the template's source (the structs' fields are Core's business)
        rec["integrators"] = described
        rec["systems"] = [Dict{String,Any}("levels" => size(s.H_drift, 1), "H_drift_re" => real.(s.H_drift), "H_drift_im" => imag.(s.H_drift),
                                           "H_drives_re" => [real.(H) for H in s.H_drives], "H_drives_im" => [imag.(H) for H in s.H_drives])
                          for s in systems]
    end
    if dynamics.μ∂²F !== nothing
        rec["mu_d2F"] = dynamics.μ∂²F(Z⃗, μ)
        rec["mu_d2F_rows"] = first.(dynamics.μ∂²F_structure)
        rec["mu_d2F_cols"] = last.(dynamics.μ∂²F_structure)
    end
    hip_check(name, integrators, traj, dynamics, Z⃗, μ)
    return rec, dynamics
end

function smooth_pulse_traj(system, U_goal, T, Δt)
    prob = UnitarySmoothPulseProblem(system, U_goal, T, Δt; ipopt_options=IpoptOptions(print_level=1),
                                     piccolo_options=PiccoloOptions(verbose=false))
    return prob.trajectory
end

verdicts = Pair{String,Any}[]
mkpath(out_dir)

# ---- fixture --------------------------------------------------------------------------------------------------
sys1 = QuantumSystem(0.1 * PAULIS[:Z], [PAULIS[:X], PAULIS[:Y]])                          # test_utils.jl:123
traj = named_trajectory_type_1(free_time=true)
rec, dyn = reference_record(sys1, traj; name="fixture")
push!(verdicts, "rows of an interval: length(F) == Z.dims.states * (T - 1)" => (length(rec["F"]) == rec["rows_declared"]))
push!(verdicts, "COO order inside an interval: first 6 (row, col) of dF_structure" => collect(zip(rec["dF_rows"][1:6], rec["dF_cols"][1:6])))
push!(verdicts, "Hessian structure is upper-triangular" => all(r <= c for (r, c) in zip(rec["mu_d2F_rows"], rec["mu_d2F_cols"])))
# The order of the value BLOCKS inside an interval, as Core's structures have it: what to pass as `jac_block_order` / `hess_block_order`
# (qc_desc, ABI 0.6) so that the library's value vectors come in Core's order without a gather on the host.  Kinds as in include/qcolloc.h:
# Jacobian 0 dU_t, 1 dU_t+1, 2 da, 3 dDt, 4 derivative integrators; Hessian 0 (U_t,a) 1 (a,U_t+1) 2 (U_t,Dt) 3 (Dt,U_t+1) 4 (a,a) 5 (a,Dt)
# 6 (Dt,Dt) 7 (dx,Dt).  An order with a kind appearing in several runs is not a block order: the binding then needs a permutation.
let zd = traj.dim, sU = length(traj.components[:Ũ⃗]), cU = traj.components[:Ũ⃗], ca = traj.components[:a], cdt = traj.components[:Δt]
    knot(c) = (c - 1) ÷ zd; pos(c) = (c - 1) % zd + 1
    jkind(r, c) = (r - 1) % traj.dims.states + 1 > sU ? 4 :
                  (pos(c) in cU ? (knot(c) == 0 ? 0 : 1) : pos(c) in ca ? 2 : pos(c) in cdt ? 3 : -1)
    runs(ks) = [ks[i] for i in eachindex(ks) if i == 1 || ks[i] != ks[i-1]]
    nj = length(rec["dF_rows"]) ÷ (traj.T - 1)
    jk = [jkind(r, c) for (r, c) in zip(rec["dF_rows"][1:nj], rec["dF_cols"][1:nj])]
    push!(verdicts, "order of the Jacobian value blocks inside an interval (jac_block_order)" => runs(jk))
    hkind(r, c) = begin
        pr, pc, kc = pos(r), pos(c), knot(c)
        pr in cU && pc in ca ? 0 : pr in ca && pc in cU && kc == 1 ? 1 : pr in cU && pc in cdt ? 2 : pr in cdt && pc in cU && kc == 1 ? 3 :
        pr in ca && pc in ca ? 4 : pr in ca && pc in cdt ? 5 : pr in cdt && pc in cdt ? 6 : pc in cdt ? 7 : -1
    end
    nh = length(rec["mu_d2F_rows"]) ÷ (traj.T - 1)
    hk = [hkind(r, c) for (r, c) in zip(rec["mu_d2F_rows"][1:nh], rec["mu_d2F_cols"][1:nh])]
    push!(verdicts, "order of the Hessian value blocks inside an interval (hess_block_order)" => runs(hk))
end
# scalar definitions (INTEGRATION.md table)
a = traj.a; Δt = vec(traj.Δt)
L = QuadraticRegularizer(:a, traj, 1.0; timestep_name=:Δt).L(traj.datavec, traj)
rec["regularizer_a_R1"] = L
rec["regularizer_plain"] = 0.5 * sum(abs2, a)
rec["regularizer_dt_scaled"] = 0.5 * sum(abs2, a .* Δt')
push!(verdicts, "quadratic regulariser" => (isapprox(L, rec["regularizer_dt_scaled"]; rtol=1e-12) ? "QC_REG_DT_SCALED (dt inside the square)" :
                                            isapprox(L, rec["regularizer_plain"]; rtol=1e-12) ? "QC_REG_PLAIN" : "NEITHER: $L"))
Ũ⃗ = traj.Ũ⃗[:, 3]; Ũ⃗g = traj.goal.Ũ⃗
U = iso_vec_to_operator(Ũ⃗); Ug = iso_vec_to_operator(Ũ⃗g); n = size(U, 1)
fid = iso_vec_unitary_fidelity(Ũ⃗, Ũ⃗g)
rec["fidelity_state"] = Ũ⃗; rec["fidelity_goal"] = Ũ⃗g; rec["fidelity"] = fid
push!(verdicts, "unitary fidelity" => (isapprox(fid, abs(tr(Ug' * U)) / n; rtol=1e-12) ? "QC_FID_FORM_ABS  |tr|/n" :
                                       isapprox(fid, abs2(tr(Ug' * U)) / n^2; rtol=1e-12) ? "QC_FID_FORM_ABS2  |tr|^2/n^2" : "NEITHER: $fid"))
ψ = ComplexF64[0.6, 0.8im]; ψg = ComplexF64[1, 1] / sqrt(2)
kf = iso_fidelity(ket_to_iso(ψ), ket_to_iso(ψg))
rec["ket_fidelity"] = kf
push!(verdicts, "ket fidelity" => (isapprox(kf, abs2(ψg' * ψ); rtol=1e-12) ? "|<goal|psi>|^2" : isapprox(kf, abs(ψg' * ψ); rtol=1e-12) ? "|<goal|psi>|" : "NEITHER: $kf"))
open(io -> write(io, json(rec)), joinpath(out_dir, "ref_fixture.json"), "w")

# ---- exponential integrator: Core's Hessian for it (the templates solve :exponential problems with eval_hessian on,
#      unitary_smooth_pulse_problem.jl:224-266) and the structure it declares -------------------------------------------------
recE, dynE = reference_record(sys1, traj; name="fixture_exponential",
                              integrators=[UnitaryExponentialIntegrator(:Ũ⃗, :a, sys1, traj), DerivativeIntegrator(:a, :da, traj),
                                           DerivativeIntegrator(:da, :dda, traj)])
recE["integrator"] = "exponential"          # tests/test_reference_golden.py builds the mirror's UnitaryExponentialIntegrator from this
push!(verdicts, "Hessian of the exponential integrator is present" => (dynE.μ∂²F !== nothing))
if dynE.μ∂²F !== nothing
    zd = traj.dim
    touches_next = any(((r - 1) ÷ zd) != ((c - 1) ÷ zd) for (r, c) in dynE.μ∂²F_structure)
    push!(verdicts, "exponential Hessian structure has entries at knot t+1 (this library: none; they would be explicit zeros)" => touches_next)
end
open(io -> write(io, json(recE)), joinpath(out_dir, "ref_fixture_exponential.json"), "w")

# ---- BASELINE configs 1 and 2 ---------------------------------------------------------------------------------
for (name, system, gate, T) in (("config1", QuantumSystem(GATES[:Z], [GATES[:X], GATES[:Y]]), GATES[:H], 50),
                                ("config2", QuantumSystem(0.1 * kron(PAULIS[:Z], PAULIS[:Z]),
                                                          [kron(PAULIS[:X], PAULIS[:I]), kron(PAULIS[:Y], PAULIS[:I]),
                                                           kron(PAULIS[:I], PAULIS[:X]), kron(PAULIS[:I], PAULIS[:Y])]), GATES[:CX], 200))
    r, _ = reference_record(system, smooth_pulse_traj(system, gate, T, 0.2); name=name)
    open(io -> write(io, json(r)), joinpath(out_dir, "ref_$(name).json"), "w")
end

# ---- the MFMA paths of the metric workloads, in small (tests/test_reference_golden.py asserts which kernel served each file) ----
pauli_on(P, i, n) = reduce(kron, [k == i ? PAULIS[P] : PAULIS[:I] for k in 1:n])             # P on qubit i of n
function qubit_system(n)    # drift 0.1 sum_i Z_i Z_i+1, drives X_i, Y_i: the systems of this repository's BASELINE configs 2 - 5
    H0 = sum(0.1 * pauli_on(:Z, i, n) * pauli_on(:Z, i + 1, n) for i in 1:n-1)
    return QuantumSystem(H0, reduce(vcat, [[pauli_on(:X, i, n), pauli_on(:Y, i, n)] for i in 1:n]))
end
toffoli = Matrix{ComplexF64}(I, 8, 8); toffoli[7:8, 7:8] = [0 1; 1 0]
qft16 = [exp(2π * im * j * k / 16) / 4 for j in 0:15, k in 0:15]
for (name, system, gate, T, order) in (("toffoli3", qubit_system(3), toffoli, 12, 4), ("qft4", qubit_system(4), qft16, 6, 4),
                                       ("order6", qubit_system(2), GATES[:CX], 10, 6))
    r, _ = reference_record(system, smooth_pulse_traj(system, gate, T, 0.2); order=order, name=name)
    open(io -> write(io, json(r)), joinpath(out_dir, "ref_$(name).json"), "w")
end

# ---- integrator lists of the other templates: the row order of lists with several integrators ----------------------------
quiet = (ipopt_options=IpoptOptions(print_level=1), piccolo_options=PiccoloOptions(verbose=false))
upade(state, control, k; order=4) = Dict{String,Any}("kind" => "unitary_pade", "state" => String(state), "control" => String(control), "system" => k, "order" => order)
deriv(x, dx) = Dict{String,Any}("kind" => "derivative", "x" => String(x), "dx" => String(dx))
# UnitarySamplingProblem: [U_1, U_2, D(a, da), D(da, dda)] over components Ũ⃗_system_k, shared controls (unitary_sampling_problem.jl:103-107,134-155)
sysA = QuantumSystem(0.3 * GATES[:Z], [GATES[:X], GATES[:Y]]); sysB = QuantumSystem(-0.3 * GATES[:Z], [GATES[:X], GATES[:Y]])
probS = UnitarySamplingProblem([sysA, sysB], GATES[:H], 8, 0.2; quiet...)
r, _ = reference_record(sysA, probS.trajectory; integrators=probS.integrators, systems=[sysA, sysB], name="sampling2",
                        described=[upade(:Ũ⃗_system_1, :a, 1), upade(:Ũ⃗_system_2, :a, 2), deriv(:a, :da), deriv(:da, :dda)])
open(io -> write(io, json(r)), joinpath(out_dir, "ref_sampling2.json"), "w")
# UnitaryDirectSumProblem: [U_1, D, D, U_2, D, D] over the members' suffixed components, own controls (unitary_direct_sum_problem.jl:104,127-130)
fixed = (ipopt_options=IpoptOptions(print_level=1), piccolo_options=PiccoloOptions(verbose=false, free_time=false))   # as its own test (:196)
sysD = QuantumSystem(0.01 * GATES[:Z], [GATES[:X], GATES[:Y]])
probD = UnitaryDirectSumProblem([UnitarySmoothPulseProblem(sysD, GATES[:X], 8, 0.2; fixed...), UnitarySmoothPulseProblem(sysD, GATES[:Y], 8, 0.2; fixed...)], 0.99;
                                ipopt_options=IpoptOptions(print_level=1))
r, _ = reference_record(sysD, probD.trajectory; integrators=probD.integrators, systems=[sysD, sysD], name="directsum2",
                        described=[upade(:Ũ⃗1, :a1, 1), deriv(:a1, :da1), deriv(:da1, :dda1), upade(:Ũ⃗2, :a2, 2), deriv(:a2, :da2), deriv(:da2, :dda2)])
open(io -> write(io, json(r)), joinpath(out_dir, "ref_directsum2.json"), "w")
# UnitaryBangBangProblem: [U, D(u, du)], Pade order 12, control_name = :u as in its own test (unitary_bang_bang_problem.jl:163-175,205-215)
probB = UnitaryBangBangProblem(sysD, GATES[:H], 8, 0.2; R_bang_bang=10.0, ipopt_options=IpoptOptions(print_level=1),
                               piccolo_options=PiccoloOptions(verbose=false, pade_order=12), control_name=:u)
r, _ = reference_record(sysD, probB.trajectory; integrators=probB.integrators, systems=[sysD], name="bangbang",
                        described=[upade(:Ũ⃗, :u, 1; order=12), deriv(:u, :du)])
open(io -> write(io, json(r)), joinpath(out_dir, "ref_bangbang.json"), "w")

println("\nreconcile.jl -- verdicts for INTEGRATION.md \"Choices this repository cannot verify\":")
for (k, v) in verdicts
    println("  ", rpad(k, 72), " => ", v)
end
if HIP
    println("\nQCollocHIP.QuantumDynamics(integrators, traj) against Core's QuantumDynamics on the same arguments (libqcolloc_hip.so on this machine):")
    for (k, v) in hip_verdicts
        println("  ", rpad(k, 72), " => ", v)
    end
end
println("wrote ref_fixture.json, ref_fixture_exponential.json, ref_config1.json, ref_config2.json, ref_toffoli3.json, ref_qft4.json, ref_order6.json, ",
        "ref_sampling2.json, ref_directsum2.json, ref_bangbang.json to ", abspath(out_dir))
println("now run:  python -m pytest tests/test_reference_golden.py -m gpu")
