#!/usr/bin/env python3
"""The reference's own micro-benchmark harness (reference test/scripts/integrator_test_script.jl and
integrator_test_1qubit.jl: @btime of dynamics.F, dynamics.∂F, dynamics.μ∂²F on host vectors), line for line, on the
MI355X library: same systems, same trajectory components (Ũ⃗, a, g, da, Δt with controls = (da,)), same integrator list
[UnitaryPadeIntegrator, DerivativeIntegrator(a, da)], same μ = ones.  Times are host-visible (vectors in and out of host
memory, what @btime would see) with the device-resident time beside them.

    python examples/integrator_test_script.py
"""
from __future__ import annotations

import os
import sys
import time
from functools import reduce

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

qc = g.load_package()
G = qc.GATES
I2 = np.eye(2, dtype=complex)
kron = lambda *ops: reduce(np.kron, ops)


def btime(fn, reps=50):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


def harness(name, H_drift, H_drives, U_goal, T, dt):
    rng = np.random.default_rng(0)
    n_drives = len(H_drives)
    system = qc.QuantumSystem(H_drift, H_drives)
    N = system.levels
    Z = qc.NamedTrajectory(
        {
            "Ũ⃗": qc.unitary_geodesic(np.eye(N, dtype=complex), U_goal, T),
            "a": rng.standard_normal((n_drives, T)),
            "g": rng.standard_normal((n_drives, T)),
            "da": rng.standard_normal((n_drives, T)),
            "Δt": np.full((1, T), dt),
        },
        controls=("da",), timestep="Δt", goal={"Ũ⃗": qc.operator_to_iso_vec(U_goal)})
    P = qc.UnitaryPadeIntegrator("Ũ⃗", "a", system, Z)
    D = qc.DerivativeIntegrator("a", "da", Z)
    # the script's `g` component is a state WITHOUT an integrator: rows = "by_component" puts every integrator's rows at its state
    # component's position and leaves g's rows structurally empty, so the shapes are exactly the script's
    # (the constructor call itself is the script's line 41, `dynamics = QuantumDynamics(f, Z)`: the list and the trajectory, nothing else --
    #  `QCollocHIP.QuantumDynamics(f, Z)` on the Julia side)
    f = [P, D]
    dynamics = qc.QuantumDynamics(f, Z, rows="by_component")
    shape = (Z.dims.states * (Z.T - 1), Z.dim * Z.T + Z.global_dim)
    assert (int(dynamics.dims.n_rows), int(dynamics.dims.n_cols)) == shape
    z = Z.datavec
    mu = np.ones(Z.dims.states * (Z.T - 1))         # the script's own multiplier vector
    tF = btime(lambda: dynamics.F(z))
    tJ = btime(lambda: (getattr(dynamics, "∂F")(z), getattr(dynamics, "∂F_structure")))
    tH = btime(lambda: (getattr(dynamics, "μ∂²F")(z, mu), getattr(dynamics, "μ∂²F_structure")))
    # device-resident times of the same three evaluations
    dz = torch.from_numpy(z).cuda()
    dmu = torch.from_numpy(mu).cuda()
    dF = torch.empty(int(dynamics.dims.F_len), dtype=torch.float64, device="cuda")
    dJ = torch.empty(int(dynamics.dims.jac_nnz), dtype=torch.float64, device="cuda")
    dH = torch.empty(int(dynamics.dims.hess_nnz), dtype=torch.float64, device="cuda")

    def dev(fn, reps=200):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    dFt = dev(lambda: dynamics.F_dF_device(dz, dF, None))
    dJt = dev(lambda: dynamics.F_dF_device(dz, dF, dJ))
    dHt = dev(lambda: dynamics.mu_d2F_device(dz, dmu, dH))
    print(f"{name}: shape {shape}, kernel {dynamics.kernel}")
    print(f"   dynamics.F(Z.datavec)          {tF:9.1f} us host-visible   {dFt:7.1f} us on the device")
    print(f"   dynamics.∂F(Z.datavec)         {tJ:9.1f} us host-visible   {dJt:7.1f} us on the device (F and ∂F fused)")
    print(f"   dynamics.μ∂²F(Z.datavec, μ)    {tH:9.1f} us host-visible   {dHt:7.1f} us on the device")
    dynamics.close()


if __name__ == "__main__":
    # integrator_test_1qubit.jl: H_drift = Z, drives X, Y, goal = Hadamard-like, T = 100, dt = 0.1
    harness("1 qubit (integrator_test_1qubit.jl)", G["Z"], [G["X"], G["Y"]], G["H"], 100, 0.1)
    # integrator_test_script.jl: 4 qubits, H_drift = Z (x) X (x) X (x) X, two drives, goal X (x) X (x) X (x) X, T = 100, dt = 0.1
    harness("4 qubits (integrator_test_script.jl)", kron(G["Z"], G["X"], G["X"], G["X"]),
            [kron(G["X"], I2, I2, I2), kron(G["Y"], G["Y"], I2, G["Y"])], kron(G["X"], G["X"], G["X"], G["X"]), 100, 0.1)
