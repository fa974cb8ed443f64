#!/usr/bin/env python3
"""One-off randomized stress run of the row-gather forms at 2N = 16 (not collected by pytest): python tests/stress_gather_gpu.py [trials] [seed]
Random three-qubit systems whose drives are weighted Pauli strings (one entry per generator row: qc_mfma16_ell_build accepts them), 1 .. 6 of
them over a dense random drift, trajectories of 1 .. 2.6 device rounds, free and fixed time steps, hess_align 0 / 16: mu_d2F alone (the
one-wave kernel's gathers beyond 1024 intervals), the one-call launch (its gather form for 1025 .. 4096 intervals) and F + dF, every value
against the C oracle; the one call bit for bit against the two launches."""
import itertools
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as g
import oracle.qc_oracle_c as oc
from oracle_bridge import assert_same_hessian_values, problem_from_inputs

qc = g.load_package()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
strings = ["".join(s) for s in itertools.product("IXYZ", repeat=3)][1:]
t0 = time.time()
worst = {"F": 0.0, "dF": 0.0, "H": 0.0}
names = {}
for trial in range(trials):
    m = int(rng.integers(1, 7))
    T = int(rng.choice([600, 1026, 1100, 1537, 1900, 2049, 2300, 2700, 4200]))
    A = rng.standard_normal((8, 8)) + 1j * rng.standard_normal((8, 8))
    drift = 0.3 * (A + A.conj().T) / 2 if rng.random() < 0.7 else 0.1 * qc.operator_from_string("ZZI")
    drives = [float(rng.uniform(0.3, 1.5)) * qc.operator_from_string(str(s)) for s in rng.choice(strings, size=m, replace=False)]
    free_time = bool(rng.integers(0, 2))
    align = int(rng.choice([0, 16]))
    inp = qc.unitary_smooth_pulse_inputs(qc.QuantumSystem(drift, drives), qc.GATES["TOFFOLI"], T, free_time=free_time)
    prob = problem_from_inputs(inp)
    prob.hess_align = align or 1
    co = oc.COracle(prob)
    Z = inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)
    mu = rng.standard_normal(prob.n_rows)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=align)
    tag = f"trial {trial}: m={m} T={T} free_time={free_time} align={align} kernels={dyn.kernel_names} / {dyn.fused_kernel_name}"
    names[dyn.fused_kernel_name] = names.get(dyn.fused_kernel_name, 0) + 1
    F, J = dyn.F_dF(Z)
    H = dyn.mu_d2F(Z, mu)
    Fr, Jr = co.F_dF(Z)
    Hr = co.mu_d2F(Z, mu)
    for k, a, b in (("F", F, Fr), ("dF", J, Jr), ("H", H, Hr)):
        e = np.abs(a - b).max() / max(1.0, np.abs(b).max())
        assert a.shape == b.shape and e < 1e-10, (tag, k, e)
        worst[k] = max(worst[k], e)
    dZ, dmu = torch.from_numpy(Z).cuda(), torch.from_numpy(mu).cuda()
    dF, dJ, dH = (torch.full((int(n),), float("nan"), dtype=torch.float64, device="cuda") for n in (dyn.dims.F_len, dyn.dims.jac_nnz, dyn.dims.hess_nnz))
    dyn.F_dF_mu_d2F_device(dZ, dmu, dF, dJ, dH)
    torch.cuda.synchronize()
    assert np.array_equal(dF.cpu().numpy(), F) and np.array_equal(dJ.cpu().numpy(), J), (tag, "one call differs from two launches")
    assert_same_hessian_values(dH.cpu().numpy(), H, dyn, tag)      # bit for bit but the (a, a) sums (round 6: Gram form in the stand-alone launch)
    dyn.close()
print(f"{trials} trials ok in {time.time() - t0:.0f} s; one-call kernels {names}; worst relative errors {worst}")
