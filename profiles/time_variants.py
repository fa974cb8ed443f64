#!/usr/bin/env python3
"""Device time of the secondary paths (extra evidence, not the metric): python profiles/time_variants.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

qc = g.load_package()


def time_dyn(name, inp, reps=200, kernel="auto", hess=True):
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, kernel=kernel)
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    d = dyn.dims
    F = torch.empty(int(d.F_len), dtype=torch.float64, device="cuda")
    J = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device="cuda") for _ in range(4)]
    for i in range(10):
        dyn.F_dF_device(Z, F, J[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        dyn.F_dF_device(Z, F, J[i % 4])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    mb = 8 * (int(d.F_len) + int(d.jac_nnz)) / 1e6
    line = f"{name:58s} F+dF {us:8.1f} us  ({mb:7.1f} MB out, {mb / us:5.2f} TB/s)"
    if hess and int(d.hess_nnz):
        mu = torch.randn(int(d.n_rows), dtype=torch.float64, device="cuda")
        H = torch.empty(int(d.hess_nnz), dtype=torch.float64, device="cuda")
        for i in range(5):
            dyn.mu_d2F_device(Z, mu, H)
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            dyn.mu_d2F_device(Z, mu, H)
        e1.record()
        torch.cuda.synchronize()
        line += f"   mu_d2F {e0.elapsed_time(e1) * 1e3 / reps:8.1f} us"
    print(line, flush=True)
    dyn.close()


s3 = qc.multi_qubit_system(3)
time_dyn("config 3 Pade-4 (mfma16)", qc.config_inputs(3))
time_dyn("config 3 Pade-4 (lds)", qc.config_inputs(3), kernel="lds")
for order in (2, 6, 12, 20):
    time_dyn(f"config 3 Pade-{order} (mfma16 any-order)", qc.unitary_smooth_pulse_inputs(s3, qc.GATES["TOFFOLI"], 1000, pade_order=order))
time_dyn("config 3 Pade-12 (lds)", qc.unitary_smooth_pulse_inputs(s3, qc.GATES["TOFFOLI"], 1000, pade_order=12), kernel="lds", reps=50)
time_dyn("config 3 exponential (mfma16-exp)", qc.unitary_smooth_pulse_inputs(s3, qc.GATES["TOFFOLI"], 1000, integrator="exponential"))
time_dyn("config 3 exponential (lds)", qc.unitary_smooth_pulse_inputs(s3, qc.GATES["TOFFOLI"], 1000, integrator="exponential"), reps=30, kernel="lds")
time_dyn("config 5 Pade-4 (mfma32)", qc.config_inputs(5))
time_dyn("config 5 exponential (mfma32-exp)", qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(4), qc.GATES["QFT16"], 500, integrator="exponential"), reps=50)
time_dyn("config 5 exponential (lds)", qc.unitary_smooth_pulse_inputs(qc.multi_qubit_system(4), qc.GATES["QFT16"], 500, integrator="exponential"), reps=5, kernel="lds")
s2 = qc.multi_qubit_system(2)
time_dyn("config 2 Pade-4 (mfma16, padded)", qc.config_inputs(2))
time_dyn("config 2 exponential (mfma16-exp, padded)", qc.unitary_smooth_pulse_inputs(s2, qc.GATES["CX"], 200, integrator="exponential"))
kets0 = [np.eye(8)[:, k] for k in range(4)]
kets1 = [np.eye(8)[:, (k + 1) % 8] for k in range(4)]
time_dyn("4 kets on 3 qubits, T=1000, Pade-4 (mfma16, masked)", qc.quantum_state_smooth_pulse_inputs(s3, kets0, kets1, 1000))
time_dyn("4 kets on 3 qubits, T=1000, Pade-4 (lds)", qc.quantum_state_smooth_pulse_inputs(s3, kets0, kets1, 1000), kernel="lds")
systems = [qc.QuantumSystem(s3.H_drift * f, s3.H_drives) for f in (0.95, 1.0, 1.05)]
time_dyn("sampling problem: 3 systems x config 3 (1 batched launch)", qc.unitary_sampling_inputs(systems, qc.GATES["TOFFOLI"], 1000))

# trajectory cost terms
inp = qc.config_inputs(3)
spec = qc.QuadraticRegularizer("a", inp.traj, 1e-2) + qc.QuadraticRegularizer("da", inp.traj, 1e-2) + qc.QuadraticRegularizer("dda", inp.traj, 1e-2)
obj = qc.TrajectoryObjective(spec, inp.traj)
Z = torch.from_numpy(inp.traj.datavec).cuda()
Jv = torch.zeros(1, dtype=torch.float64, device="cuda")
gr = torch.empty(Z.numel(), dtype=torch.float64, device="cuda")
Hv = torch.empty(obj.hess_nnz, dtype=torch.float64, device="cuda")
for _ in range(10):
    obj.eval_device(Z, Jv, gr, Hv)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    obj.eval_device(Z, Jv, gr, Hv)
e1.record()
torch.cuda.synchronize()
print(f"{'regularisers a/da/dda, config 3: J + grad + hess':58s}      {e0.elapsed_time(e1) * 1e3 / 200:8.1f} us (2 launches)")

# rollout
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
init = torch.from_numpy(qc.operator_to_iso_vec(np.eye(8, dtype=complex))).cuda()
out = torch.empty(128 * 1000, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    qc._lib.check(qc._lib.lib.qc_rollout_dev(dyn._h, Z.data_ptr(), init.data_ptr(), out.data_ptr(), s), dyn._h)
torch.cuda.synchronize()
e0.record()
for _ in range(50):
    qc._lib.check(qc._lib.lib.qc_rollout_dev(dyn._h, Z.data_ptr(), init.data_ptr(), out.data_ptr(), s), dyn._h)
e1.record()
torch.cuda.synchronize()
print(f"{'unitary rollout, config 3 (T=1000), 4 launches':58s}      {e0.elapsed_time(e1) * 1e3 / 50:8.1f} us")
