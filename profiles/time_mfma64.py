#!/usr/bin/env python3
"""Device time of the 4 x 4-tile MFMA kernel (5 qubits) against the global-workspace kernel: python profiles/time_mfma64.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
qc = g.load_package()
def run(name, inp, kernel, reps):
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, kernel=kernel)
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    d = dyn.dims
    F = torch.empty(int(d.F_len), dtype=torch.float64, device="cuda")
    J = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device="cuda") for _ in range(2)]
    for i in range(3): dyn.F_dF_device(Z, F, J[i % 2])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): dyn.F_dF_device(Z, F, J[i % 2])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    mb = 8 * (int(d.F_len) + int(d.jac_nnz)) / 1e6
    e0.record()
    for i in range(reps): dyn.F_dF_device(Z, F, None)
    e1.record(); torch.cuda.synchronize()
    usF = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:40s} {kernel:5s} F+dF {us:9.1f} us ({mb:8.1f} MB, {mb/us:5.2f} TB/s)  F {usF:8.1f} us", flush=True)
    dyn.close()
s5 = qc.multi_qubit_system(5)
U = np.eye(32, dtype=complex)[:, ::-1].copy()
for T in (100, 257, 1000):
    inp = qc.unitary_smooth_pulse_inputs(s5, U, T)
    for k in ("auto", "lds"):
        if k == "lds" and T > 300: continue
        run(f"5 qubits T={T} m={s5.n_drives}", inp, k, 20 if k == "auto" else 5)
