#!/usr/bin/env python3
"""Device time of mu_d2F at 5 qubits: 4 x 4-tile MFMA Hessian kernel against the global-workspace kernel.
python profiles/time_mfma64_hess.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

qc = g.load_package()
s5 = qc.multi_qubit_system(5)
U = np.eye(32, dtype=complex)[:, ::-1].copy()
for T in (100, 257, 1000):
    inp = qc.unitary_smooth_pulse_inputs(s5, U, T)
    for kernel in ("auto", "lds"):
        if kernel == "lds" and T > 300:
            continue
        dyn = qc.QuantumDynamics(inp.integrators, inp.traj, kernel=kernel)
        Z = torch.from_numpy(inp.traj.datavec).cuda()
        d = dyn.dims
        mu = torch.randn(int(d.n_rows), dtype=torch.float64, device="cuda")
        H = torch.empty(int(d.hess_nnz), dtype=torch.float64, device="cuda")
        reps = 20 if kernel == "auto" else 3
        for _ in range(2):
            dyn.mu_d2F_device(Z, mu, H)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dyn.mu_d2F_device(Z, mu, H)
        e1.record()
        torch.cuda.synchronize()
        print(f"5 qubits T={T:5d} m={s5.n_drives} {kernel:5s} mu_d2F {e0.elapsed_time(e1) * 1e3 / reps:9.1f} us  ({int(d.hess_nnz) * 8 / 1e6:7.1f} MB out)", flush=True)
        dyn.close()
