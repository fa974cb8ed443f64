#!/usr/bin/env python3
"""Diagnostic timeline of the 4 x 4-tile Hessian kernel (5 qubits): s_memrealtime checkpoints of wave 0 (matrix wave) and
wave 8 (aux wave).  Run on the GPU box: python profiles/stamps64h.py [T]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
import __graft_entry__ as g

qc = g.load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
s5 = qc.multi_qubit_system(5)
U = np.eye(32, dtype=complex)[:, ::-1].copy()
inp = qc.unitary_smooth_pulse_inputs(s5, U, T)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = torch.from_numpy(inp.traj.datavec).cuda()
mu = torch.randn(dyn.dims.n_rows, dtype=torch.float64, device="cuda")
Hs = [torch.empty(dyn.dims.hess_nnz, dtype=torch.float64, device="cuda") for _ in range(2)]
s = torch.cuda.current_stream().cuda_stream
for i in range(4):
    qc._lib.check(qc._lib.lib.qc_eval_hess_dev(dyn._h, Z.data_ptr(), mu.data_ptr(), Hs[i % 2].data_ptr(), s), dyn._h)
torch.cuda.synchronize()
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 2, 8).astype(np.int64)
t0 = st[st > 0].min()
rel = (st - t0) * 0.01
names = ["start", "phase 0 done (barrier 1)", "M1 / G D done (barrier 2)", "M2 + (U,h) blocks issued", "drive 0 done", "drive 1 done",
         "all drives done", "Gram + scalar blocks done"]
print(f"T={T}: {n} intervals; span {rel.max():.2f} us")
for w, role in enumerate(["wave 0 (matrix)", "wave 8 (aux)"]):
    print(" ", role)
    for k in range(8):
        c = rel[:, w, k]
        print(f"     {names[k]:28s} min {c.min():7.2f} median {np.median(c):7.2f} max {c.max():7.2f} us")
d = rel[:, :, 1:] - rel[:, :, :-1]
print("  phase durations (median over intervals), wave 0 / wave 8:")
for k in range(7):
    print(f"     {names[k]:28s} -> {names[k+1]:28s} {np.median(d[:, 0, k]):7.2f} / {np.median(d[:, 1, k]):7.2f} us")
