#!/usr/bin/env python3
"""Diagnostic (variant build -DQC_ELL_STAMP_BARRIER, QCOLLOC_HIP_VARIANT=barrier, QC_STAMPS=1): when does each of the eight waves of a
workgroup of the sparse-drive 2N = 32 kernels arrive at barrier A?  usage: python profiles/stamps_ell32_barrier.py [jac|hess|fused] [T]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
os.environ["QCOLLOC_HIP_VARIANT"] = "barrier"
import __graft_entry__ as g

qc = g.load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "fused"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
inp = qc.config_inputs(5, T=T)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
d = dyn.dims
Z = torch.from_numpy(inp.traj.datavec).cuda()
mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(d.n_rows))).cuda()
nb = 6
Fs = [torch.empty(int(d.F_len), dtype=torch.float64, device="cuda") for _ in range(nb)]
Js = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device="cuda") for _ in range(nb)]
Hs = [torch.empty(int(d.hess_nnz), dtype=torch.float64, device="cuda") for _ in range(nb)]
st = torch.cuda.current_stream()
call = {"jac": lambda i: dyn.F_dF_device(Z, Fs[i], Js[i], st), "hess": lambda i: dyn.mu_d2F_device(Z, mu, Hs[i], st),
        "fused": lambda i: dyn.F_dF_mu_d2F_device(Z, mu, Fs[i], Js[i], Hs[i], st)}[which]
for i in range(24):
    call(i % nb)
torch.cuda.synchronize()
n = int(d.n_intervals)
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
s = out.reshape(n, 16).astype(np.int64)[:, :8]
t0 = s[s > 0].min()
rel = (s - t0) * 10.0 / 1e3
last = rel.max(axis=1)
print(f"{which}, T={T}: arrival of each wave at barrier A (us after the launch's first stamp); workgroups whose LAST wave arrives after 2x the median: "
      f"{int((last > 2 * np.median(last)).sum())} of {n}")
for w in range(8):
    col = rel[:, w]
    print(f"   wave {w}: median {np.median(col):6.2f}  90% {np.quantile(col, 0.9):6.2f}  99% {np.quantile(col, 0.99):6.2f}  max {col.max():6.2f}   last to arrive in {int((rel.argmax(axis=1) == w).sum()):4d} workgroups")
late = np.where(last > 2 * np.median(last))[0]
if late.size:
    print("   late workgroups: which wave is last:", np.bincount(rel[late].argmax(axis=1), minlength=8).tolist(), " their intervals (first 20):", late[:20].tolist())
    print("   in the late workgroups, median arrival per wave:", np.round(np.median(rel[late], axis=0), 2).tolist())
