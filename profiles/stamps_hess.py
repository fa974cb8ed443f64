#!/usr/bin/env python3
"""Diagnostic timeline of the 2N = 16 Hessian kernel (QC_STAMPS=1): per-wave s_memrealtime checkpoints.
Run on the GPU box:  python profiles/stamps_hess.py [T]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
os.environ.setdefault("QC_HESS_TWO_WAVES", "0")     # the one-wave kernel (the two-wave kernel: profiles/stamps_fused.py T hess)
import __graft_entry__ as g

qc = g.load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
inp = qc.config_inputs(3, T=T)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = torch.from_numpy(inp.traj.datavec).cuda()
mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(dyn.dims.n_rows))).cuda()
Hs = [torch.empty(dyn.dims.hess_nnz, dtype=torch.float64, device="cuda") for _ in range(24)]
for i in range(24):
    dyn.mu_d2F_device(Z, mu, Hs[i])
torch.cuda.synchronize()
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 16).astype(np.int64)
names = {0: "kernel entry", 10: "kernel arguments read", 11: "first loads requested (amplitudes, timestep)", 1: "all loads requested",
         2: "loads back", 3: "stage A issued (-M1, T_k)", 5: "stage B issued (M2, Q_p)", 6: "tiles transposed",
         7: "matrix stores issued", 8: "all stores issued", 9: "drained"}
order = [0, 10, 11, 1, 2, 3, 5, 6, 7, 8, 9]
t0 = st[:, order][st[:, order] > 0].min()
rel = (st - t0) * 10.0 / 1e3
print(f"T={T}: {n} intervals; kernels {dyn.kernel_names}; span (first entry -> last drain) = {rel[:, 9].max():.2f} us")
prev = None
for k in order:
    ok = st[:, k] > 0
    if not ok.any():
        continue
    col = rel[:, k][ok]
    step = "" if prev is None else f"   (+{np.median(rel[:, k][ok] - rel[:, prev][ok]):.2f} per wave)"
    print(f"  {k:2d} {names[k]:44s} min {col.min():7.2f}  median {np.median(col):7.2f}  max {col.max():7.2f} us{step}")
    prev = k
