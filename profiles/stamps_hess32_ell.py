#!/usr/bin/env python3
"""Diagnostic timeline of the sparse-drive 2N = 32 Hessian kernel (qc_mfma32_ell.hip; QC_STAMPS=1; config 5): s_memrealtime
checkpoints of waves 0 and 5 of every interval's workgroup.  Run on the GPU box:  python profiles/stamps_hess32_ell.py [T]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
import __graft_entry__ as g

qc = g.load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 500
inp = qc.config_inputs(5, T=T)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = torch.from_numpy(inp.traj.datavec).cuda()
mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(dyn.dims.n_rows))).cuda()
Hs = [torch.empty(dyn.dims.hess_nnz, dtype=torch.float64, device="cuda") for _ in range(6)]
for i in range(12):
    dyn.mu_d2F_device(Z, mu, Hs[i % 6])
torch.cuda.synchronize()
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 16).astype(np.int64)
names = ["workgroup start", "loads back, G half tile and state tiles published", "barrier 1 passed", "phase-1 products issued", "barrier 2 passed",
         "matrix blocks of the drive stored", "scalar blocks done", "workgroup done"]
t0 = st[st > 0].min()
rel = (st - t0) * 10.0 / 1e3
print(f"T={T}: {n} intervals, one per workgroup; kernels {dyn.kernel_names}; span = {rel.max():.2f} us")
start = rel[:, 0]
order = np.argsort(start)
print(f" workgroup start: min {start.min():.2f}, median {np.median(start):.2f}, 90 % {np.quantile(start, 0.9):.2f}, max {start.max():.2f} us")
for wv, off in (("wave 0", 0), ("wave 5", 8)):
    prev = None
    for k in range(8):
        ok = st[:, off + k] > 0
        if not ok.any():
            continue
        col = rel[ok, off + k]
        step = "" if prev is None else f"  (+{np.median(rel[ok, off + k] - rel[ok, off + prev]):.2f})"
        print(f"   {wv} {k} {names[k]:52s} median {np.median(col):6.2f}  max {col.max():6.2f} us{step}")
        prev = k
