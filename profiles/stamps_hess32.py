#!/usr/bin/env python3
"""Diagnostic timeline of the 2N = 32 Hessian kernel (QC_STAMPS=1; config 5): s_memrealtime checkpoints of waves 0 and 5 of
every interval's workgroup.  Run on the GPU box:  python profiles/stamps_hess32.py [T]"""
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
names = ["interval start", "loads back, G half tile published", "barrier 1 passed", "phase-1 products issued", "barrier 2 passed",
         "first half of phase 2 (w < 4: matrix blocks, else scalars)", "drive blocks and scalars done", "interval done"]
t0 = st[st > 0].min()
rel = (st - t0) * 10.0 / 1e3
per_wg = (n + 255) // 256
first = np.arange(n) % per_wg == 0          # (the XCD remap permutes workgroups, not the intervals inside a workgroup's run)
print(f"T={T}: {n} intervals, {per_wg} per workgroup; kernels {dyn.kernel_names}; span = {rel.max():.2f} us")
for label, sel in (("first interval of a workgroup", first), ("second interval", ~first)):
    print(f" {label}:")
    for wv, off in (("wave 0", 0), ("wave 5", 8)):
        prev = None
        for k in range(8):
            ok = sel & (st[:, off + k] > 0)
            if not ok.any():
                continue
            col = rel[ok, off + k]
            step = "" if prev is None else f"  (+{np.median(rel[ok, off + k] - rel[ok, off + prev]):.2f})"
            print(f"   {wv} {k} {names[k]:58s} median {np.median(col):6.2f}  max {col.max():6.2f} us{step}")
            prev = k
