#!/usr/bin/env python3
"""Diagnostic timeline of the fused F + dF + mu_d2F kernel (qc_mfma_fused.hip, QC_STAMPS=1): per-wave s_memrealtime checkpoints of
both roles.  Run on the GPU box:  python profiles/stamps_fused.py [T] [hess]   (hess: the Hessian-only form, qc_eval_hess_dev)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
import __graft_entry__ as g

qc = g.load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
inp = qc.config_inputs(3, T=T)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=int(os.environ.get("QC_BENCH_HESS_ALIGN", "16")))
Z = torch.from_numpy(inp.traj.datavec).cuda()
mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(dyn.dims.n_rows))).cuda()
nb = 16
Fs = [torch.empty(dyn.dims.F_len, dtype=torch.float64, device="cuda") for _ in range(nb)]
Js = [torch.empty(dyn.dims.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(nb)]
Hs = [torch.empty(dyn.dims.hess_nnz, dtype=torch.float64, device="cuda") for _ in range(nb)]
hess_only = len(sys.argv) > 2 and sys.argv[2] == "hess"
def call(i):
    if hess_only:
        dyn.mu_d2F_device(Z, mu, Hs[i])
    else:
        dyn.F_dF_mu_d2F_device(Z, mu, Fs[i], Js[i], Hs[i])


for i in range(3 * nb):
    call(i % nb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10 * nb):
    call(i % nb)
e1.record()
torch.cuda.synchronize()
l2l = e0.elapsed_time(e1) * 1e3 / (10 * nb)
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 16).astype(np.int64)
names = {10: "copy wave: entry", 11: "copy wave: loads back, G assembled", 12: "copy wave: tile copies issued", 13: "copy wave: all its stores issued",
         14: "copy wave: second barrier passed", 15: "copy wave: (a, a) sums stored, done",
         0: "compute wave: entry", 1: "compute wave: hand-off received", 2: "compute wave: F + dF products, transposes done",
         3: "compute wave: F + dF stores issued", 4: "compute wave: stage A done, tiles parked", 5: "compute wave: second barrier passed",
         6: "compute wave: stage B issued", 7: "compute wave: Hessian matrix stores issued", 8: "compute wave: all stores issued", 9: "compute wave: drained"}
if hess_only:   # qc_mfma_hess2.hip: wave 0 does the one-wave kernel's work in a new order, wave 1 the (a, a) sums
    names = {0: "wave 0: entry", 1: "wave 0: every load requested", 2: "wave 0: loads back", 3: "wave 0: stage A issued",
             4: "wave 0: T_k parked, barrier passed", 5: "wave 0: stage B, first part issued", 6: "wave 0: first part's blocks stored",
             7: "wave 0: every matrix block stored", 8: "wave 0: every store issued", 9: "wave 0: drained",
             10: "wave 1: entry", 11: "wave 1: released", 14: "wave 1: its drive pair's blocks stored", 12: "wave 1: (a, a) products through", 13: "wave 1: done"}
if os.environ.get("QC_STAMP_DUMP"):      # raw stamps (row = interval index b = qc_xcd_remap(blockIdx.x, n)) for offline analysis
    np.save(os.environ["QC_STAMP_DUMP"], st)
t0 = st[st > 0].min()
rel = (st - t0) * 10.0 / 1e3
print(f"launch-to-launch of this (stamped) instantiation: {l2l:.2f} us")
print(f"T={T}: {n} intervals; one call = {dyn.kernel_names[1] if hess_only else dyn.fused_kernel_name}; span (first entry -> last compute wave drained) = {rel[:, 9].max():.2f} us, last stamp of any wave {rel[st > 0].max():.2f} us")
for order in (([10, 11, 14, 12, 13] if hess_only else [10, 11, 12, 13, 14, 15]), [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]):
    prev = None
    for k in order:
        ok = st[:, k] > 0
        if not ok.any():
            continue
        col = rel[:, k][ok]
        step = "" if prev is None else f"   (+{np.median(rel[:, k][ok] - rel[:, prev][ok]):.2f} per wave)"
        print(f"  {k:2d} {names[k]:52s} min {col.min():7.2f}  median {np.median(col):7.2f}  max {col.max():7.2f} us{step}")
        prev = k
