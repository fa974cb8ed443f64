#!/usr/bin/env python3
"""Diagnostic timeline of the sparse-drive 2N = 32 kernels (qc_mfma32_ell.hip; QC_STAMPS=1; config 5): s_memrealtime checkpoints of
the first compute wave and of copy wave 0 (F + dF, one-call form) or the last compute wave (mu_d2F alone) of every interval's
workgroup.  usage: python profiles/stamps_ell32.py [jac|hess|fused] [T]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
import __graft_entry__ as g

qc = g.load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "fused"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
inp = qc.config_inputs(5, T=T)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=int(os.environ.get("QC_BENCH_HESS_ALIGN", "16")))
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
for i in range(12):
    call(i % nb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for i in range(60):
    call(i % nb)
e1.record(st)
torch.cuda.synchronize()
n = int(d.n_intervals)
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
s = out.reshape(n, 16).astype(np.int64)
if os.environ.get("QC_STAMP_DUMP"):      # raw stamps (row = interval index b = qc_xcd_remap(blockIdx.x, n)) for offline analysis
    np.save(os.environ["QC_STAMP_DUMP"], s)
t0 = s[s > 0].min()
rel = (s - t0) * 10.0 / 1e3
print(f"{which}, T={T}: {n} intervals, one per workgroup; kernels {dyn.kernel_names} / {dyn.fused_kernel_name}; launch-to-launch of this (stamped) "
      f"instantiation {e0.elapsed_time(e1) * 1e3 / 60:.2f} us; span = {rel[s > 0].max():.2f} us")
comp = ["start", "G half tile written (at barrier A)", "barrier A passed", "state tiles seen (U, M)", "first drive: products issued",
        "first drive: G D / E seen", "first drive: blocks stored", "wave done"]
copy = ["start", "G half tile written (at barrier A)", "barrier A passed", "B^T / F^T tiles ready, first store next", "every copy's stores issued", "", "", ""]
second = copy if which != "hess" else comp
for name, off, labels in (("first compute wave", 0, comp), ("copy wave 0" if which != "hess" else "last compute wave", 8, second)):
    prev = None
    for k in range(8):
        ok = s[:, off + k] > 0
        if not ok.any() or not labels[k]:
            continue
        col = rel[ok, off + k]
        step = "" if prev is None else f"  (+{np.median(rel[ok, off + k] - rel[ok, off + prev]):.2f})"
        print(f"   {name:18s} {k} {labels[k]:44s} min {col.min():6.2f} median {np.median(col):6.2f}  90% {np.quantile(col, 0.9):6.2f}  99% {np.quantile(col, 0.99):6.2f}  max {col.max():6.2f} us{step}")
        prev = k
