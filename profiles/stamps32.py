#!/usr/bin/env python3
"""Diagnostic timeline of the MFMA32 kernel (config 5): per wave [loads done + G tile published, after barrier,
all issued, all acknowledged].  Run on the GPU box: python profiles/stamps32.py"""
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
Fs = [torch.empty(dyn.dims.F_len, dtype=torch.float64, device="cuda") for _ in range(6)]
Js = [torch.empty(dyn.dims.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(6)]
for i in range(12):
    dyn.F_dF_device(Z, Fs[i % 6], Js[i % 6])
torch.cuda.synchronize()
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 4, 4).astype(np.int64)
t0 = st[st > 0].min()
rel = (st - t0) * 0.01
names = ["G tile published", "after barrier", "all issued", "all acknowledged"]
roles = ["compute wave 0 (+residual)", "compute wave 1", "copy wave 2 (+deriv)", "copy wave 3"]
print(f"T={T}: {n} intervals; span {rel.max():.2f} us")
for w in range(4):
    print(" ", roles[w])
    for k in range(4):
        c = rel[:, w, k]
        print(f"     {names[k]:18s} min {c.min():6.2f} median {np.median(c):6.2f} max {c.max():6.2f} us")
