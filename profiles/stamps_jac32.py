#!/usr/bin/env python3
"""Diagnostic timeline of the 2N = 32 F + dF kernel (QC_STAMPS=1; config 5): per interval, compute waves 0 / 1 and copy waves
0 / 1: loads requested, G assembled (barrier 2 passed), every store issued, every store acknowledged.
Run on the GPU box:  python profiles/stamps_jac32.py [T]"""
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
Fs = [torch.empty(dyn.dims.F_len, dtype=torch.float64, device="cuda") for _ in range(5)]
Js = [torch.empty(dyn.dims.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(5)]
for i in range(10):
    dyn.F_dF_device(Z, Fs[i % 5], Js[i % 5])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(50):
    dyn.F_dF_device(Z, Fs[i % 5], Js[i % 5])
e1.record()
torch.cuda.synchronize()
print(f"launch-to-launch time of this (stamped) instantiation: {e0.elapsed_time(e1) * 1e3 / 50:.2f} us")
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 4, 4).astype(np.int64)
t0 = st[st > 0].min()
rel = (st - t0) * 10.0 / 1e3
print(f"T={T}: {n} intervals; kernels {dyn.kernel_names}; span = {rel.max():.2f} us")
for wi, wn in enumerate(["compute wave 0", "compute wave 1", "copy wave 0", "copy wave 1"]):
    labels = ["loads landed (barrier 1)", "G assembled (barrier 2)", "every store issued", "every store acknowledged"]
    if wi == 1:      # this wave records the prologue instead (first pass of a persistent workgroup: kernel entry)
        labels = ["kernel entry", "arguments read, addresses known", "loads landed (barrier 1)", "every store acknowledged"]
    for k, nm in enumerate(labels):
        col = rel[:, wi, k][st[:, wi, k] > 0]
        if col.size:
            print(f"  {wn:15s} {nm:26s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")

# finish time of the copy waves by XCD (workgroup vb runs on XCD vb % 8; interval pair 2 * remap(vb) -> invert the remap)
n_wg = (n + 1) // 2
q, r = divmod(n_wg, 8)
def xcd_of_pair(p):          # inverse of qc_xcd_remap: pair index -> blockIdx % 8
    for x in range(8):
        lo = x * (q + 1) if x < r else r * (q + 1) + (x - r) * q
        hi = lo + (q + 1 if x < r else q)
        if lo <= p < hi:
            return x
    return -1
fin = np.maximum(rel[:, 2, 3], rel[:, 3, 3])
xcd = np.array([xcd_of_pair(b // 2) for b in range(n)])
print("copy waves' last acknowledged store by XCD (median / max us):",
      "  ".join(f"{x}: {np.median(fin[xcd == x]):.1f}/{fin[xcd == x].max():.1f}" for x in range(8)))
