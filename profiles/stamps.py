#!/usr/bin/env python3
"""Diagnostic timeline of the MFMA kernel (QC_STAMPS=1): per-wave s_memrealtime checkpoints.
Run on the GPU box:  QC_STAMPS=1 python profiles/stamps.py [T]"""
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
dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=16)
Z = torch.from_numpy(inp.traj.datavec).cuda()
Fs = [torch.empty(dyn.dims.F_len, dtype=torch.float64, device="cuda") for _ in range(20)]
Js = [torch.empty(dyn.dims.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(20)]
for i in range(20):
    dyn.F_dF_device(Z, Fs[i], Js[i])
torch.cuda.synchronize()
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 16).astype(np.int64)
t0 = st[:, 0].min()
rel = (st - t0) * 10.0 / 1e3   # microseconds (100 MHz)
names = ["copy: start", "copy: loads done, G assembled", "copy: stores issued", "copy: drained", "comp: start", "comp: hand-off received",
         "comp: P1,P2", "comp: E^T ready", "comp: pair0 ready", "copy: kernel entry", "copy: kernel arguments read (QC_DEBUG_SKIP=4: scalar loads back)", "copy: vector loads back", "comp: drained"]
tcols = [c for c in range(13) if c != 11]      # slot 11 (and 15) hold HW_REG_HW_ID, not a time
t0 = st[:, tcols][st[:, tcols] > 0].min()
rel = (st - t0) * 10.0 / 1e3
print(f"T={T}: {n} intervals; kernel span (first start -> last drain) = {max(rel[:, 3].max(), rel[:, 12].max()):.2f} us")
for k, nm in enumerate(names):
    if k == 11 and not (int(os.environ.get("QC_DEBUG_SKIP", "0")) & 4):
        continue
    col = rel[:, k]
    col = col[st[:, k] > 0]
    if col.size:
        print(f"  {k:2d} {nm:16s} min {col.min():7.2f}  median {np.median(col):7.2f}  max {col.max():7.2f} us")

print("raw stamps of interval 0:", st[0].tolist())
dt_real = (st[:, 12] - st[:, 4]).astype(float) * 10e-9
dt_cyc = (st[:, 14] - st[:, 13]).astype(float)
ok = (dt_real > 0) & (dt_cyc > 0)
if ok.any():
    print(f"shader clock over the compute waves: median {np.median(dt_cyc[ok] / dt_real[ok]) / 1e9:.3f} GHz "
          f"(min {np.min(dt_cyc[ok] / dt_real[ok]) / 1e9:.3f}, max {np.max(dt_cyc[ok] / dt_real[ok]) / 1e9:.3f})")

# drain time of the copy waves by XCD (workgroup vb -> XCD vb % 8; interval = qc_xcd_remap(vb): invert it)
q8, r8 = divmod(n, 8)
def xcd_of_interval(b):
    for x in range(8):
        lo = x * (q8 + 1) if x < r8 else r8 * (q8 + 1) + (x - r8) * q8
        if lo <= b < lo + (q8 + 1 if x < r8 else q8):
            return x
    return -1
xcd = np.array([xcd_of_interval(b) for b in range(n)])
print("copy waves drained, by XCD (median / max us):", "  ".join(f"{x}: {np.median(rel[xcd == x, 3]):.2f}/{rel[xcd == x, 3].max():.2f}" for x in range(8)))
