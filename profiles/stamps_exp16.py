#!/usr/bin/env python3
"""Diagnostic timeline of qc_mfma16_exp_hess_kernel (a -DQC_XH_STAMPS variant build: profiles/build_variant.sh xhs qc_mfma_exp_hess.hip -DQC_XH_STAMPS,
QCOLLOC_HIP_VARIANT=xhs, QC_STAMPS=1): config 3 with the exponential integrator, the two waves of every workgroup.
    python profiles/stamps_exp16.py [T]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
import __graft_entry__ as g  # noqa: E402

qc = g.load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
inp = qc.config_inputs(3, T=T, integrator="exponential")
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = torch.from_numpy(inp.traj.datavec).cuda()
mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(dyn.dims.n_rows))).cuda()
Hs = [torch.empty(dyn.dims.hess_nnz, dtype=torch.float64, device="cuda") for _ in range(5)]
for i in range(10):
    dyn.mu_d2F_device(Z, mu, Hs[i % 5])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(50):
    dyn.mu_d2F_device(Z, mu, Hs[i % 5])
e1.record()
torch.cuda.synchronize()
print(f"launch-to-launch time of this (stamped) build: {e0.elapsed_time(e1) * 1e3 / 50:.2f} us; kernels {dyn.kernel_names}")
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 2, 8).astype(np.int64)
t0 = st[st > 0].min()
rel = (st - t0) * 10.0 / 1e3
labels = ["kernel entry", "loads, G, norm, W / V", "Horner steps done", "squarings done", "(U, a) blocks stored", "every output issued", "every store acknowledged"]
early = rel[:, :, 1] < 6.0          # waves whose first products got matrix-pipe slots at once / waves that waited for an older wave
for grp, sel in (("waves that start their chains at once", early), ("waves that start behind an older wave of their SIMD", ~early)):
    cols = [rel[:, :, k][sel & (st[:, :, k] > 0)] for k in range(len(labels))]
    if cols[0].size:
        print(f"  {grp} ({cols[0].size} of {2 * n}): " + ", ".join(f"{nm} {float(np.median(c)):.1f}" for nm, c in zip(labels, cols) if c.size))
for wi in range(2):
    prev = None
    for k, nm in enumerate(labels):
        col = rel[:, wi, k][st[:, wi, k] > 0]
        if col.size:
            med = float(np.median(col))
            print(f"  wave {wi} {nm:28s} median {med:7.2f} us  (min {col.min():6.2f}, max {col.max():6.2f})" + (f"   (+{med - prev:5.2f})" if prev is not None else ""))
            prev = med
