#!/usr/bin/env python3
"""Diagnostic timeline of qc_mfma32_exp_kernel (a -DQC_X32_STAMPS variant build: profiles/build_variant.sh x32s qc_mfma32_exp.hip -DQC_X32_STAMPS,
QCOLLOC_HIP_VARIANT=x32s, QC_STAMPS=1): config 5's system with the exponential integrator, waves 0 and 5 of every workgroup.
    python profiles/stamps_exp32.py [T]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
import __graft_entry__ as g  # noqa: E402

qc = g.load_package()
hess = len(sys.argv) > 1 and sys.argv[1] == "hess"      # python profiles/stamps_exp32.py hess [T]: qc_mfma32_exp_hess_kernel (its own variant build)
if hess:
    sys.argv.pop(1)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
inp = qc.config_inputs(5, T=T, integrator="exponential")
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = torch.from_numpy(inp.traj.datavec).cuda()
Fs = [torch.empty(dyn.dims.F_len, dtype=torch.float64, device="cuda") for _ in range(5)]
Js = [torch.empty(dyn.dims.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(5)]
mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(dyn.dims.n_rows))).cuda()
Hs = [torch.empty(dyn.dims.hess_nnz, dtype=torch.float64, device="cuda") for _ in range(5)]
run = (lambda i: dyn.mu_d2F_device(Z, mu, Hs[i % 5])) if hess else (lambda i: dyn.F_dF_device(Z, Fs[i % 5], Js[i % 5]))
for i in range(10):
    run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(50):
    run(i)
e1.record()
torch.cuda.synchronize()
print(f"launch-to-launch time of this (stamped) build: {e0.elapsed_time(e1) * 1e3 / 50:.2f} us; kernels {dyn.kernel_names}")
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 2, 8).astype(np.int64)
t0 = st[st > 0].min()
rel = (st - t0) * 10.0 / 1e3
labels_h = ["kernel entry", "loads, W / V, norm, chains set up", "Horner steps done", "squarings done", "shared outputs (E V, G^T W, G E) published",
            "(U, a) block stored", "(a, a) row done", "wave done"]
labels = labels_h if hess else ["kernel entry", "loads in, G half tile published", "barrier passed", "norm, Y, chains set up", "Horner steps done", "squarings done",
          "every output issued", "every store acknowledged"]
for wi, wn in enumerate(["wave 0", "wave 7" if hess else "wave 5"]):
    prev = None
    for k, nm in enumerate(labels):
        col = rel[:, wi, k][st[:, wi, k] > 0]
        if col.size:
            med = float(np.median(col))
            print(f"  {wn} {nm:34s} median {med:7.2f} us" + (f"   (+{med - prev:5.2f})" if prev is not None else ""))
            prev = med
