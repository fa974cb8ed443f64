#!/usr/bin/env python3
"""Does the previous launch's write-back slow the next launch's loads?  Stamps of the config-5 mu_d2F kernel launched ALONE (device idle
for a millisecond before it) against back to back.  usage: QC_STAMP_DUMP=prefix python profiles/single_launch_stamps.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
import __graft_entry__ as g

qc = g.load_package()
inp = qc.config_inputs(5, T=500)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=16)
d = dyn.dims
Z = torch.from_numpy(inp.traj.datavec).cuda()
mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(d.n_rows))).cuda()
Hs = [torch.empty(int(d.hess_nnz), dtype=torch.float64, device="cuda") for _ in range(8)]
st = torch.cuda.current_stream()
n = int(d.n_intervals)


def read():
    out = np.zeros(n * 16, dtype=np.uint64)
    qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
    return out.reshape(n, 16).astype(np.int64)


for i in range(16):
    dyn.mu_d2F_device(Z, mu, Hs[i % 8], st)
torch.cuda.synchronize()
np.save(os.environ.get("QC_STAMP_DUMP", "gpurun_out/r05_single") + "_b2b.npy", read())
time.sleep(0.01)
dyn.mu_d2F_device(Z, mu, Hs[3], st)
torch.cuda.synchronize()
np.save(os.environ.get("QC_STAMP_DUMP", "gpurun_out/r05_single") + "_alone.npy", read())
