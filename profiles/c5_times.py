#!/usr/bin/env python3
"""Config 5 (4-qubit QFT, 2N = 32) on the device: F + dF, mu_d2F, and the one-call form, per launch, over rings of output vectors beyond
the Infinity Cache.  usage: python profiles/c5_times.py [T ...]   (QC_NO_ELL=1 for the dense-image kernels)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

qc = g.load_package()
Ts = [int(x) for x in sys.argv[1:]] or [500]
for T in Ts:
    inp = qc.config_inputs(5, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=int(os.environ.get("QC_BENCH_HESS_ALIGN", "16")))   # the device consumers' layout
    d = dyn.dims
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    mu = torch.from_numpy(np.random.default_rng(5).standard_normal(int(d.n_rows))).cuda()
    nb = max(2, min(8, int((800 << 20) // (8 * int(d.jac_nnz))) + 1))
    nh = max(2, min(24, int((700 << 20) // (8 * int(d.hess_nnz))) + 1))
    Fs = [torch.empty(int(d.F_len), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Js = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Hs = [torch.empty(int(d.hess_nnz), dtype=torch.float64, device="cuda") for _ in range(nh)]
    st = torch.cuda.current_stream()

    def timed(calls, n=300):
        for i in range(30):
            calls[i % len(calls)]()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(n):
            calls[i % len(calls)]()
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    jac = timed([dyn.bind_F_dF_device(Z, Fs[i], Js[i], st) for i in range(nb)])
    hes = timed([dyn.bind_mu_d2F_device(Z, mu, Hs[i], st) for i in range(nh)])
    both = timed([dyn.bind_F_dF_mu_d2F_device(Z, mu, Fs[i % nb], Js[i % nb], Hs[i % nh], st) for i in range(int(np.lcm(nb, nh)))])
    f_only = timed([dyn.bind_F_dF_device(Z, Fs[i], None, st) for i in range(nb)])
    n_int = int(d.n_intervals)
    jb = 8 * (inp.traj.dim * (n_int + 1) + (int(d.ddim) + int(d.jac_nnz_interval)) * n_int)
    hb = 8 * (inp.traj.dim * (n_int + 1) + (int(d.ddim) + int(d.hess_nnz_interval)) * n_int)
    print(f"T={T}: kernels {dyn.kernel_names} / {dyn.fused_kernel_name}: F+dF {jac:.2f} us ({jb / jac / 1e6:.2f} TB/s), mu_d2F {hes:.2f} us "
          f"({hb / hes / 1e6:.2f} TB/s), one call {both:.2f} us ({(jb + hb) / both / 1e6:.2f} TB/s), F only {f_only:.2f} us", flush=True)
    dyn.close()
