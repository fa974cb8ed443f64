#!/usr/bin/env python3
"""mu_d2F (and F + dF) launch times of the BASELINE configs with stream events over a ring of output buffers.
Run on the GPU box:  [QCOLLOC_HIP_VARIANT=name] python profiles/hess_bench.py [config ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

qc = g.load_package()
# arguments: config ids, optionally with a suffix: 3e = exponential integrator, 3p6 / 3p8 = Pade order 6 / 8
cfgs = sys.argv[1:] or ["3", "5"]
for spec in cfgs:
    kw = {}
    if spec.endswith("e"):
        spec, kw = spec[:-1], {"integrator": "exponential"}
    elif "p" in spec:
        spec, order = spec.split("p")
        kw = {"pade_order": int(order)}
    c = int(spec)
    inp = qc.config_inputs(c, **kw)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    dims = dyn.dims
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(dims.n_rows))).cuda()
    nb = max(4, int((640 << 20) // (8 * int(dims.jac_nnz))) + 1)
    nb = min(nb, 24)
    nh = max(2, int((640 << 20) // (8 * max(1, int(dims.hess_nnz)))) + 1) if int(dims.hess_nnz) else 1   # its own ring: > 2 x the Infinity Cache
    nh = min(nh, 64)
    Hs = [torch.empty(int(dims.hess_nnz), dtype=torch.float64, device="cuda") for _ in range(nh)]
    Fs = [torch.empty(int(dims.F_len), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Js = [torch.empty(int(dims.jac_nnz), dtype=torch.float64, device="cuda") for _ in range(nb)]
    st = torch.cuda.current_stream()

    def timed(fn, steps=600):
        best = []
        for rep in range(3):
            for i in range(50):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for i in range(steps):
                fn(i)
            e1.record(st)
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3 / steps)
        return min(best), float(np.median(best))

    h = timed(lambda i: dyn.mu_d2F_device(Z, mu, Hs[i % nh], st)) if int(dims.hess_nnz) else (float("nan"), float("nan"))
    j = timed(lambda i: dyn.F_dF_device(Z, Fs[i % nb], Js[i % nb], st))
    print(f"variant {os.environ.get('QCOLLOC_HIP_VARIANT', 'product')} config {c} {kw} T={inp.traj.T} kernels {dyn.kernel_names}: "
          f"mu_d2F {h[0]:.2f} us (median of 3: {h[1]:.2f}), F+dF {j[0]:.2f} us (median {j[1]:.2f})", flush=True)
    dyn.close()
