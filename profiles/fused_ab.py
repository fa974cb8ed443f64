#!/usr/bin/env python3
"""The one-call 2N = 16 launch, dense-image form against row-gather form, INSIDE one process on the same buffers (a build with
-DQC_FUSED_ELL_DYNAMIC reads QC_FUSED_ELL at every launch): long streams are bimodal from process to process (where the buffers land),
so an A/B across processes says little.   QCOLLOC_HIP_VARIANT=dyn python profiles/fused_ab.py T [T ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

qc = g.load_package()
for T in [int(t) for t in sys.argv[1:]]:
    inp = qc.config_inputs(3, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=16)
    d = dyn.dims
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(d.n_rows))).cuda()
    nb = max(2, -(-(640 << 20) // (8 * int(d.jac_nnz))))
    Fb = [torch.empty(int(d.F_len), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Jb = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Hb = [torch.empty(int(d.hess_nnz), dtype=torch.float64, device="cuda") for _ in range(nb)]
    st = torch.cuda.current_stream()
    fus = [dyn.bind_F_dF_mu_d2F_device(Z, mu, Fb[i], Jb[i], Hb[i], st) for i in range(nb)]
    res = {"0": [], "1": []}
    for rep in range(4):
        for mode in ("0", "1"):
            os.environ["QC_FUSED_ELL"] = mode
            for i in range(3 * nb):
                fus[i % nb]()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = max(200, 20 * nb)
            e0.record(st)
            for i in range(n):
                fus[i % nb]()
            e1.record(st)
            torch.cuda.synchronize()
            res[mode].append(e0.elapsed_time(e1) * 1e3 / n)
    print(f"T={T}: one call, images " + " ".join(f"{x:.2f}" for x in res["0"]) + "  |  row gathers " + " ".join(f"{x:.2f}" for x in res["1"]) + " us", flush=True)
