#!/usr/bin/env python3
"""Residual-only launches (the line-search call on the device), stream events.   python profiles/f_only.py [config=3] [T ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

qc = g.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for T in [int(t) for t in sys.argv[2:]] or [0]:
    inp = qc.config_inputs(cfg, T=T or None)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    rng = np.random.default_rng(0)
    Zs = [torch.from_numpy(inp.traj.datavec + (1e-3 * rng.standard_normal(inp.traj.datavec.size) if k else 0.0)).cuda() for k in range(4)]
    Fb = [torch.empty(int(dyn.dims.F_len), dtype=torch.float64, device="cuda") for _ in range(8)]
    st = torch.cuda.current_stream()
    fl = [dyn.bind_F_dF_device(Zs[i & 3], Fb[i & 7], None, st) for i in range(8)]
    res = []
    for rep in range(3):
        for i in range(100):
            fl[i & 7]()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(2000):
            fl[i & 7]()
        e1.record(st)
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 2000)
    print(f"config {cfg} T={inp.traj.T} [{dyn.kernel_names[0]}]: F only " + " ".join(f"{x:.2f}" for x in res) + " us", flush=True)
