#!/usr/bin/env python3
"""Residual-only launch (what an Ipopt line-search trial costs on the device), stream events, configs 1, 2, 3.
    [QCOLLOC_HIP_VARIANT=name] python profiles/f_only_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

qc = g.load_package()
for cfg in (3, 1, 2, 5):
    inp = qc.config_inputs(cfg)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    Fs = [torch.empty(int(dyn.dims.F_len), dtype=torch.float64, device="cuda") for _ in range(8)]
    st = torch.cuda.current_stream()
    best = []
    for rep in range(3):
        for i in range(50):
            dyn.F_dF_device(Z, Fs[i % 8], None, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(1000):
            dyn.F_dF_device(Z, Fs[i % 8], None, st)
        e1.record(st)
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1))
    print(f"variant {os.environ.get('QCOLLOC_HIP_VARIANT', 'product')} config {cfg}: F only {min(best):.2f} us", flush=True)
    dyn.close()
