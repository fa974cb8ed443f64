import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
qc = g.load_package()
for T in (64, 126, 251, 501, 1001, 2001):
    inp = qc.config_inputs(5, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    d = dyn.dims
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    nb = max(2, min(8, int((800 << 20) // (8 * int(d.jac_nnz))) + 1))
    Fs = [torch.empty(int(d.F_len), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Js = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device="cuda") for _ in range(nb)]
    st = torch.cuda.current_stream()
    for i in range(20): dyn.F_dF_device(Z, Fs[i % nb], Js[i % nb], st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    e0.record(st)
    for i in range(n): dyn.F_dF_device(Z, Fs[i % nb], Js[i % nb], st)
    e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    mb = 8 * int(d.jac_nnz) / 1e6
    print(f"T={T:5d} intervals={d.n_intervals:5d} {mb:8.1f} MB  {us:8.2f} us  {mb / us / 1e3 * 1e3:6.2f} GB/ms = {mb/us:.3f} TB/s  nb={nb}", flush=True)
    dyn.close()
