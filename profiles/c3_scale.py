import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
qc = g.load_package()
for T in (1000, 2000, 8000, 32000):
    inp = qc.config_inputs(3, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    d = dyn.dims
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    nb = max(2, min(17, int((800 << 20) // (8 * int(d.jac_nnz))) + 1))
    Fs = [torch.empty(int(d.F_len), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Js = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device="cuda") for _ in range(nb)]
    st = torch.cuda.current_stream()
    n = 400 if T <= 2000 else 60
    for i in range(20): dyn.F_dF_device(Z, Fs[i % nb], Js[i % nb], st)
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(n): dyn.F_dF_device(Z, Fs[i % nb], Js[i % nb], st)
        e1.record(st); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / n)
    us = min(best)
    mb = (8 * (int(d.jac_nnz) + int(d.F_len)) + 8 * Z.numel()) / 1e6
    print(f"variant {os.environ.get('QCOLLOC_HIP_VARIANT','product')} T={T:6d} {mb:8.1f} MB {us:9.2f} us {mb/us:6.3f} TB/s", flush=True)
    dyn.close()
