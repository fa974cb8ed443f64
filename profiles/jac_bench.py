import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
qc = g.load_package()
for cfg, T in ((3, 1000), (3, 2000), (3, 8000), (1, 50), (2, 200)):
    inp = qc.config_inputs(cfg, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    d = dyn.dims
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    nb = max(2, min(17, int((800 << 20) // (8 * int(d.jac_nnz))) + 1))
    Fs = [torch.empty(int(d.F_len), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Js = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device="cuda") for _ in range(nb)]
    st = torch.cuda.current_stream()
    n = 600 if T <= 2000 else 80
    for i in range(30): dyn.F_dF_device(Z, Fs[i % nb], Js[i % nb], st)
    torch.cuda.synchronize()
    best = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(n): dyn.F_dF_device(Z, Fs[i % nb], Js[i % nb], st)
        e1.record(st); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / n)
    print(f"variant {os.environ.get('QCOLLOC_HIP_VARIANT','product')} config {cfg} T={T}: F+dF {min(best):.2f} us", flush=True)
    dyn.close()
