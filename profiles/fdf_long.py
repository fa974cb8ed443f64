import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
qc = g.load_package()
for T in [int(t) for t in sys.argv[1:]]:
    inp = qc.config_inputs(3, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=16)
    d = dyn.dims
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    nb = max(2, -(-(640 << 20) // (8 * int(d.jac_nnz))))
    Fb = [torch.empty(int(d.F_len), dtype=torch.float64, device="cuda") for _ in range(nb)]
    Jb = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device="cuda") for _ in range(nb)]
    st = torch.cuda.current_stream()
    jl = [dyn.bind_F_dF_device(Z, Fb[i], Jb[i], st) for i in range(nb)]
    for i in range(3 * nb): jl[i % nb]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = max(300, 20 * nb)
    e0.record(st)
    for i in range(n): jl[i % nb]()
    e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"T={T}: F+dF {us:.2f} us ({8 * (int(d.jac_nnz) + int(d.F_len)) / us / 1e6:.2f} TB/s)", flush=True)
