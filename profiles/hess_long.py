import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
qc = g.load_package()
for T in [int(t) for t in sys.argv[1:]]:
    inp = qc.config_inputs(3, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=16)
    d = dyn.dims
    Z = torch.from_numpy(inp.traj.datavec).cuda()
    mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(d.n_rows))).cuda()
    nh = max(2, -(-(640 << 20) // (8 * int(d.hess_nnz))))
    Hb = [torch.empty(int(d.hess_nnz), dtype=torch.float64, device="cuda") for _ in range(nh)]
    st = torch.cuda.current_stream()
    hes = [dyn.bind_mu_d2F_device(Z, mu, Hb[i], st) for i in range(nh)]
    for i in range(50): hes[i % nh]()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    e0.record(st)
    for i in range(n): hes[i % nh]()
    e1.record(st); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"T={T}: mu_d2F {us:.2f} us  ({8 * int(d.hess_nnz) / us / 1e6:.2f} TB/s) kernel {dyn.kernel_names[1]}", flush=True)
