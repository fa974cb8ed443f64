#!/usr/bin/env python3
"""Device time of dF + mu_d2F at one point: qc_eval_F_jac_hess_dev (one fused launch at 2N = 16) against the two launches, over
rings of output vectors, stream events.   python profiles/fused_bench.py [config] [T]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

qc = g.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
T = int(sys.argv[2]) if len(sys.argv) > 2 else 0
inp = qc.config_inputs(cfg, T=T or None)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=int(os.environ.get("QC_BENCH_HESS_ALIGN", "16")))   # the device consumers' layout
d = dyn.dims
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
Z = torch.from_numpy(inp.traj.datavec).to(dev)
NZ = int(os.environ.get("QC_BENCH_NZ", "1"))     # distinct trajectory vectors cycled through (bench.py cycles 4)
Zs = [Z] + [torch.from_numpy(inp.traj.datavec + 1e-3 * rng.standard_normal(inp.traj.datavec.size)).to(dev) for _ in range(NZ - 1)]
mu = torch.from_numpy(rng.standard_normal(int(d.n_rows))).to(dev)
nb = max(2, -(-(640 << 20) // (8 * int(d.jac_nnz))))
nh = max(2, -(-(640 << 20) // (8 * int(d.hess_nnz))))
Fb = [torch.empty(int(d.F_len), dtype=torch.float64, device=dev) for _ in range(nb)]
Jb = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device=dev) for _ in range(nb)]
Hb = [torch.empty(int(d.hess_nnz), dtype=torch.float64, device=dev) for _ in range(nh)]
st = torch.cuda.current_stream(dev)
n = int(np.lcm(nb, nh))
jac = [dyn.bind_F_dF_device(Zs[i % NZ], Fb[i % nb], Jb[i % nb], st) for i in range(n)]
hes = [dyn.bind_mu_d2F_device(Zs[i % NZ], mu, Hb[i % nh], st) for i in range(n)]
fus = [dyn.bind_F_dF_mu_d2F_device(Zs[i % NZ], mu, Fb[i % nb], Jb[i % nb], Hb[i % nh], st) for i in range(n)]


def timed(fn, steps=1000):
    for i in range(100):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(steps):
        fn(i)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


for rep in range(3):
    a = timed(lambda i: jac[i % n]())
    b = timed(lambda i: hes[i % n]())
    c = timed(lambda i: (jac[i % n](), hes[i % n]()))
    f = timed(lambda i: fus[i % n]())
    print(f"config {cfg} T={inp.traj.T} [{dyn.fused_kernel_name}]: F+dF {a:.2f} us, mu_d2F {b:.2f} us, both back to back {c:.2f} us, one call {f:.2f} us", flush=True)
dyn.close()
