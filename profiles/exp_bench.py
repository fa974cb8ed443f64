#!/usr/bin/env python3
"""Device time of the exponential integrator's launches (F + dF, mu_d2F) on a BASELINE-shaped problem with
`integrator=:exponential` (reference unitary_smooth_pulse_problem.jl:224-240), stream events over rings of output vectors.
    python profiles/exp_bench.py [config] [T]"""
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
inp = qc.config_inputs(cfg, T=T or None, integrator="exponential")
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
d = dyn.dims
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
Zh = inp.traj.datavec.copy()
if os.environ.get("QC_EXP_BENCH_DT"):      # every timestep set to this value: the squaring count follows ||dt G||_1 (0 squarings below 1/8)
    off = inp.traj.components["Δt"].start
    Zh[off::inp.traj.dim][:inp.traj.T] = float(os.environ["QC_EXP_BENCH_DT"])
Z = torch.from_numpy(Zh).to(dev)
mu = torch.from_numpy(rng.standard_normal(int(d.n_rows))).to(dev)
nb = max(2, -(-(640 << 20) // (8 * int(d.jac_nnz))))
nh = max(2, -(-(640 << 20) // (8 * int(d.hess_nnz))))
Fb = [torch.empty(int(d.F_len), dtype=torch.float64, device=dev) for _ in range(nb)]
Jb = [torch.empty(int(d.jac_nnz), dtype=torch.float64, device=dev) for _ in range(nb)]
Hb = [torch.empty(int(d.hess_nnz), dtype=torch.float64, device=dev) for _ in range(nh)]
st = torch.cuda.current_stream(dev)
jac = [dyn.bind_F_dF_device(Z, Fb[i], Jb[i], st) for i in range(nb)]
hes = [dyn.bind_mu_d2F_device(Z, mu, Hb[i], st) for i in range(nh)]


def timed(fn, steps=200):
    for i in range(20):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for i in range(steps):
        fn(i)
    e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


P = dyn._desc
n, m, n_int = 2 * P.N, P.m, int(d.n_intervals)
for rep in range(3):
    a = timed(lambda i: jac[i % nb]())
    b = timed(lambda i: hes[i % nh]())
    print(f"config {cfg} exponential T={inp.traj.T} [{dyn.kernel_names[0]} / {dyn.kernel_names[1]}]: F+dF {a:.2f} us, mu_d2F {b:.2f} us", flush=True)
dyn.close()
