#!/usr/bin/env python3
"""Host-visible (PCIe-inclusive) times of the host-buffer entry points, one JSON line per process (the library reads its
QC_HOST_* switches once per process): qc_eval_F (new x), qc_eval_F_jac (new x), qc_eval_jac / qc_eval_hess with
qc_set_new_x(h, 0) (Ipopt's accepted point), and Ipopt's per-iteration sequence F(new x) -> dF -> mu_d2F.

    QC_HOST_THREADS=12 python profiles/host_path_r03.py [config] [T] [n_devices]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

qc = g.load_package()
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
T = int(sys.argv[2]) if len(sys.argv) > 2 else (1000 if cfg == 3 else 0)
ndev = int(sys.argv[3]) if len(sys.argv) > 3 else 0
inp = qc.config_inputs(cfg, T=T or None)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj, devices=[0] * ndev if ndev else None)
d = dyn.dims
rng = np.random.default_rng(0)
Zs = [inp.traj.datavec + 1e-3 * k * rng.standard_normal(inp.traj.datavec.size) for k in range(3)]
F, J = np.empty(int(d.F_len)), np.empty(int(d.jac_nnz))
H, mu = np.empty(int(d.hess_nnz)), rng.standard_normal(int(d.n_rows))


def timed(fn, reps=40):
    for i in range(4):
        fn(i)
    ts = []
    for i in range(reps):
        t0 = time.perf_counter()
        fn(i)
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    return float(np.median(ts)), float(ts.min())


out = {"config": cfg, "T": int(inp.traj.T), "devices": ndev or 1,
       "env": {k: v for k, v in os.environ.items() if k.startswith("QC_HOST")}}
dyn.set_new_x(True)
out["F_new_x_ms"], out["F_new_x_min"] = timed(lambda i: dyn.F(Zs[i % 3], out=F))
out["F_dF_new_x_ms"], out["F_dF_new_x_min"] = timed(lambda i: dyn.F_dF(Zs[i % 3], out=(F, J)))
out["dF_new_x_ms"], _ = timed(lambda i: dyn.dF(Zs[i % 3], out=J))
if d.hess_nnz:
    out["hess_new_x_ms"], out["hess_new_x_min"] = timed(lambda i: dyn.mu_d2F(Zs[i % 3], mu, out=H))
dyn.F(Zs[0], out=F)
dyn.set_new_x(False)
out["dF_same_x_ms"], out["dF_same_x_min"] = timed(lambda i: dyn.dF(Zs[0], out=J))
out["F_dF_same_x_ms"], _ = timed(lambda i: dyn.F_dF(Zs[0], out=(F, J)))
if d.hess_nnz:
    out["hess_same_x_ms"], out["hess_same_x_min"] = timed(lambda i: dyn.mu_d2F(Zs[0], mu, out=H))


def ipopt_iter(i):
    dyn.set_new_x(True)
    dyn.F(Zs[i % 3], out=F)          # the accepted trial point's residuals
    dyn.set_new_x(False)
    dyn.dF(Zs[i % 3], out=J)         # eval_jac_g at the same x
    if d.hess_nnz:
        dyn.mu_d2F(Zs[i % 3], mu, out=H)


out["ipopt_sequence_ms"], out["ipopt_sequence_min"] = timed(ipopt_iter)
dyn.set_new_x(True)
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in out.items()}), flush=True)
dyn.close()
