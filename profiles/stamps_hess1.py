#!/usr/bin/env python3
"""Timeline of the one-wave mu_d2F kernel (qc_mfma16_pade4_hess_anti_kernel, QC_STAMPS=1) at a trajectory of more than one device round.
python profiles/stamps_hess1.py [T=2000]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
import __graft_entry__ as g
qc = g.load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
inp = qc.config_inputs(3, T=T)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj, hess_align=16)
Z = torch.from_numpy(inp.traj.datavec).cuda()
mu = torch.from_numpy(np.random.default_rng(0).standard_normal(int(dyn.dims.n_rows))).cuda()
Hs = [torch.empty(dyn.dims.hess_nnz, dtype=torch.float64, device="cuda") for _ in range(16)]
for i in range(48): dyn.mu_d2F_device(Z, mu, Hs[i % 16])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(160): dyn.mu_d2F_device(Z, mu, Hs[i % 16])
e1.record(); torch.cuda.synchronize()
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 16).astype(np.int64)
names = {0: "entry", 11: "first two loads requested", 1: "every load requested", 2: "loads back", 3: "stage A issued", 5: "stage B issued", 4: "(a, a) stored (g2 kernel)",
         6: "tiles transposed (g2: first pair stored)", 7: "matrix blocks' stores issued", 8: "every store issued", 9: "drained"}
t0 = st[:, 0][st[:, 0] > 0].min()
rel = (st - t0) * 10.0 / 1e3
print(f"T={T}: kernel {dyn.kernel_names[1]}; launch-to-launch (stamped) {e0.elapsed_time(e1) * 1e3 / 160:.2f} us; span {rel[:, 9].max():.2f} us")
prev = None
for k in [0, 11, 1, 2, 3, 5, 4, 6, 7, 8, 9]:
    ok = st[:, k] > 0
    if not ok.any(): continue
    col = rel[:, k][ok]
    step = "" if prev is None else f"   (+{np.median(rel[:, k][ok] - rel[:, prev][ok]):.2f} per wave)"
    print(f"  {k:2d} {names[k]:30s} min {col.min():7.2f}  median {np.median(col):7.2f}  90% {np.percentile(col, 90):7.2f}  max {col.max():7.2f} us{step}")
    prev = k
life = rel[:, 9] - rel[:, 0]
print(f"wave lifetime: median {np.median(life):.2f} us, 90% {np.percentile(life, 90):.2f}, max {life.max():.2f}; started after 1 us: {(rel[:, 0] > 1.0).sum()} of {n}")
