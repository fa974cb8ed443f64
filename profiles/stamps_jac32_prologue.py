import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["QC_STAMPS"] = "1"
import __graft_entry__ as g
qc = g.load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 500
inp = qc.config_inputs(5, T=T)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = torch.from_numpy(inp.traj.datavec).cuda()
Fs = [torch.empty(dyn.dims.F_len, dtype=torch.float64, device="cuda") for _ in range(5)]
Js = [torch.empty(dyn.dims.jac_nnz, dtype=torch.float64, device="cuda") for _ in range(5)]
for i in range(10):
    dyn.F_dF_device(Z, Fs[i % 5], Js[i % 5])
torch.cuda.synchronize()
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 4, 4).astype(np.int64)
t0 = st[st > 0].min()
rel = (st - t0) * 10.0 / 1e3
for wi, wn in enumerate(["compute wave 0", "compute wave 1", "copy wave 0 (deriv)", "copy wave 1"]):
    labels = ["addresses known", "every load requested", "own loads back", "at barrier 1"]
    if wi == 1: labels[0] = "after barrier 1"
    if os.environ.get("QC_PHASE2"): labels = ["behind barrier 1", "G tile written", "knot loads requested (at barrier 2)", "behind barrier 2"]
    for k, nm in enumerate(labels):
        col = rel[:, wi, k][st[:, wi, k] > 0]
        if col.size: print(f"  {wn:20s} {nm:22s} min {col.min():6.2f}  median {np.median(col):6.2f}  max {col.max():6.2f} us")
# first-round workgroups (cold instruction cache) against later ones
for wi, wn in enumerate(["compute wave 0", "compute wave 1", "copy wave 0 (deriv)", "copy wave 1"]):
    if wi == 1: continue
    ok = (st[:, wi, 0] > 0) & (st[:, wi, 1] > 0)
    d_issue = (rel[:, wi, 1] - rel[:, wi, 0])[ok]
    d_back = (rel[:, wi, 2] - rel[:, wi, 1])[ok]
    start = rel[:, wi, 0][ok]
    first = start < 3.0
    if first.any() and (~first).any():
        print(f"  {wn:20s} addresses known -> every load requested: first round {np.median(d_issue[first]):.2f} us ({first.sum()} intervals), "
              f"later rounds {np.median(d_issue[~first]):.2f} us ({(~first).sum()}); -> loads back: {np.median(d_back[first]):.2f} / {np.median(d_back[~first]):.2f} us")
