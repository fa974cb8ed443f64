"""Can the device entry points be captured in a HIP graph, and what does a replay of one Ipopt iteration's launches (dF + mu_d2F fused, one F)
cost against the stream launches?   python profiles/graph_probe.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
qc = g.load_package()
inp = qc.config_inputs(3)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
d = dyn.dims
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
Z = torch.from_numpy(inp.traj.datavec).to(dev)
mu = torch.from_numpy(rng.standard_normal(int(d.n_rows))).to(dev)
new = lambda n: torch.empty(int(n), dtype=torch.float64, device=dev)
F, J, H, F2 = new(d.F_len), new(d.jac_nnz), new(d.hess_nnz), new(d.F_len)
s = torch.cuda.Stream(dev)
with torch.cuda.stream(s):
    for _ in range(3):
        dyn.F_dF_mu_d2F_device(Z, mu, F, J, H, s)
        dyn.F_dF_device(Z, F2, None, s)
    s.synchronize()
    gr = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(gr, stream=s):
            dyn.F_dF_mu_d2F_device(Z, mu, F, J, H, torch.cuda.current_stream(dev))
            dyn.F_dF_device(Z, F2, None, torch.cuda.current_stream(dev))
        print("captured")
    except Exception as e:
        print("capture failed:", repr(e)[:300]); sys.exit(0)
def timed(fn, n=2000):
    for _ in range(100): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(torch.cuda.current_stream(dev))
    for _ in range(n): fn()
    b.record(torch.cuda.current_stream(dev))
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
st = torch.cuda.current_stream(dev)
f1 = dyn.bind_F_dF_mu_d2F_device(Z, mu, F, J, H, st)
f2 = dyn.bind_F_dF_device(Z, F2, None, st)
Jc, Hc = J.clone(), H.clone()
print("stream launches: %.2f us per iteration" % timed(lambda: (f1(), f2())))
print("graph replay   : %.2f us per iteration" % timed(lambda: gr.replay()))
torch.cuda.synchronize()
print("same values:", bool(torch.equal(J, Jc) and torch.equal(H, Hc)))
