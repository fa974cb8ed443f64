"""What an idle torch.cuda.synchronize() costs, and what it costs right after work on the current stream has been seen complete through
an event (the closing bracket of bench.py's timed region at K = 20)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
import __graft_entry__ as g
qc = g.load_package()

def idle_sync(tag):
    torch.cuda.synchronize()
    ts = []
    for _ in range(200):
        t0 = time.perf_counter(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"{tag}: idle torch.cuda.synchronize() median {np.median(ts) * 1e6:.1f} us, min {np.min(ts) * 1e6:.1f}")

x = torch.zeros(1, device="cuda")
idle_sync("no handle yet")
inp = qc.config_inputs(3)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
idle_sync("one handle (its two streams idle)")
Z = torch.from_numpy(inp.traj.datavec).cuda()
F = torch.empty(int(dyn.dims.F_len), dtype=torch.float64, device="cuda"); J = torch.empty(int(dyn.dims.jac_nnz), dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream()
call = dyn.bind_F_dF_device(Z, F, J, st)
for _ in range(50): call()
torch.cuda.synchronize()
ev = torch.cuda.Event(enable_timing=True); ev.record(st); torch.cuda.synchronize()
res = {"sync": [], "stream_sync": [], "event_sync": []}
for kind in res:
    for _ in range(100):
        for _ in range(20): call()
        ev.record(st)
        while not ev.query(): pass
        t0 = time.perf_counter()
        if kind == "sync": torch.cuda.synchronize()
        elif kind == "stream_sync": st.synchronize()
        else: ev.synchronize()
        res[kind].append(time.perf_counter() - t0)
    print(f"after 20 launches seen complete by event query: {kind} median {np.median(res[kind]) * 1e6:.1f} us, min {np.min(res[kind]) * 1e6:.1f}")
dyn.close()
idle_sync("handle destroyed")
