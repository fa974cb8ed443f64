import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import __graft_entry__ as g
qc = g.load_package()
rng = np.random.default_rng(5)
for nq, K, T in [(3, 3, 1000), (2, 4, 200), (1, 2, 50)]:
    base = qc.multi_qubit_system(nq)
    systems = [qc.QuantumSystem(base.H_drift * (1.0 + 0.1 * rng.standard_normal()), base.H_drives) for _ in range(K)]
    gate = {1: "H", 2: "CNOT", 3: "TOFFOLI"}[nq]
    inp = qc.unitary_sampling_inputs(systems, qc.GATES[gate], T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    Z = inp.traj.datavec
    mu = rng.standard_normal(int(dyn.dims.n_rows))
    out = {}
    for name, f in [("F", lambda: dyn.F(Z)), ("F_dF", lambda: dyn.F_dF(Z)), ("hess", lambda: dyn.mu_d2F(Z, mu))]:
        for _ in range(5): f()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        out[name] = round(float(np.median(ts)) * 1e3, 4)
    mb = (int(dyn.dims.jac_nnz) + int(dyn.dims.F_len)) * 8 / 1e6
    print(f"sampling nq={nq} K={K} T={T}: ms {out}; F+dF bytes {mb:.1f} MB -> {mb / out['F_dF'] :.1f} GB/s; hess {int(dyn.dims.hess_nnz) * 8 / 1e6:.1f} MB", flush=True)
    dyn.close()
