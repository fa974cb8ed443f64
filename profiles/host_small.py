import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as g
qc = g.load_package()
def timed(fn, reps=200):
    for _ in range(10): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e6
for cfg, T in ((1, 50), (1, 100), (2, 200), (3, 100), (3, 1000)):
    inp = qc.config_inputs(cfg, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    Z = inp.traj.datavec
    F, J = np.empty(int(dyn.dims.F_len)), np.empty(int(dyn.dims.jac_nnz))
    H, mu = np.empty(int(dyn.dims.hess_nnz)), np.ones(int(dyn.dims.n_rows))
    t1 = timed(lambda: dyn.F_dF(Z, out=(F, J)), 200 if T < 1000 else 40)
    t2 = timed(lambda: dyn.F(Z, out=F), 200 if T < 1000 else 40)
    t3 = timed(lambda: dyn.mu_d2F(Z, mu, out=H), 200 if T < 1000 else 40)
    print(f"variant {os.environ.get('QCOLLOC_HIP_VARIANT','product')} config {cfg} T={T}: F+dF {t1:.1f} us, F {t2:.1f} us, mu_d2F {t3:.1f} us", flush=True)
    dyn.close()
