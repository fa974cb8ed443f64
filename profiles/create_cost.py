"""Set-up costs a consumer pays once per problem: qc_create, the two structure calls, the first evaluation (staging buffers, pinned ring)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
import __graft_entry__ as g
qc = g.load_package()
import torch; torch.zeros(1, device="cuda")
for cfg in (1, 3, 5, 4):
    inp = qc.config_inputs(cfg)
    t0 = time.perf_counter(); dyn = qc.QuantumDynamics(inp.integrators, inp.traj); t1 = time.perf_counter()
    s = dyn.dF_structure; t2 = time.perf_counter()
    hs = dyn.mu_d2F_structure; t3 = time.perf_counter()
    Z = inp.traj.datavec
    F, J = dyn.F_dF(Z); t4 = time.perf_counter()
    F, J = dyn.F_dF(Z); t5 = time.perf_counter()
    H = dyn.mu_d2F(Z, np.ones(int(dyn.dims.n_rows))); t6 = time.perf_counter()
    print(f"config {cfg} (T = {inp.traj.T}): create {1e3 * (t1 - t0):.1f} ms, dF_structure ({s[0].size} entries) {1e3 * (t2 - t1):.1f} ms, "
          f"mu_d2F_structure {1e3 * (t3 - t2):.1f} ms, first F_dF (ring built) {1e3 * (t4 - t3):.1f} ms, second {1e3 * (t5 - t4):.2f} ms, first mu_d2F {1e3 * (t6 - t5):.1f} ms")
    dyn.close()
