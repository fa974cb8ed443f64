import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
qc = g.load_package()
inp = qc.config_inputs(3, T=257)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
rng = np.random.default_rng(1)
Z = torch.from_numpy(inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)).cuda()
mu = torch.from_numpy(rng.standard_normal(int(dyn.dims.n_rows))).cuda()
new = lambda n: torch.zeros(int(n), dtype=torch.float64, device="cuda")
F1, J1, H1 = new(dyn.dims.F_len), new(dyn.dims.jac_nnz), new(dyn.dims.hess_nnz)
F2, J2, H2 = new(dyn.dims.F_len), new(dyn.dims.jac_nnz), new(dyn.dims.hess_nnz)
dyn.F_dF_device(Z, F1, J1); dyn.mu_d2F_device(Z, mu, H1)
dyn.F_dF_mu_d2F_device(Z, mu, F2, J2, H2)
torch.cuda.synchronize()
hs = int(dyn.dims.hess_nnz_interval)
d = (H1 != H2).cpu().numpy().reshape(-1, hs)
cols = np.nonzero(d.any(axis=0))[0]
print("hess stride", hs, "differing positions within an interval:", cols[:60], len(cols))
# block offsets: (U_t,a) s*m | (a,U_t+1) s*m | (a,a) m(m+1)/2 | (a,h) m | (U_t,h) s | (h,U_t+1) s | (h,h) 1 | (dx,h)
s, m = 128, 6
offs = {"Ua": 0, "aU": s*m, "aa": 2*s*m, "ah": 2*s*m + 21, "Uh": 2*s*m+21+6, "hU": 2*s*m+27+s, "hh": 2*s*m+27+2*s, "d": 2*s*m+28+2*s}
print(offs)
print("J equal", torch.equal(J1, J2), "F equal", torch.equal(F1, F2))
