import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
os.environ["QC_HOST_TRACE"] = "1"
import __graft_entry__ as g
qc = g.load_package()
inp = qc.config_inputs(3, T=1000)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = inp.traj.datavec
F, J = np.empty(int(dyn.dims.F_len)), np.empty(int(dyn.dims.jac_nnz))
for i in range(12):
    t0 = time.perf_counter(); dyn.F_dF(Z, out=(F, J)); print("call %.0f us" % ((time.perf_counter()-t0)*1e6), file=sys.stderr)
