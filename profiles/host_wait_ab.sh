#!/bin/bash
# host-visible F+dF at config 3: polling (default) against blocking waits (QC_HOST_WAIT=1), alternating, medians of 40 calls
for i in 1 2 3; do for w in 0 1; do
  QC_HOST_WAIT=$w python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
qc = g.load_package()
inp = qc.config_inputs(3, T=1000)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = inp.traj.datavec
F, J = np.empty(int(dyn.dims.F_len)), np.empty(int(dyn.dims.jac_nnz))
ts = []
for i in range(50):
    t0 = time.perf_counter(); dyn.F_dF(Z, out=(F, J)); ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts[10:])
print(f"QC_HOST_WAIT={os.environ['QC_HOST_WAIT']}: median {np.median(ts):.0f} us, min {ts.min():.0f}, mean {ts.mean():.0f}, max {ts.max():.0f}")
PY
done; done 2>&1 | grep QC_HOST_WAIT
