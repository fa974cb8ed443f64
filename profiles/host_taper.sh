#!/bin/bash
# host-visible F+dF at config 3 over chunk counts and taper factors (QC_HOST_CHUNKS, QC_HOST_TAPER); median call time of 30
for c in 16 12 10 8 6; do for t in 1 2 4 8; do
  QC_HOST_CHUNKS=$c QC_HOST_TAPER=$t python - <<'PY'
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
for i in range(40):
    t0 = time.perf_counter(); dyn.F_dF(Z, out=(F, J)); ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts[8:])
print(f"chunks {os.environ['QC_HOST_CHUNKS']:>2s} taper {os.environ['QC_HOST_TAPER']}: median {np.median(ts):.0f} us, min {ts.min():.0f}, mean {ts.mean():.0f}")
PY
done; done 2>&1 | grep chunks
