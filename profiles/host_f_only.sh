#!/bin/bash
# host-visible F (line-search trial) at config 3 over QC_HOST_F_CHUNKS
for c in 1 2 3 4 6; do
  QC_HOST_F_CHUNKS=$c python - <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
qc = g.load_package()
inp = qc.config_inputs(3, T=1000)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = inp.traj.datavec
F = np.empty(int(dyn.dims.F_len))
ts = []
for i in range(60):
    t0 = time.perf_counter(); dyn.F(Z, out=F); ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts[10:])
print(f"F chunks {os.environ['QC_HOST_F_CHUNKS']}: median {np.median(ts):.0f} us, min {ts.min():.0f}, mean {ts.mean():.0f}")
PY
done 2>&1 | grep "F chunks"
