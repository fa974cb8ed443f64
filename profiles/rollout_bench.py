#!/usr/bin/env python3
"""The unitary rollout of config 3 (T = 1000) fifty times, for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rollout -- python3 profiles/rollout_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

qc = g.load_package()
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
inp = qc.config_inputs(3, T=T)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = torch.from_numpy(inp.traj.datavec).cuda()
init = torch.from_numpy(qc.operator_to_iso_vec(np.eye(8, dtype=complex))).cuda()
out = torch.empty(128 * T, dtype=torch.float64, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    qc._lib.check(qc._lib.lib.qc_rollout_dev(dyn._h, Z.data_ptr(), init.data_ptr(), out.data_ptr(), s), dyn._h)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    qc._lib.check(qc._lib.lib.qc_rollout_dev(dyn._h, Z.data_ptr(), init.data_ptr(), out.data_ptr(), s), dyn._h)
e1.record()
torch.cuda.synchronize()
print(f"unitary rollout, config 3, T = {T}: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us per rollout (4 launches)")
dyn.close()
