#!/usr/bin/env python3
"""Where do the two waves of every workgroup of the config-3 F + dF launch run?  (QC_STAMPS=1: the DIAG instantiation records
HW_REG_HW_ID of the copy wave in stamp slot 11 and of the compute wave in slot 15.)  Prints, per compute unit, how many
compute / copy waves each SIMD hosts.   QC_STAMPS=1 python profiles/placement.py"""
import collections
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QC_STAMPS"] = "1"
import __graft_entry__ as g

qc = g.load_package()
inp = qc.config_inputs(3, T=1000)
dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
Z = torch.from_numpy(inp.traj.datavec).cuda()
F = torch.empty(dyn.dims.F_len, dtype=torch.float64, device="cuda")
J = torch.empty(dyn.dims.jac_nnz, dtype=torch.float64, device="cuda")
for _ in range(5):
    dyn.F_dF_device(Z, F, J)
torch.cuda.synchronize()
n = dyn.dims.n_intervals
out = np.zeros(n * 16, dtype=np.uint64)
qc._lib.check(qc._lib.lib.qc_debug_read_stamps(dyn._h, out.ctypes.data_as(C.POINTER(C.c_uint64)), out.size), dyn._h)
st = out.reshape(n, 16)


def fields(hw):   # gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ... (XCC from blockIdx % 8)
    hw = int(hw) & 0xffffffff
    return dict(wave=hw & 15, simd=(hw >> 4) & 3, cu=(hw >> 8) & 15, sh=(hw >> 12) & 1, se=(hw >> 13) & 7)


per_cu = collections.defaultdict(lambda: {"compute": collections.Counter(), "copy": collections.Counter()})
pairs = collections.Counter()
for b in range(n):
    c, k = fields(st[b, 15]), fields(st[b, 11])
    # blocks are dealt round-robin over the XCDs: XCD = (dispatch order) % 8; the interval -> block map is qc_xcd_remap, so use the HW fields only
    key_c = (c["se"], c["sh"], c["cu"])
    key_k = (k["se"], k["sh"], k["cu"])
    per_cu[key_c]["compute"][c["simd"]] += 1
    per_cu[key_k]["copy"][k["simd"]] += 1
    pairs[(c["simd"], k["simd"])] += 1
print("(compute SIMD, copy SIMD) of a workgroup:", dict(pairs))
hist = collections.Counter()
for key, d in per_cu.items():
    hist[tuple(sorted(d["compute"].values(), reverse=True))] += 1
print("compute waves per SIMD, per (SE, SH, CU) id [XCDs folded together]:", dict(hist))
tot_c, tot_k = collections.Counter(), collections.Counter()
for d in per_cu.values():
    tot_c.update(d["compute"]); tot_k.update(d["copy"])
print("all compute waves by SIMD:", dict(tot_c), " all copy waves by SIMD:", dict(tot_k))
