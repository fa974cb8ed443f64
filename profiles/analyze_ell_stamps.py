#!/usr/bin/env python3
"""Offline view of a raw stamp dump of qc_mfma32_ell_kernel (profiles/stamps_ell32.py with QC_STAMP_DUMP): the workgroups that were
placed FIRST on their compute unit (blockIdx.x < 256: tests/hip/wg_placement.hip) against those that share it as the second one.
usage: python profiles/analyze_ell_stamps.py dump.npy [...]"""
import sys

import numpy as np


def remap(b, nb):
    q, r, x, i = nb >> 3, nb & 7, b & 7, b >> 3
    return x * (q + 1) + i if x < r else r * (q + 1) + (x - r) * q + i


LABELS = ["start", "G written", "barrier A", "state seen", "products issued", "GD/E seen", "blocks stored", "wave done"]
for path in sys.argv[1:]:
    s = np.load(path)
    n = s.shape[0]
    inv = np.zeros(n, dtype=int)
    for blk in range(n):
        inv[remap(blk, n)] = blk
    rel = (s - s[s > 0].min()) * 10.0 / 1e3
    first = inv < 256
    print(f"{path}: span {rel[s > 0].max():.2f} us")
    for k, lab in enumerate(LABELS):
        a, b = rel[first, k], rel[~first, k]
        la, lb = rel[first, 8 + k], rel[~first, 8 + k]
        print(f"  {lab:16s} first on its CU: {np.median(a):5.2f} (max {a.max():5.2f})   second: {np.median(b):5.2f} (max {b.max():5.2f})   |   "
              f"other stamped wave: {np.median(la):5.2f} ({la.max():5.2f})   {np.median(lb):5.2f} ({lb.max():5.2f})")
