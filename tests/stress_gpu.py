#!/usr/bin/env python3
"""One-off randomized stress run (not collected by pytest): python tests/stress_gpu.py [trials] [seed]
Random systems up to 32 levels, drives 0..8 (0..12 above 16 levels), Pade orders 2/4/6 and the exponential integrator, ket columns, layouts,
non-Hermitian generators, every kernel that accepts the problem, against the oracle."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as g
from oracle_bridge import random_problem, sparse_drive_problem
from test_gpu_parity import RawHandle, kernels_for

qc, o = g.load_package(), g.load_oracle()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0 = time.time()
worst = {"F": 0.0, "dF": 0.0, "H": 0.0}
count = {}
for trial in range(trials):
    N = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 10, 12, 13, 15, 16, 16, 17, 20, 24, 27, 32]))
    m = int(rng.integers(0, 13 if N > 16 else 9))
    order = int(rng.choice([4, 4, 4, 2, 6]))
    integ = o.EXPONENTIAL if rng.random() < 0.3 else o.PADE
    if os.environ.get("QC_STRESS_SPARSE"):      # every trial a sparse-drive problem at 16 levels (qc_mfma32_ell.hip)
        N, m, order, integ = 16, int(rng.integers(1, 9)), 4, o.PADE
    T = int(rng.integers(2, 6))
    free_time = bool(rng.integers(0, 2))
    ncol = int(rng.integers(1, min(N, 16) + 1)) if rng.random() < 0.3 and not os.environ.get("QC_STRESS_SPARSE") else 0
    prob, Z = random_problem(o, N=N, m=max(m, 1), T=T, order=order, free_time=free_time, integrator=integ, seed=int(rng.integers(1 << 30)),
                             ncol=ncol, layout=str(rng.choice(["standard", "shuffled"])), hermitian=bool(rng.random() < 0.7))
    sparse = N == 16 and m >= 1 and integ == o.PADE and order == 4 and ncol == 0 and (rng.random() < 0.6 or bool(os.environ.get("QC_STRESS_SPARSE")))
    if sparse:      # sparse drive generators at 16 levels: the row-gather kernels (qc_mfma32_ell.hip) for F + dF, mu_d2F and the one-call form
        prob, Z = sparse_drive_problem(o, m=min(m, 8), T=T, R=int(rng.integers(1, 3)), free_time=free_time, layout=str(rng.choice(["standard", "shuffled"])),
                                       seed=int(rng.integers(1 << 30)), dense_drift=bool(rng.random() < 0.7),
                                       kinds=tuple(rng.choice(["real", "imag", "diag"], size=3)))
        m = prob.m
    sparse_exp = integ == o.EXPONENTIAL and N <= 8 and m >= 1 and ncol == 0 and rng.random() < 0.5
    if sparse_exp:  # one entry per drive-generator row with the exponential integrator: the row-gather Horner steps of qc_mfma_exp*.hip
        prob, Z = sparse_drive_problem(o, m=min(m, 8), T=T, R=1, N=N, free_time=free_time, layout=str(rng.choice(["standard", "shuffled"])),
                                       seed=int(rng.integers(1 << 30)), dense_drift=bool(rng.random() < 0.7),
                                       kinds=tuple(rng.choice(["real", "imag", "diag"], size=3)), integrator=o.EXPONENTIAL)
        m = prob.m
    if m == 0:
        prob.m = 0
        prob.G_drives = prob.G_drives[:0]
        prob.derivs = []
    if free_time and rng.random() < 0.3:
        Z[prob.off_dt::prob.zdim] *= rng.choice([0.01, 3.0, 10.0])
    tag = f"trial {trial}: N={N} m={m} order={order} integ={integ} T={T} ft={free_time} ncol={ncol}"
    Fr, Jr = o.F(prob, Z), o.dF(prob, Z)
    do_h = (integ == o.PADE and (N <= 12 or sparse)) or (integ == o.EXPONENTIAL and N <= 8)     # (round 6: the exponential integrator's mu_d2F)
    cross_h = integ == o.PADE and order == 4 and N > 16       # 4 x 4-tile Hessian kernel against the global-workspace kernel
    if do_h or cross_h:
        mu = rng.standard_normal(prob.n_rows)
    if do_h:
        Hr = o.mu_d2F(prob, Z, mu)
    Hk = {}
    for kernel in kernels_for(qc, prob):
        h = RawHandle(qc, prob, kernel=kernel)
        F, J = h.F_jac(Z)
        eF = np.abs(F - Fr).max() / max(1.0, np.abs(Fr).max())
        eJ = np.abs(J - Jr).max() / max(1.0, np.abs(Jr).max())
        assert eF < 1e-9 and eJ < 1e-9, (tag, kernel, eF, eJ)
        Fo = h.F(Z)
        dFo = np.abs(Fo - F).max() / max(1.0, np.abs(Fr).max())
        assert dFo < 1e-12, (tag, kernel, "F-only differs", dFo)
        worst["F_only_vs_fused"] = max(worst.get("F_only_vs_fused", 0.0), dFo)
        worst["F"], worst["dF"] = max(worst["F"], eF), max(worst["dF"], eJ)
        if do_h and (kernel == "lds" or prob.m <= 8):
            H = h.hess(Z, mu)
            eH = np.abs(H - Hr).max() / max(1.0, np.abs(Hr).max()) if Hr.size else 0.0
            assert eH < 1e-9, (tag, kernel, eH)
            key = "H" if integ == o.PADE else "H_exponential"
            worst[key] = max(worst.get(key, 0.0), eH)
        if cross_h and int(h.dims.hess_nnz):
            Hk[kernel] = h.hess(Z, mu)
        if sparse_exp and kernel == "mfma":
            assert qc._lib.lib.qc_kernel_name(h.h, 0) == b"mfma16-exp-gather" and qc._lib.lib.qc_kernel_name(h.h, 1) == b"mfma16-exp-hess-gather", tag
            count["mfma16-exp-gather"] = count.get("mfma16-exp-gather", 0) + 1
        if sparse and kernel == "mfma":
            ell = qc._lib.lib.qc_kernel_name(h.h, 1) == b"mfma32-pade4-hess-ell"       # (not when five drives share an entry of G)
            count["mfma32-ell"] = count.get("mfma32-ell", 0) + int(ell)
        count[kernel] = count.get(kernel, 0) + 1
        h.close()
    if len(Hk) == 2:
        eH = np.abs(Hk["mfma"] - Hk["lds"]).max() / max(1.0, np.abs(Hk["lds"]).max())
        assert eH < 1e-9, (tag, "mfma64 hessian vs lds", eH)
        worst["H_mfma64_vs_lds"] = max(worst.get("H_mfma64_vs_lds", 0.0), eH)
print(f"{trials} trials ok in {time.time() - t0:.0f} s; handles per kernel {count}; worst relative errors {worst}")
