#!/usr/bin/env python3
"""Host-visible (PCIe-inclusive) time of qc_eval_F_jac / qc_eval_hess / qc_eval_F at BASELINE configs 3 and 5, across the
transfer modes of the host-buffer path (QC_HOST_COMPACT = 0 full copy, 1 direct-to-host compact, 2 packed compact), worker
counts and chunk counts, and through multi-device handles with several shards on device 0.

    python profiles/host_visible.py [--quick]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

qc = g.load_package()


def timed(fn, reps):
    for _ in range(3):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def run(cfg, T, env, devices=None, reps=30):
    for k in ("QC_HOST_COMPACT", "QC_HOST_THREADS", "QC_HOST_CHUNKS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    inp = qc.config_inputs(cfg, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, devices=devices)
    Z = inp.traj.datavec
    F, J = np.empty(int(dyn.dims.F_len)), np.empty(int(dyn.dims.jac_nnz))
    H = np.empty(int(dyn.dims.hess_nnz))
    mu = np.ones(int(dyn.dims.n_rows))
    out = {
        "cfg": cfg, "T": T, "env": env, "devices": devices,
        "F_dF_ms": timed(lambda: dyn.F_dF(Z, out=(F, J)), reps),
        "hess_ms": timed(lambda: dyn.mu_d2F(Z, mu, out=H), reps),
        "F_ms": timed(lambda: dyn.F(Z, out=F), reps),
        "compact_MB": None,
    }
    dyn.close()
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    res = []
    for cfg, T in ((3, 1000), (5, 500)):
        res.append(run(cfg, T, {"QC_HOST_COMPACT": "0"}))
        res.append(run(cfg, T, {"QC_HOST_COMPACT": "2"}))
        res.append(run(cfg, T, {}))
        if quick:
            continue
        for th in ("4", "8", "12", "16"):
            for ch in ("8", "16", "32"):
                res.append(run(cfg, T, {"QC_HOST_THREADS": th, "QC_HOST_CHUNKS": ch}))
        for shards in (2, 4):
            res.append(run(cfg, T, {}, devices=[0] * shards))
    res.append(run(3, 8000, {}))
    res.append(run(3, 8000, {}, devices=[0] * 8))
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "host_visible.json"), "w"), indent=1)
