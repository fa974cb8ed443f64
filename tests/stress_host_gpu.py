#!/usr/bin/env python3
"""One-off randomized stress run of the round-3 paths (not collected by pytest): python tests/stress_host_gpu.py [trials] [seed]

Random Hermitian systems with 1 .. 16 levels and 0 .. 8 drives, unitaries and kets, free and fixed timestep, T from 2 to ~1500 knots:
  * host-buffer entry points (one kernel + one watched copy, ring of pinned blocks, deferred re-arm, qc_set_new_x) against the plain
    full copy (QC_HOST_COMPACT=0), bit for bit, over several calls with changing and unchanged x;
  * a multi-device handle with 2 .. 5 shards on device 0 against the single handle, bit for bit;
  * qc_eval_F_jac_hess_dev against the two launches, bit for bit (one fused launch where the handle has one);
  * a window of the result against the numpy oracle (rtol 1e-10);
  * integrator lists (sampling problems, direct sums of unequal members) through qc_eval_*_list: the compact, watched transfer against
    plain copies bit for bit, new_x on the list, and against the oracle."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import __graft_entry__ as g
from oracle_bridge import problem_from_inputs

qc, o = g.load_package(), g.load_oracle()
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
t0 = time.time()
stats = {"handles": 0, "fused": 0, "two_launches": 0, "multi": 0, "max_T": 0, "worst_F": 0.0, "worst_dF": 0.0, "worst_H": 0.0}


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


def herm(n):
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    return (A + A.conj().T) / 2


for trial in range(trials):
    N = int(rng.choice([1, 2, 3, 4, 4, 5, 7, 8, 8, 8, 8, 8, 8, 9, 12, 16, 16]))
    m = int(rng.integers(1, 9))       # (no drives: covered by tests/stress_gpu.py through the raw descriptor; the templates here want at least one)
    T = int(rng.choice([2, 3, 5, 17, 64, 65, 130, 257, 500, 1000, 1025, 1500])) if N <= 8 else int(rng.choice([2, 3, 9, 33, 120, 251, 500]))
    free_time = bool(rng.random() < 0.7)
    kets = int(rng.integers(1, min(N, 6) + 1)) if rng.random() < 0.25 else 0
    system = qc.QuantumSystem(herm(N), [herm(N) for _ in range(max(m, 1))][:m] if m else [])
    tag = f"trial {trial}: N={N} m={m} T={T} ft={free_time} kets={kets}"
    if os.environ.get("QC_STRESS_VERBOSE"):
        print(tag, flush=True)
    try:
        if kets:
            basis = np.eye(N, dtype=complex)
            inp = qc.quantum_state_smooth_pulse_inputs(system, [basis[:, k % N] for k in range(kets)], [basis[:, (k + 1) % N] for k in range(kets)], T,
                                                       free_time=free_time)
        else:
            inp = qc.unitary_smooth_pulse_inputs(system, np.eye(N, dtype=complex), T, free_time=free_time)
    except Exception as exc:   # noqa: BLE001  (a template that does not take this combination, e.g. no drives)
        print(tag, "skipped:", repr(exc)[:80])
        continue
    Zs = [inp.traj.datavec + 1e-2 * k * rng.standard_normal(inp.traj.datavec.size) for k in range(3)]
    os.environ["QC_HOST_COMPACT"] = "0"
    ref = qc.QuantumDynamics(inp.integrators, inp.traj)
    os.environ.pop("QC_HOST_COMPACT")
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    d = dyn.dims
    has_h = bool(d.hess_nnz)
    mu = rng.standard_normal(int(d.n_rows))
    F, J = np.empty(int(d.F_len)), np.empty(int(d.jac_nnz))
    H = np.empty(int(d.hess_nnz))
    # ---- host path against the full copy, Ipopt-like call order with new_x on and off --------------------------------------
    for rep in range(4):
        Z = Zs[rep % 3]
        Fr, Jr = ref.F_dF(Z)
        Hr = ref.mu_d2F(Z, mu) if has_h else None
        dyn.set_new_x(True)
        dyn.F(Z, out=F)
        assert np.array_equal(bits(F), bits(Fr)), (tag, "F")
        if rng.random() < 0.5:
            dyn.set_new_x(False)
            dyn.dF(np.full_like(Z, np.nan), out=J)            # Z is not read
        else:
            F[:] = 0.0
            dyn.F_dF(Z, out=(F, J))
            assert np.array_equal(bits(F), bits(Fr)), (tag, "F of F_dF")
        assert np.array_equal(bits(J), bits(Jr)), (tag, "dF", rep)
        if has_h:
            dyn.mu_d2F(Z, mu, out=H)
            assert np.array_equal(bits(H), bits(Hr)), (tag, "mu_d2F", rep)
        dyn.set_new_x(True)
    stats["handles"] += 1
    stats["max_T"] = max(stats["max_T"], T)
    # ---- a window against the oracle -------------------------------------------------------------------------------------
    prob = problem_from_inputs(inp)
    t1 = min(T - 1, 3)
    Z = Zs[3 % 3]
    Fr, Jr = ref.F_dF(Z)
    Fo, Jo = o.F(prob, Z, 0, t1), o.dF(prob, Z, 0, t1)
    eF = np.abs(Fr[:Fo.size] - Fo).max() / max(1.0, np.abs(Fo).max())
    eJ = np.abs(Jr[:Jo.size] - Jo).max() / max(1.0, np.abs(Jo).max())
    assert eF < 1e-10 and eJ < 1e-10, (tag, "oracle", eF, eJ)
    stats["worst_F"], stats["worst_dF"] = max(stats["worst_F"], eF), max(stats["worst_dF"], eJ)
    if has_h:
        Ho = o.mu_d2F(prob, Z, mu, 0, t1)
        Hr = ref.mu_d2F(Z, mu)
        eH = np.abs(Hr[:Ho.size] - Ho).max() / max(1.0, np.abs(Ho).max())
        assert eH < 1e-10, (tag, "oracle H", eH)
        stats["worst_H"] = max(stats["worst_H"], eH)
    # ---- several shards on device 0 ----------------------------------------------------------------------------------------
    if rng.random() < 0.4 and T > 2:
        shards = int(rng.integers(2, 6))
        many = qc.QuantumDynamics(inp.integrators, inp.traj, devices=[0] * shards)
        Fm, Jm = many.F_dF(Z)
        assert np.array_equal(bits(Fm), bits(Fr)) and np.array_equal(bits(Jm), bits(Jr)), (tag, "multi", shards)
        if has_h:
            assert np.array_equal(bits(many.mu_d2F(Z, mu)), bits(ref.mu_d2F(Z, mu))), (tag, "multi H", shards)
        many.set_new_x(False)
        assert np.array_equal(bits(many.dF(Zs[1])), bits(Jr)), (tag, "multi new_x")      # still the knots of Z
        many.close()
        stats["multi"] += 1
    # ---- one call for dF + mu_d2F on the device ----------------------------------------------------------------------------
    if has_h:
        dZ = torch.from_numpy(Z).cuda()
        dmu = torch.from_numpy(mu).cuda()
        new = lambda n: torch.full((int(n),), float("nan"), dtype=torch.float64, device="cuda")
        F1, J1, H1, F2, J2, H2 = new(d.F_len), new(d.jac_nnz), new(d.hess_nnz), new(d.F_len), new(d.jac_nnz), new(d.hess_nnz)
        dyn.F_dF_device(dZ, F1, J1)
        dyn.mu_d2F_device(dZ, dmu, H1)
        dyn.F_dF_mu_d2F_device(dZ, dmu, F2, J2, H2)
        torch.cuda.synchronize()
        # (rows of state components without an integrator are never written by the device entry points: compare what is)
        same = lambda a, b: bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())
        assert same(F1, F2) and same(J1, J2) and same(H1, H2), (tag, "one call vs two launches", dyn.fused_kernel_name)
        stats["fused" if dyn.fused_kernel_name.endswith("fused") else "two_launches"] += 1
    dyn.close()
    ref.close()
    # ---- an integrator list (sampling problem: shared controls; direct sum: own controls per member) through qc_eval_*_list -------
    if not kets and rng.random() < 0.35 and N <= 8:
        from oracle_bridge import composed_oracle
        K = int(rng.integers(2, 4))
        Tl = min(T, 300)
        if rng.random() < 0.5:
            systems = [qc.QuantumSystem(herm(N), system.H_drives) for _ in range(K)]
            lst = qc.unitary_sampling_inputs(systems, np.eye(N, dtype=complex), Tl, free_time=free_time, seed=int(rng.integers(1 << 30)))
            kind = "sampling"
        else:
            ft = bool(free_time and rng.random() < 0.5)
            parts = []
            for k in range(K):
                Nk = int(rng.choice([1, 2, 4, N]))
                sk = qc.QuantumSystem(herm(Nk), [herm(Nk) for _ in range(int(rng.integers(1, 4)))])
                parts.append(qc.unitary_smooth_pulse_inputs(sk, np.eye(Nk, dtype=complex), Tl, free_time=ft, seed=int(rng.integers(1 << 30)),
                                                            pade_order=int(rng.choice([4, 4, 6]))))
            if ft:
                for pk in parts[1:]:
                    pk.traj.data[pk.traj.components["Δt"].start, :] = parts[0].traj["Δt"][0]
            lst = qc.unitary_direct_sum_inputs(parts)
            kind = "direct sum"
        os.environ["QC_HOST_LANDING"] = "0"
        lref = qc.QuantumDynamics(lst.integrators, lst.traj)
        os.environ.pop("QC_HOST_LANDING")
        ldyn = qc.QuantumDynamics(lst.integrators, lst.traj)
        Zl = lst.traj.datavec
        lmu = rng.standard_normal(int(ldyn.dims.n_rows))
        for rep in range(3):
            Zr = Zl + 1e-2 * rep * rng.standard_normal(Zl.size)
            Fr, Jr = lref.F_dF(Zr, fresh=True)
            Fl, Jl = ldyn.F_dF(Zr, fresh=True)
            assert np.array_equal(bits(Fl), bits(Fr)) and np.array_equal(bits(Jl), bits(Jr)), (tag, kind, "list F_dF", rep)
            ldyn.set_new_x(False)
            assert np.array_equal(bits(ldyn.dF(np.full_like(Zr, np.nan), fresh=True)), bits(Jr)), (tag, kind, "list dF at the device's knots")
            ldyn.set_new_x(True)
            if ldyn.dims.hess_nnz:
                assert np.array_equal(bits(ldyn.mu_d2F(Zr, lmu, fresh=True)), bits(lref.mu_d2F(Zr, lmu, fresh=True))), (tag, kind, "list H")
        oref = composed_oracle(lst)
        eF = np.abs(Fr - oref.F(Zr)).max() / max(1.0, np.abs(Fr).max())
        eJ = np.abs(Jr - oref.dF(Zr)).max() / max(1.0, np.abs(Jr).max())
        assert eF < 1e-10 and eJ < 1e-10, (tag, kind, "list oracle", eF, eJ)
        stats["lists"] = stats.get("lists", 0) + 1
        ldyn.close()
        lref.close()
print(f"{trials} trials ok in {time.time() - t0:.0f} s; {stats}")
