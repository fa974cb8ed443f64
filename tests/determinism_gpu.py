#!/usr/bin/env python3
"""One-off check (not collected by pytest): python tests/determinism_gpu.py [trials] [seed]
Every entry point called several times on the same input must return the same BITS: random systems over every kernel that accepts them
(F + dF, F alone, mu_d2F, host and device forms), final-knot fidelities (value, gradient, Hessian), objective terms, rollouts at chunk
boundaries, integrator lists.  A difference means two writers of one word or an unordered reduction somewhere (the rollout had one
until round 4: tests/test_rollout.py::test_rollout_is_bit_reproducible)."""
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
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
REP = 4
t0 = time.time()
count = {"handles": 0, "fidelities": 0, "rollouts": 0, "lists": 0, "terms": 0}


def same(xs, what):
    for x in xs[1:]:
        assert np.array_equal(np.ascontiguousarray(x).view(np.uint64), np.ascontiguousarray(xs[0]).view(np.uint64)), what


for trial in range(trials):
    N = int(rng.choice([1, 2, 3, 4, 5, 8, 8, 9, 12, 16, 16, 17, 24, 32]))
    m = int(rng.integers(1, 13 if N > 16 else 9))
    order = int(rng.choice([4, 4, 4, 2, 6, 12]))
    integ = o.EXPONENTIAL if rng.random() < 0.25 else o.PADE
    T = int(rng.choice([2, 3, 5, 9, 17, 33, 70]))
    if N > 16:
        T = min(T, 9)
    free_time = bool(rng.integers(0, 2))
    ncol = int(rng.integers(1, min(N, 16) + 1)) if rng.random() < 0.25 else 0
    if N == 16 and integ == o.PADE and order == 4 and ncol == 0 and rng.random() < 0.6:
        prob, Z = sparse_drive_problem(o, m=min(m, 8), T=T, R=int(rng.integers(1, 3)), free_time=free_time, layout="standard", seed=int(rng.integers(1 << 30)),
                                       dense_drift=True, kinds=tuple(rng.choice(["real", "imag", "diag"], size=3)))
    else:
        prob, Z = random_problem(o, N=N, m=m, T=T, order=order, free_time=free_time, integrator=integ, seed=int(rng.integers(1 << 30)), ncol=ncol,
                                 layout=str(rng.choice(["standard", "shuffled"])), hermitian=bool(rng.random() < 0.7))
    tag = f"trial {trial}: N={N} m={prob.m} order={order} integ={integ} T={T} ft={free_time} ncol={ncol}"
    mu = rng.standard_normal(prob.n_rows)
    for kernel in kernels_for(qc, prob):
        h = RawHandle(qc, prob, kernel=kernel)
        outs = [h.F_jac(Z) for _ in range(REP)]
        same([x[0] for x in outs], (tag, kernel, "F"))
        same([x[1] for x in outs], (tag, kernel, "dF"))
        # (the exponential integrator's residual-only launch scales exp(dt G) by the norm of dt G alone, the F + dF launch by the norm of
        #  the augmented Frechet matrix: the same residuals to the last bits, not bit for bit; the Pade kernels' are identical)
        same([h.F(Z) for _ in range(REP)] + ([outs[0][0]] if integ == o.PADE else []), (tag, kernel, "F alone"))
        if h.dims.hess_nnz and ((integ == o.PADE and (N <= 16 or order == 4)) or (integ == o.EXPONENTIAL and N <= 8)):
            same([h.hess(Z, mu) for _ in range(REP)], (tag, kernel, "mu_d2F"))
        h.close()
        count["handles"] += 1
    # rollouts: the scan's chunk boundaries
    if ncol == 0 and N <= 16:
        h = RawHandle(qc, prob)
        init = rng.standard_normal(2 * N * N)
        outs = []
        for _ in range(REP):
            out = np.empty((T, 2 * N * N))
            qc._lib.check(qc._lib.lib.qc_rollout(h.h, qc._lib.dptr(Z), qc._lib.dptr(init), qc._lib.dptr(out)), h.h)
            outs.append(out)
        same(outs, (tag, "rollout"))
        h.close()
        count["rollouts"] += 1
    # final-knot fidelity: value, gradient, Hessian
    if N <= 16 and trial % 2 == 0:
        goal = rng.standard_normal(2 * N * N)
        sub = None if rng.random() < 0.5 or N < 3 else sorted(rng.choice(N, size=max(2, N // 2), replace=False).tolist())
        fid = qc.objectives._Fidelity(goal, subspace=sub, form=str(rng.choice(["abs", "abs2"])))
        u = rng.standard_normal(2 * N * N)
        ev = [fid.eval(u) for _ in range(REP)]
        assert len({e[0] for e in ev}) == 1 and len({e[1] for e in ev}) == 1, (tag, "fidelity value")
        same([e[2] for e in ev], (tag, "fidelity gradient"))
        same([e[3] for e in ev], (tag, "fidelity hessian"))
        fid.close()
        count["fidelities"] += 1

# integrator lists and objective terms through the mirror
for trial in range(max(4, trials // 6)):
    nq = int(rng.choice([1, 2, 3]))
    K = int(rng.integers(2, 4))
    T = int(rng.choice([3, 9, 40, 130]))
    base = qc.multi_qubit_system(nq)
    systems = [qc.QuantumSystem(base.H_drift * (1.0 + 0.2 * rng.standard_normal()), base.H_drives) for _ in range(K)]
    inp = qc.unitary_sampling_inputs(systems, np.eye(2 ** nq, dtype=complex), T, seed=int(rng.integers(1 << 30)))
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    Z = inp.traj.datavec
    mu = rng.standard_normal(int(dyn.dims.n_rows))
    outs = [dyn.F_dF(Z, fresh=True) for _ in range(REP)]
    same([x[0] for x in outs], ("list F", nq, K, T))
    same([x[1] for x in outs], ("list dF", nq, K, T))
    same([dyn.mu_d2F(Z, mu, fresh=True) for _ in range(REP)], ("list H", nq, K, T))
    dyn.close()
    count["lists"] += 1
    names = [(n, float(rng.uniform(0.1, 2.0))) for n in ("a", "da", "dda")]
    spec = None
    for n, R in names:
        term = qc.QuadraticRegularizer(n, inp.traj, R)
        spec = term if spec is None else spec + term
    spec = spec + qc.MinimumTimeObjective(inp.traj, D=1.5)
    obj = qc.TrajectoryObjective(spec, inp.traj)
    ev = [obj.L_grad_hess(Z) for _ in range(REP)]
    assert len({e[0] for e in ev}) == 1, "terms value"
    same([e[1] for e in ev], "terms gradient")
    same([e[2] for e in ev], "terms hessian")
    obj.close()
    count["terms"] += 1
print(f"{trials} trials: every repeated call bit-identical, {time.time() - t0:.0f} s; {count}")
