#!/usr/bin/env python3
"""BASELINE config 1 (1-qubit Hadamard UnitarySmoothPulseProblem, T = 50) solved with EXACT second derivatives through the
evaluator in Ipopt's call order.  Ipopt is not in this image; the driver below is a textbook primal-dual interior-point
iteration (log barrier on the bounds, Newton steps on the primal-dual KKT system with the Lagrangian Hessian
sigma d2f + sum mu_i d2c_i assembled from `eval_hessian_lagrangian`, inertia correction by diagonal shifts, fraction to the
boundary, backtracking on an l1 merit function whose trial points ask for `eval_objective` / `eval_constraint` only, monotone
barrier updates: Waechter & Biegler 2006, sections 2-3, without the filter and the restoration phase).  It is NOT part of the
build and says nothing about Ipopt's robustness; it exists so that `F`, `dF` AND `mu_d2F` are exercised together in a
converging solve, in the order and with the buffer ownership the reference's consumer uses (the reference's integration tests
do the same with Ipopt: src/problem_templates/unitary_smooth_pulse_problem.jl:205-222).

    python examples/ipm_solve.py [max_iter]
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g


def solve(max_iter: int = 80, T: int = 50, verbose: bool = True, tol: float = 1e-6, drift_scales=None, integrator: str = "pade"):
    """drift_scales = None: BASELINE config 1.  drift_scales = (0.7, 1.3, ...): a `UnitarySamplingProblem` over copies of the config-1
    system whose drifts are scaled so (reference unitary_sampling_problem.jl:44-167: one pulse that makes the gate on EVERY system) --
    an integrator list with several state integrators, evaluated through qc_eval_*_list, one infidelity objective per system."""
    qc = g.load_package()
    U_goal = qc.GATES["H"]
    if drift_scales is None:
        # integrator = "exponential": `PiccoloOptions(integrator=:exponential)`, which the reference solves with the Hessian left on
        # (unitary_smooth_pulse_problem.jl:224-240): mu_d2F of the exponential integrator in a converging solve
        inp = qc.config_inputs(1, T=T, integrator=integrator)
        state_names = ["Ũ⃗"]
    else:
        base = qc.multi_qubit_system(1)
        systems = [qc.QuantumSystem(base.H_drift * float(sc), base.H_drives) for sc in drift_scales]
        inp = qc.unitary_sampling_inputs(systems, U_goal, T)
        state_names = [f"Ũ⃗_system_{k + 1}" for k in range(len(systems))]
    traj = inp.traj
    dyn = qc.QuantumDynamics(inp.integrators, traj)
    objs = [qc.UnitaryInfidelityObjective(nm, traj, Q=100.0, form="abs2") for nm in state_names]      # smooth at the optimum
    R = 1e-2
    reg = qc.TrajectoryObjective(qc.QuadraticRegularizer("a", traj, R) + qc.QuadraticRegularizer("da", traj, R)
                                 + qc.QuadraticRegularizer("dda", traj, R), traj)
    ev = qc.QuantumControlEvaluator(dyn, objs + [reg])
    nv, m = ev.n_variables, ev.n_constraints
    zdim, comps = traj.dim, traj.components

    # pinned variables (initial state, initial / final controls) are eliminated; bounds as in the problem template
    z_full = traj.datavec.copy()
    pinned = np.zeros(nv, dtype=bool)
    for nm in state_names:
        pinned[comps[nm].start:comps[nm].stop] = True
    for t_pin in (0, T - 1):
        pinned[t_pin * zdim + comps["a"].start:t_pin * zdim + comps["a"].stop] = True
    free = np.flatnonzero(~pinned)
    lb, ub = np.full(nv, -np.inf), np.full(nv, np.inf)
    for t in range(T):
        for nm, bnd in (("a", 1.0), ("dda", 1.0)):
            sl = slice(t * zdim + comps[nm].start, t * zdim + comps[nm].stop)
            lb[sl], ub[sl] = -bnd, bnd
        i = t * zdim + comps["Δt"].start
        lb[i], ub[i] = 0.1, 0.3
    for t in range(1, T - 1):      # a start well inside the bounds
        z_full[t * zdim + comps["a"].start:t * zdim + comps["a"].stop] *= 0.2
    for nm in ("da", "dda"):
        for t in range(T):
            z_full[t * zdim + comps[nm].start:t * zdim + comps[nm].stop] *= 0.2
    lo, hi = lb[free], ub[free]
    has_lo, has_hi = np.isfinite(lo), np.isfinite(hi)
    n = free.size

    def full(x):
        z = z_full.copy()
        z[free] = x
        return z

    cbuf, gbuf = np.empty(m), np.empty(nv)

    def f_c(x):                       # what a line-search trial asks for
        z = full(x)
        ev.eval_constraint(cbuf, z)
        return ev.eval_objective(z), cbuf.copy()

    def barrier(x, mu):
        return -mu * (np.log(x[has_lo] - lo[has_lo]).sum() + np.log(hi[has_hi] - x[has_hi]).sum())

    def rollout_fidelity(z):          # of the pulse under every system's own generators: the smallest
        init, goal = qc.operator_to_iso_vec(np.eye(U_goal.shape[0], dtype=complex)), qc.operator_to_iso_vec(U_goal)
        if drift_scales is None:
            return qc.iso_vec_unitary_fidelity(dyn.rollout(z, init)[:, -1], goal)
        return min(qc.iso_vec_unitary_fidelity(dyn.rollout(z, init, part=k)[:, -1], goal) for k in range(len(state_names)))

    with np.errstate(invalid="ignore"):                 # (inf - inf on the unbounded variables, masked by the where)
        width = np.where(has_lo & has_hi, hi - lo, 1.0)
    x = np.clip(z_full[free], np.where(has_lo, lo + 1e-2 * width, -np.inf), np.where(has_hi, hi - 1e-2 * width, np.inf))
    lam = np.zeros(m)
    mu = 0.1
    zl = np.where(has_lo, mu / np.maximum(x - lo, 1e-12), 0.0)
    zu = np.where(has_hi, mu / np.maximum(hi - x, 1e-12), 0.0)
    nu = 10.0                                              # penalty of the l1 merit function
    f_before = rollout_fidelity(full(x))
    fval, c = f_c(x)
    it = 0
    for it in range(1, max_iter + 1):
        z = full(x)
        ev.eval_objective_gradient(gbuf, z)                 # accepted point: gradient, Jacobian, Hessian at the same x
        gx = gbuf[free]
        J = ev.jacobian_matrix(z).tocsc()[:, free]
        W = ev.hessian_lagrangian_matrix(z, 1.0, lam)[free][:, free]
        dl, du = np.where(has_lo, x - lo, 1.0), np.where(has_hi, hi - x, 1.0)
        r_dual = gx + J.T @ lam - zl + zu
        e0 = max(np.abs(r_dual).max(), np.abs(c).max(), np.abs(dl * zl)[has_lo].max(initial=0.0), np.abs(du * zu)[has_hi].max(initial=0.0))
        emu = max(np.abs(r_dual).max(), np.abs(c).max(), np.abs(dl * zl - mu)[has_lo].max(initial=0.0),
                  np.abs(du * zu - mu)[has_hi].max(initial=0.0))
        if verbose and (it <= 3 or it % 5 == 0):
            print(f"  it {it:3d}  f {fval:10.4e}  |c| {np.abs(c).max():8.2e}  dual {np.abs(r_dual).max():8.2e}  mu {mu:7.1e}")
        if e0 < tol:
            break
        if emu < 10.0 * mu and mu > tol / 10:
            mu = max(tol / 10, min(0.2 * mu, mu ** 1.5))
            continue
        Sigma = np.where(has_lo, zl / dl, 0.0) + np.where(has_hi, zu / du, 0.0)
        rhs_x = -(gx + J.T @ lam - np.where(has_lo, mu / dl, 0.0) + np.where(has_hi, mu / du, 0.0))
        dw = 0.0
        while True:                                         # inertia correction: shift until the step is a descent direction
            K = sp.bmat([[W + sp.diags(Sigma + dw), J.T], [J, -1e-9 * sp.identity(m)]], format="csc")
            try:
                sol = spla.splu(K).solve(np.concatenate([rhs_x, -c]))
            except RuntimeError:
                sol = None
            if sol is not None and np.isfinite(sol).all():
                dx, dlam = sol[:n], sol[n:]
                if dx @ (W @ dx + (Sigma + dw) * dx) > 1e-10 * (dx @ dx):
                    break
            dw = 1e-4 if dw == 0.0 else 10.0 * dw
            if dw > 1e8:
                raise RuntimeError("inertia correction failed")
        dzl = np.where(has_lo, mu / dl - zl - zl / dl * dx, 0.0)
        dzu = np.where(has_hi, mu / du - zu + zu / du * dx, 0.0)
        tau = max(0.99, 1.0 - mu)

        def max_step(v, dv):
            neg = dv < 0
            return min(1.0, (-tau * v[neg] / dv[neg]).min(initial=1.0))
        a_pr = min(max_step(dl[has_lo], dx[has_lo]), max_step(du[has_hi], -dx[has_hi]))
        a_du = min(max_step(zl[has_lo], dzl[has_lo]), max_step(zu[has_hi], dzu[has_hi]))
        # l1 merit function on the barrier problem; the penalty dominates the multipliers
        nu = max(nu, 1.1 * np.abs(lam + dlam).max())
        phi0 = fval + barrier(x, mu) + nu * np.abs(c).sum()
        dphi = (gx - np.where(has_lo, mu / dl, 0.0) + np.where(has_hi, mu / du, 0.0)) @ dx - nu * np.abs(c).sum()
        alpha = a_pr
        for _ in range(25):
            xt = x + alpha * dx
            ft, ct = f_c(xt)                                # trial point: objective and constraints only
            if ft + barrier(xt, mu) + nu * np.abs(ct).sum() <= phi0 + 1e-4 * alpha * min(dphi, 0.0):
                break
            alpha *= 0.5
        x, fval, c = xt, ft, ct
        lam = lam + alpha * dlam
        zl, zu = zl + a_du * dzl, zu + a_du * dzu
        dl, du = np.where(has_lo, x - lo, 1.0), np.where(has_hi, hi - x, 1.0)
        zl = np.where(has_lo, np.clip(zl, mu / (1e10 * dl), 1e10 * mu / dl), 0.0)
        zu = np.where(has_hi, np.clip(zu, mu / (1e10 * du), 1e10 * mu / du), 0.0)
    z = full(x)
    f_after = rollout_fidelity(z)
    viol = float(np.abs(dyn.F(z)).max())
    if verbose:
        print(f"iterations {it}  launches {ev.stats}  rollout fidelity {f_before:.6f} -> {f_after:.6f}  max |dynamics residual| {viol:.2e}")
    stats = dict(ev.stats)
    for o in [dyn, reg] + objs:
        o.close()
    return f_before, f_after, viol, stats


if __name__ == "__main__":
    # python examples/ipm_solve.py [max_iter] [drift scale ...]     e.g.  ipm_solve.py 80 0.7 1.3  = a two-system sampling problem
    solve(int(sys.argv[1]) if len(sys.argv) > 1 else 80, drift_scales=[float(x) for x in sys.argv[2:]] or None)
