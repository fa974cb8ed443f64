#!/usr/bin/env python3
"""End-to-end plumbing check in the spirit of the reference's solve-and-compare tests
(reference src/problem_templates/unitary_smooth_pulse_problem.jl:205-222: build the problem, solve a few iterations,
assert that the rollout fidelity improved): BASELINE config 1 — 1-qubit Hadamard UnitarySmoothPulseProblem, T = 50,
dt = 0.2, X/Y drives — with the dynamics constraint, its Jacobian and Lagrangian Hessian, the infidelity objective,
the regularisers and the rollout fidelity all served by the MI355X library and the NLP driven by a CPU solver.
Ipopt is not available in this image, so scipy's SLSQP (default; 0.61 -> 0.9999 rollout fidelity in 60 iterations) or
`trust-constr` (sparse equality constraints, exact Hessian; it stalls on this problem) stands in for it; the solver
is not part of the build.

    python examples/solve_hadamard.py [max_iter]
"""
from __future__ import annotations

import os
import sys

import numpy as np
from scipy.optimize import Bounds, NonlinearConstraint, minimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g


def rollout_fidelity(qc, dyn, z, U_goal):
    """unitary_rollout_fidelity (reference unitary_smooth_pulse_problem.jl:218): roll U out with exp(dt G(a_t)) on the
    device (`qc_rollout`) and compare the last knot with the goal (`qc_fidelity_eval`)."""
    N = U_goal.shape[0]
    states = dyn.rollout(z, qc.operator_to_iso_vec(np.eye(N, dtype=complex)))
    return qc.iso_vec_unitary_fidelity(states[:, -1], qc.operator_to_iso_vec(U_goal))


def solve(max_iter: int = 60, T: int = 50, verbose: bool = True, method: str = "SLSQP", _debug_hook=None):
    qc = g.load_package()
    inp = qc.config_inputs(1, T=T)   # geodesic + N(0, 1e-2) noise: exactly on the geodesic the loss |1 - F| sits on its kink
    traj, system = inp.traj, inp.system
    U_goal = qc.GATES["H"]
    dyn = qc.QuantumDynamics(inp.integrators, traj)
    obj = qc.UnitaryInfidelityObjective("Ũ⃗", traj, Q=100.0)
    nv = int(dyn.dims.Z_len)
    zdim = traj.dim
    comps = traj.components
    R = 1e-2   # R_a = R_da = R_dda (reference unitary_smooth_pulse_problem.jl:151-153), evaluated by qc_terms_*
    reg = qc.TrajectoryObjective(qc.QuadraticRegularizer("a", traj, R) + qc.QuadraticRegularizer("da", traj, R)
                                 + qc.QuadraticRegularizer("dda", traj, R), traj)
    # the NLP evaluator Ipopt's interface would drive (quantumcollocation.jl_amd/evaluator.py): objective terms, dynamics rows,
    # Lagrangian Hessian over one fixed structure
    ev = qc.QuantumControlEvaluator(dyn, [obj, reg])

    # pinned variables (initial state, initial/final controls) are eliminated: the solver sees only the free ones
    z_full = traj.datavec.copy()
    pinned = np.zeros(nv, dtype=bool)
    pinned[comps["Ũ⃗"].start:comps["Ũ⃗"].stop] = True
    for t_pin in (0, T - 1):
        pinned[t_pin * zdim + comps["a"].start:t_pin * zdim + comps["a"].stop] = True
    free = np.flatnonzero(~pinned)
    # a milder initial guess than the reference's U(-1, 1) controls keeps the interior-point start well inside the bounds
    for t in range(1, T - 1):
        z_full[t * zdim + comps["a"].start:t * zdim + comps["a"].stop] *= 0.2
    for nm in ("da", "dda"):
        for t in range(T):
            z_full[t * zdim + comps[nm].start:t * zdim + comps[nm].stop] *= 0.2

    def full(x):
        z = z_full.copy()
        z[free] = x
        return z

    def fun(x):
        return ev.eval_objective(full(x))

    def grad(x):
        gvec = np.empty(nv)
        ev.eval_objective_gradient(gvec, full(x))
        return gvec[free]

    def hess_obj(x):
        return ev.hessian_lagrangian_matrix(full(x), 1.0, np.zeros(ev.n_constraints))[free][:, free]

    def cons(x):
        c = np.empty(ev.n_constraints)
        ev.eval_constraint(c, full(x))
        return c

    def cons_jac(x):
        return ev.jacobian_matrix(full(x)).tocsc()[:, free]

    def cons_hess(x, v):
        return ev.hessian_lagrangian_matrix(full(x), 0.0, v)[free][:, free]

    # bounds: |a| <= 1, |dda| <= 1, dt in [0.1, 0.3]
    lb, ub = np.full(nv, -np.inf), np.full(nv, np.inf)
    for t in range(T):
        for nm, bnd in (("a", 1.0), ("dda", 1.0)):
            sl = slice(t * zdim + comps[nm].start, t * zdim + comps[nm].stop)
            lb[sl], ub[sl] = -bnd, bnd
        i = t * zdim + comps["Δt"].start
        lb[i], ub[i] = 0.1, 0.3
    x0 = z_full[free]
    if _debug_hook is not None:
        _debug_hook(fun, grad, hess_obj, cons, cons_jac, cons_hess, x0, lb[free], ub[free])
    f_before = rollout_fidelity(qc, dyn, z_full, U_goal)
    if method == "trust-constr":      # sparse Jacobian + exact Hessian of the Lagrangian (F, dF, mu_d2F all exercised)
        res = minimize(fun, x0, jac=grad, hess=hess_obj, method="trust-constr", bounds=Bounds(lb[free], ub[free]),
                       constraints=[NonlinearConstraint(cons, 0.0, 0.0, jac=cons_jac, hess=cons_hess)],
                       options={"maxiter": max_iter, "verbose": 0, "gtol": 1e-8, "xtol": 1e-12})
    else:                             # SLSQP: dense Jacobian, quasi-Newton (F and dF exercised)
        res = minimize(fun, x0, jac=grad, method="SLSQP", bounds=Bounds(lb[free], ub[free]),
                       constraints=[{"type": "eq", "fun": cons, "jac": lambda x: cons_jac(x).toarray()}],
                       options={"maxiter": max_iter, "ftol": 1e-10})
        res.constr_nfev = [res.nfev]
    res.x = full(res.x)
    f_after = rollout_fidelity(qc, dyn, res.x, U_goal)
    viol = float(np.max(np.abs(dyn.F(res.x))))
    if verbose:
        print(f"iterations {res.nit}  constraint evaluations {res.constr_nfev}  rollout fidelity {f_before:.6f} -> {f_after:.6f}  "
              f"max |dynamics residual| {viol:.2e}")
    dyn.close()
    obj.close()
    reg.close()
    return f_before, f_after, viol


if __name__ == "__main__":
    solve(int(sys.argv[1]) if len(sys.argv) > 1 else 60, method=sys.argv[2] if len(sys.argv) > 2 else "SLSQP")
