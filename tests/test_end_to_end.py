"""Solve-and-compare, like the reference's integration tests (unitary_smooth_pulse_problem.jl:205-222): config 1 with
the GPU evaluator behind a CPU NLP solver; the rollout fidelity must improve."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_hadamard_solve_improves_rollout_fidelity(qc):
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import solve_hadamard
    before, after, viol = solve_hadamard.solve(max_iter=40, T=30, verbose=False)
    assert after > before, (before, after)
    assert after > 0.9 or after - before > 0.2
    assert viol < 1e-2


@pytest.mark.gpu
def test_interior_point_solve_with_exact_hessians(qc):
    """The same problem through the NLP evaluator in Ipopt's call order with the Lagrangian Hessian (examples/ipm_solve.py):
    F, dF and mu_d2F together in a converging solve; one fused F + dF and one mu_d2F launch per accepted point, residual-only
    launches for the line-search trials."""
    sys.path.insert(0, os.path.join(ROOT, "examples"))
    import ipm_solve
    before, after, viol, stats = ipm_solve.solve(max_iter=60, T=30, verbose=False)
    assert after > before and after > 0.99, (before, after)
    assert viol < 1e-2
    # every accepted point: one Jacobian evaluation (values only when its residuals were the last thing evaluated, fused otherwise)
    # and one mu_d2F
    assert stats["F_dF"] + stats["dF"] == stats["mu_d2F"] and 10 <= stats["mu_d2F"] <= 60
    assert stats["F"] >= stats["mu_d2F"] and stats["uploads_elided"] >= stats["dF"]
