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
