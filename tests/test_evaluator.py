"""The NLP evaluator on the caller's side of the hot path (quantumcollocation.jl_amd/evaluator.py): bookkeeping on the CPU
with stand-in terms, values and Ipopt's call pattern on the GPU."""
import numpy as np
import pytest
import scipy.sparse as sp


class _FakeDims:
    def __init__(self, Z_len, n_rows, jac_nnz, hess_nnz):
        self.Z_len, self.n_rows, self.jac_nnz, self.hess_nnz = Z_len, n_rows, jac_nnz, hess_nnz


class _FakeDynamics:
    """rows: c_r(x) = x_r * x_{r+1} - 1 for r < n-1 (a bilinear chain): Jacobian 2 entries per row, Hessian 1 per row."""

    def __init__(self, n):
        self.n = n
        self.dims = _FakeDims(n, n - 1, 2 * (n - 1), n - 1)
        r = np.arange(n - 1)
        self.dF_structure = (np.repeat(r, 2), np.stack([r, r + 1], axis=1).ravel())
        self.mu_d2F_structure = (r, r + 1)
        self.calls = []

    def F(self, Z, out=None):
        self.calls.append("F")
        v = Z[:-1] * Z[1:] - 1.0
        if out is not None:
            out[:] = v
            return out
        return v

    def F_dF(self, Z, out=None):
        self.calls.append("F_dF")
        F, J = out
        F[:] = Z[:-1] * Z[1:] - 1.0
        J[:] = np.stack([Z[1:], Z[:-1]], axis=1).ravel()
        return F, J

    def mu_d2F(self, Z, mu, out=None):
        self.calls.append("mu_d2F")
        out[:] = mu
        return out


class _QuadObjective:
    """0.5 * w * sum x^2 over all variables (full-length gradient, diagonal Hessian)."""

    def __init__(self, n, w):
        self.n, self.w = n, w
        self.hess_structure = (np.arange(n), np.arange(n))

    def L(self, Z):
        return 0.5 * self.w * float(Z @ Z)

    def grad_L(self, Z):
        return self.w * Z

    def hess_L(self, Z):
        return np.full(self.n, self.w)


class _LastPairObjective:
    """(x_{n-2} * x_{n-1})^2 on `state_indices` (a final-knot term: gradient on its own variables only)."""

    def __init__(self, n):
        self.state_indices = np.array([n - 2, n - 1])
        self.hess_structure = (np.array([n - 2, n - 2, n - 1]), np.array([n - 2, n - 1, n - 1]))   # column-major upper triangle

    def L(self, Z):
        a, b = Z[self.state_indices]
        return (a * b) ** 2

    def grad_L(self, Z):
        a, b = Z[self.state_indices]
        return np.array([2 * a * b * b, 2 * a * a * b])

    def hess_L(self, Z):
        a, b = Z[self.state_indices]
        return np.array([2 * b * b, 4 * a * b, 2 * a * a])


class _SumConstraint:
    """g = sum of two variables - 1: one dense row over `state_indices`, no second derivative."""
    dim = 1

    def __init__(self, i, j):
        self.state_indices = np.array([i, j])

    def g(self, Z):
        return np.array([Z[self.state_indices].sum() - 1.0])

    def dg(self, Z):
        return np.ones(2)


def _load_evaluator():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("qc_evaluator_only", os.path.join(root, "quantumcollocation.jl_amd", "evaluator.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.QuantumControlEvaluator


def test_evaluator_bookkeeping_with_stand_in_terms():   # (CPU: the evaluator module imports nothing from the library)
    Evaluator = _load_evaluator()
    n = 7
    dyn = _FakeDynamics(n)
    ev = Evaluator(dyn, [_QuadObjective(n, 3.0), _LastPairObjective(n)], [_SumConstraint(0, 3)])
    assert (ev.n_variables, ev.n_constraints, ev.jac_nnz, ev.hess_nnz) == (n, n, 2 * (n - 1) + 2, n + 3 + (n - 1))
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n)
    mu = rng.standard_normal(ev.n_constraints)
    sigma = 0.7
    c = np.empty(ev.n_constraints)
    ev.eval_constraint(c, x)
    np.testing.assert_allclose(c[:n - 1], x[:-1] * x[1:] - 1.0)
    np.testing.assert_allclose(c[n - 1], x[0] + x[3] - 1.0)
    g = np.empty(n)
    ev.eval_objective_gradient(g, x)

    def lagrangian(z):
        cc = np.empty(ev.n_constraints)
        ev._x_F = None
        ev.eval_constraint(cc, z)
        return sigma * ev.eval_objective(z) + mu @ cc

    def lag_grad(z):
        gg = np.empty(n)
        ev.eval_objective_gradient(gg, z)
        return sigma * gg + ev.jacobian_matrix(z).T @ mu

    eps = 1e-6
    fd = np.array([(lagrangian(x + eps * e) - lagrangian(x - eps * e)) / (2 * eps) for e in np.eye(n)])
    np.testing.assert_allclose(lag_grad(x), fd, rtol=1e-6, atol=1e-8)
    H = ev.hessian_lagrangian_matrix(x, sigma, mu).toarray()
    fdH = np.array([(lag_grad(x + eps * e) - lag_grad(x - eps * e)) / (2 * eps) for e in np.eye(n)])
    np.testing.assert_allclose(H, fdH, rtol=1e-6, atol=1e-7)
    r1, c1 = ev.jacobian_structure(one_based=True)
    assert r1.min() == 1 and c1.max() == n
    hr, hc = ev.hessian_lagrangian_structure()
    assert (hr <= hc).all()


def test_evaluator_call_pattern_with_stand_in_terms():
    """Ipopt's order: trial points ask for f and c; the accepted point then for grad f, jac c, Hessian at the same x."""
    Evaluator = _load_evaluator()
    n = 6
    dyn = _FakeDynamics(n)
    ev = Evaluator(dyn, [_QuadObjective(n, 1.0)], [])
    rng = np.random.default_rng(1)
    c = np.empty(ev.n_constraints)
    J = np.empty(ev.jac_nnz)
    H = np.empty(ev.hess_nnz)
    g = np.empty(n)
    xs = [rng.standard_normal(n) for _ in range(3)]
    for x in xs:                       # three line-search trials
        ev.eval_objective(x)
        ev.eval_constraint(c, x)
    x = xs[-1]                         # the last one is accepted
    ev.eval_objective_gradient(g, x)
    ev.eval_constraint(c, x)           # (a repeated request at the same x: served from the cache)
    ev.eval_constraint_jacobian(J, x)
    ev.eval_hessian_lagrangian(H, x, 1.0, np.ones(ev.n_constraints))
    ev.eval_constraint(c, x)           # after the fused call the residuals are still those of x
    assert dyn.calls == ["F", "F", "F", "F_dF", "mu_d2F"]
    assert ev.stats == {"F": 3, "F_dF": 1, "dF": 0, "mu_d2F": 1, "reused_F": 2, "uploads_elided": 0}   # (a dynamics without set_new_x: nothing elided)
    np.testing.assert_allclose(c, x[:-1] * x[1:] - 1.0)
    with pytest.raises(ValueError):
        ev.eval_constraint(c, np.zeros(n + 1))
    ev2 = Evaluator(dyn, [_QuadObjective(n, 1.0)], [], eval_hessian=False)
    with pytest.raises(RuntimeError):
        ev2.eval_hessian_lagrangian(H, x, 1.0, np.ones(ev.n_constraints))


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["pade", "exponential"])
def test_evaluator_on_the_library_matches_finite_differences_of_the_lagrangian(qc, integrator):
    """Config 1 (T = 8): infidelity objective + regularisers, dynamics rows + a final-fidelity constraint row, all served by the
    library; the assembled Lagrangian Hessian against central differences of the assembled Lagrangian gradient.  (The exponential
    integrator too, round 6: its Hessian has no entry at knot t+1, which the assembly must not mind.)"""
    inp = qc.config_inputs(1, T=8, integrator=integrator)
    traj = inp.traj
    dyn = qc.QuantumDynamics(inp.integrators, traj)
    obj = qc.UnitaryInfidelityObjective("Ũ⃗", traj, Q=100.0, form="abs2")     # smooth at F = 1 (|1 - F| has a kink)
    reg = qc.TrajectoryObjective(qc.QuadraticRegularizer("a", traj, 1e-2) + qc.QuadraticRegularizer("da", traj, 1e-2)
                                 + qc.QuadraticRegularizer("dda", traj, 1e-2), traj)
    con = qc.FinalUnitaryFidelityConstraint("Ũ⃗", 0.99, traj, form="abs2")
    ev = qc.QuantumControlEvaluator(dyn, [obj, reg], [con])
    assert ev.n_constraints == int(dyn.dims.n_rows) + 1
    rng = np.random.default_rng(3)
    x = traj.datavec + 1e-2 * rng.standard_normal(traj.datavec.size)
    mu = rng.standard_normal(ev.n_constraints)
    sigma = 0.3

    def lag_grad(z):
        g = np.empty(ev.n_variables)
        ev.eval_objective_gradient(g, z)
        return sigma * g + ev.jacobian_matrix(z).T @ mu

    H = ev.hessian_lagrangian_matrix(x, sigma, mu)
    eps = 1e-6
    for _ in range(6):
        v = rng.standard_normal(ev.n_variables)
        fd = (lag_grad(x + eps * v) - lag_grad(x - eps * v)) / (2 * eps)
        np.testing.assert_allclose(H @ v, fd, rtol=2e-6, atol=2e-6 * np.abs(fd).max())
    # the pieces equal the library's own entry points
    c = np.empty(ev.n_constraints)
    ev.eval_constraint(c, x)
    np.testing.assert_array_equal(c[:-1], dyn.F(x))
    assert c[-1] == con.g(x)[0]
    J = np.empty(ev.jac_nnz)
    ev.eval_constraint_jacobian(J, x)
    np.testing.assert_array_equal(J[:int(dyn.dims.jac_nnz)], dyn.dF(x))
    for o in (dyn, obj, reg, con):
        o.close()


@pytest.mark.gpu
def test_evaluator_drives_an_ipopt_ordered_iteration_on_the_library(qc):
    """One interior-point-style iteration at config 2 in Ipopt's call order, with the launches counted."""
    inp = qc.config_inputs(2, T=40)
    traj = inp.traj
    dyn = qc.QuantumDynamics(inp.integrators, traj)
    obj = qc.UnitaryInfidelityObjective("Ũ⃗", traj, Q=100.0)
    ev = qc.QuantumControlEvaluator(dyn, [obj])
    rng = np.random.default_rng(5)
    x0 = traj.datavec.copy()
    step = 1e-3 * rng.standard_normal(x0.size)
    c, g = np.empty(ev.n_constraints), np.empty(ev.n_variables)
    J, H = np.empty(ev.jac_nnz), np.empty(ev.hess_nnz)
    lam = rng.standard_normal(ev.n_constraints)
    for alpha in (1.0, 0.5, 0.25):     # backtracking line search: f and c only
        ev.eval_objective(x0 + alpha * step)
        ev.eval_constraint(c, x0 + alpha * step)
    xa = x0 + 0.25 * step
    ev.eval_objective_gradient(g, xa)
    ev.eval_constraint_jacobian(J, xa)
    ev.eval_hessian_lagrangian(H, xa, 1.0, lam)
    ev.eval_constraint(c, xa)
    # the accepted point's residuals were the last thing evaluated: Jacobian and Hessian run on the knots already on the device
    assert ev.stats == {"F": 3, "F_dF": 0, "dF": 1, "mu_d2F": 1, "reused_F": 1, "uploads_elided": 2}
    np.testing.assert_array_equal(c, dyn.F(xa))
    np.testing.assert_array_equal(J, dyn.dF(xa))
    off, cnt = ev._hess_dyn
    np.testing.assert_array_equal(H[off:off + cnt], dyn.mu_d2F(xa, lam))
    assert np.isfinite(H).all() and np.isfinite(g).all()
    dyn.close()
    obj.close()


@pytest.mark.gpu
def test_evaluator_with_linear_and_free_phase_constraints(qc):
    """Config 1 with a `TimeStepsAllEqualConstraint` (T - 1 constant rows given by a local structure) behind the dynamics rows and a
    fidelity row behind those: rows, Jacobian entries and the Lagrangian Hessian land where the structures say."""
    inp = qc.config_inputs(1, T=6)
    traj = inp.traj
    dyn = qc.QuantumDynamics(inp.integrators, traj)
    obj = qc.UnitaryInfidelityObjective("Ũ⃗", traj, Q=10.0, form="abs2")
    teq = qc.TimeStepsAllEqualConstraint("Δt", traj)
    fid = qc.FinalUnitaryFidelityConstraint("Ũ⃗", 0.9, traj, form="abs2")
    ev = qc.QuantumControlEvaluator(dyn, [obj], [teq, fid])
    n_dyn = int(dyn.dims.n_rows)
    assert ev.n_constraints == n_dyn + (traj.T - 1) + 1
    rng = np.random.default_rng(8)
    x = traj.datavec + 1e-2 * rng.standard_normal(traj.datavec.size)
    c = np.empty(ev.n_constraints)
    ev.eval_constraint(c, x)
    np.testing.assert_array_equal(c[n_dyn:n_dyn + traj.T - 1], teq.g(x))
    assert c[-1] == fid.g(x)[0]
    J = ev.jacobian_matrix(x).toarray()
    rows, cols = teq.jac_structure
    Jt = np.zeros((traj.T - 1, x.size))
    Jt[rows, cols] = teq.dg(x)
    np.testing.assert_array_equal(J[n_dyn:n_dyn + traj.T - 1], Jt)
    np.testing.assert_array_equal(J[-1, fid.state_indices], fid.dg(x))
    assert not np.delete(J[-1], fid.state_indices).any()
    # central differences of every constraint row against the assembled Jacobian, a few random directions
    eps = 1e-6
    for _ in range(4):
        v = rng.standard_normal(x.size)
        cp, cm = np.empty_like(c), np.empty_like(c)
        ev.eval_constraint(cp, x + eps * v)
        ev.eval_constraint(cm, x - eps * v)
        np.testing.assert_allclose(J @ v, (cp - cm) / (2 * eps), rtol=1e-6, atol=1e-7)
    mu = rng.standard_normal(ev.n_constraints)
    H = ev.hessian_lagrangian_matrix(x, 0.5, mu)
    lag_grad = lambda z: 0.5 * _grad(ev, z) + ev.jacobian_matrix(z).T @ mu     # noqa: E731
    for _ in range(4):
        v = rng.standard_normal(x.size)
        fd = (lag_grad(x + eps * v) - lag_grad(x - eps * v)) / (2 * eps)
        np.testing.assert_allclose(H @ v, fd, rtol=2e-6, atol=2e-6 * max(1.0, np.abs(fd).max()))
    for o in (dyn, obj, fid):
        o.close()


def _grad(ev, z):
    g = np.empty(ev.n_variables)
    ev.eval_objective_gradient(g, z)
    return g


@pytest.mark.gpu
def test_evaluator_over_a_sampling_problem_fills_the_callers_buffers(qc):
    """Several unitary integrators (ComposedQuantumDynamics): the evaluator hands its own arrays in as `out=` and reads the results
    from them -- the composed methods must fill those arrays, not return fresh ones."""
    systems = [qc.multi_qubit_system(2, zz=z) for z in (0.08, 0.1, 0.12)]
    inp = qc.unitary_sampling_inputs(systems, qc.GATES["CX"], 12)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj)
    assert isinstance(dyn, qc.ComposedQuantumDynamics)
    ev = qc.QuantumControlEvaluator(dyn, [])
    rng = np.random.default_rng(11)
    x = inp.traj.datavec + 1e-2 * rng.standard_normal(inp.traj.datavec.size)
    mu = rng.standard_normal(ev.n_constraints)
    c = np.full(ev.n_constraints, np.nan)
    J = np.full(ev.jac_nnz, np.nan)
    H = np.full(ev.hess_nnz, np.nan)
    ev.eval_constraint(c, x)
    ev.eval_constraint_jacobian(J, x)
    ev.eval_hessian_lagrangian(H, x, 1.0, mu)
    F_ref, J_ref = dyn.F_dF(x)
    np.testing.assert_array_equal(c, F_ref)
    np.testing.assert_array_equal(J, J_ref)
    np.testing.assert_array_equal(H, dyn.mu_d2F(x, mu))
    np.testing.assert_array_equal(J, dyn.dF(x))
    with pytest.raises(ValueError):
        dyn.F(x, out=np.empty(3))
    dyn.close()
