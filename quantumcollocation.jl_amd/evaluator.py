"""The caller of the hot path: an NLP evaluator in the shape Ipopt's interface drives.

Reference side (SURVEY 8b "who calls it"): `QuantumControlProblem(traj, J, integrators; constraints, ...)`
(src/problem_templates/unitary_smooth_pulse_problem.jl:181-190) hands its `QuantumDynamics`, its objective and its nonlinear
constraints to QuantumCollocationCore's `MOI.AbstractNLPEvaluator`; Ipopt.jl then calls `eval_objective`,
`eval_objective_gradient`, `eval_constraint`, `eval_constraint_jacobian`, `eval_hessian_lagrangian` and, once,
`jacobian_structure` / `hessian_lagrangian_structure`.  That evaluator is not in /root/reference (un-vendored package); this
class restates its observable contract on this side of the boundary so that the library's entry points can be exercised in
the order and with the buffer ownership Ipopt uses:

* one serial caller, one evaluation in flight, caller-owned output arrays of fixed length;
* constraint rows: the dynamics rows first, then the nonlinear constraints in the order given ("appended after the
  dynamics rows"); linear constraints (bounds, pinned knots, `TimeStepsAllEqualConstraint`) stay with the solver;
* the Hessian of the Lagrangian as ONE value vector over a fixed structure: objective terms (scaled by sigma), then the
  dynamics' `mu_d2F`, then the constraints' `mu_d2g` -- duplicates allowed and summed by the consumer, as MOI specifies;
* Ipopt's call pattern: every trial point of the line search asks for `eval_objective` and `eval_constraint` only; the
  accepted point then asks for the gradient, the Jacobian and the Hessian AT THE SAME x.  `eval_constraint` therefore runs the
  residual-only launch, `eval_constraint_jacobian` the fused `qc_eval_F_jac` (refreshing the cached residuals for free), and
  nothing is recomputed when a callback repeats the x of the previous one.  When the Jacobian or the Hessian is asked for at
  the x whose residuals were the last thing evaluated, the trajectory vector is not sent again (`dynamics.set_new_x(False)`,
  the library's counterpart of the `new_x` flag of Ipopt's C callbacks): the knots are on the device already.

Everything numeric happens in the library (dynamics, fidelity and trajectory-term kernels); this file only routes buffers.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


class QuantumControlEvaluator:
    """`MOI.AbstractNLPEvaluator`-shaped view of a quantum-control NLP.

    dynamics     `QuantumDynamics` (rows 0 .. n_dyn-1)
    objectives   terms with `L(Z)`, `grad_L(Z)`, `hess_L(Z)`, `hess_structure`; a term with `state_indices` returns its gradient
                 on those variables only (final-knot terms), otherwise over all variables (`TrajectoryObjective`)
    constraints  nonlinear constraints with `dim`, `g(Z)`, `dg(Z)` and either `state_indices` (one dense row over those
                 variables: the fidelity constraints) or `jac_structure` (rows local to the constraint); optional
                 `mu_d2g(Z, mu)` + `hess_structure`
    """

    def __init__(self, dynamics, objectives: Sequence, constraints: Sequence = (), eval_hessian: bool = True):
        self.dynamics = dynamics
        self.objectives = list(objectives)
        self.constraints = list(constraints)
        self.eval_hessian = bool(eval_hessian)
        d = dynamics.dims
        self.n_variables = int(d.Z_len)
        self.n_dynamics_rows = int(d.n_rows)
        self.n_constraints = self.n_dynamics_rows + sum(int(c.dim) for c in self.constraints)
        self._F = np.zeros(self.n_dynamics_rows)
        if hasattr(dynamics, "result_ring"):            # the library's dynamics: the residual cache in pinned memory, written in place by the kernel
            from .dynamics import pinned_zeros
            self._F = pinned_zeros(self.n_dynamics_rows)
        self._x_F: Optional[np.ndarray] = None          # the x the cached residuals belong to == the x whose knots are on the device
        self._gen_F = -1                                # ... as long as the handle's upload count is still this one (knot_generation)
        self.stats = {"F": 0, "F_dF": 0, "dF": 0, "mu_d2F": 0, "reused_F": 0, "uploads_elided": 0}
        self._can_elide = hasattr(dynamics, "set_new_x") and hasattr(dynamics, "knot_generation")
        # ---- Jacobian structure -------------------------------------------------------------------
        jr, jc = dynamics.dF_structure
        rows, cols = [np.asarray(jr, dtype=np.int64)], [np.asarray(jc, dtype=np.int64)]
        self._jac_dyn = int(d.jac_nnz)
        self._con_jac: List[Tuple[int, int]] = []      # (offset, count) of each constraint's values in the Jacobian vector
        row0, off = self.n_dynamics_rows, self._jac_dyn
        for c in self.constraints:
            if hasattr(c, "jac_structure"):
                r, cc = c.jac_structure
                r, cc = np.asarray(r, dtype=np.int64) + row0, np.asarray(cc, dtype=np.int64)
            else:
                cc = np.asarray(c.state_indices, dtype=np.int64)
                r = np.full(cc.size, row0, dtype=np.int64)
            rows.append(r)
            cols.append(cc)
            self._con_jac.append((off, r.size))
            off += r.size
            row0 += int(c.dim)
        self._jac_rows, self._jac_cols = np.concatenate(rows), np.concatenate(cols)
        self.jac_nnz = int(self._jac_rows.size)
        # ---- Hessian-of-the-Lagrangian structure ---------------------------------------------------------
        hr, hc = [], []
        self._obj_hess: List[Tuple[int, int]] = []
        off = 0
        for o in self.objectives:
            r, c = o.hess_structure
            hr.append(np.asarray(r, dtype=np.int64))
            hc.append(np.asarray(c, dtype=np.int64))
            self._obj_hess.append((off, hr[-1].size))
            off += hr[-1].size
        r, c = dynamics.mu_d2F_structure
        self._hess_dyn = (off, int(d.hess_nnz))
        hr.append(np.asarray(r, dtype=np.int64))
        hc.append(np.asarray(c, dtype=np.int64))
        off += int(d.hess_nnz)
        self._con_hess: List[Optional[Tuple[int, int]]] = []
        for cobj in self.constraints:
            if hasattr(cobj, "mu_d2g") and hasattr(cobj, "hess_structure"):
                r, c = cobj.hess_structure
                hr.append(np.asarray(r, dtype=np.int64))
                hc.append(np.asarray(c, dtype=np.int64))
                self._con_hess.append((off, hr[-1].size))
                off += hr[-1].size
            else:
                self._con_hess.append(None)
        self._hess_rows, self._hess_cols = np.concatenate(hr), np.concatenate(hc)
        self.hess_nnz = int(self._hess_rows.size)

    # -- structures (asked for once) -----------------------------------------------------------------------
    def jacobian_structure(self, one_based: bool = False):
        """(rows, cols) of the constraint Jacobian's value vector (0-based; `one_based=True`: as MOI / the reference count)."""
        k = 1 if one_based else 0
        return self._jac_rows + k, self._jac_cols + k

    def hessian_lagrangian_structure(self, one_based: bool = False):
        """(rows, cols) of the Lagrangian Hessian's value vector: upper triangle, duplicates summed by the consumer."""
        k = 1 if one_based else 0
        return self._hess_rows + k, self._hess_cols + k

    # -- callbacks ------------------------------------------------------------------------------------------
    def _x(self, x) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        if x.size != self.n_variables:
            raise ValueError(f"x has length {x.size}, expected {self.n_variables}")
        return x

    def _same_x(self, x: np.ndarray) -> bool:
        return self._x_F is not None and np.array_equal(self._x_F, x)

    def eval_objective(self, x) -> float:
        x = self._x(x)
        return float(sum(o.L(x) for o in self.objectives))

    def eval_objective_gradient(self, g: np.ndarray, x) -> None:
        x = self._x(x)
        g[:] = 0.0
        for o in self.objectives:
            v = o.grad_L(x)
            if hasattr(o, "state_indices"):
                g[o.state_indices] += v
            else:
                g += v

    def eval_constraint(self, c: np.ndarray, x) -> None:
        x = self._x(x)
        n = self.n_dynamics_rows
        if self._same_x(x):
            self.stats["reused_F"] += 1
        else:
            if self._can_elide:
                self.dynamics.set_new_x(True)
            self.dynamics.F(x, out=self._F)            # residual-only launch: what a line-search trial costs
            self._remember(x)
            self.stats["F"] += 1
        c[:n] = self._F
        off = n
        for cobj in self.constraints:
            c[off:off + cobj.dim] = cobj.g(x)
            off += cobj.dim

    def _remember(self, x: np.ndarray) -> None:
        self._x_F = x.copy()
        if self._can_elide:
            self._gen_F = self.dynamics.knot_generation()

    def _at_device_x(self, x: np.ndarray) -> bool:
        """True (and the library told so) when x is the vector the dynamics evaluated last: its knots are on the device.  The
        handle may be shared (the same `QuantumDynamics` called directly, bound host calls, another evaluator): the upload is
        elided only while the handle's own upload count is the one seen after this evaluator's last upload."""
        same = self._can_elide and self._same_x(x) and self.dynamics.knot_generation() == self._gen_F
        if self._can_elide:
            self.dynamics.set_new_x(not same)
        if same:
            self.stats["uploads_elided"] += 1
        return same

    def eval_constraint_jacobian(self, J: np.ndarray, x) -> None:
        x = self._x(x)
        if self._at_device_x(x):
            # Ipopt's accepted point: the residuals are cached and the knots are on the device -- Jacobian values only, no upload
            self.dynamics.dF(x, out=J[:self._jac_dyn])
            self.stats["dF"] += 1
        else:
            # fused: the residuals come with the Jacobian at no extra cost and refresh the cache
            self.dynamics.F_dF(x, out=(self._F, J[:self._jac_dyn]))
            self._remember(x)
            self.stats["F_dF"] += 1
        if self._can_elide:
            self.dynamics.set_new_x(True)
        for cobj, (off, cnt) in zip(self.constraints, self._con_jac):
            J[off:off + cnt] = cobj.dg(x)

    def eval_hessian_lagrangian(self, H: np.ndarray, x, sigma: float, mu) -> None:
        if not self.eval_hessian:
            raise RuntimeError("the evaluator was built with eval_hessian=False (quasi-Newton solve)")
        x = self._x(x)
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        if mu.size != self.n_constraints:
            raise ValueError(f"mu has length {mu.size}, expected {self.n_constraints}")
        for o, (off, cnt) in zip(self.objectives, self._obj_hess):
            H[off:off + cnt] = sigma * np.asarray(o.hess_L(x))
        off, cnt = self._hess_dyn
        if cnt:
            at_x = self._at_device_x(x)
            self.dynamics.mu_d2F(x, mu[:self.n_dynamics_rows], out=H[off:off + cnt])
            if self._can_elide:
                self.dynamics.set_new_x(True)
                if not at_x:            # the Hessian call just put x's knots on the device, but the cached residuals are another x's
                    self._x_F = None
            self.stats["mu_d2F"] += 1
        r0 = self.n_dynamics_rows
        for cobj, slot in zip(self.constraints, self._con_hess):
            if slot is not None:
                H[slot[0]:slot[0] + slot[1]] = cobj.mu_d2g(x, mu[r0:r0 + cobj.dim])
            r0 += cobj.dim

    # -- conveniences for solvers that want matrices -------------------------------------------------------------------
    def jacobian_matrix(self, x):
        import scipy.sparse as sp
        J = np.empty(self.jac_nnz)
        self.eval_constraint_jacobian(J, x)
        return sp.coo_matrix((J, (self._jac_rows, self._jac_cols)), shape=(self.n_constraints, self.n_variables)).tocsr()

    def hessian_lagrangian_matrix(self, x, sigma: float, mu):
        """Full symmetric matrix (the value vector holds the upper triangle)."""
        import scipy.sparse as sp
        H = np.empty(self.hess_nnz)
        self.eval_hessian_lagrangian(H, x, sigma, mu)
        U = sp.coo_matrix((H, (self._hess_rows, self._hess_cols)), shape=(self.n_variables, self.n_variables)).tocsr()
        return U + sp.triu(U, 1).T
