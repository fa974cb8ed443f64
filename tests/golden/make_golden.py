"""Generates tests/golden/fixture_outputs.npz: outputs of the BUILD'S ORACLE (oracle/qc_oracle.py) on
the reference's data fixture (named_trajectory_type_1.json, from reference test/test_utils.jl:54-70)
with the system of test_utils.jl:123 (0.1 Z drift, X/Y drives), order-4 Pade, free time.
These are NOT outputs of the reference (it cannot be run here: SURVEY.md section 8c)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle.qc_oracle as o  # noqa: E402

fx = json.load(open(os.path.join(HERE, "named_trajectory_type_1.json")))
data = np.array(fx["data"])
X = np.array([[0, 1], [1, 0]], dtype=complex)
Y = np.array([[0, -1j], [1j, 0]])
Zp = np.array([[1, 0], [0, -1]], dtype=complex)
prob = o.Problem(N=2, m=2, T=5, zdim=15, off_U=0, off_a=8, off_dt=14, G_drift=o.generator(0.1 * Zp),
                 G_drives=np.array([o.generator(X), o.generator(Y)]), order=4,
                 derivs=[o.DerivSpec(8, 10, 2), o.DerivSpec(10, 12, 2)])
Zv = data.reshape(-1, order="F")
mu = np.ones(prob.n_rows)            # reference script uses mu = ones (integrator_test_1qubit.jl:50)
rows, cols = o.jac_structure(prob)
prob.hess_align = 1            # the committed Hessian vectors are the unpadded layout
hr, hc = o.hess_structure(prob)
np.savez(os.path.join(HERE, "fixture_outputs.npz"), F=o.F(prob, Zv), dF=o.dF(prob, Zv), dF_rows=rows, dF_cols=cols,
         mu_d2F=o.mu_d2F(prob, Zv, mu), mu_d2F_rows=hr, mu_d2F_cols=hc)
print("wrote fixture_outputs.npz")

# Rows 8f of the scope table on the same fixture: exponential integrator, rollout, final-knot fidelity against the
# fixture's goal, and the regularisers on a / da plus a minimum-time term.  Build-oracle outputs as well.
pe = o.Problem(N=2, m=2, T=5, zdim=15, off_U=0, off_a=8, off_dt=14, G_drift=o.generator(0.1 * Zp),
               G_drives=np.array([o.generator(X), o.generator(Y)]), integrator=o.EXPONENTIAL,
               derivs=[o.DerivSpec(8, 10, 2), o.DerivSpec(10, 12, 2)])
init = np.array(fx["initial_U"], dtype=float)      # the fixture's own initial / goal iso-vecs (test_utils.jl:102-107)
goal = np.array(fx["goal_U"], dtype=float)
roll = o.rollout(pe, Zv, init)
fid, gfid, hfid = o.fidelity_value_grad_hess(roll[:, -1], goal)
tm = o.Terms(T=5, zdim=15, off_dt=14, reg_index=np.arange(8, 12), reg_R=np.array([1e-2, 1e-2, 2e-2, 3e-2]), D=1.5, n_mt=4, dt_scaled=True)
thr, thc = o.terms_hess_structure(tm)
np.savez(os.path.join(HERE, "fixture_outputs_8f.npz"), exp_F=o.F(pe, Zv), exp_dF=o.dF(pe, Zv), rollout=roll, init=init, goal=goal,
         fidelity=fid, fidelity_grad=gfid, fidelity_hess=hfid, terms_J=o.terms_value(tm, Zv), terms_grad=o.terms_grad(tm, Zv),
         terms_hess=o.terms_hess(tm, Zv), terms_hess_rows=thr, terms_hess_cols=thc, terms_R=tm.reg_R, terms_index=tm.reg_index)
print("wrote fixture_outputs_8f.npz")
