"""Test helper: turn the host layer's inputs (integrators + trajectory) into the oracle's Problem.
Lives under tests/ because only tests, smoke() and bench's cpu_baseline may touch oracle/."""
from __future__ import annotations

import numpy as np


def problem_from_inputs(inp, T=None):
    import __graft_entry__ as g

    o = g.load_oracle()
    qc = g.load_package()
    P = inp.integrators[0]
    traj = inp.traj
    sys_ = P.system
    derivs = [o.DerivSpec(traj.offset(D.x), traj.offset(D.dx), D.dim) for D in inp.integrators if isinstance(D, qc.DerivativeIntegrator)]
    free = isinstance(traj.timestep, str)
    is_pade = isinstance(P, (qc.UnitaryPadeIntegrator, qc.QuantumStatePadeIntegrator))
    kets = [I for I in inp.integrators if isinstance(I, (qc.QuantumStatePadeIntegrator, qc.QuantumStateExponentialIntegrator,
                                                         qc.DensityOperatorExponentialIntegrator))]
    return o.Problem(
        N=sys_.state_levels, m=sys_.n_drives, T=traj.T if T is None else T, zdim=traj.dim,
        off_U=traj.offset(P.state_name), off_a=traj.offset(P.control_name),
        off_dt=traj.offset(traj.timestep) if free else -1,
        G_drift=np.array(sys_.G_drift), G_drives=np.array(sys_.G_drives).reshape(sys_.n_drives, 2 * sys_.state_levels, 2 * sys_.state_levels),
        dt_fixed=0.0 if free else float(traj.timestep),
        integrator=o.PADE if is_pade else o.EXPONENTIAL,
        order=P.order if is_pade else 4,
        derivs=derivs, global_dim=traj.global_dim, ncol=len(kets),
    )


def random_problem(o, N, m, T, order=4, free_time=True, integrator=None, seed=0, layout="standard", hermitian=True, ncol=0):
    """A random oracle Problem + trajectory vector, independent of the host layer."""
    rng = np.random.default_rng(seed)
    n, s = 2 * N, 2 * N * (ncol if ncol > 0 else N)

    def rand_H():
        A = rng.standard_normal((N, N)) + 1j * rng.standard_normal((N, N))
        return (A + A.conj().T) / 2 if hermitian else A

    G0 = o.generator(rand_H())
    Gd = np.array([o.generator(rand_H()) for _ in range(m)]).reshape(m, n, n)
    if layout == "standard":      # [U, a, da, dda, dt]
        off_U, off_a, off_da, off_dda = 0, s, s + m, s + 2 * m
        zdim = s + 3 * m + (1 if free_time else 0)
        off_dt = s + 3 * m if free_time else -1
    elif layout == "script":      # (U, a, g, da, dt) as in integrator_test_1qubit.jl:24-30, one deriv integrator
        off_U, off_a, off_da = 0, s, s + 2 * m
        off_dda = None
        zdim = s + 3 * m + (1 if free_time else 0)
        off_dt = s + 3 * m if free_time else -1
    elif layout == "shuffled":    # [dt, a, da, U, dda]
        off_dt = 0 if free_time else -1
        b = 1 if free_time else 0
        off_a, off_da, off_U, off_dda = b, b + m, b + 2 * m, b + 2 * m + s
        zdim = b + 3 * m + s
    else:
        raise ValueError(layout)
    derivs = [o.DerivSpec(off_a, off_da, m)]
    if off_dda is not None:
        derivs.append(o.DerivSpec(off_da, off_dda, m))
    prob = o.Problem(N=N, m=m, T=T, zdim=zdim, off_U=off_U, off_a=off_a, off_dt=off_dt, G_drift=G0, G_drives=Gd,
                     dt_fixed=0.17, integrator=o.PADE if integrator is None else integrator, order=order, derivs=derivs, ncol=ncol)
    Z = rng.standard_normal(zdim * T) * 0.5
    if free_time:
        Z[off_dt::zdim] = rng.uniform(0.1, 0.3, size=T)
    return prob, Z


def sparse_drive_problem(o, m, T, R=1, free_time=True, layout="standard", seed=0, N=16, dense_drift=True, kinds=("real", "imag", "diag"),
                         integrator=None):
    """Like random_problem, with SPARSE Hermitian drive Hamiltonians whose real-iso generators have exactly R entries per row:
    R = 1: a perfect matching of the N levels with real or imaginary couplings (a Pauli-string-like signed permutation) or a real
    diagonal; R = 2: complex couplings on a matching, or two matchings (a ladder pair a + a^dagger has this shape).  The drift
    is dense unless dense_drift=False."""
    prob, Z = random_problem(o, N=N, m=m, T=T, order=4, free_time=free_time, layout=layout, seed=seed, integrator=integrator)
    rng = np.random.default_rng(1000 + seed)

    def matching_H(kind):
        perm = rng.permutation(N)
        H = np.zeros((N, N), dtype=complex)
        for a, b in zip(perm[0::2], perm[1::2]):
            c = rng.uniform(0.5, 1.5) * rng.choice([-1.0, 1.0])
            if kind == "imag":
                c = 1j * c
            elif kind == "complex":
                c = c * np.exp(1j * rng.uniform(0.2, 1.2))
            H[a, b], H[b, a] = c, np.conj(c)
        return H

    Hs = []
    for k in range(m):
        if R == 1:
            kind = kinds[k % len(kinds)]
            Hs.append(np.diag(rng.uniform(-1.5, 1.5, size=N)).astype(complex) if kind == "diag" else matching_H(kind))
        else:
            Hs.append(matching_H("complex") if k % 2 == 0 else matching_H("real") + matching_H("real" if k % 4 == 1 else "imag"))
    n = 2 * N
    prob.G_drives = np.array([o.generator(H) for H in Hs]).reshape(m, n, n)
    if not dense_drift:
        prob.G_drift = o.generator(np.diag(rng.uniform(-1.0, 1.0, size=N)).astype(complex))
    per_row = max(int(np.count_nonzero(G, axis=1).max()) for G in prob.G_drives)
    assert per_row <= R, per_row
    return prob, Z


def composed_oracle(inp, hess_align=1):
    """Oracle evaluation of an integrator list with several unitary integrators (sampling problem): one oracle
    Problem per group, outputs interleaved per interval in integrator order."""
    import __graft_entry__ as g
    from types import SimpleNamespace
    o = g.load_oracle()
    qc = g.load_package()
    groups = qc.split_groups(inp.integrators)
    probs = [problem_from_inputs(SimpleNamespace(integrators=grp, traj=inp.traj)) for grp in groups]
    for p in probs:
        p.hess_align = 1          # the groups share one per-interval block: it is padded as a whole (below), not per group
    T = inp.traj.T
    hess_total = sum(len(o.hess_structure_local(p)) for p in probs)
    hess_padn = (-hess_total) % max(1, hess_align) if all(len(o.hess_structure_local(p)) for p in probs) else 0

    def interleave(parts):
        return np.concatenate([p.reshape(T - 1, -1) for p in parts], axis=1).reshape(-1)

    rows = sum(p.ddim for p in probs)

    def F(Z):
        return interleave([o.F(p, Z) for p in probs])

    def dF(Z):
        return interleave([o.dF(p, Z) for p in probs])

    def structure():
        rs, cs, ro = [], [], 0
        for p in probs:
            loc = np.array(o.jac_structure_local(p), dtype=np.int64).reshape(-1, 2)
            ts = np.arange(T - 1, dtype=np.int64)
            rs.append(ts[:, None] * rows + ro + loc[None, :, 0])
            cs.append(ts[:, None] * p.zdim + loc[None, :, 1])
            ro += p.ddim
        return np.concatenate(rs, axis=1).reshape(-1), np.concatenate(cs, axis=1).reshape(-1)

    def mu_d2F(Z, mu):
        mus = mu.reshape(T - 1, rows)
        out, ro = [], 0
        for p in probs:
            out.append(o.mu_d2F(p, Z, np.ascontiguousarray(mus[:, ro:ro + p.ddim]).reshape(-1)))
            ro += p.ddim
        if hess_padn:
            out.append(np.zeros((T - 1) * hess_padn))
        return interleave(out)

    def hess_structure():
        rs, cs = [], []
        for p in probs:
            r, c = o.hess_structure(p)
            rs.append(r.reshape(T - 1, -1))
            cs.append(c.reshape(T - 1, -1))
        if hess_padn:   # the last handle repeats ITS first entry for the padding
            rs.append(np.repeat(rs[-1][:, :1], hess_padn, axis=1))
            cs.append(np.repeat(cs[-1][:, :1], hess_padn, axis=1))
        return np.concatenate(rs, axis=1).reshape(-1), np.concatenate(cs, axis=1).reshape(-1)

    return SimpleNamespace(F=F, dF=dF, structure=structure, mu_d2F=mu_d2F, hess_structure=hess_structure, rows=rows, probs=probs)


def aa_columns(dyn):
    """Columns of the per-interval Hessian value block that hold (a_i, a_j) entries (for every member of a composed handle).
    Round 6: mu_d2F ALONE takes these from the Gram matrix M D^T where the handle's drives have one entry per row
    (qc_mfma_hess_g2.hip), the one-call / batched launches from the stage-A tiles: the same numbers to rounding, not the same sums."""
    descs = [p[0] for p in dyn._parts] if hasattr(dyn, "_parts") else [dyn._desc]
    stride = int(dyn.dims.hess_nnz_interval)
    mask = np.zeros(stride, dtype=bool)
    for d in descs:
        if d.integrator != 0:
            continue
        s = 2 * d.N * (d.state_cols or d.N)
        ft = d.off_dt >= 0
        o = int(d.hess_offset) + 2 * s * d.m + (2 * s if ft else 0)
        mask[o:o + d.m * (d.m + 1) // 2] = True
    return mask


def assert_same_hessian_values(Ha, Hb, dyn, what=""):
    """Bit for bit outside the (a, a) columns; there to rounding (1e-12 of the block's scale: sums of at most 256 terms)."""
    Ha, Hb = np.asarray(Ha), np.asarray(Hb)
    assert Ha.shape == Hb.shape, what
    mask = aa_columns(dyn)
    A, B = Ha.reshape(-1, mask.size), Hb.reshape(-1, mask.size)
    rest = np.array_equal(A[:, ~mask], B[:, ~mask])
    assert rest, f"{what}: {(A[:, ~mask] != B[:, ~mask]).sum()} values outside the (a, a) block differ"
    scale = max(1.0, float(np.abs(B[:, mask]).max())) if mask.any() and B.size else 1.0
    np.testing.assert_allclose(A[:, mask], B[:, mask], rtol=1e-11, atol=1e-12 * scale, err_msg=f"{what}: (a, a) block")
