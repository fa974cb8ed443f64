"""CPU ORACLE (test infrastructure, NOT product code) for the knot-point dynamics evaluator.

PARITY UNPINNED.  The arithmetic of this path lives in the un-vendored Julia dependency
QuantumCollocationCore 0.3 (reference `Project.toml:31`), which is absent from `/root/reference`
and cannot be run here (no Julia).  The reference's own tests pin no number on this path
(`SURVEY.md` §8c).  This file therefore restates the *published mathematics* of the path:

* NLP statement / integrator roles ........ reference `src/problem_templates/unitary_smooth_pulse_problem.jl:10-30`
* Schroedinger step, exponential form ..... reference `README.md:74-80`
* iso-vec layout vec([Re U; Im U]) ........ reference `src/trajectory_initialization.jl:137`
* knot component order [U, a, da, dda, dt]  reference `src/trajectory_initialization.jl:357-382`
* integrator (row) order .................. reference `src/problem_templates/unitary_smooth_pulse_problem.jl:175-179`
* shapes of F / dF / mu_d2F ............... reference `test/scripts/integrator_test_1qubit.jl:41-52`
* COO accumulate convention ............... reference `test/test_utils.jl:14-27`

and is pinned against mathematics instead (tests/test_oracle_math.py): complex-step and
finite-difference derivatives, scipy `expm`, mpmath 50-digit spot checks, Pade order-of-accuracy.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this module.

Everything here is written for clarity, with dense per-interval blocks and explicit Kronecker
products (the way the reference builds them per knot), not for speed.  All indices are 0-based;
`structure(..., one_based=True)` converts for the Julia convention.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import numpy as np

PADE = 0
EXPONENTIAL = 1


# --------------------------------------------------------------------------------------------
#  Isomorphism (reference trajectory_initialization.jl:137)
# --------------------------------------------------------------------------------------------
def operator_to_iso_vec(U: np.ndarray) -> np.ndarray:
    """vec(vcat(real(U), imag(U))), column-major."""
    U = np.asarray(U, dtype=complex)
    return np.vstack([U.real, U.imag]).reshape(-1, order="F").copy()


def iso_vec_to_operator(v: np.ndarray) -> np.ndarray:
    v = np.asarray(v, dtype=float)
    N = int(round(math.sqrt(v.size / 2)))
    M = v.reshape(2 * N, N, order="F")
    return M[:N] + 1j * M[N:]


def generator(H: np.ndarray) -> np.ndarray:
    """G = iso(-iH) = [[Im H, Re H], [-Re H, Im H]]: d/dt [Re U; Im U] = G [Re U; Im U]."""
    H = np.asarray(H, dtype=complex)
    return np.block([[H.imag, H.real], [-H.real, H.imag]])


# --------------------------------------------------------------------------------------------
#  Problem description (mirrors the fields of the C descriptor `qc_desc`)
# --------------------------------------------------------------------------------------------
@dataclass
class DerivSpec:
    """DerivativeIntegrator(x, dx, traj): x_{t+1} - x_t - dt * dx_t = 0
    (reference unitary_smooth_pulse_problem.jl:15-16,177-178)."""
    x_off: int
    dx_off: int
    dim: int


@dataclass
class Problem:
    N: int
    m: int
    T: int
    zdim: int
    off_U: int
    off_a: int
    off_dt: int                  # -1 => fixed timestep
    G_drift: np.ndarray          # (n, n)
    G_drives: np.ndarray         # (m, n, n)
    dt_fixed: float = 0.0
    integrator: int = PADE
    order: int = 4
    derivs: List[DerivSpec] = field(default_factory=list)
    global_dim: int = 0
    ncol: int = 0                # columns of the iso state: 0 -> N (unitary); K -> K kets stored back to back
                                 # (QuantumStatePadeIntegrator per ket, reference quantum_state_smooth_pulse_problem.jl:146-152)
    hess_align: int = 1          # 1 (default, as qc_desc.hess_align = 0): exactly the structural entries; k > 1: the per-interval
                                 # Hessian value block is padded with explicit zeros (structure: duplicates of its first entry) to
                                 # a multiple of k entries (qc_desc.hess_align = k, the layout of device-resident consumers)
    # Row placement (qc_desc.row_placement = QC_ROWS_BY_COMPONENT): rows_per_interval = Z.dims.states, the state
    # integrator's rows at row_offset, derivative integrator i's rows at deriv_rows[i]; rows of state components without an
    # integrator stay structurally empty (reference test/scripts/integrator_test_script.jl:23-44).  None = stacked rows.
    rows_per_interval: int = 0
    row_offset: int = 0
    deriv_rows: List[int] | None = None

    @property
    def n(self) -> int:
        return 2 * self.N

    @property
    def nc(self) -> int:
        return self.ncol if self.ncol > 0 else self.N

    @property
    def s(self) -> int:
        return 2 * self.N * self.nc

    @property
    def free_time(self) -> bool:
        return self.off_dt >= 0

    @property
    def ddim(self) -> int:
        return self.s + sum(d.dim for d in self.derivs)

    @property
    def n_vars(self) -> int:
        return self.zdim * self.T + self.global_dim

    @property
    def row_stride(self) -> int:
        """Rows of the F / mu vectors per interval (ddim unless rows_per_interval says otherwise)."""
        return self.rows_per_interval if self.rows_per_interval > 0 else self.ddim

    @property
    def row_map(self) -> np.ndarray:
        """Row inside the per-interval block of each of the ddim stacked rows [state integrator | derivative integrators]."""
        rm = np.empty(self.ddim, dtype=np.int64)
        rm[:self.s] = self.row_offset + np.arange(self.s)
        r = self.s
        for i, d in enumerate(self.derivs):
            base = self.deriv_rows[i] if self.deriv_rows is not None else self.row_offset + r
            rm[r:r + d.dim] = base + np.arange(d.dim)
            r += d.dim
        return rm

    @property
    def n_rows(self) -> int:
        return self.row_stride * (self.T - 1)


def pade_coeffs(order: int) -> List[float]:
    """Diagonal (p,p) Pade coefficients c_0..c_p of exp, order 2p."""
    if order % 2 or order < 2:
        raise ValueError("Pade order must be even and >= 2")
    p = order // 2
    f = math.factorial
    return [f(2 * p - k) * f(p) / (f(2 * p) * f(k) * f(p - k)) for k in range(p + 1)]


def _G_of(prob: Problem, a: np.ndarray) -> np.ndarray:
    return prob.G_drift + np.tensordot(a, prob.G_drives, axes=(0, 0))


def _split(prob: Problem, z0: np.ndarray, z1: np.ndarray):
    n, N = prob.n, prob.nc
    U0 = z0[prob.off_U:prob.off_U + prob.s].reshape(n, N, order="F")
    U1 = z1[prob.off_U:prob.off_U + prob.s].reshape(n, N, order="F")
    a = z0[prob.off_a:prob.off_a + prob.m]
    h = z0[prob.off_dt] if prob.free_time else prob.dt_fixed
    return U0, U1, a, h


def _vec(X: np.ndarray) -> np.ndarray:
    return X.reshape(-1, order="F")


# --------------------------------------------------------------------------------------------
#  Matrix exponential + Frechet derivative (for the exponential integrator), self-contained:
#  scaled Taylor + squaring on the block-triangular augmented matrix.
# --------------------------------------------------------------------------------------------
def expm_taylor(X: np.ndarray) -> np.ndarray:
    nrm = np.linalg.norm(X, 1)
    sq = 0 if nrm <= 0.25 else max(0, int(math.ceil(math.log2(nrm / 0.25))))
    Y = X / (2.0 ** sq)
    E = np.eye(X.shape[0], dtype=X.dtype)
    term = np.eye(X.shape[0], dtype=X.dtype)
    for k in range(1, 25):
        term = term @ Y / k
        E = E + term
    for _ in range(sq):
        E = E @ E
    return E


def expm_frechet_block(X: np.ndarray, E: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(exp(X), L_exp(X; E)) via exp([[X, E], [0, X]]) top-right block."""
    k = X.shape[0]
    A = np.zeros((2 * k, 2 * k), dtype=np.result_type(X, E))
    A[:k, :k] = X
    A[k:, k:] = X
    A[:k, k:] = E
    EA = expm_taylor(A)
    return EA[:k, :k], EA[:k, k:]


def expm_frechet2_block(X: np.ndarray, A: np.ndarray, B: np.ndarray) -> np.ndarray:
    """Second Frechet derivative L2_exp(X; A, B) = d2/ds dt exp(X + sA + tB) at 0 (symmetric, bilinear).
    The top-right block of exp([[X, A, 0], [0, X, B], [0, 0, X]]) is the ordered double integral with A to
    the left of B; the derivative is that block plus the one with the directions swapped."""
    k = X.shape[0]
    dt = np.result_type(X, A, B)

    def ordered(P, Q):
        T = np.zeros((3 * k, 3 * k), dtype=dt)
        T[:k, :k] = T[k:2 * k, k:2 * k] = T[2 * k:, 2 * k:] = X
        T[:k, k:2 * k] = P
        T[k:2 * k, 2 * k:] = Q
        return expm_taylor(T)[:k, 2 * k:]

    return ordered(A, B) + ordered(B, A)


# --------------------------------------------------------------------------------------------
#  Per-interval residual
# --------------------------------------------------------------------------------------------
def interval_residual(prob: Problem, z0: np.ndarray, z1: np.ndarray) -> np.ndarray:
    """delta_t(z_t, z_{t+1}) in R^ddim: unitary rows first, then each derivative integrator."""
    U0, U1, a, h = _split(prob, z0, z1)
    G = _G_of(prob, a)
    out = np.zeros(prob.ddim, dtype=np.result_type(z0, z1, float))
    if prob.integrator == PADE:
        c = pade_coeffs(prob.order)
        B = sum(((-1) ** k) * c[k] * h ** k * np.linalg.matrix_power(G, k) for k in range(len(c)))
        F = sum(c[k] * h ** k * np.linalg.matrix_power(G, k) for k in range(len(c)))
        # explicit Kronecker form, as the reference forms it per knot: (I_N (x) B) U1vec - (I_N (x) F) U0vec
        IN = np.eye(prob.nc)
        out[:prob.s] = np.kron(IN, B) @ _vec(U1) - np.kron(IN, F) @ _vec(U0)
    else:
        E = expm_taylor(h * G)
        out[:prob.s] = _vec(U1) - np.kron(np.eye(prob.nc), E) @ _vec(U0)
    r = prob.s
    for d in prob.derivs:
        x0 = z0[d.x_off:d.x_off + d.dim]
        x1 = z1[d.x_off:d.x_off + d.dim]
        dx0 = z0[d.dx_off:d.dx_off + d.dim]
        out[r:r + d.dim] = x1 - x0 - h * dx0
        r += d.dim
    return out


# --------------------------------------------------------------------------------------------
#  Per-interval dense Jacobian (ddim x 2*zdim), columns [z_t | z_{t+1}]
# --------------------------------------------------------------------------------------------
def _dGpow(G: np.ndarray, Gj: np.ndarray, k: int) -> np.ndarray:
    """d(G^k)[Gj] = sum_{i=0}^{k-1} G^i Gj G^{k-1-i}."""
    mp = np.linalg.matrix_power
    return sum(mp(G, i) @ Gj @ mp(G, k - 1 - i) for i in range(k)) if k > 0 else np.zeros_like(G)


def _d2Gpow(G: np.ndarray, Gi: np.ndarray, Gj: np.ndarray, k: int) -> np.ndarray:
    """d^2(G^k)[Gi, Gj] = sum over ordered insertions of Gi and Gj (both orders)."""
    mp = np.linalg.matrix_power
    out = np.zeros_like(G)
    for al in range(k - 1):
        for be in range(k - 1 - al):
            ga = k - 2 - al - be
            out = out + mp(G, al) @ Gi @ mp(G, be) @ Gj @ mp(G, ga)
            out = out + mp(G, al) @ Gj @ mp(G, be) @ Gi @ mp(G, ga)
    return out


def interval_jacobian_dense(prob: Problem, z0: np.ndarray, z1: np.ndarray) -> np.ndarray:
    U0, U1, a, h = _split(prob, z0, z1)
    G = _G_of(prob, a)
    n, N, s, m, zd = prob.n, prob.nc, prob.s, prob.m, prob.zdim
    J = np.zeros((prob.ddim, 2 * zd), dtype=np.result_type(z0, z1, float))
    IN = np.eye(N)
    mp = np.linalg.matrix_power
    if prob.integrator == PADE:
        c = pade_coeffs(prob.order)
        p = len(c) - 1
        B = sum(((-1) ** k) * c[k] * h ** k * mp(G, k) for k in range(p + 1))
        F = sum(c[k] * h ** k * mp(G, k) for k in range(p + 1))
        J[:s, prob.off_U:prob.off_U + s] = -np.kron(IN, F)
        J[:s, zd + prob.off_U:zd + prob.off_U + s] = np.kron(IN, B)
        for j in range(m):
            dB = sum(((-1) ** k) * c[k] * h ** k * _dGpow(G, prob.G_drives[j], k) for k in range(1, p + 1))
            dF = sum(c[k] * h ** k * _dGpow(G, prob.G_drives[j], k) for k in range(1, p + 1))
            J[:s, prob.off_a + j] = _vec(dB @ U1 - dF @ U0)
        if prob.free_time:
            dB = sum(((-1) ** k) * c[k] * k * h ** (k - 1) * mp(G, k) for k in range(1, p + 1))
            dF = sum(c[k] * k * h ** (k - 1) * mp(G, k) for k in range(1, p + 1))
            J[:s, prob.off_dt] = _vec(dB @ U1 - dF @ U0)
    else:
        E = expm_taylor(h * G)
        J[:s, prob.off_U:prob.off_U + s] = -np.kron(IN, E)
        J[:s, zd + prob.off_U:zd + prob.off_U + s] = np.eye(s)
        for j in range(m):
            _, L = expm_frechet_block(h * G, h * prob.G_drives[j])
            J[:s, prob.off_a + j] = -_vec(L @ U0)
        if prob.free_time:
            J[:s, prob.off_dt] = -_vec(G @ E @ U0)
    r = s
    for d in prob.derivs:
        dx0 = z0[d.dx_off:d.dx_off + d.dim]
        for i in range(d.dim):
            J[r + i, d.x_off + i] += -1.0
            J[r + i, zd + d.x_off + i] += 1.0
            J[r + i, d.dx_off + i] += -h
            if prob.free_time:
                J[r + i, prob.off_dt] += -dx0[i]
        r += d.dim
    return J


# --------------------------------------------------------------------------------------------
#  Per-interval dense Hessian of mu^T delta (2*zdim x 2*zdim, full symmetric)
# --------------------------------------------------------------------------------------------
def _interval_hessian_dense_exp(prob: Problem, z0: np.ndarray, z1: np.ndarray, mu: np.ndarray) -> np.ndarray:
    """Exponential integrator, delta = U1 - exp(h G(a)) U0 (reference README.md:79; the templates build
    `PiccoloOptions(integrator=:exponential)` problems and solve them with the Hessian left on:
    unitary_smooth_pulse_problem.jl:224-240,242-266 against `eval_hessian=false` spelled out at
    unitary_robustness_problem.jl:205,247).  delta is linear in U1, so every U_{t+1} block vanishes; with
    E = exp(hG), L_j = L_exp(hG; h G_j), M = reshape(mu, n, nc):
        (a_i, a_j) = -<M, L2_exp(hG; h G_i, h G_j) U0>        (U0, a_j) = -vec(L_j^T M)
        (a_j, h)   = -<M, (G_j E + G L_j) U0>                 (U0, h)   = -vec((G E)^T M)
        (h, h)     = -<M, G^2 E U0>                           (dx_i, h) = -mu_i
    [DERIVED]; certified by complex step of the Jacobian and an mpmath check (tests/test_oracle_math.py)."""
    U0, U1, a, h = _split(prob, z0, z1)
    G = _G_of(prob, a)
    n, N, s, m, zd = prob.n, prob.nc, prob.s, prob.m, prob.zdim
    M = mu[:s].reshape(n, N, order="F")
    Gd = prob.G_drives
    Hs = np.zeros((2 * zd, 2 * zd))

    def sym_set(i, j, v):
        Hs[i, j] += v
        if i != j:
            Hs[j, i] += v

    iU0 = prob.off_U
    E = expm_taylor(h * G)
    Ls = [expm_frechet_block(h * G, h * Gd[j])[1] for j in range(m)]
    for i in range(m):
        for j in range(i, m):
            L2 = expm_frechet2_block(h * G, h * Gd[i], h * Gd[j])
            sym_set(prob.off_a + i, prob.off_a + j, -np.sum(M * (L2 @ U0)))
    for j in range(m):
        v = -_vec(Ls[j].T @ M)
        for i in range(s):
            sym_set(iU0 + i, prob.off_a + j, v[i])
    if prob.free_time:
        ih = prob.off_dt
        for j in range(m):
            sym_set(prob.off_a + j, ih, -np.sum(M * ((Gd[j] @ E + G @ Ls[j]) @ U0)))
        sym_set(ih, ih, -np.sum(M * (G @ G @ E @ U0)))
        v = -_vec((G @ E).T @ M)
        for i in range(s):
            sym_set(iU0 + i, ih, v[i])
        r = s
        for d in prob.derivs:
            for i in range(d.dim):
                sym_set(d.dx_off + i, ih, -mu[r + i])
            r += d.dim
    return Hs


def interval_hessian_dense(prob: Problem, z0: np.ndarray, z1: np.ndarray, mu: np.ndarray) -> np.ndarray:
    if prob.integrator != PADE:
        return _interval_hessian_dense_exp(prob, z0, z1, mu)
    U0, U1, a, h = _split(prob, z0, z1)
    G = _G_of(prob, a)
    n, N, s, m, zd = prob.n, prob.nc, prob.s, prob.m, prob.zdim
    M = mu[:s].reshape(n, N, order="F")
    c = pade_coeffs(prob.order)
    p = len(c) - 1
    mp = np.linalg.matrix_power
    Hs = np.zeros((2 * zd, 2 * zd))
    Gd = prob.G_drives

    def sym_set(i, j, v):
        Hs[i, j] += v
        if i != j:
            Hs[j, i] += v

    iU0 = prob.off_U
    iU1 = zd + prob.off_U
    # (a_i, a_j)
    for i in range(m):
        for j in range(i, m):
            d2B = sum(((-1) ** k) * c[k] * h ** k * _d2Gpow(G, Gd[i], Gd[j], k) for k in range(2, p + 1))
            d2F = sum(c[k] * h ** k * _d2Gpow(G, Gd[i], Gd[j], k) for k in range(2, p + 1))
            v = np.sum(M * (d2B @ U1 - d2F @ U0)) if p >= 2 else 0.0
            sym_set(prob.off_a + i, prob.off_a + j, v)
    # (U, a_j)
    for j in range(m):
        dB = sum(((-1) ** k) * c[k] * h ** k * _dGpow(G, Gd[j], k) for k in range(1, p + 1))
        dF = sum(c[k] * h ** k * _dGpow(G, Gd[j], k) for k in range(1, p + 1))
        vB = _vec(dB.T @ M)
        vF = -_vec(dF.T @ M)
        for i in range(s):
            sym_set(iU1 + i, prob.off_a + j, vB[i])
            sym_set(iU0 + i, prob.off_a + j, vF[i])
    if prob.free_time:
        ih = prob.off_dt
        # (a_j, h)
        for j in range(m):
            dB = sum(((-1) ** k) * c[k] * k * h ** (k - 1) * _dGpow(G, Gd[j], k) for k in range(1, p + 1))
            dF = sum(c[k] * k * h ** (k - 1) * _dGpow(G, Gd[j], k) for k in range(1, p + 1))
            sym_set(prob.off_a + j, ih, np.sum(M * (dB @ U1 - dF @ U0)))
        # (h, h)
        d2B = sum(((-1) ** k) * c[k] * k * (k - 1) * h ** (k - 2) * mp(G, k) for k in range(2, p + 1))
        d2F = sum(c[k] * k * (k - 1) * h ** (k - 2) * mp(G, k) for k in range(2, p + 1))
        if p >= 2:
            sym_set(ih, ih, np.sum(M * (d2B @ U1 - d2F @ U0)))
        # (U, h)
        dB = sum(((-1) ** k) * c[k] * k * h ** (k - 1) * mp(G, k) for k in range(1, p + 1))
        dF = sum(c[k] * k * h ** (k - 1) * mp(G, k) for k in range(1, p + 1))
        vB = _vec(dB.T @ M)
        vF = -_vec(dF.T @ M)
        for i in range(s):
            sym_set(iU1 + i, ih, vB[i])
            sym_set(iU0 + i, ih, vF[i])
        # derivative integrators: d2/d(dx_i) dh = -mu_i
        r = s
        for d in prob.derivs:
            for i in range(d.dim):
                sym_set(d.dx_off + i, ih, -mu[r + i])
            r += d.dim
    return Hs


# --------------------------------------------------------------------------------------------
#  Canonical per-interval sparsity structure.  Local coordinates:
#  rows 0..ddim-1, cols 0..2*zdim-1 (col >= zdim  <=>  knot t+1).
# --------------------------------------------------------------------------------------------
def jac_structure_local(prob: Problem) -> List[Tuple[int, int]]:
    n, N, s, m, zd = prob.n, prob.nc, prob.s, prob.m, prob.zdim
    st: List[Tuple[int, int]] = []
    # 1. d/dU_t = -(I_N (x) F): N dense n x n diagonal blocks, column-major inside each block
    for q in range(N):
        for cc in range(n):
            for rr in range(n):
                st.append((q * n + rr, prob.off_U + q * n + cc))
    # 2. d/dU_{t+1} = I_N (x) B   (exponential integrator: identity, s diagonal entries)
    if prob.integrator == PADE:
        for q in range(N):
            for cc in range(n):
                for rr in range(n):
                    st.append((q * n + rr, zd + prob.off_U + q * n + cc))
    else:
        for i in range(s):
            st.append((i, zd + prob.off_U + i))
    # 3. d/da_j: dense s x m, column-major
    for j in range(m):
        for i in range(s):
            st.append((i, prob.off_a + j))
    # 4. d/ddt: dense s x 1
    if prob.free_time:
        for i in range(s):
            st.append((i, prob.off_dt))
    # 5. derivative integrators
    r = s
    for d in prob.derivs:
        for i in range(d.dim):
            st.append((r + i, d.x_off + i))
        for i in range(d.dim):
            st.append((r + i, zd + d.x_off + i))
        for i in range(d.dim):
            st.append((r + i, d.dx_off + i))
        if prob.free_time:
            for i in range(d.dim):
                st.append((r + i, prob.off_dt))
        r += d.dim
    return st


def hess_structure_local(prob: Problem) -> List[Tuple[int, int]]:
    """Upper-triangular (row <= col) local pairs over [z_t ; z_{t+1}] (size 2*zdim).  The exponential
    integrator is linear in U_{t+1}: its blocks 2 and 4 are structurally empty."""
    pade = prob.integrator == PADE
    s, m, zd = prob.s, prob.m, prob.zdim
    st: List[Tuple[int, int]] = []

    def up(i, j):
        st.append((min(i, j), max(i, j)))

    for j in range(m):                       # 1. (U_t, a_j)
        for i in range(s):
            up(prob.off_U + i, prob.off_a + j)
    for j in range(m if pade else 0):        # 2. (a_j, U_{t+1})
        for i in range(s):
            up(prob.off_a + j, zd + prob.off_U + i)
    if prob.free_time:
        for i in range(s):                   # 3. (U_t, h)
            up(prob.off_U + i, prob.off_dt)
        for i in range(s if pade else 0):    # 4. (h, U_{t+1})
            up(prob.off_dt, zd + prob.off_U + i)
    for j in range(m):                       # 5. (a_i, a_j), i <= j
        for i in range(j + 1):
            up(prob.off_a + i, prob.off_a + j)
    if prob.free_time:
        for j in range(m):                   # 6. (a_j, h)
            up(prob.off_a + j, prob.off_dt)
        up(prob.off_dt, prob.off_dt)         # 7. (h, h)
        for d in prob.derivs:                # 8. (dx_i, h)
            for i in range(d.dim):
                up(d.dx_off + i, prob.off_dt)
    return st


def jac_nnz_interval(prob: Problem) -> int:
    return len(jac_structure_local(prob))


def hess_pad(prob: Problem) -> int:
    """Explicit zero entries appended to an interval's Hessian values (alignment of the value blocks)."""
    k = len(hess_structure_local(prob))
    a = max(1, prob.hess_align)
    return (-k) % a if k else 0


def hess_nnz_interval(prob: Problem) -> int:
    return len(hess_structure_local(prob)) + hess_pad(prob)


def jac_structure(prob: Problem, one_based: bool = False, t_begin: int = 0, t_end: int | None = None):
    """Global COO (rows, cols) of dF, knot-major; row = t*ddim + r, col = t*zdim + c."""
    loc = np.array(jac_structure_local(prob), dtype=np.int64).reshape(-1, 2)
    t_end = prob.T - 1 if t_end is None else t_end
    ts = np.arange(t_begin, t_end, dtype=np.int64)
    rows = (ts[:, None] * prob.row_stride + prob.row_map[loc[None, :, 0]]).reshape(-1)
    cols = (ts[:, None] * prob.zdim + loc[None, :, 1]).reshape(-1)
    o = 1 if one_based else 0
    return rows + o, cols + o


def hess_structure(prob: Problem, one_based: bool = False, t_begin: int = 0, t_end: int | None = None):
    loc = np.array(hess_structure_local(prob), dtype=np.int64).reshape(-1, 2)
    if hess_pad(prob):
        loc = np.concatenate([loc, np.repeat(loc[:1], hess_pad(prob), axis=0)])
    t_end = prob.T - 1 if t_end is None else t_end
    ts = np.arange(t_begin, t_end, dtype=np.int64)
    rows = (ts[:, None] * prob.zdim + loc[None, :, 0]).reshape(-1)
    cols = (ts[:, None] * prob.zdim + loc[None, :, 1]).reshape(-1)
    o = 1 if one_based else 0
    return rows + o, cols + o


# --------------------------------------------------------------------------------------------
#  Full-trajectory evaluators (the three closures of QuantumDynamics; reference
#  test/scripts/integrator_test_1qubit.jl:41-52)
# --------------------------------------------------------------------------------------------
def _knots(prob: Problem, Z: np.ndarray, t: int):
    zd = prob.zdim
    return Z[t * zd:(t + 1) * zd], Z[(t + 1) * zd:(t + 2) * zd]


def F(prob: Problem, Z: np.ndarray, t_begin: int = 0, t_end: int | None = None) -> np.ndarray:
    t_end = prob.T - 1 if t_end is None else t_end
    out = np.zeros((t_end - t_begin, prob.row_stride))
    rm = prob.row_map
    for k, t in enumerate(range(t_begin, t_end)):
        z0, z1 = _knots(prob, Z, t)
        out[k, rm] = interval_residual(prob, z0, z1)
    return out.reshape(-1)


def dF(prob: Problem, Z: np.ndarray, t_begin: int = 0, t_end: int | None = None) -> np.ndarray:
    t_end = prob.T - 1 if t_end is None else t_end
    loc = np.array(jac_structure_local(prob), dtype=np.int64).reshape(-1, 2)
    nnz = loc.shape[0]
    out = np.empty((t_end - t_begin) * nnz)
    for k, t in enumerate(range(t_begin, t_end)):
        z0, z1 = _knots(prob, Z, t)
        J = interval_jacobian_dense(prob, z0, z1)
        out[k * nnz:(k + 1) * nnz] = J[loc[:, 0], loc[:, 1]]
    return out


def mu_d2F(prob: Problem, Z: np.ndarray, mu: np.ndarray, t_begin: int = 0, t_end: int | None = None) -> np.ndarray:
    t_end = prob.T - 1 if t_end is None else t_end
    loc = np.array(hess_structure_local(prob), dtype=np.int64).reshape(-1, 2)
    nnz = loc.shape[0]
    stride = nnz + hess_pad(prob)
    out = np.zeros((t_end - t_begin) * stride)
    rs, rm = prob.row_stride, prob.row_map
    for k, t in enumerate(range(t_begin, t_end)):
        z0, z1 = _knots(prob, Z, t)
        Hd = interval_hessian_dense(prob, z0, z1, mu[t * rs:(t + 1) * rs][rm])
        out[k * stride:k * stride + nnz] = Hd[loc[:, 0], loc[:, 1]]
    return out


def dense_from_coo(vals: np.ndarray, rows: np.ndarray, cols: np.ndarray, shape, symmetric=False) -> np.ndarray:
    """`dense(vals, structure, shape)` of reference test/test_utils.jl:14-27 (0-based here)."""
    Mx = np.zeros(shape)
    np.add.at(Mx, (rows, cols), vals)
    if symmetric:
        Mx = np.triu(Mx) + np.triu(Mx, 1).T
    return Mx


# --------------------------------------------------------------------------------------------
#  fidelity of the final knot (objective / constraint), [DERIVED] from the docstring
#  reference unitary_smooth_pulse_problem.jl:23-28:  l(U_T, U_goal) = | 1 - |tr(U_goal' U_T)| / N |
#  and the call site unitary_minimum_time_problem.jl:77 `iso_vec_unitary_fidelity(U_T, U_G, subspace=...)`.
#  The exact normalisation lives in the un-vendored PiccoloQuantumObjects 0.3 (|tr|/n vs |tr|^2/n^2): UNVERIFIED.
# --------------------------------------------------------------------------------------------
def fidelity_vectors(goal_iso: np.ndarray, N: int, subspace=None):
    """Constant vectors g_r, g_i with tr(U_goal' U) = g_r.u + i g_i.u over the subspace block; n = len(subspace)."""
    sub = list(range(N)) if subspace is None else list(subspace)
    G = iso_vec_to_operator(goal_iso)
    gr = np.zeros(2 * N * N)
    gi = np.zeros(2 * N * N)
    for j in sub:
        for i in sub:
            re_idx, im_idx = j * 2 * N + i, j * 2 * N + N + i
            gr[re_idx], gr[im_idx] = G[i, j].real, G[i, j].imag
            gi[re_idx], gi[im_idx] = -G[i, j].imag, G[i, j].real
    return gr, gi, len(sub)


def iso_vec_unitary_fidelity(u: np.ndarray, goal_iso: np.ndarray, subspace=None):
    N = int(round(math.sqrt(goal_iso.size / 2)))
    gr, gi, n = fidelity_vectors(goal_iso, N, subspace)
    tr, ti = gr @ u, gi @ u
    return np.sqrt(tr * tr + ti * ti) / n


def fidelity_value_grad_hess(u: np.ndarray, goal_iso: np.ndarray, subspace=None):
    """F, dF/du (s), d2F/du2 (s x s dense)."""
    N = int(round(math.sqrt(goal_iso.size / 2)))
    gr, gi, n = fidelity_vectors(goal_iso, N, subspace)
    tr, ti = gr @ u, gi @ u
    Fv = np.sqrt(tr * tr + ti * ti) / n
    grad = (tr * gr + ti * gi) / (n * n * Fv)
    hess = (np.outer(gr, gr) + np.outer(gi, gi)) / (n * n * Fv) - np.outer(grad, grad) / Fv
    return Fv, grad, hess


def infidelity_value_grad_hess(u: np.ndarray, goal_iso: np.ndarray, subspace=None):
    """l = |1 - F| with its gradient and Hessian."""
    Fv, g, Hm = fidelity_value_grad_hess(u, goal_iso, subspace)
    sg = 1.0 if 1.0 - Fv >= 0 else -1.0
    return abs(1.0 - Fv), -sg * g, -sg * Hm


# --------------------------------------------------------------------------------------------
#  Trajectory cost terms: quadratic regularisers + minimum-time term
#  J = sum_t 1/2 sum_k R_k (sc_t (v_tk - b_tk))^2 + D sum_{t < n_mt} dt_t, sc_t = dt_t (dt_scaled) or 1.
#  Reference call sites: unitary_smooth_pulse_problem.jl:151-153 (QuadraticRegularizer on a, da, dda),
#  unitary_minimum_time_problem.jl:67-69 (MinimumTimeObjective).  Definitions live in QuantumCollocationCore 0.3
#  (not vendored): the dt-scaling is recalled, not verified -- "parity unpinned" applies here as well.
# --------------------------------------------------------------------------------------------
@dataclass
class Terms:
    T: int
    zdim: int
    off_dt: int                       # -1: fixed timestep
    reg_index: np.ndarray             # offsets inside a knot
    reg_R: np.ndarray
    baseline: np.ndarray | None = None   # (T, n_reg)
    dt_scaled: bool = True       # default: dt inside the square (the templates pass timestep_name=, unitary_smooth_pulse_problem.jl:151-153); False: the docstring form 1/2 sum R x^2 (:13)
    dt_fixed: float = 0.0
    D: float = 0.0
    n_mt: int = 0
    global_dim: int = 0

    @property
    def cross(self) -> bool:
        return self.dt_scaled and self.off_dt >= 0 and len(self.reg_index) > 0


def _terms_knot(tm: Terms, Z, t):
    z = Z[t * tm.zdim:(t + 1) * tm.zdim]
    dt = z[tm.off_dt] if tm.off_dt >= 0 else tm.dt_fixed
    dv = z[np.asarray(tm.reg_index, dtype=int)] if len(tm.reg_index) else np.zeros(0, dtype=Z.dtype)
    if tm.baseline is not None:
        dv = dv - tm.baseline[t]
    return dt, dv


def terms_value(tm: Terms, Z):
    """Works on complex Z too (no abs / conj), so complex-step differentiation applies."""
    J = 0.0
    for t in range(tm.T):
        dt, dv = _terms_knot(tm, Z, t)
        sc = dt if tm.dt_scaled else 1.0
        J = J + 0.5 * np.sum(tm.reg_R * (sc * dv) ** 2)
        if tm.off_dt >= 0 and t < tm.n_mt:
            J = J + tm.D * dt
    return J


def terms_grad(tm: Terms, Z: np.ndarray) -> np.ndarray:
    g = np.zeros(tm.T * tm.zdim + tm.global_dim)
    for t in range(tm.T):
        dt, dv = _terms_knot(tm, Z, t)
        sc = dt if tm.dt_scaled else 1.0
        base = t * tm.zdim
        for k, j in enumerate(tm.reg_index):
            g[base + j] = tm.reg_R[k] * sc * sc * dv[k]
        if tm.off_dt >= 0:
            q = float(np.sum(tm.reg_R * dv * dv))
            g[base + tm.off_dt] = (dt * q if tm.dt_scaled else 0.0) + (tm.D if t < tm.n_mt else 0.0)
    return g


def terms_hess_structure(tm: Terms, one_based: bool = False):
    rows, cols = [], []
    b = 1 if one_based else 0
    for t in range(tm.T):
        c0 = t * tm.zdim + b
        for j in tm.reg_index:
            rows.append(c0 + j); cols.append(c0 + j)
        if tm.cross:
            for j in tm.reg_index:
                rows.append(c0 + min(j, tm.off_dt)); cols.append(c0 + max(j, tm.off_dt))
            rows.append(c0 + tm.off_dt); cols.append(c0 + tm.off_dt)
    return np.asarray(rows, dtype=np.int64), np.asarray(cols, dtype=np.int64)


def terms_hess(tm: Terms, Z: np.ndarray) -> np.ndarray:
    out = []
    for t in range(tm.T):
        dt, dv = _terms_knot(tm, Z, t)
        sc = dt if tm.dt_scaled else 1.0
        out.extend(tm.reg_R * sc * sc)
        if tm.cross:
            out.extend(2.0 * dt * tm.reg_R * dv)
            out.append(float(np.sum(tm.reg_R * dv * dv)))
    return np.asarray(out, dtype=np.float64)


# --------------------------------------------------------------------------------------------
#  Rollout: x_{t+1} = exp(dt_t G(a_t)) x_t  (`unitary_rollout` / `rollout` / `open_rollout`,
#  reference call sites trajectory_initialization.jl:426,493,547).  Returns the (n nc) x T state matrix.
# --------------------------------------------------------------------------------------------
def rollout(prob: Problem, Z: np.ndarray, init: np.ndarray) -> np.ndarray:
    n, nc = prob.n, prob.nc
    X = np.asarray(init, dtype=np.float64).reshape(n, nc, order="F")
    out = np.empty((n * nc, prob.T))
    out[:, 0] = _vec(X)
    for t in range(prob.T - 1):
        z = Z[t * prob.zdim:(t + 1) * prob.zdim]
        a = z[prob.off_a:prob.off_a + prob.m]
        h = z[prob.off_dt] if prob.free_time else prob.dt_fixed
        X = expm_taylor(h * _G_of(prob, a)) @ X
        out[:, t + 1] = _vec(X)
    return out


# --------------------------------------------------------------------------------------------
#  Ket and density-operator fidelities of the final knot (reference quantum_state_minimum_time_problem.jl:50 `iso_fidelity`,
#  quantum_state_smooth_pulse_problem.jl:133, density_operator_smooth_pulse_problem.jl:55).  The definitions
#  (|<g|psi>|^2, psi' rho psi) are recalled from PiccoloQuantumObjects 0.3, which is not vendored.
# --------------------------------------------------------------------------------------------
def ket_fidelity_value_grad_hess(psi_iso: np.ndarray, goal_iso: np.ndarray):
    N = goal_iso.size // 2
    g = goal_iso[:N] + 1j * goal_iso[N:]
    gr = np.concatenate([g.real, g.imag])
    gi = np.concatenate([-g.imag, g.real])
    tr, ti = gr @ psi_iso, gi @ psi_iso
    F = tr * tr + ti * ti
    return F, 2.0 * (tr * gr + ti * gi), 2.0 * (np.outer(gr, gr) + np.outer(gi, gi))


def density_fidelity_value_grad(rho_iso: np.ndarray, goal_ket_iso: np.ndarray):
    N = goal_ket_iso.size // 2
    g = goal_ket_iso[:N] + 1j * goal_ket_iso[N:]
    P = np.outer(g, g.conj())
    gr = np.concatenate([P.real.reshape(-1, order="F"), P.imag.reshape(-1, order="F")])
    return float(gr @ rho_iso), gr


# --------------------------------------------------------------------------------------------
#  Free-phase fidelity (reference unitary_minimum_time_problem.jl:86-100: `iso_vec_unitary_free_phase_fidelity(U_T, U_G,
#  phases, phase_operators; subspace)`, `FinalUnitaryFreePhaseFidelityConstraint`; phases are global variables of the
#  trajectory, trajectory_initialization.jl:370-380).  [RECALL] PiccoloQuantumObjects 0.3:
#      R(phi) = reduce(kron, [exp(i phi_k Op_k)]),   F = fidelity(R(phi) U, U_goal)   (phase gates applied AFTER the pulse)
#  Dense restatement over x = [u ; phi]: t = tr(G' R U) on the subspace, dt/dphi_k = i tr(G' O_k R U),
#  d2t/dphi_k dphi_l = -tr(G' O_k O_l R U), O_k = I (x) .. Op_k .. (x) I.  `form`: "abs" |t| / n (docstring), "abs2" |t|^2 / n^2.
# --------------------------------------------------------------------------------------------
def free_phase_operator(phases, phase_ops) -> np.ndarray:
    R = np.ones((1, 1), dtype=complex)
    for ph, Op in zip(phases, phase_ops):
        R = np.kron(R, expm_taylor(1j * ph * np.asarray(Op, dtype=complex)))
    return R


def _embedded_phase_ops(phase_ops):
    dims = [np.asarray(Op).shape[0] for Op in phase_ops]
    out = []
    for k, Op in enumerate(phase_ops):
        M = np.ones((1, 1), dtype=complex)
        for j, d in enumerate(dims):
            M = np.kron(M, np.asarray(Op, dtype=complex) if j == k else np.eye(d, dtype=complex))
        out.append(M)
    return out


def free_phase_fidelity_value_grad_hess(x: np.ndarray, goal_iso: np.ndarray, phase_ops, subspace=None, form: str = "abs"):
    """F, dF/dx, d2F/dx2 (dense) over x = [u (2N^2) ; phi (K)]."""
    N = int(round(math.sqrt(goal_iso.size / 2)))
    s, K = 2 * N * N, len(phase_ops)
    sub = list(range(N)) if subspace is None else list(subspace)
    n = len(sub)
    u, phi = np.asarray(x[:s], dtype=float), np.asarray(x[s:], dtype=float)
    G = iso_vec_to_operator(goal_iso)[np.ix_(sub, sub)]
    R = free_phase_operator(phi, phase_ops) if K else np.eye(n, dtype=complex)
    Ok = _embedded_phase_ops(phase_ops)
    assert R.shape == (n, n), "the phase operators' dimensions must multiply to the subspace size"

    def coeff_vectors(A):
        """t = tr(A' U_sub) = a . u + i b . u  ->  (a, b) over the iso-vec u."""
        a, b = np.zeros(s), np.zeros(s)
        for jb, j in enumerate(sub):
            for ia, i in enumerate(sub):
                c = np.conj(A[ia, jb])                 # coefficient of U_ij
                re, im = j * 2 * N + i, j * 2 * N + N + i
                a[re], b[re] = c.real, c.imag          # Re U_ij
                a[im], b[im] = -c.imag, c.real         # Im U_ij enters as i c
        return a, b

    P = s + K
    a1, b1 = np.zeros(P), np.zeros(P)
    a2, b2 = np.zeros((P, P)), np.zeros((P, P))
    a1[:s], b1[:s] = coeff_vectors(R.conj().T @ G)
    t = a1[:s] @ u + 1j * (b1[:s] @ u)
    for k in range(K):
        ak, bk = coeff_vectors((1j * Ok[k] @ R).conj().T @ G)
        a1[s + k], b1[s + k] = ak @ u, bk @ u
        a2[:s, s + k] = a2[s + k, :s] = ak
        b2[:s, s + k] = b2[s + k, :s] = bk
        for l in range(K):
            akl, bkl = coeff_vectors((-(Ok[k] @ Ok[l]) @ R).conj().T @ G)
            a2[s + k, s + l], b2[s + k, s + l] = akl @ u, bkl @ u
    tr, ti = t.real, t.imag
    S = tr * tr + ti * ti
    q = np.outer(a1, a1) + np.outer(b1, b1) + tr * a2 + ti * b2
    if form == "abs2":
        return S / n ** 2, 2.0 * (tr * a1 + ti * b1) / n ** 2, 2.0 * q / n ** 2
    Fv = math.sqrt(S) / n
    grad = (tr * a1 + ti * b1) / (n * n * Fv)
    return Fv, grad, q / (n * n * Fv) - np.outer(grad, grad) / Fv


def unitary_free_phase_fidelity(U: np.ndarray, U_goal: np.ndarray, phases, phase_ops, subspace=None, form: str = "abs") -> float:
    """Direct statement on operators (no derivatives): |tr(U_goal' R(phi) U)| / n on the subspace."""
    sub = list(range(U.shape[0])) if subspace is None else list(subspace)
    t = np.trace(U_goal[np.ix_(sub, sub)].conj().T @ free_phase_operator(phases, phase_ops) @ U[np.ix_(sub, sub)])
    return abs(t) ** 2 / len(sub) ** 2 if form == "abs2" else abs(t) / len(sub)
