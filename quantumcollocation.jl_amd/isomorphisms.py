"""Real isomorphism of complex operators (reference trajectory_initialization.jl:137,
`operator_to_iso_vec` / `iso_vec_to_operator` call sites :40-41,:96,:413-418).  Thin wrappers over
the C ABI so that Python and a Julia `ccall` see the same layout code."""
from __future__ import annotations

import numpy as np

from . import _lib


def _planes(A: np.ndarray):
    A = np.asarray(A, dtype=complex)
    if A.ndim != 2 or A.shape[0] != A.shape[1]:
        raise ValueError("expected a square matrix")
    return np.asfortranarray(A.real.copy()), np.asfortranarray(A.imag.copy()), A.shape[0]


def operator_to_iso_vec(U: np.ndarray) -> np.ndarray:
    """vec(vcat(real(U), imag(U))), column-major; length 2 N^2."""
    re, im, N = _planes(U)
    out = np.empty(2 * N * N)
    _lib.check(_lib.lib.qc_operator_to_iso_vec(N, _lib.dptr(re), _lib.dptr(im), _lib.dptr(out)))
    return out


def iso_vec_to_operator(v: np.ndarray) -> np.ndarray:
    v = np.ascontiguousarray(v, dtype=np.float64)
    N = int(round((v.size / 2) ** 0.5))
    if 2 * N * N != v.size:
        raise ValueError("iso-vec length must be 2 N^2")
    re = np.empty((N, N), order="F")
    im = np.empty((N, N), order="F")
    _lib.check(_lib.lib.qc_iso_vec_to_operator(N, _lib.dptr(v), _lib.dptr(re), _lib.dptr(im)))
    return re + 1j * im


def iso_generator(H: np.ndarray) -> np.ndarray:
    """G = iso(-iH) (2N x 2N real): d/dt [Re U; Im U] = G [Re U; Im U]."""
    re, im, N = _planes(H)
    G = np.empty((2 * N, 2 * N), order="F")
    _lib.check(_lib.lib.qc_generator_from_hamiltonian(N, _lib.dptr(re), _lib.dptr(im), _lib.dptr(G)))
    return G


def pade_coefficients(order: int) -> np.ndarray:
    out = np.empty(order // 2 + 1)
    _lib.check(_lib.lib.qc_pade_coefficients(order, _lib.dptr(out)))
    return out


def iso_operator(A: np.ndarray) -> np.ndarray:
    """Real isomorphism [[Re A, -Im A], [Im A, Re A]] of a complex matrix acting on [Re v; Im v]
    (`iso_generator(H)` is `iso_operator(-1j * H)`)."""
    A = np.asarray(A, dtype=complex)
    return np.asfortranarray(np.block([[A.real, -A.imag], [A.imag, A.real]]))


def density_to_iso_vec(rho: np.ndarray) -> np.ndarray:
    """[vec(Re rho); vec(Im rho)], column-major vec: the state of `DensityOperatorExponentialIntegrator`
    (reference density_operator_smooth_pulse_problem.jl:38-51, component `ρ⃗̃`); length 2 N^2."""
    rho = np.asarray(rho, dtype=complex)
    v = rho.reshape(-1, order="F")
    return np.concatenate([v.real, v.imag])


def iso_vec_to_density(v: np.ndarray) -> np.ndarray:
    v = np.asarray(v, dtype=np.float64)
    N = int(round((v.size / 2) ** 0.5))
    if 2 * N * N != v.size:
        raise ValueError("iso-vec length must be 2 N^2")
    return (v[:N * N] + 1j * v[N * N:]).reshape(N, N, order="F")
