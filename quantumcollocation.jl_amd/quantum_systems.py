"""`QuantumSystem(H_drift, H_drives)` (reference call sites unitary_smooth_pulse_problem.jl:199,
test/test_utils.jl:123): holds the Hamiltonians and their real-isomorphism generators
`G_drift`, `G_drives`, `n_drives`, `G(a)` consumed by the integrators."""
from __future__ import annotations

from typing import Sequence

import numpy as np

from .isomorphisms import iso_generator, iso_operator


class QuantumSystem:
    def __init__(self, H_drift: np.ndarray, H_drives: Sequence[np.ndarray]):
        self.H_drift = np.asarray(H_drift, dtype=complex)
        self.H_drives = [np.asarray(H, dtype=complex) for H in H_drives]
        self.levels = self.H_drift.shape[0]
        self.state_levels = self.levels          # dimension the generators act on (N; N^2 for an open system)
        for H in self.H_drives:
            if H.shape != self.H_drift.shape:
                raise ValueError("drive Hamiltonians must have the drift's shape")
        self.n_drives = len(self.H_drives)
        self.G_drift = iso_generator(self.H_drift)
        self.G_drives = [iso_generator(H) for H in self.H_drives]

    def G(self, a: Sequence[float]) -> np.ndarray:
        out = self.G_drift.copy()
        for aj, Gj in zip(a, self.G_drives):
            out += aj * Gj
        return out


class OpenQuantumSystem:
    """`OpenQuantumSystem(H_drift, H_drives, dissipation_operators)` (reference call site
    density_operator_smooth_pulse_problem.jl:4,104-106): Lindblad dynamics of vec(rho) (column-major),
        d vec(rho)/dt = [ -i (I (x) H - H^T (x) I) + sum_L ( conj(L) (x) L - 1/2 (I (x) L'L + (L'L)^T (x) I) ) ] vec(rho),
    in the real isomorphism on [Re vec(rho); Im vec(rho)].  Only the drift carries the dissipators; the drives enter
    through their commutator superoperators.  `G_drift` / `G_drives` are (2 N^2) x (2 N^2)."""

    def __init__(self, H_drift: np.ndarray, H_drives: Sequence[np.ndarray], dissipation_operators: Sequence[np.ndarray] = ()):
        self.H_drift = np.asarray(H_drift, dtype=complex)
        self.H_drives = [np.asarray(H, dtype=complex) for H in H_drives]
        self.dissipation_operators = [np.asarray(L, dtype=complex) for L in dissipation_operators]
        self.levels = self.H_drift.shape[0]
        self.state_levels = self.levels ** 2
        for H in self.H_drives + self.dissipation_operators:
            if H.shape != self.H_drift.shape:
                raise ValueError("drive Hamiltonians and dissipation operators must have the drift's shape")
        self.n_drives = len(self.H_drives)
        self.L_drift = self.hamiltonian_superoperator(self.H_drift)
        for L in self.dissipation_operators:
            self.L_drift = self.L_drift + self.dissipator_superoperator(L)
        self.L_drives = [self.hamiltonian_superoperator(H) for H in self.H_drives]
        self.G_drift = iso_operator(self.L_drift)
        self.G_drives = [iso_operator(L) for L in self.L_drives]

    @staticmethod
    def hamiltonian_superoperator(H: np.ndarray) -> np.ndarray:
        I = np.eye(H.shape[0])
        return -1j * (np.kron(I, H) - np.kron(H.T, I))

    @staticmethod
    def dissipator_superoperator(L: np.ndarray) -> np.ndarray:
        I = np.eye(L.shape[0])
        LdL = L.conj().T @ L
        return np.kron(L.conj(), L) - 0.5 * (np.kron(I, LdL) + np.kron(LdL.T, I))

    def G(self, a: Sequence[float]) -> np.ndarray:
        out = self.G_drift.copy()
        for aj, Gj in zip(a, self.G_drives):
            out += aj * Gj
        return out
