"""`QuantumSystem(H_drift, H_drives)` (reference call sites unitary_smooth_pulse_problem.jl:199,
test/test_utils.jl:123): holds the Hamiltonians and their real-isomorphism generators
`G_drift`, `G_drives`, `n_drives`, `G(a)` consumed by the integrators."""
from __future__ import annotations

from typing import Sequence

import numpy as np

from .isomorphisms import iso_generator


class QuantumSystem:
    def __init__(self, H_drift: np.ndarray, H_drives: Sequence[np.ndarray]):
        self.H_drift = np.asarray(H_drift, dtype=complex)
        self.H_drives = [np.asarray(H, dtype=complex) for H in H_drives]
        self.levels = self.H_drift.shape[0]
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
