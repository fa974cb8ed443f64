"""Gate / Pauli tables used by the reference's templates and tests (`GATES[:H]`, `PAULIS[:Z]`,
reference test/test_utils.jl:123-126, unitary_smooth_pulse_problem.jl:206-207)."""
from __future__ import annotations

from functools import reduce

import numpy as np

_I = np.eye(2, dtype=complex)
_X = np.array([[0, 1], [1, 0]], dtype=complex)
_Y = np.array([[0, -1j], [1j, 0]], dtype=complex)
_Z = np.array([[1, 0], [0, -1]], dtype=complex)

PAULIS = {"I": _I, "X": _X, "Y": _Y, "Z": _Z}


def operator_from_string(s: str) -> np.ndarray:
    """Kronecker product of single-qubit Paulis, e.g. "XIZ"."""
    return reduce(np.kron, [PAULIS[ch] for ch in s])


def qft(n_qubits: int) -> np.ndarray:
    d = 2 ** n_qubits
    w = np.exp(2j * np.pi / d)
    j, k = np.meshgrid(np.arange(d), np.arange(d), indexing="ij")
    return w ** (j * k) / np.sqrt(d)


def _controlled(U: np.ndarray, n_controls: int) -> np.ndarray:
    d = U.shape[0] * 2 ** n_controls
    out = np.eye(d, dtype=complex)
    out[-U.shape[0]:, -U.shape[0]:] = U
    return out


GATES = {
    "I": _I,
    "X": _X,
    "Y": _Y,
    "Z": _Z,
    "H": np.array([[1, 1], [1, -1]], dtype=complex) / np.sqrt(2),
    "CX": _controlled(_X, 1),
    "CNOT": _controlled(_X, 1),
    "CZ": _controlled(_Z, 1),
    "XI": np.kron(_X, _I),
    "TOFFOLI": _controlled(_X, 2),
    "QFT16": qft(4),
}
