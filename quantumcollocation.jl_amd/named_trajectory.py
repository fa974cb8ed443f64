"""Minimal `NamedTrajectory` (NamedTrajectories.jl 0.2) carrying exactly what the dynamics path
reads: `data` (dim x T), `datavec = vec(data)` (knot-major), `names`, `components` (row ranges),
`dims`, `dim`, `T`, `timestep`, `global_dim` (reference integrator_test_1qubit.jl:22-34,44-48;
fixture test/test_utils.jl:52-118)."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Mapping, Sequence, Union

import numpy as np


class NamedTrajectory:
    def __init__(
        self,
        components: Mapping[str, np.ndarray],
        controls: Sequence[str] = (),
        timestep: Union[str, float] = 1.0,
        bounds: Mapping[str, tuple] | None = None,
        initial: Mapping[str, np.ndarray] | None = None,
        final: Mapping[str, np.ndarray] | None = None,
        goal: Mapping[str, np.ndarray] | None = None,
        global_data: Mapping[str, np.ndarray] | None = None,
    ):
        self.names = tuple(components.keys())
        mats = []
        self.components: Dict[str, range] = {}
        row = 0
        T = None
        for name in self.names:
            M = np.atleast_2d(np.asarray(components[name], dtype=np.float64))
            T = M.shape[1] if T is None else T
            if M.shape[1] != T:
                raise ValueError(f"component {name} has {M.shape[1]} knots, expected {T}")
            self.components[name] = range(row, row + M.shape[0])
            row += M.shape[0]
            mats.append(M)
        self.T = int(T)
        self.dim = row
        self.data = np.asfortranarray(np.vstack(mats))
        if isinstance(timestep, str) and timestep not in self.components:
            raise ValueError(f"timestep component {timestep} not in trajectory")
        # NamedTrajectories treats a free timestep as a control variable
        if isinstance(timestep, str) and timestep not in controls:
            controls = tuple(controls) + (timestep,)
        self.controls = tuple(controls)
        self.timestep = timestep
        self.bounds = dict(bounds or {})
        self.initial = dict(initial or {})
        self.final = dict(final or {})
        self.goal = dict(goal or {})
        self.global_data = {k: np.asarray(v, dtype=np.float64).ravel() for k, v in (global_data or {}).items()}
        self.global_dim = int(sum(v.size for v in self.global_data.values()))
        control_dim = sum(len(self.components[c]) for c in self.controls)
        dims = {name: len(r) for name, r in self.components.items()}
        self.dims = SimpleNamespace(**{k: v for k, v in dims.items() if k.isidentifier()})
        self.dims.by_name = dims
        self.dims.controls = control_dim
        self.dims.states = self.dim - control_dim  # Z.dims.states (integrator_test_1qubit.jl:44)

    @property
    def datavec(self) -> np.ndarray:
        """vec(data) followed by the global components: length dim*T + global_dim."""
        v = self.data.reshape(-1, order="F")
        if self.global_dim:
            v = np.concatenate([v] + [self.global_data[k] for k in self.global_data])
        return np.ascontiguousarray(v)

    def __getitem__(self, name: str) -> np.ndarray:
        return self.data[self.components[name].start:self.components[name].stop, :]

    def offset(self, name: str) -> int:
        return self.components[name].start
