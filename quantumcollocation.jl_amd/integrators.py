"""Integrator descriptions with the reference's constructor shapes:

    UnitaryPadeIntegrator(state_name, control_name, system, traj; order=4)   unitary_smooth_pulse_problem.jl:165-167
    UnitaryExponentialIntegrator(state_name, control_name, system, traj)     unitary_smooth_pulse_problem.jl:168-170
    DerivativeIntegrator(x, dx, traj)                                        unitary_smooth_pulse_problem.jl:177-178
    QuantumStatePadeIntegrator / QuantumStateExponentialIntegrator           quantum_state_smooth_pulse_problem.jl:146-152
    DensityOperatorExponentialIntegrator(state_name, control_name, system, traj)   density_operator_smooth_pulse_problem.jl:104-106

They carry no arithmetic: `QuantumDynamics` turns a list of them into a C descriptor and the HIP
kernels evaluate them.
"""
from __future__ import annotations

from dataclasses import dataclass

from .named_trajectory import NamedTrajectory
from .quantum_systems import OpenQuantumSystem, QuantumSystem


@dataclass
class _UnitaryIntegrator:
    state_name: str
    control_name: str
    system: QuantumSystem
    traj: NamedTrajectory

    def __post_init__(self):
        s = 2 * self.system.levels ** 2
        if len(self.traj.components[self.state_name]) != s:
            raise ValueError(f"state component {self.state_name} must have length 2 N^2 = {s}")
        if len(self.traj.components[self.control_name]) != self.system.n_drives:
            raise ValueError("control component length must equal system.n_drives")

    @property
    def dim(self) -> int:
        return 2 * self.system.levels ** 2


@dataclass
class UnitaryPadeIntegrator(_UnitaryIntegrator):
    order: int = 4

    def __post_init__(self):
        super().__post_init__()
        if self.order < 2 or self.order % 2 or self.order > 20:
            raise ValueError("Pade order must be even, 2..20")


@dataclass
class UnitaryExponentialIntegrator(_UnitaryIntegrator):
    pass


@dataclass
class _KetIntegrator:
    """One ket psi~ = [Re psi; Im psi] (length 2N) under the same generators: the 1-column case of the unitary
    integrator (reference quantum_state_smooth_pulse_problem.jl:146-152, one integrator per state)."""
    state_name: str
    control_name: str
    system: QuantumSystem
    traj: NamedTrajectory

    def __post_init__(self):
        if len(self.traj.components[self.state_name]) != 2 * self.system.levels:
            raise ValueError(f"ket component {self.state_name} must have length 2 N = {2 * self.system.levels}")
        if len(self.traj.components[self.control_name]) != self.system.n_drives:
            raise ValueError("control component length must equal system.n_drives")

    @property
    def dim(self) -> int:
        return 2 * self.system.levels


@dataclass
class QuantumStatePadeIntegrator(_KetIntegrator):
    order: int = 4

    def __post_init__(self):
        super().__post_init__()
        if self.order < 2 or self.order % 2 or self.order > 20:
            raise ValueError("Pade order must be even, 2..20")


@dataclass
class QuantumStateExponentialIntegrator(_KetIntegrator):
    pass


@dataclass
class DensityOperatorExponentialIntegrator:
    """rho~_{t+1} = exp(dt G(a_t)) rho~_t on the iso-vec of vec(rho) (length 2 N^2) with the Lindblad generators of an
    `OpenQuantumSystem`: to the kernels this is the exponential ket integrator on a state of N^2 levels."""
    state_name: str
    control_name: str
    system: OpenQuantumSystem
    traj: NamedTrajectory

    def __post_init__(self):
        if not isinstance(self.system, OpenQuantumSystem):
            raise TypeError("DensityOperatorExponentialIntegrator needs an OpenQuantumSystem")
        if len(self.traj.components[self.state_name]) != 2 * self.system.levels ** 2:
            raise ValueError(f"density component {self.state_name} must have length 2 N^2 = {2 * self.system.levels ** 2}")
        if len(self.traj.components[self.control_name]) != self.system.n_drives:
            raise ValueError("control component length must equal system.n_drives")

    @property
    def dim(self) -> int:
        return 2 * self.system.levels ** 2


@dataclass
class DerivativeIntegrator:
    x: str
    dx: str
    traj: NamedTrajectory

    def __post_init__(self):
        if len(self.traj.components[self.x]) != len(self.traj.components[self.dx]):
            raise ValueError("x and dx must have equal dimension")

    @property
    def dim(self) -> int:
        return len(self.traj.components[self.x])
