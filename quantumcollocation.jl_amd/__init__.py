"""MI355X-native knot-point evaluator for QuantumCollocation.jl's direct-collocation NLP.

Host-side mirror of the reference interface for the hot path over the C ABI
of include/qcolloc.h.  The directory name carries a dot, so load it through
`__graft_entry__.load_package()` (registers it as module `qcolloc_amd`).
"""
from . import _lib
from ._lib import QCollocError
from .evaluator import QuantumControlEvaluator
from .dynamics import ComposedQuantumDynamics, QuantumDynamics, desc_dims, desc_structures, make_desc, pinned_zeros, split_groups, state_row_offset
from .gates import GATES, PAULIS, operator_from_string
from .integrators import (DensityOperatorExponentialIntegrator, DerivativeIntegrator, QuantumStateExponentialIntegrator, QuantumStatePadeIntegrator,
                          UnitaryExponentialIntegrator, UnitaryPadeIntegrator)
from .isomorphisms import (density_to_iso_vec, iso_generator, iso_operator, iso_vec_to_density, iso_vec_to_operator,
                           operator_to_iso_vec, pade_coefficients)
from .named_trajectory import NamedTrajectory
from .objectives import (DensityOperatorPureStateInfidelityObjective, FinalQuantumStateFidelityConstraint, FinalUnitaryFidelityConstraint,
                         QuantumStateObjective, iso_fidelity, MinimumTimeObjective, QuadraticRegularizer, TimeStepsAllEqualConstraint,
                         TrajectoryObjective, UnitaryInfidelityObjective, iso_vec_unitary_fidelity, iso_vec_unitary_free_phase_fidelity,
                         UnitaryFreePhaseInfidelityObjective, FinalUnitaryFreePhaseFidelityConstraint)
from .problems import (CONFIGS, config_inputs, density_operator_smooth_pulse_inputs, multi_qubit_system, quantum_state_sampling_inputs, quantum_state_smooth_pulse_inputs,
                       unitary_bang_bang_inputs, unitary_direct_sum_inputs, unitary_sampling_inputs, unitary_smooth_pulse_inputs)
from .quantum_systems import OpenQuantumSystem, QuantumSystem
from .rollouts import open_rollout, rollout, rollout_fidelity, unitary_rollout, unitary_rollout_fidelity
from .trajectory_initialization import initialize_trajectory, unitary_geodesic

__all__ = [
    "QuantumControlEvaluator",
    "QuantumDynamics", "QuantumSystem", "NamedTrajectory", "UnitaryPadeIntegrator",
    "UnitaryExponentialIntegrator", "DerivativeIntegrator", "QuantumStatePadeIntegrator",
    "QuantumStateExponentialIntegrator", "quantum_state_smooth_pulse_inputs", "quantum_state_sampling_inputs", "unitary_sampling_inputs", "unitary_bang_bang_inputs", "unitary_direct_sum_inputs", "ComposedQuantumDynamics", "split_groups", "operator_to_iso_vec", "iso_vec_to_operator",
    "iso_generator", "pade_coefficients", "GATES", "PAULIS", "operator_from_string", "config_inputs",
    "unitary_smooth_pulse_inputs", "multi_qubit_system", "CONFIGS", "initialize_trajectory",
    "unitary_geodesic", "iso_vec_unitary_fidelity", "iso_vec_unitary_free_phase_fidelity", "UnitaryFreePhaseInfidelityObjective",
    "FinalUnitaryFreePhaseFidelityConstraint", "UnitaryInfidelityObjective", "FinalUnitaryFidelityConstraint", "QuantumStateObjective", "FinalQuantumStateFidelityConstraint", "DensityOperatorPureStateInfidelityObjective", "iso_fidelity",
    "QuadraticRegularizer", "MinimumTimeObjective", "TrajectoryObjective", "TimeStepsAllEqualConstraint",
    "OpenQuantumSystem", "DensityOperatorExponentialIntegrator", "density_operator_smooth_pulse_inputs",
    "density_to_iso_vec", "iso_vec_to_density", "iso_operator", "unitary_rollout", "rollout", "open_rollout", "unitary_rollout_fidelity", "rollout_fidelity", "make_desc", "desc_dims", "desc_structures", "state_row_offset", "QCollocError",
]
