"""`QuantumDynamics(integrators, traj)`: the object the reference's MOI evaluator consumes, with the
field names and call shapes of reference test/scripts/integrator_test_1qubit.jl:41-52:

    dynamics.F(Z.datavec)                    -> residuals, length Z.dims.states*(Z.T-1)
    dynamics.dF(Z.datavec)                   -> Jacobian values in dF_structure order      ("∂F")
    dynamics.mu_d2F(Z.datavec, mu)           -> Hessian-of-Lagrangian values               ("μ∂²F")
    dynamics.dF_structure / mu_d2F_structure -> (rows, cols) global COO indices

(`getattr(dynamics, "∂F")` etc. resolve to the same members; `∂` is not a Python identifier.)
Every evaluation is a call into libqcolloc_hip.so; this class only builds the C descriptor.

RESULT LIFETIME of the vector-returning calls (the only shape the reference's evaluator uses, script lines 45-52): a returned
vector is the caller's for as long as the caller holds it (or any view of it), exactly as with the reference's closures, which
return a fresh vector.  What is recycled is only what the caller has LET GO OF: each closure keeps up to `result_ring` (default 3)
pinned, already-faulted-in vectors and hands one out again once the array of its last hand-out and every view of it have been
garbage-collected (a lease object under each hand-out, asked through a weak reference: explicit ownership, no reference counts;
holders of a raw address are invisible to it); while every vector of the ring is still held -- `[dyn.F(Z + h * e_i) for i in ...]` -- the call returns a newly
allocated array instead.  The evaluator's pattern (copy into Ipopt's buffer, drop) therefore never pays the 2.7 - 4 ms of
first-touch page faults a fresh 40 MB vector costs at config 3, ten times the evaluation, and nobody ever sees a result change
under their hands.  `fresh=True` on a call (or `result_ring=0` at construction) always allocates; `out=` writes into the caller's
own array.  `dF` and `F_dF` share one ring of Jacobian vectors; `close()` releases the rings.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import weakref
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .integrators import (DensityOperatorExponentialIntegrator, DerivativeIntegrator, QuantumStateExponentialIntegrator, QuantumStatePadeIntegrator,
                          UnitaryExponentialIntegrator, UnitaryPadeIntegrator, _KetIntegrator, _UnitaryIntegrator)
from .named_trajectory import NamedTrajectory

_KERNELS = {"auto": _lib.QC_KERNEL_AUTO, "lds": _lib.QC_KERNEL_LDS, "mfma": _lib.QC_KERNEL_MFMA}
_ALIASES = {"∂F": "dF", "μ∂²F": "mu_d2F", "∂F_structure": "dF_structure", "μ∂²F_structure": "mu_d2F_structure"}


class _PinnedBlock:
    """Owner of one block of qc_host_alloc, and its array-interface front: numpy keeps this object as the base of every array that
    views the block, so the block is freed when the last of them is gone (never during interpreter shutdown)."""

    def __init__(self, ptr: int, n: int):
        self.ptr = ptr
        self.nbytes = 8 * n
        _pinned_live[0] += self.nbytes
        self.__array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            if self.ptr and not sys.is_finalizing():
                _lib.lib.qc_host_free(C.c_void_p(self.ptr))
                _pinned_live[0] -= self.nbytes
        except Exception:   # noqa: BLE001
            pass
        self.ptr = 0


_pinned_live = [0]           # bytes of qc_host_alloc memory currently owned by _PinnedBlocks of this process
_pinned_warned = [False]


def pinned_cap_bytes() -> int:
    """Upper limit of page-locked result memory this process takes from the library (QC_PINNED_CAP_MB, default 2048): pinned pages
    cannot be swapped, and every dynamics object keeps a few result vectors (config 3: 40 MB each)."""
    try:
        return max(0, int(float(os.environ.get("QC_PINNED_CAP_MB", "2048")) * (1 << 20)))
    except ValueError:
        return 2048 << 20


def pinned_zeros(n: int) -> np.ndarray:
    """A float64 vector of n zeros in pinned host memory of the library (qc_host_alloc): the copy engine moves data from / into it
    without pinning pages per call, and `F(Z, out=...)` has the kernel write the residuals into it in place.  Ordinary zeros where there
    is no GPU, n is too small to matter, or the process-wide cap (`pinned_cap_bytes`) is reached -- said once on stderr, since the
    calls then pay the runtime's per-call pinning.  The memory is freed when the last view of it is garbage-collected."""
    n = int(n)
    if n * 8 < (64 << 10) or os.environ.get("QC_NO_PINNED"):
        return np.zeros(n)
    if _pinned_live[0] + n * 8 > pinned_cap_bytes():
        if not _pinned_warned[0]:
            _pinned_warned[0] = True
            print(f"qcolloc: {_pinned_live[0] >> 20} MiB of pinned result memory in use, the cap is {pinned_cap_bytes() >> 20} MiB "
                  "(QC_PINNED_CAP_MB): further result vectors are ordinary (pageable) arrays", file=sys.stderr)
        return np.zeros(n)
    p = C.c_void_p()
    if _lib.lib.qc_host_alloc(n * 8, C.byref(p)) != 0 or not p.value:
        return np.zeros(n)
    out = np.asarray(_PinnedBlock(p.value, n))
    out[:] = 0.0
    return out


class _Lease:
    """One hand-out of a ring vector: the array the caller receives is made over this object (array interface), so this object is the
    ultimate `.base` of that array and of every view derived from it, and it dies exactly when the last of them does.  The ring asks a
    weak reference whether the lease is still alive -- explicit ownership instead of CPython reference counts (free-
    threaded builds and PyPy count differently).  Holders of a raw address (`.ctypes.data`) are invisible to either scheme."""
    __slots__ = ("__array_interface__", "owner", "__weakref__")

    def __init__(self, owner: np.ndarray):
        self.owner = owner           # keeps the (pinned) block alive while the lease is
        self.__array_interface__ = {"shape": owner.shape, "typestr": "<f8", "data": (owner.ctypes.data, False), "version": 3}


class _RingSlot:
    __slots__ = ("owner", "lease", "quarantined")

    def __init__(self, owner: np.ndarray):
        self.owner, self.lease, self.quarantined = owner, None, False

    def free(self) -> bool:
        return not self.quarantined and (self.lease is None or self.lease() is None)

    def lend(self) -> np.ndarray:
        lease = _Lease(self.owner)
        self.lease = weakref.ref(lease)
        return np.asarray(lease)


def split_groups(integrators: Sequence):
    """Integrator list -> groups, one per unitary integrator (or run of ket integrators of one system); derivative
    integrators go with the state integrator they follow, so that their rows follow its rows.  Covers the lists the
    reference's templates build: [U, D, D] (unitary_smooth_pulse_problem.jl:175-179), [U_1 .. U_K, D, D]
    (unitary_sampling_problem.jl:149-155), [U, D] (unitary_bang_bang_problem.jl:171-175) and the direct sum's
    [U_1, D, D, U_2, D, D, ...] (unitary_direct_sum_problem.jl:127-130)."""
    state_types = (_UnitaryIntegrator, _KetIntegrator, DensityOperatorExponentialIntegrator)
    groups = []
    i, n = 0, len(integrators)
    while i < n:
        I = integrators[i]
        if isinstance(I, (_UnitaryIntegrator, DensityOperatorExponentialIntegrator)):
            groups.append([I])
            i += 1
        elif isinstance(I, _KetIntegrator):
            run = [I]
            i += 1
            while i < n and isinstance(integrators[i], _KetIntegrator) and integrators[i].system is run[0].system:
                run.append(integrators[i])
                i += 1
            groups.append(run)
        elif isinstance(I, DerivativeIntegrator):
            if not groups:
                raise NotImplementedError("the integrator list must start with a unitary or ket integrator")
            groups[-1].append(I)
            i += 1
        else:
            raise NotImplementedError("only DerivativeIntegrators may follow the state integrators")
    if not groups:
        raise NotImplementedError("the integrator list must start with a unitary or ket integrator")
    assert all(isinstance(g[0], state_types) for g in groups)
    return groups


def state_row_offset(traj: NamedTrajectory, name: str) -> int:
    """First row of state component `name` among the trajectory's STATE components in trajectory order (controls, the free
    timestep included, carry no dynamics rows): the row convention under which `Z.dims.states` rows per interval hold every
    integrator at its component's position (reference test/scripts/integrator_test_script.jl:23-44)."""
    if name in traj.controls:
        raise ValueError(f"{name} is a control component: it has no dynamics rows")
    r = 0
    for other in traj.names:
        if other == name:
            return r
        if other not in traj.controls:
            r += len(traj.components[other])
    raise KeyError(name)


def make_desc(integrators: Sequence, traj: NamedTrajectory, *, device: int = 0, kernel: str = "auto",
              t_range: Optional[Tuple[int, int]] = None, placement: Optional[dict] = None, rows: str = "stacked",
              hess_align: int = 0, jac_block_order: Optional[Sequence[int]] = None, hess_block_order: Optional[Sequence[int]] = None):
    """Translate (integrators, traj) into a qc_desc.  Returns (desc, keepalive).

    rows = "stacked": rows of an interval are the integrators' rows in integrator order (default; what every problem
    template produces, since it lists one integrator per state component in component order).  rows = "by_component":
    every integrator's rows sit at its state component's position inside Z.dims.states rows per interval; state components
    without an integrator leave structurally empty rows (qc_desc.row_placement = QC_ROWS_BY_COMPONENT).
    hess_align: qc_desc.hess_align (0 / 1 = exactly the structural entries, the reference's structure and the default; 16 = every
    interval's value block padded to whole 128-byte lines with explicit zero duplicates, for device-resident consumers).
    jac_block_order / hess_block_order: qc_desc's (ABI 0.6) -- permutations of _lib.QC_JB_* / QC_HB_*, the order of the value blocks
    inside an interval; None = the library's default order."""
    if not integrators or not isinstance(integrators[0], (_UnitaryIntegrator, _KetIntegrator, DensityOperatorExponentialIntegrator)):
        raise NotImplementedError("the first integrator must be the unitary (or the first ket) integrator "
                                  "(row order of reference unitary_smooth_pulse_problem.jl:175-179)")
    P = integrators[0]
    n_state = 1
    if isinstance(P, _KetIntegrator):
        # K ket integrators over the same system and controls = one 2N x K iso state; their trajectory components
        # must lie back to back in integrator order (quantum_state_smooth_pulse_problem.jl:146-152)
        while n_state < len(integrators) and isinstance(integrators[n_state], _KetIntegrator):
            Q = integrators[n_state]
            if type(Q) is not type(P) or Q.system is not P.system or Q.control_name != P.control_name or \
                    getattr(Q, "order", None) != getattr(P, "order", None):
                raise NotImplementedError("all ket integrators must share the system, the control and the integrator type")
            if traj.offset(Q.state_name) != traj.offset(P.state_name) + n_state * P.dim:
                raise NotImplementedError("ket components must be contiguous and in integrator order")
            n_state += 1
    derivs = list(integrators[n_state:])
    for D in derivs:
        if not isinstance(D, DerivativeIntegrator):
            raise NotImplementedError("only DerivativeIntegrators may follow the unitary integrator")
    if len(derivs) > _lib.QC_MAX_DERIV:
        raise ValueError("too many derivative integrators")
    sys = P.system
    d = _lib.qc_desc()
    d.N = sys.state_levels
    d.m = sys.n_drives
    d.T = traj.T
    d.zdim = traj.dim
    d.global_dim = traj.global_dim
    d.off_U = traj.offset(P.state_name)
    d.off_a = traj.offset(P.control_name)
    if isinstance(traj.timestep, str):
        d.off_dt = traj.offset(traj.timestep)
        d.dt_fixed = 0.0
    else:
        d.off_dt = -1
        d.dt_fixed = float(traj.timestep)
    d.state_cols = n_state if isinstance(P, (_KetIntegrator, DensityOperatorExponentialIntegrator)) else 0
    if isinstance(P, (UnitaryPadeIntegrator, QuantumStatePadeIntegrator)):
        d.integrator = _lib.QC_PADE
        d.pade_order = P.order
    elif isinstance(P, (UnitaryExponentialIntegrator, QuantumStateExponentialIntegrator, DensityOperatorExponentialIntegrator)):
        d.integrator = _lib.QC_EXPONENTIAL
        d.pade_order = 0
    else:
        raise NotImplementedError(type(P).__name__)
    d.n_deriv = len(derivs)
    for i, D in enumerate(derivs):
        d.deriv_x_off[i] = traj.offset(D.x)
        d.deriv_dx_off[i] = traj.offset(D.dx)
        d.deriv_dim[i] = D.dim
    n = 2 * sys.state_levels
    G0 = np.asfortranarray(sys.G_drift, dtype=np.float64)
    Gd = np.empty((max(1, sys.n_drives), n * n))
    for j, Gj in enumerate(sys.G_drives):
        Gd[j] = np.asarray(Gj, dtype=np.float64).reshape(-1, order="F")
    Gd = np.ascontiguousarray(Gd)
    d.G_drift = _lib.dptr(G0)
    d.G_drives = _lib.dptr(Gd)
    d.device = device
    d.kernel = _KERNELS[kernel]
    d.hess_align = int(hess_align)
    for name, order, n in (("jac_block_order", jac_block_order, _lib.QC_JAC_BLOCKS), ("hess_block_order", hess_block_order, _lib.QC_HESS_BLOCKS)):
        if order is not None:
            if sorted(int(x) for x in order) != list(range(n)):
                raise ValueError(f"{name} must be a permutation of 0..{n - 1}")
            for i, x in enumerate(order):
                getattr(d, name)[i] = int(x)
    if rows == "by_component":
        if placement:
            raise NotImplementedError("rows='by_component' with several unitary integrators")
        d.row_placement = _lib.QC_ROWS_BY_COMPONENT
        d.rows_per_interval = int(traj.dims.states)
        d.row_offset = state_row_offset(traj, P.state_name)
        for i, D in enumerate(derivs):
            d.deriv_row_off[i] = state_row_offset(traj, D.x)
    elif rows != "stacked":
        raise ValueError("rows must be 'stacked' or 'by_component'")
    if placement:
        for k, v in placement.items():
            setattr(d, k, int(v))
    if t_range is None:
        d.t_begin, d.t_end = 0, 0
    else:
        d.t_begin, d.t_end = int(t_range[0]), int(t_range[1])
        if d.t_begin == 0 and d.t_end == 0:
            raise ValueError("an empty shard must be expressed with t_begin = t_end > 0 or skipped")
    return d, (G0, Gd)


def desc_dims(desc) -> _lib.qc_dims_t:
    dims = _lib.qc_dims_t()
    _lib.check(_lib.lib.qc_desc_dims(C.byref(desc), C.byref(dims)))
    return dims


def desc_structures(desc, one_based: bool = False):
    """(jac_rows, jac_cols, hess_rows, hess_cols) implied by a descriptor; needs no GPU."""
    dims = desc_dims(desc)
    jr = np.empty(dims.jac_nnz, dtype=np.int64)
    jc = np.empty(dims.jac_nnz, dtype=np.int64)
    _lib.check(_lib.lib.qc_desc_jac_structure(C.byref(desc), _lib.iptr(jr), _lib.iptr(jc), int(one_based)))
    hr = np.empty(dims.hess_nnz, dtype=np.int64)
    hc = np.empty(dims.hess_nnz, dtype=np.int64)
    if dims.hess_nnz:
        _lib.check(_lib.lib.qc_desc_hess_structure(C.byref(desc), _lib.iptr(hr), _lib.iptr(hc), int(one_based)))
    return jr, jc, hr, hc


class QuantumDynamics:
    def __new__(cls, integrators: Sequence, traj: NamedTrajectory, **kw):
        if cls is QuantumDynamics and len(split_groups(integrators)) > 1:
            return super().__new__(ComposedQuantumDynamics)
        return super().__new__(cls)

    def __init__(self, integrators: Sequence, traj: NamedTrajectory, *, device: int = 0, kernel: str = "auto",
                 t_range: Optional[Tuple[int, int]] = None, eval_hessian: bool = True, devices: Optional[Sequence[int]] = None,
                 rows: str = "stacked", hess_align: int = 0, result_ring: int = 3,
                 jac_block_order: Optional[Sequence[int]] = None, hess_block_order: Optional[Sequence[int]] = None):
        """devices = [d0, d1, ...]: ONE evaluator over several GPUs (qc_create_multi): the interval range is split into
        len(devices) contiguous shards, shard i on HIP device devices[i] (ordinals may repeat); F / dF / mu_d2F behave
        exactly as on one device and return the same arrays.
        result_ring: vectors per closure that the vector-returning calls recycle once the caller has let go of them (module
        docstring); 0 = a fresh vector per call.
        jac_block_order / hess_block_order: the order of the value blocks inside an interval (make_desc)."""
        self.integrators = list(integrators)
        self.traj = traj
        self.eval_hessian = eval_hessian
        self._init_ring(result_ring)
        self.devices = None if devices is None else [int(x) for x in devices]
        if self.devices is not None:
            if not self.devices:
                raise ValueError("devices must name at least one device")
            device = self.devices[0]
        self._desc, self._keep = make_desc(integrators, traj, device=device, kernel=kernel, t_range=t_range, rows=rows, hess_align=hess_align,
                                           jac_block_order=jac_block_order, hess_block_order=hess_block_order)
        self._h = C.c_void_p()
        if self.devices is None:
            _lib.check(_lib.lib.qc_create(C.byref(self._desc), C.byref(self._h)))
        else:
            ids = (C.c_int32 * len(self.devices))(*self.devices)
            _lib.check(_lib.lib.qc_create_multi(C.byref(self._desc), len(self.devices), ids, C.byref(self._h)))
        dims = _lib.qc_dims_t()
        _lib.check(_lib.lib.qc_dims(self._h, C.byref(dims)), self._h)
        self.dims = dims
        self.dim = int(dims.ddim)          # dynamics rows per interval (= Z.dims.states)
        self.device = device
        self.kernel = {_lib.QC_KERNEL_LDS: "lds", _lib.QC_KERNEL_MFMA: "mfma"}[dims.kernel]
        self.kernel_names = (_lib.lib.qc_kernel_name(self._h, 0).decode(), _lib.lib.qc_kernel_name(self._h, 1).decode())
        self._structs = None

    # -- lifetime --------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib.qc_destroy(self._h)
            self._h = None
        self._drop_rings()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __getattr__(self, name):
        if name in _ALIASES:
            return getattr(self, _ALIASES[name])
        raise AttributeError(name)

    # -- structure --------------------------------------------------------------------------------
    def _structure(self, one_based=False):
        d = self.dims
        jr = np.empty(d.jac_nnz, dtype=np.int64)
        jc = np.empty(d.jac_nnz, dtype=np.int64)
        _lib.check(_lib.lib.qc_jac_structure(self._h, _lib.iptr(jr), _lib.iptr(jc), int(one_based)), self._h)
        hr = np.empty(d.hess_nnz, dtype=np.int64)
        hc = np.empty(d.hess_nnz, dtype=np.int64)
        if d.hess_nnz:
            _lib.check(_lib.lib.qc_hess_structure(self._h, _lib.iptr(hr), _lib.iptr(hc), int(one_based)), self._h)
        return jr, jc, hr, hc

    @property
    def dF_structure(self):
        """(rows, cols), 0-based global indices, in value order."""
        if self._structs is None:
            self._structs = self._structure(False)
        return self._structs[0], self._structs[1]

    @property
    def mu_d2F_structure(self):
        if self._structs is None:
            self._structs = self._structure(False)
        return self._structs[2], self._structs[3]

    def structure_tuples(self, which: str = "dF", one_based: bool = True) -> List[Tuple[int, int]]:
        """Vector of (row, col) tuples as the reference stores it (1-based by default)."""
        jr, jc, hr, hc = self._structure(one_based)
        r, c = (jr, jc) if which in ("dF", "∂F") else (hr, hc)
        return list(zip(r.tolist(), c.tolist()))

    def rollout(self, Z: np.ndarray, init: np.ndarray) -> np.ndarray:
        """States x_{t+1} = exp(dt_t G(a_t)) x_t for every knot, controls and timesteps taken from Z, x_0 = init:
        the (2N cols) x T matrix of `unitary_rollout` / `rollout` / `open_rollout` (trajectory_initialization.jl:426)."""
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        init = np.ascontiguousarray(init, dtype=np.float64).ravel()
        if Z.size != self.dims.Z_len:
            raise ValueError(f"Z has length {Z.size}, expected {self.dims.Z_len}")
        s = 2 * self._desc.N * (self._desc.state_cols or self._desc.N)
        if init.size != s:
            raise ValueError(f"initial state has length {init.size}, expected {s}")
        out = np.empty((self._desc.T, s))
        _lib.check(_lib.lib.qc_rollout(self._h, _lib.dptr(Z), _lib.dptr(init), _lib.dptr(out)), self._h)
        return np.ascontiguousarray(out.T)

    def F_dF_into(self, Z: np.ndarray, F: np.ndarray, J: np.ndarray) -> None:
        """qc_eval_F_jac into caller-owned arrays (no allocation): the call shape of an MOI callback."""
        _lib.check(_lib.lib.qc_eval_F_jac(self._h, _lib.dptr(Z), _lib.dptr(F), _lib.dptr(J)), self._h)

    # -- host-buffer evaluation (what Ipopt's callbacks use) --------------------------------------
    def _Z(self, Z):
        Z = np.ascontiguousarray(Z, dtype=np.float64)
        if Z.size != self.dims.Z_len:
            raise ValueError(f"Z has length {Z.size}, expected {self.dims.Z_len}")
        return Z

    def _init_ring(self, result_ring: int) -> None:
        if int(result_ring) < 0:
            raise ValueError("result_ring must be 0 (a fresh vector per call) or the number of recycled vectors per closure")
        self.result_ring = int(result_ring)
        self._rings = {}           # closure slot -> [_RingSlot]  (at most result_ring each, built as they are needed)
        self._lent_now = []        # slots handed out for the call in progress
        self._quarantine = False

    def _drop_rings(self) -> None:
        self._rings = {}           # vectors the caller still holds live on; the pinned blocks go with their last view

    def _out(self, name: str, n: int, out: Optional[np.ndarray] = None, fresh: bool = False, slot: Optional[str] = None) -> np.ndarray:
        """Result array of `n` doubles: the caller's `out`; else (fresh=True, or no ring) a newly allocated array; else a vector of
        closure `slot`'s ring that nobody outside the ring refers to any more (module docstring) -- a result is never written
        under a holder's hands.  The ring's vectors are written once when they are made, so a recycled one costs no page faults
        (2.7 - 4 ms per 40 MB at config 3); with every vector of a full ring still held the call allocates."""
        if out is not None:
            if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.flags.c_contiguous and out.size == n):
                raise ValueError(f"out for {name} must be a contiguous float64 array of {n} elements")
            return out
        if fresh or not self.result_ring or n == 0:
            return np.empty(n)
        ring = self._rings.setdefault(slot or name, [])
        for k in range(len(ring)):
            if ring[k].free():                 # nobody holds the array of its last hand-out (nor a view of it) any more
                ring.append(ring.pop(k))       # least recently handed out first
                self._lent_now.append(ring[-1])
                return ring[-1].lend()
        if len(ring) < self.result_ring:
            ring.append(_RingSlot(pinned_zeros(n)))
            self._lent_now.append(ring[-1])
            return ring[-1].lend()
        return np.empty(n)

    def _check(self, rc: int, h=None) -> None:
        """_lib.check for the host-buffer calls.  A call that came back with QC_ERR_HIP (a wait on the device ran into
        QC_HOST_TIMEOUT_MS) may have left work queued that still writes into its result vectors: their ring slots are quarantined until
        a later call on the handle has succeeded, i.e. the library has drained its streams."""
        lent, self._lent_now = self._lent_now, []
        try:
            _lib.check(rc, self._h if h is None else h)
        except _lib.QCollocError as e:
            if e.code == _lib.QC_ERR_HIP:
                for ent in lent:
                    ent.quarantined = True
                self._quarantine = True
            raise
        if self._quarantine:
            for ring in self._rings.values():
                for ent in ring:
                    ent.quarantined = False
            self._quarantine = False

    def F(self, Z, out: Optional[np.ndarray] = None, *, fresh: bool = False) -> np.ndarray:
        Z = self._Z(Z)
        out = self._out("F", int(self.dims.F_len), out, fresh)
        self._check(_lib.lib.qc_eval_F(self._h, _lib.dptr(Z), _lib.dptr(out)))
        return out

    def dF(self, Z, out: Optional[np.ndarray] = None, *, fresh: bool = False) -> np.ndarray:
        Z = self._Z(Z)
        out = self._out("J", int(self.dims.jac_nnz), out, fresh)
        self._check(_lib.lib.qc_eval_jac(self._h, _lib.dptr(Z), _lib.dptr(out)))
        return out

    def F_dF(self, Z, out: Optional[Tuple[np.ndarray, np.ndarray]] = None, *, fresh: bool = False):
        Z = self._Z(Z)
        F = self._out("F", int(self.dims.F_len), None if out is None else out[0], fresh)
        J = self._out("J", int(self.dims.jac_nnz), None if out is None else out[1], fresh)
        self._check(_lib.lib.qc_eval_F_jac(self._h, _lib.dptr(Z), _lib.dptr(F), _lib.dptr(J)))
        return F, J

    def mu_d2F(self, Z, mu, out: Optional[np.ndarray] = None, *, fresh: bool = False) -> np.ndarray:
        Z = self._Z(Z)
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        if mu.size != self.dims.n_rows:
            raise ValueError(f"mu has length {mu.size}, expected {self.dims.n_rows}")
        out = self._out("H", int(self.dims.hess_nnz), out, fresh)
        self._check(_lib.lib.qc_eval_hess(self._h, _lib.dptr(Z), _lib.dptr(mu), _lib.dptr(out)))
        return out

    def bind_host(self, which: str, Z: np.ndarray, *, mu: Optional[np.ndarray] = None, F: Optional[np.ndarray] = None,
                  J: Optional[np.ndarray] = None, H: Optional[np.ndarray] = None):
        """Pre-validated host-buffer call: a zero-argument callable returning the C status code, for timing loops (the per-call
        argument checks and pointer conversions of `F` / `dF` / `F_dF` / `mu_d2F` cost 8 - 10 us in Python, a tenth of a residual
        evaluation; a `ccall` from Julia has none of that).  which = "F" | "dF" | "F_dF" | "mu_d2F"; the callable keeps its arrays
        alive (`fn.keep`)."""
        import functools
        Z = self._Z(Z)
        if Z.ctypes.data % 8:
            raise ValueError("Z must be 8-byte aligned")
        zp = _lib.dptr(Z)
        # arrays the caller did not pass are the callable's own (never the closures' rings: a raw address is invisible to the rings'
        # ownership) and live as long as it does
        own = lambda given, n: given if given is not None else pinned_zeros(int(n))
        chk = lambda a, n, what: self._out(what, int(n), a)          # validates a caller's array
        if which == "F":
            F = chk(own(F, self.dims.F_len), self.dims.F_len, "F")
            fn = functools.partial(_lib.lib.qc_eval_F, self._h, zp, _lib.dptr(F))
            fn.keep = (Z, F)
            return fn
        if which == "dF":
            J = chk(own(J, self.dims.jac_nnz), self.dims.jac_nnz, "J")
            fn = functools.partial(_lib.lib.qc_eval_jac, self._h, zp, _lib.dptr(J))
            fn.keep = (Z, J)
            return fn
        if which == "F_dF":
            F = chk(own(F, self.dims.F_len), self.dims.F_len, "F")
            J = chk(own(J, self.dims.jac_nnz), self.dims.jac_nnz, "J")
            fn = functools.partial(_lib.lib.qc_eval_F_jac, self._h, zp, _lib.dptr(F), _lib.dptr(J))
            fn.keep = (Z, F, J)
            return fn
        if which == "mu_d2F":
            mu = np.ascontiguousarray(mu, dtype=np.float64)
            if mu.size != self.dims.n_rows:
                raise ValueError(f"mu has length {mu.size}, expected {self.dims.n_rows}")
            H = chk(own(H, self.dims.hess_nnz), self.dims.hess_nnz, "H")
            fn = functools.partial(_lib.lib.qc_eval_hess, self._h, zp, _lib.dptr(mu), _lib.dptr(H))
            fn.keep = (Z, mu, H)
            return fn
        raise ValueError(which)

    def set_new_x(self, new_x: bool) -> None:
        """Ipopt's `new_x`: False declares that the following host-buffer calls receive the x of the previous one (the accepted
        trial point: residuals, then Jacobian and Hessian at the same x), so the knots already on the device are used and Z
        is not read; stays in force until set_new_x(True) (qc_set_new_x)."""
        _lib.check(_lib.lib.qc_set_new_x(self._h, int(bool(new_x))), self._h)

    def knot_generation(self) -> int:
        """Uploads of a trajectory vector's knots this handle has done so far (qc_knot_generation).  A caller that elides uploads
        (set_new_x(False)) remembers it after the call that put ITS x on the device and elides only while it is unchanged: any other
        host-buffer call on the shared handle in between moves it."""
        return int(_lib.lib.qc_knot_generation(self._h))

    def host_expand_rate(self, reps: int = 5) -> float:
        """GB/s at which this host replicates the compact Jacobian form into the full value array (diagnostic, no GPU work)."""
        r = C.c_double()
        _lib.check(_lib.lib.qc_debug_host_expand_rate(self._h, reps, C.byref(r)), self._h)
        return r.value

    # -- multi-device handles ---------------------------------------------------------------------
    @property
    def n_shards(self) -> int:
        return int(_lib.lib.qc_multi_count(self._h))

    def shard_info(self, i: int) -> Tuple[int, int, int]:
        """(device, t_begin, t_end) of shard i of a multi-device evaluator."""
        dev, t0, t1 = C.c_int32(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib.qc_multi_shard_info(self._h, i, C.byref(dev), C.byref(t0), C.byref(t1)), self._h)
        return dev.value, t0.value, t1.value

    # -- device-resident evaluation (torch tensors are only the memory/stream plumbing) -----------
    def _dev_ptr(self, t: Optional[torch.Tensor], length: int, name: str):
        if t is None:
            return None
        if getattr(self, "devices", None) is not None:
            raise ValueError("device-resident calls on a multi-device evaluator go through qc_multi_eval_*_dev (one pointer per shard)")
        if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
            raise ValueError(f"{name} must be a contiguous float64 tensor on the GPU")
        if t.device.index != self.device:
            raise ValueError(f"{name} is on {t.device}, handle is bound to cuda:{self.device}")
        if t.numel() < length:
            raise ValueError(f"{name} has {t.numel()} elements, needs {length}")
        return C.c_void_p(t.data_ptr())

    def F_dF_device(self, Z: torch.Tensor, F: Optional[torch.Tensor], J: Optional[torch.Tensor], stream=None) -> None:
        """Asynchronous on `stream` (default: torch's current stream on the handle's device)."""
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        _lib.check(_lib.lib.qc_eval_F_jac_dev(
            self._h, self._dev_ptr(Z, self.dims.Z_len, "Z"), self._dev_ptr(F, self.dims.F_len, "F"),
            self._dev_ptr(J, self.dims.jac_nnz, "J"), C.c_void_p(st.cuda_stream)), self._h)

    def bind_F_dF_device(self, Z: torch.Tensor, F: Optional[torch.Tensor], J: Optional[torch.Tensor], stream=None):
        """Pre-validated launcher: returns a zero-argument callable that enqueues one evaluation (returns the
        C status code).  Keeps Python's per-call argument checking out of launch-bound loops; the tensors must
        outlive the callable."""
        import functools
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        return functools.partial(
            _lib.lib.qc_eval_F_jac_dev, self._h, self._dev_ptr(Z, self.dims.Z_len, "Z"),
            self._dev_ptr(F, self.dims.F_len, "F"), self._dev_ptr(J, self.dims.jac_nnz, "J"), C.c_void_p(st.cuda_stream))

    def bind_F_dF_mu_d2F_device(self, Z: torch.Tensor, mu: torch.Tensor, F: Optional[torch.Tensor], J: torch.Tensor, H: torch.Tensor, stream=None):
        """Pre-validated launcher of qc_eval_F_jac_hess_dev: dF and mu_d2F (and F) at one point in one call -- one kernel launch
        where a fused kernel serves the handle (`fused_kernel_name`), two otherwise, same values."""
        import functools
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        return functools.partial(
            _lib.lib.qc_eval_F_jac_hess_dev, self._h, self._dev_ptr(Z, self.dims.Z_len, "Z"), self._dev_ptr(mu, self.dims.n_rows, "mu"),
            self._dev_ptr(F, self.dims.F_len, "F"), self._dev_ptr(J, self.dims.jac_nnz, "J"), self._dev_ptr(H, self.dims.hess_nnz, "H"),
            C.c_void_p(st.cuda_stream))

    def F_dF_mu_d2F_device(self, Z, mu, F, J, H, stream=None) -> None:
        _lib.check(self.bind_F_dF_mu_d2F_device(Z, mu, F, J, H, stream)(), self._h)

    @property
    def fused_kernel_name(self) -> str:
        return _lib.lib.qc_kernel_name(self._h, 2).decode()

    def bind_mu_d2F_device(self, Z: torch.Tensor, mu: torch.Tensor, H: torch.Tensor, stream=None):
        """Pre-validated launcher of mu_d2F_device (see bind_F_dF_device)."""
        import functools
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        return functools.partial(
            _lib.lib.qc_eval_hess_dev, self._h, self._dev_ptr(Z, self.dims.Z_len, "Z"), self._dev_ptr(mu, self.dims.n_rows, "mu"),
            self._dev_ptr(H, self.dims.hess_nnz, "H"), C.c_void_p(st.cuda_stream))

    def mu_d2F_device(self, Z: torch.Tensor, mu: torch.Tensor, H: torch.Tensor, stream=None) -> None:
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        _lib.check(_lib.lib.qc_eval_hess_dev(
            self._h, self._dev_ptr(Z, self.dims.Z_len, "Z"), self._dev_ptr(mu, self.dims.n_rows, "mu"),
            self._dev_ptr(H, self.dims.hess_nnz, "H"), C.c_void_p(st.cuda_stream)), self._h)


class ComposedQuantumDynamics(QuantumDynamics):
    """`QuantumDynamics` over an integrator list with SEVERAL unitary integrators (UnitarySamplingProblem: K systems
    share the controls of a merged trajectory, reference unitary_sampling_problem.jl:134-155).  One HIP handle per
    unitary integrator; every handle writes its rows / Jacobian values / Hessian values straight into its slot of the
    problem's per-interval blocks (qc_desc.rows_per_interval / row_offset / jac_* / hess_*), so the value vectors come
    out in the reference's order: interval-major, integrator-major inside an interval."""

    def __init__(self, integrators: Sequence, traj: NamedTrajectory, *, device: int = 0, kernel: str = "auto",
                 t_range: Optional[Tuple[int, int]] = None, eval_hessian: bool = True, devices: Optional[Sequence[int]] = None,
                 rows: str = "stacked", hess_align: int = 0, result_ring: int = 3):
        """devices = [d0, d1, ...]: every member of the list is created over the same device list (qc_create_multi on a composed
        descriptor), so shard s of every member covers the same intervals on the same GPU; the host-buffer calls (`F`, `dF`, `F_dF`,
        `mu_d2F`: the "_list" entry points) then evaluate shard by shard, each GPU landing its slice of the caller's arrays over its
        own PCIe link.  Same arrays, bit for bit, as on one device."""
        if rows != "stacked":
            raise NotImplementedError("several unitary integrators: stacked rows")
        self._init_ring(result_ring)
        self.integrators = list(integrators)
        self.traj = traj
        self.eval_hessian = eval_hessian
        self.devices = None if devices is None else [int(x) for x in devices]
        if self.devices is not None:
            if not self.devices:
                raise ValueError("devices must name at least one device")
            device = self.devices[0]
        self.device = device
        groups = split_groups(integrators)
        own = []
        for g in groups:
            d0, keep = make_desc(g, traj, device=device, kernel=kernel, t_range=t_range, hess_align=1)
            own.append(desc_dims(d0))
        rows = sum(int(x.ddim) for x in own)
        jac = sum(int(x.jac_nnz_interval) for x in own)
        hess_own = sum(int(x.hess_nnz_interval) for x in own) if all(x.hess_nnz_interval for x in own) else 0
        # with hess_align > 1 the shared per-interval Hessian block is padded through its LAST handle (qc_desc.hess_tail_zeros)
        al = max(1, int(hess_align))        # 0 / 1: exactly the structural entries (the default)
        hess = -(-hess_own // al) * al if hess_own else 0
        self._parts = []
        ro = jo = ho = 0
        for gi, (g, x) in enumerate(zip(groups, own)):
            place = dict(rows_per_interval=rows, row_offset=ro, jac_per_interval=jac, jac_offset=jo,
                         hess_per_interval=hess if hess else 0, hess_offset=ho if hess else 0,
                         hess_tail_zeros=(hess - hess_own) if (hess and gi == len(groups) - 1) else 0)
            desc, keep = make_desc(g, traj, device=device, kernel=kernel, t_range=t_range, placement=place)
            h = C.c_void_p()
            if self.devices is None:
                _lib.check(_lib.lib.qc_create(C.byref(desc), C.byref(h)))
            else:
                ids = (C.c_int32 * len(self.devices))(*self.devices)
                _lib.check(_lib.lib.qc_create_multi(C.byref(desc), len(self.devices), ids, C.byref(h)))
            dims = _lib.qc_dims_t()
            _lib.check(_lib.lib.qc_dims(h, C.byref(dims)), h)
            self._parts.append((desc, keep, h, dims))
            ro += int(x.ddim)
            jo += int(x.jac_nnz_interval)
            ho += int(x.hess_nnz_interval)
        self._handles = (C.c_void_p * len(self._parts))(*[p[2] for p in self._parts])
        n_int = int(self._parts[0][3].n_intervals)
        d = _lib.qc_dims_t()
        d.n_rows, d.n_cols, d.ddim = self._parts[0][3].n_rows, self._parts[0][3].n_cols, rows
        d.jac_nnz_interval, d.hess_nnz_interval, d.n_intervals = jac, hess, n_int
        d.F_len, d.jac_nnz, d.hess_nnz, d.Z_len = rows * n_int, jac * n_int, hess * n_int, self._parts[0][3].Z_len
        d.kernel = self._parts[0][3].kernel
        self.dims = d
        self.dim = rows
        self.kernel = "+".join({_lib.QC_KERNEL_LDS: "lds", _lib.QC_KERNEL_MFMA: "mfma"}[p[3].kernel] for p in self._parts)
        self._h = None
        self._structs = None
        self._dev = torch.device("cuda", device)

    @property
    def n_shards(self) -> int:
        return int(_lib.lib.qc_multi_count(self._parts[0][2]))

    def shard_info(self, i: int) -> Tuple[int, int, int]:
        dev, t0, t1 = C.c_int32(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib.qc_multi_shard_info(self._parts[0][2], i, C.byref(dev), C.byref(t0), C.byref(t1)), self._parts[0][2])
        return dev.value, t0.value, t1.value

    def close(self):
        for _, _, h, _ in getattr(self, "_parts", []):
            if h:
                _lib.lib.qc_destroy(h)
        self._parts = []
        self._drop_rings()

    def _structure(self, one_based=False):
        n_int = int(self.dims.n_intervals)
        jr_l, jc_l, hr_l, hc_l = [], [], [], []
        for desc, keep, h, dims in self._parts:
            jr = np.empty(dims.jac_nnz, dtype=np.int64)
            jc = np.empty(dims.jac_nnz, dtype=np.int64)
            _lib.check(_lib.lib.qc_jac_structure(h, _lib.iptr(jr), _lib.iptr(jc), int(one_based)), h)
            jr_l.append(jr.reshape(n_int, -1))
            jc_l.append(jc.reshape(n_int, -1))
            if self.dims.hess_nnz:
                hr = np.empty(dims.hess_nnz, dtype=np.int64)
                hc = np.empty(dims.hess_nnz, dtype=np.int64)
                _lib.check(_lib.lib.qc_hess_structure(h, _lib.iptr(hr), _lib.iptr(hc), int(one_based)), h)
                hr_l.append(hr.reshape(n_int, -1))
                hc_l.append(hc.reshape(n_int, -1))
        cat = lambda L: np.ascontiguousarray(np.concatenate(L, axis=1).reshape(-1)) if L else np.empty(0, dtype=np.int64)
        return cat(jr_l), cat(jc_l), cat(hr_l), cat(hc_l)

    def F_dF_device(self, Z: torch.Tensor, F: Optional[torch.Tensor], J: Optional[torch.Tensor], stream=None) -> None:
        st = stream if stream is not None else torch.cuda.current_stream(self.device)     # (a list over several devices: _dev_ptr refuses)
        # one launch for all systems when their shapes allow it (qc_eval_F_jac_dev_multi falls back to one per handle)
        _lib.check(_lib.lib.qc_eval_F_jac_dev_multi(
            self._handles, len(self._parts), self._dev_ptr(Z, self.dims.Z_len, "Z"), self._dev_ptr(F, self.dims.F_len, "F"),
            self._dev_ptr(J, self.dims.jac_nnz, "J"), C.c_void_p(st.cuda_stream)), self._parts[0][2])

    def mu_d2F_device(self, Z: torch.Tensor, mu: torch.Tensor, H: torch.Tensor, stream=None) -> None:
        if not self.dims.hess_nnz:
            raise _lib.QCollocError(_lib.QC_ERR_UNSUPPORTED, "no analytic Hessian for this integrator list")
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        _lib.check(_lib.lib.qc_eval_hess_dev_multi(
            self._handles, len(self._parts), self._dev_ptr(Z, self.dims.Z_len, "Z"), self._dev_ptr(mu, self.dims.n_rows, "mu"),
            self._dev_ptr(H, self.dims.hess_nnz, "H"), C.c_void_p(st.cuda_stream)), self._parts[0][2])

    def bind_F_dF_device(self, Z, F, J, stream=None):
        return lambda: (self.F_dF_device(Z, F, J, stream), 0)[1]

    def bind_mu_d2F_device(self, Z, mu, H, stream=None):
        return lambda: (self.mu_d2F_device(Z, mu, H, stream), 0)[1]

    def rollout(self, Z, init, part: int = 0):
        """Rollout under the system and controls of the list's `part`-th state integrator (`unitary_rollout_fidelity(traj, sys_k)` of a
        sampling problem, reference unitary_sampling_problem.jl:187-193; a member of a direct sum): the (2N cols) x T state matrix."""
        desc, _, h, _ = self._parts[part]
        Z = self._Z(Z)
        init = np.ascontiguousarray(init, dtype=np.float64).ravel()
        s = 2 * desc.N * (desc.state_cols or desc.N)
        if init.size != s:
            raise ValueError(f"initial state has length {init.size}, expected {s}")
        out = np.empty((desc.T, s))
        _lib.check(_lib.lib.qc_rollout(h, _lib.dptr(Z), _lib.dptr(init), _lib.dptr(out)), h)
        return np.ascontiguousarray(out.T)

    def F_dF_into(self, Z, F, J):
        self.F_dF(Z, out=(F, J))

    # -- host buffers: the "_list" entry points of the C ABI (one upload, the batched launch, copies straight into the arrays) ----
    def F_dF(self, Z, out=None, *, fresh: bool = False):
        Z = self._Z(Z)
        F = self._out("F", int(self.dims.F_len), None if out is None else out[0], fresh)
        J = self._out("J", int(self.dims.jac_nnz), None if out is None else out[1], fresh)
        self._check(_lib.lib.qc_eval_F_jac_list(self._handles, len(self._parts), _lib.dptr(Z), _lib.dptr(F), _lib.dptr(J)), self._parts[0][2])
        return F, J

    def F(self, Z, out=None, *, fresh: bool = False):
        Z = self._Z(Z)
        F = self._out("F", int(self.dims.F_len), out, fresh)
        self._check(_lib.lib.qc_eval_F_list(self._handles, len(self._parts), _lib.dptr(Z), _lib.dptr(F)), self._parts[0][2])
        return F

    def dF(self, Z, out=None, *, fresh: bool = False):
        Z = self._Z(Z)
        J = self._out("J", int(self.dims.jac_nnz), out, fresh)
        self._check(_lib.lib.qc_eval_jac_list(self._handles, len(self._parts), _lib.dptr(Z), _lib.dptr(J)), self._parts[0][2])
        return J

    def mu_d2F(self, Z, mu, out=None, *, fresh: bool = False):
        if not self.dims.hess_nnz:
            raise _lib.QCollocError(_lib.QC_ERR_UNSUPPORTED, "no analytic Hessian for this integrator list")
        Z = self._Z(Z)
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        if mu.size != self.dims.n_rows:
            raise ValueError(f"mu has length {mu.size}, expected {self.dims.n_rows}")
        H = self._out("H", int(self.dims.hess_nnz), out, fresh)
        self._check(_lib.lib.qc_eval_hess_list(self._handles, len(self._parts), _lib.dptr(Z), _lib.dptr(mu), _lib.dptr(H)), self._parts[0][2])
        return H

    def knot_generation(self) -> int:
        return int(_lib.lib.qc_knot_generation(self._parts[0][2]))

    def set_new_x(self, new_x: bool) -> None:
        """Ipopt's `new_x` for the whole list (the list's first handle owns the knots on the device)."""
        _lib.check(_lib.lib.qc_set_new_x(self._parts[0][2], int(bool(new_x))), self._parts[0][2])
