"""Host-side mirror of the reference interface: trajectory layout, configs, geodesic known answers."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_named_trajectory_layout(qc):
    with open(os.path.join(GOLD, "named_trajectory_type_1.json")) as f:
        fx = json.load(f)
    data = np.array(fx["data"])
    comps = {"Ũ⃗": data[0:8], "a": data[8:10], "da": data[10:12], "dda": data[12:14], "Δt": data[14:15]}
    Z = qc.NamedTrajectory(comps, controls=("dda", "Δt"), timestep="Δt")
    assert Z.dim == 15 and Z.T == 5 and Z.global_dim == 0
    assert Z.dims.states == 12 and Z.dims.a == 2        # rows of the dynamics per interval
    np.testing.assert_array_equal(Z.datavec, data.reshape(-1, order="F"))   # knot-major
    np.testing.assert_array_equal(Z.datavec[15:30], data[:, 1])
    assert [Z.offset(k) for k in ("Ũ⃗", "a", "da", "dda", "Δt")] == [0, 8, 10, 12, 14]
    Zf = qc.NamedTrajectory({k: v for k, v in comps.items() if k != "Δt"}, controls=("dda",), timestep=0.2)
    assert Zf.dim == 14 and Zf.dims.states == 12


@pytest.mark.parametrize("cfg,N,m,zdim,ddim", [(1, 2, 2, 15, 12), (2, 4, 4, 45, 40), (3, 8, 6, 147, 140), (5, 16, 8, 537, 528)])
def test_config_shapes_match_survey_table(qc, cfg, N, m, zdim, ddim):
    inp = qc.config_inputs(cfg, T=6)
    assert inp.system.levels == N and inp.system.n_drives == m
    assert inp.traj.dim == zdim and inp.traj.dims.states == ddim
    assert inp.traj.names == ("Ũ⃗", "a", "da", "dda", "Δt")      # trajectory_initialization.jl:357-362
    a = inp.traj["a"]
    assert not a[:, 0].any() and not a[:, -1].any() and np.abs(a).max() <= 1.0
    for G in [inp.system.G_drift] + inp.system.G_drives:
        np.testing.assert_array_equal(G.T, -G)


def test_unitary_geodesic_known_answers(qc):
    # reference trajectory_initialization.jl:588-642: endpoints are the iso-vecs, ||H|| = pi (X gate)
    X = qc.GATES["X"]
    geo, H = qc.unitary_geodesic(np.eye(2, dtype=complex), X, 10, return_generator=True)
    np.testing.assert_allclose(geo[:, 0], qc.operator_to_iso_vec(np.eye(2)), atol=1e-12)
    np.testing.assert_allclose(geo[:, -1], qc.operator_to_iso_vec(X), atol=1e-12)
    np.testing.assert_allclose(H, H.conj().T, atol=1e-12)
    assert abs(np.linalg.norm(H, 2) - np.pi / 2) < 1e-9 or abs(np.linalg.norm(H) - np.pi) < 1e-6


def test_dynamics_requires_unitary_integrator_first(qc):
    inp = qc.config_inputs(1, T=5)
    with pytest.raises(NotImplementedError):
        qc.make_desc(inp.integrators[1:], inp.traj)


def test_result_vectors_are_recycled_only_when_the_caller_let_go(qc):
    """QuantumDynamics._out (the vector-returning closures F / dF / F_dF / mu_d2F): a returned vector is never handed out again
    while the caller holds it or a view of it (the reference's closures return fresh vectors); once released, one of up
    to `result_ring` pre-faulted vectors per closure is recycled; `fresh=True` and `result_ring=0` allocate; `out=` is validated and
    passed through.  (No GPU: the method only manages host arrays.)"""
    import types
    obj = types.SimpleNamespace()
    qc.QuantumDynamics._init_ring(obj, 3)
    out = lambda name, n, **kw: qc.QuantumDynamics._out(obj, name, n, **kw)
    # the evaluator's pattern: use the result, drop it -> the same few vectors come round, none is allocated per call
    seen = set()
    for k in range(9):
        J = out("J", 10)
        J[:] = k
        seen.add(J.ctypes.data)
        del J
    assert len(seen) == 1                                                      # (one vector suffices for that pattern)
    # results that are HELD are never written again: a finite-difference loop keeps every one of its vectors
    held = []
    for k in range(7):
        J = out("J", 10)
        J[:] = 100 + k
        held.append(J)
    assert len({x.ctypes.data for x in held}) == 7 and [int(x[0]) for x in held] == [100 + k for k in range(7)]
    ring_ptrs = {v.owner.ctypes.data for v in obj._rings["J"]}
    assert len(ring_ptrs) == 3 and seen <= ring_ptrs and sum(x.ctypes.data in ring_ptrs for x in held) == 3   # the ring's three, then fresh arrays
    view = held[0][2:5]                                                        # a view keeps its vector out of circulation ...
    p0 = held[0].ctypes.data
    del held
    got = [out("J", 10) for _ in range(3)]
    assert p0 not in {g.ctypes.data for g in got} and int(view[0]) == 100
    del view, got
    assert p0 in {out("J", 10).ctypes.data for _ in range(3)}                  # ... and releases it with itself
    assert not np.shares_memory(out("H", 8), out("J", 10))                     # one ring per closure
    f1, f2 = out("J", 10, fresh=True), out("J", 10, fresh=True)
    assert f1.ctypes.data not in ring_ptrs and not np.shares_memory(f1, f2)
    mine = np.empty(10)
    assert out("J", 10, out=mine) is mine
    with pytest.raises(ValueError):
        out("J", 10, out=np.empty(9))
    with pytest.raises(ValueError):
        out("J", 10, out=np.empty(10, dtype=np.float32))
    none = types.SimpleNamespace()
    qc.QuantumDynamics._init_ring(none, 0)
    x, y = qc.QuantumDynamics._out(none, "J", 10), qc.QuantumDynamics._out(none, "J", 10)
    assert not np.shares_memory(x, y)
    # a call that timed out on the device (QC_ERR_HIP) quarantines the vectors it was given: queued work may still write into them; a
    # later successful call on the handle (the library has drained its streams by then) releases them
    q = types.SimpleNamespace(_h=None)
    qc.QuantumDynamics._init_ring(q, 2)
    a = qc.QuantumDynamics._out(q, "J", 10)
    pa = a.ctypes.data
    with pytest.raises(qc._lib.QCollocError):
        qc.QuantumDynamics._check(q, qc._lib.QC_ERR_HIP)
    del a
    b = qc.QuantumDynamics._out(q, "J", 10)
    assert b.ctypes.data != pa                                                 # not the quarantined vector, although nobody holds it
    qc.QuantumDynamics._check(q, qc._lib.QC_OK)
    del b
    assert pa in {qc.QuantumDynamics._out(q, "J", 10).ctypes.data for _ in range(2)}
    with pytest.raises(ValueError):
        qc.QuantumDynamics._init_ring(types.SimpleNamespace(), -1)
    qc.QuantumDynamics._drop_rings(obj)
    assert obj._rings == {}


def test_seeded_synthetic_inputs_are_pinned(qc):
    """bench.py's workload and most parity tests are the SEEDED synthetic inputs of SURVEY 8(d) (seed 20250218): a change in the order the
    random numbers are drawn in (state noise first, then controls) silently changes every such input -- it did once, in round 4, and only
    an end-to-end solve noticed.  Pinned here to the values the committed profiles were measured on."""
    for cfg, T, total, z7, zm3, zmax in ((1, 50, 103.82787723113978, 0.0033395584736109836, 0.15813731463912759, 1.0131176439937228),
                                         (3, 20, 166.28131240545315, 0.006430117657540341, 0.010376626553159643, 1.031110652365442)):
        z = qc.config_inputs(cfg, T=T).traj.datavec
        assert abs(float(z.sum()) - total) < 1e-9 and float(z[7]) == z7 and float(z[-3]) == zm3 and float(np.abs(z).max()) == zmax, (cfg, T)
