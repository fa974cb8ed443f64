#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: full-trajectory constraint+Jacobian evaluations per
second for the 3-qubit Toffoli UnitarySmoothPulseProblem, T = 1000 knots per GPU.

A "step" is ONE pass of the hot path: `qc_eval_F_jac_dev` (== `dynamics.F(Z)` + `dynamics.dF(Z)`,
reference test/scripts/integrator_test_1qubit.jl:45-46) over every interval of the trajectory, with
Z already resident in HBM and the residual + Jacobian values left in HBM.

N GPUs: one process per GPU (torch.distributed/RCCL only for the barrier and the max-over-ranks);
the trajectory has T = 1000*N knots (N = 8 is BASELINE config 4, T = 8000) and each rank evaluates
its contiguous knot shard — no collective on the data path ("scaling": "weak").  `value` is
reported in T=1000-equivalent evaluations per second: intervals processed by all ranks / 999 / time.

Each step writes into the next of a ring of output buffers whose total exceeds 2 x 256 MiB, so a
step's stores cannot be absorbed by the Infinity Cache across steps ("Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
# The device-resident legs use the line-aligned layout of the Hessian values (qc_desc.hess_align = QC_HESS_ALIGN_LINE: every interval's
# block padded to whole 128-byte lines with explicit zeros), the documented opt-in of "_dev" consumers; the host-visible legs use the
# library's default, exactly the reference's structural entries (1 832 per interval at config 3).  The metric (F + dF) has no
# Hessian in it and is the same under both.
DEVICE_HESS_ALIGN = 16
SIDE_WARM = 1000             # untimed launches in front of every timed side leg: behind the light residual-only leg a 50-launch warm-up left the
                             # one-call launch 0.4 - 0.5 us slow (13.05 against 12.6 us on the same box; profiles/fused_bench.py: 12.65)
SIDE_STEPS = 500             # launches per timed side leg (mu_d2F alone, F alone, the one-call form, one output buffer); --side-steps
T_PER_GPU = 1000
RING_BYTES = 640 << 20


def structural_hess_nnz(dyn) -> int:
    """Entries per interval of the reference's mu_d2F_structure (1 832 at config 3, 9 277 at config 5), whatever padding
    the handle's layout adds behind them (hess_align = 16: 1 840 / 9 280)."""
    P = dyn._desc
    n, nc, m = 2 * P.N, (P.state_cols or P.N), P.m
    s, ft = n * nc, P.off_dt >= 0
    ub = 2 if P.integrator == 0 else 1                  # the exponential integrator has no block at knot t+1
    k = ub * s * m + m * (m + 1) // 2
    if ft:
        k += m + ub * s + 1 + sum(int(P.deriv_dim[i]) for i in range(P.n_deriv))
    assert k <= int(dyn.dims.hess_nnz_interval) < k + 4096
    return k


def usable_cores() -> int:
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(qc, inp, seconds: float):
    """The C restatement (oracle/qc_oracle.c, OpenMP over intervals) on the host cores, same workload: F + dF (the metric), then
    mu_d2F and F alone -- the three calls the reference's own harness times (test/scripts/integrator_test_1qubit.jl:45,46,52) --
    so that `ms/Ipopt-iter` has a CPU figure beside it.  `seconds` is the whole budget: 60 % F + dF, 25 % mu_d2F, 15 % F."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_bridge import problem_from_inputs
    import oracle.qc_oracle_c as oc

    prob = problem_from_inputs(inp)
    threads = usable_cores()
    co = oc.COracle(prob, threads=threads)
    Z = inp.traj.datavec
    mu = np.random.default_rng(1).standard_normal(int(prob.n_rows))

    def loop(fn, budget, min_calls=3):
        fn()  # warm-up (thread pool, page faults)
        n, t0 = 0, time.perf_counter()
        while True:
            fn()
            n += 1
            el = time.perf_counter() - t0
            if el >= budget and n >= min_calls:
                return n, el

    n, el = loop(lambda: co.F_dF(Z), 0.60 * seconds)
    rec = {"value": n / el, "unit": "evals/s", "cores": threads, "kind": "port",
           "sample": f"{n} full F+dF evaluations of the same T={prob.T} workload in {el:.1f} s (oracle/qc_oracle.c, OpenMP)",
           "F_dF_ms": el / n * 1e3}
    if prob.integrator == 0:
        nh, elh = loop(lambda: co.mu_d2F(Z, mu), 0.25 * seconds)
        nf, elf = loop(lambda: co.F_dF(Z, want_J=False), 0.15 * seconds)
        rec["hess_ms"] = elh / nh * 1e3
        rec["F_ms"] = elf / nf * 1e3
        rec["ms_per_ipopt_iter"] = rec["F_dF_ms"] + rec["hess_ms"] + rec["F_ms"]      # F + dF, mu_d2F, one line-search F
        rec["sample"] += f"; {nh} mu_d2F in {elh:.1f} s; {nf} F in {elf:.1f} s"
    return rec


PCIE_PEAK_GBS = 63.0    # MI355X_MICROARCH.md: PCIe Gen5 x16 host link
XGMI_LINK_GBS = 153.0   # MI355X_MICROARCH.md / SURVEY 8(e): one xGMI link, 7 per GPU in a full mesh


class NodeBarrier:
    """Barrier between the ranks of ONE node through a page in /dev/shm: one cache line per rank, each written by its owner only
    (a generation count), every rank spins until all lines carry the current generation.  The contract brackets the timed region
    with a barrier on both sides; the ranks exchange no data (the path shards without a collective), so the
    barrier is measurement scaffolding, and `dist.barrier()` on the RCCL backend -- an all-reduce kernel plus a stream
    synchronisation, 40 - 100 us -- would be a fifth to a half of a K = 20 measurement of 9-us steps at N > 1 and none of it at
    N = 1.  Falls back to `dist.barrier()` when the ranks are not all on this node or the page cannot be set up."""

    def __init__(self, rank, world):
        import mmap
        self.rank, self.world, self.gen, self.slots = rank, world, 0, None
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        ok = local_world == world and os.path.isdir("/dev/shm") and os.environ.get("QC_BENCH_BARRIER", "shm") != "dist"
        self.path = f"/dev/shm/qcolloc_bench_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}"
        if ok and rank == 0:
            try:
                with open(self.path, "wb") as f:
                    f.write(bytes(64 * world))
            except OSError:
                ok = False
        dist.barrier()                                       # (every rank, whatever happened above) the page exists, or does not
        if ok:
            try:
                self._f = open(self.path, "r+b")
                self._mm = mmap.mmap(self._f.fileno(), 64 * world)
                self.slots = np.frombuffer(self._mm, dtype=np.int64).reshape(world, 8)
            except (OSError, ValueError):
                self.slots = None
        flag = torch.tensor([1 if self.slots is not None else 0], dtype=torch.int32,
                            device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank or none
        if int(flag.item()) == 0:
            self.slots = None
        self.kind = "shared-memory page (/dev/shm)" if self.slots is not None else "dist.barrier"

    def wait(self):
        if self.slots is None:
            dist.barrier()
            return
        self.gen += 1
        self.slots[self.rank, 0] = self.gen
        col = self.slots[:, 0]
        spins, t_start = 0, None
        while int(col.min()) < self.gen:
            spins += 1
            if (spins & 0xFFFF) == 0:                        # a rank that died must not leave the others spinning for ever
                t_start = t_start or time.perf_counter()
                if time.perf_counter() - t_start > 300.0:
                    raise RuntimeError(f"bench.py: rank {self.rank} waited 300 s at the timing barrier (generation {self.gen}): {col.tolist()}")

    def close(self):
        if self.slots is not None:
            dist.barrier()                                   # nobody is still spinning on the page
            self.slots = None
            if self.rank == 0:
                try:
                    os.unlink(self.path)
                except OSError:
                    pass


PCIE_COPY_GBS = 54.5    # what the copy engine moves device -> host on this host (tests/hip/landing_probe.hip, profiles/r03_landing_probe.txt)


def host_visible_times(qc, dyn, Zs, reps=30):
    """PCIe-inclusive times of the host-buffer entry points (what the reference's consumer, a CPU Ipopt process, sees), caller-owned
    numpy arrays, milliseconds per call (median of `reps`).  `*_ms`: every call receives a NEW trajectory vector (Z goes up each
    time); `*_same_x_ms`: after qc_set_new_x(h, 0), the way Ipopt asks for the Jacobian and the Hessian at its accepted point;
    `ipopt_sequence_ms`: one F at a new x, then dF and mu_d2F at that x."""
    dims = dyn.dims
    Fh, Jh = np.empty(int(dims.F_len)), np.empty(int(dims.jac_nnz))
    Hh, mu = np.empty(int(dims.hess_nnz)), np.ones(int(dims.n_rows))
    nz = len(Zs)

    mins = {}

    def timed(fn, key, n=reps):   # median of `n` calls (the host is shared: one preempted call in thirty moves a mean by a tenth)
        for i in range(3):
            fn(i)
        ts = []
        for i in range(n):
            t0 = time.perf_counter()
            fn(i)
            ts.append(time.perf_counter() - t0)
        mins[key] = float(np.min(ts)) * 1e3
        return float(np.median(ts)) * 1e3

    has_h = bool(dims.hess_nnz)
    # pre-bound calls (QuantumDynamics.bind_host): the C entry point with its pointers, no per-call Python argument handling
    status = [0]
    cF = [dyn.bind_host("F", Z, F=Fh) for Z in Zs]
    cFJ = [dyn.bind_host("F_dF", Z, F=Fh, J=Jh) for Z in Zs]
    cJ = [dyn.bind_host("dF", Z, J=Jh) for Z in Zs]
    cH = [dyn.bind_host("mu_d2F", Z, mu=mu, H=Hh) for Z in Zs] if has_h else None

    def run(c):
        status[0] |= c()

    dyn.set_new_x(True)
    out = {"F_dF_ms": timed(lambda i: run(cFJ[i % nz]), "F_dF_ms"), "F_ms": timed(lambda i: run(cF[i % nz]), "F_ms")}
    if has_h:
        out["hess_ms"] = timed(lambda i: run(cH[i % nz]), "hess_ms")
    run(cF[0])
    dyn.set_new_x(False)
    out["jac_same_x_ms"] = timed(lambda i: run(cJ[0]), "jac_same_x_ms")
    if has_h:
        out["hess_same_x_ms"] = timed(lambda i: run(cH[0]), "hess_same_x_ms")

    def sequence(i):
        dyn.set_new_x(True)
        run(cF[i % nz])
        dyn.set_new_x(False)
        run(cJ[i % nz])
        if has_h:
            run(cH[i % nz])

    out["ipopt_sequence_ms"] = timed(sequence, "ipopt_sequence_ms")
    dyn.set_new_x(True)
    # the residual-only call into pinned memory of the library (qc_host_alloc; where the bindings take the vectors their closures hand
    # out and an evaluator its residual cache): the kernel writes it in place, no device-to-host copy, no pinning per call
    Fr = qc.pinned_zeros(int(dims.F_len))
    if type(Fr.base).__name__ == "_PinnedBlock":
        cFr = [dyn.bind_host("F", Z, F=Fr) for Z in Zs]
        out["F_pinned_ms"] = timed(lambda i: run(cFr[i % nz]), "F_pinned_ms")
        run(cF[0])
        run(cFr[0])
        assert np.array_equal(Fr, Fh), "pinned and plain residual arrays differ"
    out["fastest_call_ms"] = dict(mins)
    assert status[0] == 0, "a host-buffer call reported an error"
    # The reference-shaped closures: dynamics.F(Z) / dF(Z) / mu_d2F(Z, mu) RETURN a vector (integrator_test_1qubit.jl:45-52), which is
    # all QuantumCollocationCore's unchanged evaluator calls.  `closure_ms`: the bindings' default, each closure's ring of three result
    # vectors faulted in when the ring is built; `closure_fresh_ms`: a newly allocated vector per call (fresh=True; what a binding
    # without the ring does) -- the first touch of 40 MB of new pages, not the GPU, then sets the rate.  Through the Python mirror's
    # no-`out=` methods, argument checks included.
    if hasattr(dyn, "result_ring") and dyn.result_ring:
        cl, fr = {}, {}
        for name, call in (("F", lambda i, **kw: dyn.F(Zs[i % nz], **kw)), ("dF", lambda i, **kw: dyn.dF(Zs[i % nz], **kw)),
                           ("F_dF", lambda i, **kw: dyn.F_dF(Zs[i % nz], **kw)),
                           ("hess", (lambda i, **kw: dyn.mu_d2F(Zs[i % nz], mu, **kw)) if has_h else None)):
            if call is None:
                continue
            cl[name] = timed(lambda i: call(i), "closure." + name)
            fr[name] = timed(lambda i: call(i, fresh=True), "closure_fresh." + name, n=10)
        out["closure_ms"], out["closure_fresh_ms"] = cl, fr
        out["closure_result_ring"] = int(dyn.result_ring)
    return out


def host_visible_record(qc, inp, dyn, Zs, cpu_rec, t1000_equiv=1.0):
    """The `host_visible` object of the bench line (never `value`: the metric is device-resident).  Every time carries the bound
    it is measured against: the larger of the PCIe floor (bytes that must cross the link at the copy engine's measured rate) and,
    for the Jacobian, the host-replication floor (the value array written at the rate this host's worker team reaches)."""
    dims = dyn.dims
    t = host_visible_times(qc, dyn, Zs)
    n_int = int(dims.n_intervals)
    n = 2 * inp.system.levels
    nc = inp.system.levels
    compact = int(dims.jac_nnz_interval) - 2 * (nc - 1) * n * n     # one copy of the N replicated -F / B blocks
    z_bytes = 8 * inp.traj.dim * (n_int + 1)
    f_bytes = 8 * int(dims.ddim) * n_int
    jc_bytes = 8 * compact * n_int
    h_bytes = 8 * int(dims.hess_nnz)
    rec = dict(t)
    rec["statistic"] = "median of 30 calls of the pre-bound C entry points (no per-call Python argument handling: a ccall has none either)"
    try:
        expand = dyn.host_expand_rate(5)
    except Exception:   # noqa: BLE001  (no replicated blocks in this handle's Jacobian)
        expand = None
    rec["host_expand_GBps"] = expand
    rec["pcie_copy_GBps"] = PCIE_COPY_GBS
    n_links = max(1, len(set(getattr(dyn, "devices", None) or [0])))
    link = PCIE_COPY_GBS * 1e9 * n_links                                     # one link per distinct device
    expand_ms = 8 * int(dims.jac_nnz) / (expand * 1e9) * 1e3 if expand else 0.0
    # from four distinct devices on the library copies the Jacobian values in full over every link instead of replicating
    # N x 41.5 MB on one host (qc_create_multi, QC_HOST_MULTI_FULL)
    full = n_links >= int(os.environ.get("QC_HOST_MULTI_FULL", "4") or 0) > 0 and "QC_HOST_COMPACT" not in os.environ
    rec["jacobian_transfer"] = "full copy over every link" if full else "compact form + host replication"
    if full:
        jc_bytes, expand_ms = 8 * int(dims.jac_nnz), 0.0

    def bound(up, down, host_ms=0.0):     # upload and download are serial (the kernel needs the whole upload)
        return max((up + down) / link * 1e3, host_ms)

    bounds = {"F_dF_ms": bound(z_bytes, f_bytes + jc_bytes, expand_ms), "F_ms": bound(z_bytes, f_bytes),
              "jac_same_x_ms": bound(0, jc_bytes, expand_ms)}
    if "hess_ms" in t:
        bounds["hess_ms"] = bound(z_bytes + f_bytes, h_bytes)       # (mu has the length of F)
        bounds["hess_same_x_ms"] = bound(f_bytes, h_bytes)
        bounds["ipopt_sequence_ms"] = bounds["F_ms"] + bounds["jac_same_x_ms"] + bounds["hess_same_x_ms"]
    rec["bound_ms"] = bounds
    rec["frac_of_bound"] = {k: bounds[k] / t[k] for k in bounds if k in t}
    rec["evals_per_s"] = 1e3 / t["F_dF_ms"]
    rec["evals_per_s_T1000_equivalent"] = rec["evals_per_s"] * t1000_equiv
    rec["pcie_bytes_per_eval"] = z_bytes + f_bytes + jc_bytes
    rec["pcie_GBps_achieved"] = rec["pcie_bytes_per_eval"] / (t["F_dF_ms"] * 1e-3) / 1e9
    rec["pcie_GBps_peak"] = PCIE_PEAK_GBS
    if "hess_ms" in t:
        rec["ms_per_ipopt_iter"] = t["F_dF_ms"] + t["hess_ms"] + t["F_ms"]   # F + dF, mu_d2F, one line-search F: every call with a new x
    if cpu_rec:
        rec["speedup_vs_cpu_baseline"] = rec["evals_per_s_T1000_equivalent"] / cpu_rec["value"]
        if "ms_per_ipopt_iter" in rec and "ms_per_ipopt_iter" in cpu_rec:
            rec["ipopt_iter_speedup_vs_cpu_baseline"] = cpu_rec["ms_per_ipopt_iter"] * t1000_equiv / rec["ms_per_ipopt_iter"]
            rec["ipopt_sequence_speedup_vs_cpu_baseline"] = cpu_rec["ms_per_ipopt_iter"] * t1000_equiv / t["ipopt_sequence_ms"]
    return rec


def integrator_list_record(qc, cpu_rec, K=3, reps=15):
    """Host-buffer times of an integrator list with several state integrators -- a `UnitarySamplingProblem` over K copies of the
    metric's 3-qubit system with scattered drifts, shared controls, T = 1000 (reference unitary_sampling_problem.jl:134-155) -- through
    `qc_eval_*_list` (one upload, the compact watched transfer of every member's Jacobian values), the vector-returning calls of the
    Python mirror.  Extra object of `host_visible`; never `value`."""
    rng = np.random.default_rng(11)
    base = qc.multi_qubit_system(3)
    systems = [qc.QuantumSystem(base.H_drift * (1.0 + 0.1 * rng.standard_normal()), base.H_drives) for _ in range(K)]
    inp = qc.unitary_sampling_inputs(systems, qc.GATES["TOFFOLI"], T_PER_GPU)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, result_ring=3)
    Zs = [inp.traj.datavec + 1e-3 * k * rng.standard_normal(inp.traj.datavec.size) for k in range(3)]
    mu = np.ones(int(dyn.dims.n_rows))
    out = {"workload": f"UnitarySamplingProblem, {K} x 3-qubit systems, shared controls, T = {T_PER_GPU}", "K": K,
           "values_MB": {"F": 8e-6 * int(dyn.dims.F_len), "dF": 8e-6 * int(dyn.dims.jac_nnz), "mu_d2F": 8e-6 * int(dyn.dims.hess_nnz)}}
    for name, f in (("F_ms", lambda i: dyn.F(Zs[i % 3])), ("F_dF_ms", lambda i: dyn.F_dF(Zs[i % 3])), ("hess_ms", lambda i: dyn.mu_d2F(Zs[i % 3], mu))):
        for i in range(3):
            f(i)
        ts = []
        for i in range(reps):
            t0 = time.perf_counter()
            f(i)
            ts.append(time.perf_counter() - t0)
        out[name] = float(np.median(ts)) * 1e3
    if cpu_rec and "F_dF_ms" in cpu_rec:
        out["F_dF_speedup_vs_cpu_baseline"] = K * cpu_rec["F_dF_ms"] / out["F_dF_ms"]      # (the stand-in evaluates the K systems one after the other)
    dyn.close()
    return out


def _mfma_counters():
    """Counter-based MFMA figures of the newest profiles/r*_mfma_util.json (rocprofv3 --pmc runs, external to this process)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_mfma_util.json")))
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except Exception:   # noqa: BLE001
        return None, None


def config5_record(qc, dev_index, steps=300):
    """BASELINE config 5 (4-qubit QFT, T = 500, the 2N = 32 MFMA path) on the device: north_star asks for the MFMA
    utilisation of the large-n case.  The times are measured here; the MFMA instruction counts (SQ_INSTS_VALU_MFMA_F64 per
    launch, 2048 FLOP per v_mfma_f64_16x16x4_f64) and the counter-based MfmaUtil come from the newest
    profiles/r*_mfma_util.json and are labelled as external; they are omitted when that file does not hold these kernels."""
    inp = qc.config_inputs(5)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, device=dev_index, hess_align=DEVICE_HESS_ALIGN)
    dev = torch.device("cuda", dev_index)
    dims = dyn.dims
    n_int = int(dims.n_intervals)
    rng = np.random.default_rng(5)
    Z = torch.from_numpy(inp.traj.datavec).to(dev)
    mu = torch.from_numpy(rng.standard_normal(int(dims.n_rows))).to(dev)
    nb = 6     # 6 x 153 MB of values > the 256 MiB Infinity Cache
    Fb = [torch.empty(int(dims.F_len), dtype=torch.float64, device=dev) for _ in range(nb)]
    Jb = [torch.empty(int(dims.jac_nnz), dtype=torch.float64, device=dev) for _ in range(nb)]
    nh = 18    # 18 x 37 MB of Hessian values, beyond 2 x the Infinity Cache as well
    Hb = [torch.empty(int(dims.hess_nnz), dtype=torch.float64, device=dev) for _ in range(nh)]
    st = torch.cuda.current_stream(dev)

    def timed(fn):
        for i in range(200):     # (a short warm-up reads 0.3 - 0.5 us high behind a lighter leg: SIDE_WARM above)
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(steps):
            fn(i)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / steps

    jl = [dyn.bind_F_dF_device(Z, Fb[i], Jb[i], st) for i in range(nb)]
    hl = [dyn.bind_mu_d2F_device(Z, mu, Hb[i], st) for i in range(nh)]
    jac_us = timed(lambda i: jl[i % nb]())
    hess_us = timed(lambda i: hl[i % nh]())
    both = [dyn.bind_F_dF_mu_d2F_device(Z, mu, Fb[i % nb], Jb[i % nb], Hb[i % nh], st) for i in range(int(np.lcm(nb, nh)))]
    both_us = timed(lambda i: both[i % len(both)]())
    zdim, ddim = inp.traj.dim, int(dims.ddim)
    jac_bytes = 8 * (zdim * (n_int + 1) + (ddim + int(dims.jac_nnz_interval)) * n_int)
    hess_bytes = 8 * (zdim * (n_int + 1) + (ddim + structural_hess_nnz(dyn)) * n_int)     # structural entries: the padding of the line-aligned layout is not algorithmic
    peak_tf = 78.6       # f64 MFMA: 256 CUs x 4 SIMDs x 2048 FLOP / 64 cycles x 2.4 GHz
    rec = {"workload": qc.CONFIGS[5].description + f"; T={inp.traj.T}", "kernels": list(dyn.kernel_names),
           "F_dF_us": jac_us, "F_dF_hbm_frac": jac_bytes / (jac_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "hess_us": hess_us, "hess_hbm_frac": hess_bytes / (hess_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "F_dF_hess_one_call_us": both_us, "F_dF_hess_kernel": dyn.fused_kernel_name,
           "mfma_peak_TFLOPs": peak_tf}
    counters, src = _mfma_counters()
    if counters and inp.traj.T == 500:
        # the newest profiles/r*_mfma_util.json: per kernel template instantiation of config 5 (dense-image kernels: qc_mfma32_pade4_*;
        # sparse-drive kernels: qc_mfma32_ell_kernel<R, JAC, HESS, ...>)
        ell = dyn.kernel_names[0].endswith("-ell")
        keys = {"F_dF": "config5 qc_mfma32_ell_kernel<1, true, false" if ell else "config5 qc_mfma32_pade4_kernel<true",
                "hess": "config5 qc_mfma32_ell_kernel<1, false, true" if ell else "config5 qc_mfma32_pade4_hess_kernel<",
                "F_dF_hess_one_call": "config5 qc_mfma32_ell_kernel<1, true, true" if ell else None}
        ext = {"source": src, "note": "external: rocprofv3 --pmc passes recorded in that file, not measured by this run"}
        for name, us in (("F_dF", jac_us), ("hess", hess_us), ("F_dF_hess_one_call", both_us)):
            hits = [v for k, v in counters.items() if keys[name] and k.startswith(keys[name])]
            if len(hits) != 1:
                continue
            c = hits[0]
            ext[name + "_mfma_instructions_per_launch"] = c["SQ_INSTS_VALU_MFMA_F64"]
            ext[name + "_MfmaUtil_percent"] = c["MfmaUtil_percent"]
            rec[name + "_mfma_frac"] = c["SQ_INSTS_VALU_MFMA_F64"] * 2048 / (us * 1e-6) / 1e12 / peak_tf
        if len(ext) > 2:
            rec["mfma_counters"] = ext
    dyn.close()
    return rec


def exponential_record(qc, dev_index, steps=200):
    """The metric workload with `PiccoloOptions(integrator=:exponential)` (SURVEY 8 rows a6 / a6'; the reference solves such problems with
    the Hessian on, unitary_smooth_pulse_problem.jl:224-266): device times of F + dF, mu_d2F and the residual alone, and the Ipopt-order
    proxy built from them.  These launches are bound by the f64 matrix pipes; the MFMA instruction counts per launch come from the newest
    profiles/r*_mfma_util.json (external, labelled) when it holds these kernels."""
    inp = qc.config_inputs(3, T=T_PER_GPU, integrator="exponential")
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, device=dev_index)
    dev = torch.device("cuda", dev_index)
    dims = dyn.dims
    rng = np.random.default_rng(6)
    Z = torch.from_numpy(inp.traj.datavec).to(dev)
    mu = torch.from_numpy(rng.standard_normal(int(dims.n_rows))).to(dev)
    nb = max(2, -(-RING_BYTES // (8 * int(dims.jac_nnz))))
    nh = max(2, -(-RING_BYTES // (8 * max(1, int(dims.hess_nnz)))))
    Fb = [torch.empty(int(dims.F_len), dtype=torch.float64, device=dev) for _ in range(nb)]
    Jb = [torch.empty(int(dims.jac_nnz), dtype=torch.float64, device=dev) for _ in range(nb)]
    Hb = [torch.empty(int(dims.hess_nnz), dtype=torch.float64, device=dev) for _ in range(nh)]
    st = torch.cuda.current_stream(dev)

    def timed(fn):
        for i in range(30):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(steps):
            fn(i)
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / steps

    jl = [dyn.bind_F_dF_device(Z, Fb[i], Jb[i], st) for i in range(nb)]
    hl = [dyn.bind_mu_d2F_device(Z, mu, Hb[i], st) for i in range(nh)]
    fl = [dyn.bind_F_dF_device(Z, Fb[i], None, st) for i in range(nb)]      # (the residual alone: a line-search trial)
    jac_us, hess_us, f_us = timed(lambda i: jl[i % nb]()), timed(lambda i: hl[i % nh]()), timed(lambda i: fl[i % nb]())
    peak_tf = 78.6
    rec = {"workload": f"3-qubit Toffoli UnitarySmoothPulseProblem, integrator = :exponential, T={inp.traj.T}", "kernels": list(dyn.kernel_names),
           "F_dF_us": jac_us, "hess_us": hess_us, "F_only_us": f_us, "ms_per_ipopt_iter_proxy_device": (jac_us + hess_us + f_us) * 1e-3,
           "hess_nnz_interval": int(dims.hess_nnz_interval), "bound": "mfma", "mfma_peak_TFLOPs": peak_tf}
    counters, src = _mfma_counters()
    if counters:
        ext = {"source": src, "note": "external: rocprofv3 --pmc passes recorded in that file, not measured by this run"}
        for name, us, key in (("F_dF", jac_us, "config3 exponential qc_mfma16_exp_kernel<true"), ("hess", hess_us, "config3 exponential qc_mfma16_exp_hess_kernel<")):
            hits = [v for k, v in counters.items() if k.startswith(key)]
            if len(hits) != 1:
                continue
            c = hits[0]
            ext[name + "_mfma_instructions_per_launch"] = c["SQ_INSTS_VALU_MFMA_F64"]
            ext[name + "_MfmaUtil_percent"] = c["MfmaUtil_percent"]
            rec[name + "_mfma_frac"] = c["SQ_INSTS_VALU_MFMA_F64"] * 2048 / (us * 1e-6) / 1e12 / peak_tf
        if len(ext) > 2:
            rec["mfma_counters"] = ext
    dyn.close()
    return rec


KERNEL_SOURCES = ("quantumcollocation.jl_amd/csrc/qc_mfma_kernels.hip", "quantumcollocation.jl_amd/csrc/qc_mfma_common.h",
                  "quantumcollocation.jl_amd/csrc/qc_internal.h")


def kernel_source_hash() -> str:
    """sha256 over the sources of the metric's kernel (qc_mfma16_pade4_kernel): what profiles/pmc_traffic.json is stamped with."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def recorded_traffic(kernel, config, T):
    """HBM bytes per launch of the metric's kernel from profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes,
    external to this run).  Dropped -- null, with the reason -- when the record is for another workload or when the kernel's
    sources have changed since the counters were collected (the record carries their sha256 and the commit)."""
    tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tj):
        return None, None
    try:
        rec = json.load(open(tj))
    except Exception:   # noqa: BLE001
        return None, "profiles/pmc_traffic.json unreadable"
    if not (rec.get("kernel") == kernel and rec.get("config") == config and rec.get("T") == T):
        return None, None
    if rec.get("kernel_source_sha256") != kernel_source_hash():
        return None, ("profiles/pmc_traffic.json dropped: the kernel sources have changed since its counters were collected "
                      f"(commit {rec.get('commit', '?')})")
    return rec.get("hbm_bytes_per_launch"), (f"profiles/pmc_traffic.json (rocprofv3 --pmc passes at commit {rec.get('commit', '?')}, kernel sources unchanged "
                                              "since; not measured by this run)")


def long_trajectory_record(qc, dev_index, T=8000, steps=200):
    """BASELINE config 4's trajectory (3-qubit Toffoli, T = 8000) on ONE GPU: the long-trajectory regime of the metric's kernel
    (persistent grid) while no 8-GPU node is available -- F + dF, mu_d2F and the one-call form, stream events over rings of
    output vectors beyond 2 x the Infinity Cache."""
    inp = qc.config_inputs(4, T=T)
    dyn = qc.QuantumDynamics(inp.integrators, inp.traj, device=dev_index, hess_align=DEVICE_HESS_ALIGN)
    dev = torch.device("cuda", dev_index)
    dims = dyn.dims
    n_int = int(dims.n_intervals)
    rng = np.random.default_rng(4)
    Zs = [torch.from_numpy(inp.traj.datavec + (1e-3 * rng.standard_normal(inp.traj.datavec.size) if k else 0.0)).to(dev) for k in range(2)]
    mu = torch.from_numpy(rng.standard_normal(int(dims.n_rows))).to(dev)
    nb = max(2, -(-RING_BYTES // (8 * int(dims.jac_nnz))))      # 2 x 322 MB of Jacobian values
    nh = max(2, -(-RING_BYTES // (8 * int(dims.hess_nnz))))
    Fb = [torch.empty(int(dims.F_len), dtype=torch.float64, device=dev) for _ in range(nb)]
    Jb = [torch.empty(int(dims.jac_nnz), dtype=torch.float64, device=dev) for _ in range(nb)]
    Hb = [torch.empty(int(dims.hess_nnz), dtype=torch.float64, device=dev) for _ in range(nh)]
    st = torch.cuda.current_stream(dev)
    status = [0]

    def timed(calls):
        for i in range(60):
            status[0] |= calls[i % len(calls)]()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(steps):
            status[0] |= calls[i % len(calls)]()
        e1.record(st)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / steps

    jac_us = timed([dyn.bind_F_dF_device(Zs[i & 1], Fb[i % nb], Jb[i % nb], st) for i in range(2 * nb)])
    hess_us = timed([dyn.bind_mu_d2F_device(Zs[i & 1], mu, Hb[i % nh], st) for i in range(2 * nh)])
    both_us = timed([dyn.bind_F_dF_mu_d2F_device(Zs[i & 1], mu, Fb[i % nb], Jb[i % nb], Hb[i % nh], st) for i in range(int(np.lcm(nb, nh)) * 2)])
    assert status[0] == 0, "a device-resident launch of the T = 8000 record reported an error"
    zdim, ddim = inp.traj.dim, int(dims.ddim)
    jac_bytes = 8 * (zdim * (n_int + 1) + (ddim + int(dims.jac_nnz_interval)) * n_int)
    hess_bytes = 8 * (zdim * (n_int + 1) + (ddim + structural_hess_nnz(dyn)) * n_int)     # structural entries: the padding of the line-aligned layout is not algorithmic
    both_bytes = jac_bytes + 8 * (ddim + structural_hess_nnz(dyn)) * n_int
    frac = lambda b, us: b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS   # noqa: E731
    rec = {"workload": f"{qc.CONFIGS[4].description}; T={T} knots on ONE GPU", "kernels": list(dyn.kernel_names),
           "F_dF_us": jac_us, "hbm_frac": frac(jac_bytes, jac_us), "algorithmic_bytes_per_launch": jac_bytes,
           "evals_per_s_T1000_equivalent": n_int / (T_PER_GPU - 1) / (jac_us * 1e-6),
           "hess_us": hess_us, "hess_hbm_frac": frac(hess_bytes, hess_us),
           "F_dF_hess_one_call_us": both_us, "F_dF_hess_one_call_hbm_frac": frac(both_bytes, both_us), "F_dF_hess_kernel": dyn.fused_kernel_name}
    dyn.close()
    return rec


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher (the shape of the driver's N = 1 command): start
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same arguments>`
    as a CHILD process, pass its output through (rank 0 prints the one JSON line) and return its exit code.  Nothing in this process
    has touched the GPU at this point, and it never replaces itself with another program: it waits for the child."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env)
    try:
        return child.wait()
    except KeyboardInterrupt:
        child.terminate()
        return child.wait()


def main():
    global SIDE_STEPS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--side-steps", type=int, default=SIDE_STEPS, help="launches per timed side leg (2000-launch windows read the same: profiles/NOTES.md)")
    ap.add_argument("--prewarm-seconds", type=float, default=0.25,
                    help="untimed launches before the W warm-up steps (output ring touched, clocks up); 0: one pass over the ring only")
    ap.add_argument("--config", type=int, default=3, help="BASELINE config id (3 = metric workload)")
    ap.add_argument("--T", type=int, default=0, help="knots per GPU (default: the config's own T)")
    ap.add_argument("--kernel", default="auto", choices=["auto", "lds", "mfma"])
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--hessian", action=argparse.BooleanOptionalAction, default=True,
                    help="also time mu_d2F and F alone and report the ms/Ipopt-iter proxy of the metric (extra fields)")
    ap.add_argument("--allgather", dest="allgather", action="store_true", default=True,
                    help="at N > 1 also time the RCCL all-gather of the value blocks (default; reported beside the metric, never part of it)")
    ap.add_argument("--no-allgather", dest="allgather", action="store_false")
    ap.add_argument("--streams", type=int, default=0, help="also time the same steps issued round-robin on S streams "
                    "(independent evaluations overlapping their launch/drain phases; reported as an extra field, never `value`)")
    ap.add_argument("--host-visible", action=argparse.BooleanOptionalAction, default=True,
                    help="also time the host-buffer entry points (PCIe-inclusive; the `host_visible` object, never `value`)")
    ap.add_argument("--config4", action=argparse.BooleanOptionalAction, default=True,
                    help="at N = 1 also run BASELINE config 4's T = 8000 trajectory on the one device (`config4_one_gpu` object)")
    ap.add_argument("--exponential", action=argparse.BooleanOptionalAction, default=True,
                    help="at N = 1 also time the metric workload with the exponential integrator (`exponential_integrator` object)")
    ap.add_argument("--config5", action=argparse.BooleanOptionalAction, default=True,
                    help="at N = 1 also run BASELINE config 5 on the device and report its HBM / MFMA fractions (`config5` object)")
    args = ap.parse_args()
    SIDE_STEPS = max(1, args.side_steps)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))      # started without a launcher: this process becomes the parent of N ranks (no GPU call so far)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU path)"
    # QC_BENCH_BACKEND=gloo lets several ranks share one GPU: a functional check of the sharded path on a 1-GPU box
    # (RCCL refuses two ranks on one device); the driver's multi-GPU runs use the default, nccl = RCCL.
    backend = os.environ.get("QC_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    qc = g.load_package()
    # The CPU baseline is a property of the T = 1000 workload and of this host, not of N: rank 0 times it at every N, BEFORE the
    # ranks meet (the others wait in the rendezvous below, asleep), so that the N > 1 lines carry the same reference as N = 1.
    cpu_rec = None
    if rank == 0 and args.cpu_seconds > 0 and args.config in (3, 4):
        cpu_rec = cpu_baseline(qc, qc.config_inputs(3, T=T_PER_GPU), args.cpu_seconds)
    elif rank == 0 and args.cpu_seconds > 0 and world == 1:
        cpu_rec = cpu_baseline(qc, qc.config_inputs(args.config, T=args.T or None), args.cpu_seconds)
    host_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
            # Waits that last seconds (the other ranks idle while rank 0 times the multi-device handle ON THEIR GPUS) go through a
            # host-side gloo group: a barrier on the RCCL backend is a kernel spinning on every waiting rank's device, exactly the
            # devices being measured.
            host_group = dist.new_group(backend="gloo")
        else:
            dist.init_process_group(backend)

    from qcolloc_amd.sharding import ShardedDynamics

    spec = qc.CONFIGS[args.config]
    t_per_gpu = args.T or (T_PER_GPU if args.config in (3, 4) else spec.T)
    T_total = t_per_gpu * world
    inp = qc.config_inputs(args.config, T=T_total)
    sd = ShardedDynamics(inp.integrators, inp.traj, rank, world, device=dev_index, kernel=args.kernel, hess_align=DEVICE_HESS_ALIGN)
    dyn = sd.local
    dims = dyn.dims
    n_int = int(dims.n_intervals)
    zdim, ddim, nnz = inp.traj.dim, int(dims.ddim), int(dims.jac_nnz_interval)
    bytes_per_launch = 8 * (zdim * (n_int + 1) + ddim * n_int + nnz * n_int)  # each knot once; DESIGN.md

    # inputs resident in HBM before the timed region; a few distinct Z so steps are not identical
    Zh = inp.traj.datavec
    rng = np.random.default_rng(1)
    Zs = [torch.from_numpy(Zh + (1e-3 * rng.standard_normal(Zh.size) if k else 0.0)).to(dev) for k in range(4)]
    out_bytes = 8 * (ddim + nnz) * n_int
    nbuf = max(2, -(-RING_BYTES // out_bytes))
    Fb = [torch.empty(int(dims.F_len), dtype=torch.float64, device=dev) for _ in range(nbuf)]
    Jb = [torch.empty(int(dims.jac_nnz), dtype=torch.float64, device=dev) for _ in range(nbuf)]
    stream = torch.cuda.current_stream(dev)

    # one pre-bound launcher per (Z variant, ring slot): the timed loop is then ctypes call + hipLaunchKernel
    period = int(nbuf * 4 // np.gcd(nbuf, 4))
    launch = [dyn.bind_F_dF_device(Zs[i & 3], Fb[i % nbuf], Jb[i % nbuf], stream) for i in range(period)]
    status = [0]

    def step(i):
        status[0] |= launch[i % period]()

    node_barrier = NodeBarrier(rank, world) if world > 1 else None

    def barrier():
        if world > 1:
            node_barrier.wait()

    # Untimed pre-warm, IN ADDITION to the W warm-up steps: every slot of the output ring is written at least once (a first write
    # to fresh device memory pays for its page-table entries: with W = 5 and K = 20 twelve of the timed steps would be first
    # writes) and the device has ~0.25 s to reach its clocks.  Reported as `prewarm_steps`; never part of the timed region.
    prewarm = 0
    t_pre = time.perf_counter()
    while prewarm < period or time.perf_counter() - t_pre < args.prewarm_seconds:
        for i in range(period):
            step(i)
        prewarm += period
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)      # (torch creates the underlying events at their first record: not inside the timed region)
    ev1.record(stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    ev0.record(stream)      # (the opening marker of the stream-event timing goes out before the clock starts: it brackets the same
    t0 = time.perf_counter()  # K launches, and its 3 - 4 us of host time do not delay the first of them inside the wall-clock region)
    ta = time.perf_counter()
    for i in range(args.steps):
        step(i)
    tb = time.perf_counter()
    ev1.record(stream)
    if os.environ.get("QC_BENCH_POLL"):   # rounds 3 - 5: the launching thread polled the closing event before the synchronize.  Measured in
        while not ev1.query():            # round 6 (profiles/sync_cost_probe.py): a synchronize BEHIND a polled event takes 18 us on an idle
            pass                          # device (3.7 us otherwise) -- 20 steps 213 us polled, 194 us with the plain synchronize
    tc = time.perf_counter()
    torch.cuda.synchronize()
    td = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("QC_BENCH_TRACE"):
        print(f"timed region: first record +{(ta - t0) * 1e6:.1f} us, launches issued +{(tb - t0) * 1e6:.1f}, last step seen done "
              f"+{(tc - t0) * 1e6:.1f}, synchronize returned +{(td - t0) * 1e6:.1f}, end +{elapsed * 1e6:.1f}", file=sys.stderr)
    assert status[0] == 0, "qc_eval_F_jac_dev reported an error during the timed loop"
    stream_ms = ev0.elapsed_time(ev1)
    stream_ms_max = stream_ms
    if world > 1:
        tt = torch.tensor([elapsed, stream_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, stream_ms_max = float(tt[0].item()), float(tt[1].item())

    # kernel duration: HIP event pair around each launch, on the launch stream, same steps
    kn = min(args.steps, 400)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(kn)]
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(pairs):
        a.record(stream)
        step(i)
        b.record(stream)
    torch.cuda.synchronize()
    pair_us = np.array([a.elapsed_time(b) for a, b in pairs]) * 1e3
    kernel_us_pairs = float(np.median(pair_us))
    kernel_us_stream = stream_ms * 1e3 / args.steps   # includes the inter-kernel boundary

    extra = {}
    # What a device-resident consumer that reuses ONE output vector sees (the 43 MB stay in the Infinity Cache): informational,
    # never `value` -- the timed region above cycles a ring of output buffers precisely so that no step is absorbed by the cache.
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    one = dyn.bind_F_dF_device(Zs[0], Fb[0], Jb[0], stream)
    for _ in range(50):
        one()
    torch.cuda.synchronize()
    s0.record(stream)
    for _ in range(SIDE_STEPS):
        one()
    s1.record(stream)
    torch.cuda.synchronize()
    extra["step_us_one_output_buffer"] = s0.elapsed_time(s1) * 1e3 / SIDE_STEPS
    if args.hessian and dims.hess_nnz:
        mu = torch.from_numpy(rng.standard_normal(int(dims.n_rows))).to(dev)
        # a ring of value vectors beyond 2 x the Infinity Cache, like the F + dF outputs (one vector would stay cache-resident)
        nh = max(2, -(-RING_BYTES // (8 * int(dims.hess_nnz))))
        Hbs = [torch.empty(int(dims.hess_nnz), dtype=torch.float64, device=dev) for _ in range(nh)]
        # pre-bound launchers (as for the metric's steps): a Python method call with its argument checks per launch takes as long
        # as these kernels do, and the stream events would then time the interpreter
        hp = int(nh * 4 // np.gcd(nh, 4))
        hl = [dyn.bind_mu_d2F_device(Zs[i & 3], mu, Hbs[i % nh], stream) for i in range(hp)]
        fl = [dyn.bind_F_dF_device(Zs[i & 3], Fb[0], None, stream) for i in range(4)]
        # (the residual-only leg FIRST, right behind the F + dF launches above, and behind 50 untimed launches only: a long run of these 4 us
        #  launches -- or of the lighter mu_d2F launches in front of them -- lets the device clock down, 5.1 - 5.8 us instead of 4.3 - 4.9;
        #  a line search's residual calls sit between heavier ones)
        for i in range(50):
            status[0] |= fl[i & 3]()
        torch.cuda.synchronize()
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        f0.record(stream)
        for i in range(SIDE_STEPS):
            status[0] |= fl[i & 3]()
        f1.record(stream)
        torch.cuda.synchronize()
        F_us = f0.elapsed_time(f1) * 1e3 / SIDE_STEPS
        for i in range(SIDE_WARM):
            status[0] |= hl[i % hp]()
        torch.cuda.synchronize()
        h0, h1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0.record(stream)
        for i in range(SIDE_STEPS):
            status[0] |= hl[i % hp]()
        h1.record(stream)
        torch.cuda.synchronize()
        hess_us = h0.elapsed_time(h1) * 1e3 / SIDE_STEPS
        assert status[0] == 0, "a device-resident launch reported an error"
        # F + dF + mu_d2F of one accepted point in ONE call (qc_eval_F_jac_hess_dev: one launch where a fused kernel serves the
        # handle, the two launches otherwise), over the same rings of output vectors
        fp = int(np.lcm(np.lcm(nbuf, nh), 4))
        ful = [dyn.bind_F_dF_mu_d2F_device(Zs[i & 3], mu, Fb[i % nbuf], Jb[i % nbuf], Hbs[i % nh], stream) for i in range(min(fp, 4 * max(nbuf, nh)))]
        for i in range(SIDE_WARM):
            status[0] |= ful[i % len(ful)]()
        torch.cuda.synchronize()
        u0, u1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        u0.record(stream)
        for i in range(SIDE_STEPS):
            status[0] |= ful[i % len(ful)]()
        u1.record(stream)
        torch.cuda.synchronize()
        fused_us = u0.elapsed_time(u1) * 1e3 / SIDE_STEPS
        assert status[0] == 0, "qc_eval_F_jac_hess_dev reported an error"
        del Hbs, ful, hl
        extra["hess_us"] = hess_us
        extra["F_only_us"] = F_us
        extra["F_dF_hess_one_call_us"] = fused_us
        extra["F_dF_hess_kernel"] = dyn.fused_kernel_name
        # Ipopt iteration proxy, in MOI's call order: eval_constraint + eval_constraint_jacobian at the accepted point
        # (F + dF), then eval_hessian_lagrangian (mu_d2F alone: mu reaches the evaluator only AFTER the Jacobian has returned), one
        # line-search F; solver algebra excluded.  `..._custom_consumer`: the one-call launch (dF + mu_d2F with mu known up front) that
        # a consumer other than MOI / Ipopt could issue -- no binding of this repository calls it.
        extra["ms_per_ipopt_iter_proxy_device"] = (kernel_us_stream + hess_us + F_us) / 1e3
        extra["ms_per_ipopt_iter_proxy_device_custom_consumer"] = (fused_us + F_us) / 1e3
    if args.streams > 1:
        # Independent evaluations (e.g. the systems of a sampling problem, or several line-search points) may overlap the
        # ~4 us of dispatch + write-back of one launch with the store phase of the next.  A serial Ipopt loop cannot.
        # qcolloc.h: a handle has ONE evaluation in flight unless its kernel keeps no scratch in the handle.
        if not (dyn.kernel_names[0].startswith(("mfma16", "mfma32")) or dyn.kernel_names[0] == "lds"):
            raise SystemExit(f"--streams: the F+dF kernel of this handle ({dyn.kernel_names[0]}) keeps scratch in the handle; "
                             "launches of one handle on several streams would race")
        sts = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
        ln =[dyn.bind_F_dF_device(Zs[i & 3], Fb[i % nbuf], Jb[i % nbuf], sts[i % args.streams]) for i in range(period * args.streams)]
        for i in range(args.warmup):
            ln[i % len(ln)]()
        torch.cuda.synchronize()
        p0 = time.perf_counter()
        for i in range(args.steps):
            ln[i % len(ln)]()
        torch.cuda.synchronize()
        extra["pipelined_streams"] = args.streams
        extra["pipelined_ms_per_step"] = (time.perf_counter() - p0) / args.steps * 1e3
    rccl_ranks = None
    if world > 1:
        # self-verification of the N > 1 run: the number of ranks as the collective library itself counts them
        ones = torch.ones(1, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_ranks = int(round(float(ones.item())))
    if args.allgather and world > 1:
        # SURVEY 8(e) "report both": the device-resident full Jacobian on every GPU (RCCL all-gather over xGMI), timed
        # outside the metric's region.  A failure here must not lose the metric line.
        try:
            gdev = dev if backend == "nccl" else torch.device("cpu")
            pad = torch.zeros(sd.padded_len(nnz), dtype=torch.float64, device=gdev)
            pad[:Jb[0].numel()].copy_(Jb[0])
            full = torch.empty(world * pad.numel(), dtype=torch.float64, device=gdev)
            for _ in range(3):
                sd.all_gather_values(pad, nnz, out=full)
            torch.cuda.synchronize()
            barrier()
            a0 = time.perf_counter()
            for _ in range(10):
                sd.all_gather_values(pad, nnz, out=full)
            torch.cuda.synchronize()
            ag = torch.tensor([(time.perf_counter() - a0) / 10 * 1e3], dtype=torch.float64, device=gdev)
            dist.all_reduce(ag, op=dist.ReduceOp.MAX)
            extra["allgather_ms"] = float(ag.item())
            extra["allgather_GB_per_gpu_received"] = (world - 1) * pad.numel() * 8 / 1e9
            # against SURVEY 8(e)'s figure: a full xGMI mesh, every GPU receiving from its N - 1 peers on separate ~153 GB/s links
            extra["allgather_GBps_per_gpu"] = extra["allgather_GB_per_gpu_received"] / (extra["allgather_ms"] * 1e-3)
            extra["xgmi_GBps_per_gpu_peak"] = XGMI_LINK_GBS * min(world - 1, 7)
            extra["allgather_frac_of_xgmi"] = extra["allgather_GBps_per_gpu"] / extra["xgmi_GBps_per_gpu_peak"] if backend == "nccl" else None
        except Exception as exc:   # noqa: BLE001
            extra["allgather_error"] = repr(exc)[:200]

    total_intervals = (T_total - 1)
    t1000_equiv = total_intervals / (T_PER_GPU - 1) if args.config in (3, 4) else float(world)
    host_rec = None
    Zhs = [Zh + (1e-3 * k) * rng.standard_normal(Zh.size) for k in range(3)]     # the host-buffer calls get a different vector each time
    if args.host_visible:
        try:
            if world == 1:
                # a handle of its own in the bindings' default layout: mu_d2F_structure is the reference's, entry for entry
                hdyn = qc.QuantumDynamics(inp.integrators, inp.traj, device=dev_index, kernel=args.kernel, result_ring=3)
                host_rec = host_visible_record(qc, inp, hdyn, Zhs, cpu_rec, t1000_equiv)
                host_rec["devices"] = [dev_index]
                host_rec["hess_nnz_interval"] = int(hdyn.dims.hess_nnz_interval)
                hdyn.close()
                if args.config in (3, 4):
                    try:
                        host_rec["integrator_list"] = integrator_list_record(qc, cpu_rec)
                    except Exception as exc:   # noqa: BLE001
                        host_rec["integrator_list"] = {"error": repr(exc)[:300]}
            else:
                # The reference's consumer is ONE process: rank 0 builds the in-library multi-device handle
                # (qc_create_multi over all N GPUs) and times the same host-buffer calls on the whole T = 1000 N
                # trajectory while the other ranks wait at the barrier; N PCIe links work in parallel.
                torch.cuda.synchronize()            # this rank's GPU is idle ...
                dist.barrier(group=host_group)      # ... and stays idle: the wait is in a socket, not in a kernel (host_group above)
                if rank == 0:
                    try:   # (whatever happens here, rank 0 reaches the barrier the other ranks are waiting at)
                        devs = list(range(world)) if backend == "nccl" else [r % torch.cuda.device_count() for r in range(world)]
                        md = qc.QuantumDynamics(inp.integrators, inp.traj, devices=devs, kernel=args.kernel, result_ring=3)
                        host_rec = host_visible_record(qc, inp, md, Zhs, cpu_rec, t1000_equiv)
                        host_rec["devices"] = devs
                        md.close()
                    except Exception as exc:   # noqa: BLE001
                        host_rec = {"error": repr(exc)[:300]}
                dist.barrier(group=host_group)
        except Exception as exc:   # noqa: BLE001  (must not lose the metric line)
            host_rec = {"error": repr(exc)[:300]}
    c5 = None
    if args.config5 and world == 1 and rank == 0 and args.config in (3, 4):
        try:
            c5 = config5_record(qc, dev_index)
        except Exception as exc:   # noqa: BLE001
            c5 = {"error": repr(exc)[:300]}

    cx = None
    if args.exponential and world == 1 and rank == 0 and args.config in (3, 4):
        try:
            cx = exponential_record(qc, dev_index)
        except Exception as exc:   # noqa: BLE001
            cx = {"error": repr(exc)[:300]}

    c4 = None
    if args.config4 and world == 1 and rank == 0 and args.config in (3, 4) and t_per_gpu == T_PER_GPU:
        try:
            c4 = long_trajectory_record(qc, dev_index)
        except Exception as exc:   # noqa: BLE001
            c4 = {"error": repr(exc)[:300]}

    value = args.steps * t1000_equiv / elapsed
    if rank == 0:
        traffic, traffic_note = recorded_traffic(dyn.kernel, args.config, t_per_gpu)
        # kernel duration: HIP events around the timed region on the launch stream / steps (includes the
        # ~1.5 us inter-kernel boundary, so it is an upper bound of rocprofv3's per-kernel average)
        achieved = bytes_per_launch / (kernel_us_stream * 1e-6) / 1e9
        line = {
            "metric": "full-trajectory constraint+Jacobian evals/s, 3-qubit T=1000" if args.config in (3, 4)
                      else f"full-trajectory constraint+Jacobian evals/s, {spec.name}",
            "value": value,
            "unit": "evals/s (T=1000-equivalent)" if args.config in (3, 4) else "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_steps": int(prewarm),
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": f"synthetic (seed 20250218: I -> {spec.gate} geodesic + N(0,1e-2) noise, a~U(-1,1), da/dda~N(0,0.1^2), dt=0.2)",
            "config": {"workload": f"{spec.description}; T={T_total} knots ({t_per_gpu}/GPU), Pade order 4, free dt",
                       "T": T_total, "N": inp.system.levels, "n_drives": inp.system.n_drives,
                       "kernel": dyn.kernel, "parallelism": f"knot-shard x{world}", "output_ring_buffers": nbuf},
            "knot_evals_per_s": args.steps * total_intervals / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # `frac` follows from the stream-event step time (the K kernels back to back on the launch stream);
                         # `frac_wall` from this line's own ms_per_step (the contract's wall clock: launch latency of the first
                         # step and the closing synchronize included), so that the line can be checked against itself
                         "frac_wall": bytes_per_launch / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "kernel_us_event_pairs": kernel_us_pairs, "step_us_stream_events": kernel_us_stream},
        }
        if world > 1:
            # whole job against N devices' HBM: every rank's launch moves the same algorithmic bytes (equal shards); the slowest
            # rank's stream-event step time is the job's
            agg = world * bytes_per_launch / (stream_ms_max * 1e3 / args.steps * 1e-6) / 1e9
            line["roofline"].update({"scope": "rank 0's shard (achieved / frac); aggregate_*: all ranks against N x peak",
                                     "aggregate_achieved": agg, "aggregate_peak": world * HBM_PEAK_GBS,
                                     "aggregate_frac": agg / (world * HBM_PEAK_GBS),
                                     "step_us_stream_events_max_over_ranks": stream_ms_max * 1e3 / args.steps})
        # rank 0's wall-clock region taken apart (microseconds after the opening synchronize): the K launch calls, the last step's
        # completion as the event poll saw it, the return of the contract's closing torch.cuda.synchronize().  At small K the first
        # launch's dispatch latency and that synchronize (a marker round trip although the work is done) are a visible share of the
        # region (profiles/r04_sync_cost.txt); `roofline.step_us_stream_events` is the same K steps without them.
        line["timed_region_us"] = {"launches_issued": (tb - t0) * 1e6, "last_step_seen_done": (tc - t0) * 1e6,
                                   "synchronize_returned": (td - t0) * 1e6, "per_step_stream_events": kernel_us_stream}
        line.update(extra)
        line["value_device_resident_evals_per_s"] = value
        if rccl_ranks is not None:
            line["rccl_ranks"] = rccl_ranks
            line["collective_backend"] = backend
            line["timing_barrier"] = node_barrier.kind
        if host_rec is not None:
            line["host_visible"] = host_rec
        if c4 is not None:
            line["config4_one_gpu"] = c4
        if c5 is not None:
            line["config5"] = c5
        if cx is not None:
            line["exponential_integrator"] = cx
        if cpu_rec is not None:
            line["cpu_baseline"] = cpu_rec
            if "ms_per_ipopt_iter" in cpu_rec and "ms_per_ipopt_iter_proxy_device" in line:
                # both halves of BASELINE's metric beside their CPU figure (T = 1000-equivalents at N > 1)
                line["ipopt_iter_speedup_device_vs_cpu_baseline"] = cpu_rec["ms_per_ipopt_iter"] * t1000_equiv / line["ms_per_ipopt_iter_proxy_device"]
            line["speedup_vs_cpu_baseline"] = value / cpu_rec["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        node_barrier.close()
        dist.barrier()
        dist.destroy_process_group()
    dyn.close()


if __name__ == "__main__":
    main()
