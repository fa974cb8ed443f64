"""N > 1 path on CPU: world_size-2 gloo processes shard the knots, each evaluates its range (with the
oracle injected as the rank-local evaluator: this is a test of the sharding/gather logic, the product
evaluator is the HIP handle), all-gather, and compare with the unsharded oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, T, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as g
        from oracle_bridge import problem_from_inputs
        qc = g.load_package()
        o = g.load_oracle()
        from qcolloc_amd.sharding import ShardedDynamics, knot_shards
        inp = qc.config_inputs(1, T=T)
        prob = problem_from_inputs(inp)
        Z = inp.traj.datavec

        class OracleShard:                     # test stand-in for the rank-local HIP handle
            def __init__(self, t0, t1):
                self.t0, self.t1 = t0, t1

            def F_dF(self, Zv):
                return o.F(prob, Zv, self.t0, self.t1), o.dF(prob, Zv, self.t0, self.t1)

        sd = ShardedDynamics(inp.integrators, inp.traj, rank, world, make_local=OracleShard)
        assert sd.shards == knot_shards(T, world)
        nnz, dd = o.jac_nnz_interval(prob), prob.ddim
        Jl = torch.zeros(sd.padded_len(nnz), dtype=torch.float64)
        Fl = torch.zeros(sd.padded_len(dd), dtype=torch.float64)
        if not sd.empty:
            F, J = sd.local.F_dF(Z)
            Jl[:J.size] = torch.from_numpy(J)
            Fl[:F.size] = torch.from_numpy(F)
        Jg = sd.all_gather_values(Jl, nnz).numpy()
        Fg = sd.all_gather_values(Fl, dd).numpy()
        ok = np.array_equal(Jg, o.dF(prob, Z)) and np.array_equal(Fg, o.F(prob, Z))
        q.put((rank, bool(ok), sd.shards))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(2, 12), (2, 7), (3, 4)])
def test_sharded_all_gather_equals_full(world, T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_knot_shards_partition():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.load_package()
    from qcolloc_amd.sharding import knot_shards
    for T in (2, 3, 8, 1000, 8000, 8001):
        for w in (1, 2, 3, 4, 8):
            sh = knot_shards(T, w)
            assert sh[0][0] == 0 and sh[-1][1] == T - 1 and len(sh) == w
            assert all(a[1] == b[0] for a, b in zip(sh, sh[1:]))
            c = -(-(T - 1) // w)
            assert all(t0 == min(r * c, T - 1) for r, (t0, _) in enumerate(sh))


def _barrier_worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(world), QC_BENCH_BARRIER=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        import bench
        nb = bench.NodeBarrier(rank, world)
        path = nb.path
        order = []
        for it in range(200):                  # ranks take turns being late: nobody may leave a barrier before the late rank arrives
            if it % world == rank:
                time.sleep(0.0005)
            t_arrive = time.perf_counter()
            nb.wait()
            order.append((t_arrive, time.perf_counter()))
        arr = torch.tensor(order, dtype=torch.float64)
        allr = [torch.zeros_like(arr) for _ in range(world)]
        dist.all_gather(allr, arr)
        allr = torch.stack(allr).numpy()       # [rank, iteration, (arrive, leave)]  -- one clock: the processes share the host
        ok = bool((allr[:, :, 1].min(axis=0) >= allr[:, :, 0].max(axis=0) - 1e-6).all())
        kind = nb.kind
        nb.close()
        q.put((rank, ok, kind, os.path.exists(path)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,world", [("shm", 2), ("shm", 3), ("dist", 2)])
def test_bench_node_barrier(mode, world):
    """bench.py's timing barrier at N > 1 (a page in /dev/shm; `dist.barrier()` as the fallback): no rank leaves before the last
    one arrives, and the page is gone afterwards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_barrier_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] for r in res), res
    assert all(r[2] == ("shared-memory page (/dev/shm)" if mode == "shm" else "dist.barrier") for r in res), res
    assert not any(r[3] for r in sorted(res)[:1]), res     # rank 0 removed the page (checked after close() on rank 0)
