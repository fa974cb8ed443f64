"""Knot sharding across GPUs, process-per-GPU flavour (bench.py under torchrun; a consumer that lives in ONE process --
Julia + Ipopt -- uses the in-library form instead: `QuantumDynamics(..., devices=[...])` = qc_create_multi).  One process per GPU, each owning a contiguous range of the T-1
intervals (plus the halo knot z_{t_end}, read-only).  Intervals are independent given Z (the
constraint couples only z_t and z_{t+1}: reference unitary_smooth_pulse_problem.jl:14-16), so there
is NO collective on the data path.  `all_gather_values` (RCCL all-gather over xGMI when the backend
is "nccl") exists for consumers that want the whole value vector resident on every GPU, as
BASELINE.json's north_star describes; a CPU Ipopt consumer instead copies each rank's contiguous
slice straight to the host (values are knot-major, so shards are contiguous).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_chunk(n_intervals: int, world: int) -> int:
    return -(-n_intervals // world)


def knot_shards(T: int, world: int) -> List[Tuple[int, int]]:
    """Interval ranges [t0, t1) per rank; equal chunks of ceil((T-1)/world), the tail ranks short
    (possibly empty).  Rank r's values start at r * chunk * nnz in the global value vector."""
    n = T - 1
    c = shard_chunk(n, world)
    return [(min(r * c, n), min((r + 1) * c, n)) for r in range(world)]


class ShardedDynamics:
    """Rank-local view of a QuantumDynamics over the whole trajectory.

    `make_local(t0, t1)` builds the rank's evaluator; the default builds the HIP handle.  Tests inject
    a CPU evaluator here to exercise the sharding logic on gloo without a GPU; product code never does.
    """

    def __init__(self, integrators, traj, rank: int, world: int, device: int = 0,
                 make_local: Optional[Callable] = None, kernel: str = "auto", hess_align: int = 0):
        self.rank, self.world = rank, world
        self.T = traj.T
        self.shards = knot_shards(traj.T, world)
        self.t0, self.t1 = self.shards[rank]
        self.chunk = shard_chunk(traj.T - 1, world)
        if make_local is None:
            from .dynamics import QuantumDynamics

            def make_local(t0, t1):
                return QuantumDynamics(integrators, traj, device=device, kernel=kernel, t_range=(t0, t1), hess_align=hess_align)
        # an empty tail shard (more ranks than intervals) gets a zero-interval evaluator: t_begin = t_end = T-1 > 0 is a valid
        # range, its dims are all zero-length and its launches are no-ops, so every rank runs the same code path
        self.empty = self.t1 == self.t0
        self.local = make_local(self.t0, self.t1)

    @property
    def n_local(self) -> int:
        return self.t1 - self.t0

    def padded_len(self, per_interval: int) -> int:
        return self.chunk * per_interval

    def all_gather_values(self, local_padded: torch.Tensor, per_interval: int, out: Optional[torch.Tensor] = None,
                          group=None) -> torch.Tensor:
        """All-gather rank-local value blocks (each padded to chunk * per_interval) into the global
        knot-major vector.  Returns a view of length (T-1) * per_interval."""
        n = self.padded_len(per_interval)
        assert local_padded.numel() == n
        if out is None:
            out = torch.empty(self.world * n, dtype=local_padded.dtype, device=local_padded.device)
        if self.world == 1:
            out[:n].copy_(local_padded)
        else:
            dist.all_gather_into_tensor(out, local_padded, group=group)
        return out[:(self.T - 1) * per_interval]
