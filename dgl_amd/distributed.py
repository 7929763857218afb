"""Instance sharding across the GPUs of a node.

Instances are independent on the whole decode path (every kernel works on rows of one instance;
beams / multi-starts of an instance stay together), so ranks take contiguous blocks of instances
with no data-path collective.  The only exchange is the evaluation statistics: one all-reduce
(sum) of [sum of tour lengths, sum of gaps, instance count] in float64 at the end, which
reproduces the reference's AverageMeter over the whole test set (TSP/TSPTester.py:123-139).
One process per GPU; backend "nccl" (RCCL over xGMI) on GPUs, "gloo" in CPU tests.
"""
from __future__ import annotations

from typing import Tuple

import torch


def world() -> Tuple[int, int]:
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of instances for `rank`: ceil(total / world) per rank, the tail
    ranks may get fewer (or none)."""
    per = (total + world_size - 1) // world_size
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def reduce_stats(sum_len: float, sum_gap: float, count: float, device=None) -> Tuple[float, float, float]:
    """All-reduce (sum) of the three evaluation statistics; identity when not distributed."""
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return sum_len, sum_gap, count
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([sum_len, sum_gap, count], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t[0]), float(t[1]), float(t[2])


def all_reduce_gradients(flat: torch.Tensor) -> torch.Tensor:
    """Sum of the flat gradient vector over the ranks, in place (identity when not distributed).
    Every rank scales its loss by 1 / (rows of the GLOBAL batch), so the sum is the gradient of the
    reference's loss.mean() over the whole batch (CVRP/CVRPTrainer.py:269-272); it goes between
    backward and optimizer.step, once per decode step.  856 961 floats for CVRP: one small message,
    latency-bound on xGMI."""
    import torch.distributed as dist
    if world()[1] > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat
