"""Multi-GPU mode of round 1: N independent replicas of the workload, one process per GPU.

The reference has no distributed code and its direct factorization does
not shard, so there is no data-path collective: ranks only agree on the
timing scalars (max wall time, summed Newton iterations) after a barrier.
"""
from __future__ import annotations


def aggregate(elapsed_s: float, newton_its: float, dist=None, device="cpu"):
    """(max over ranks of elapsed, sum over ranks of iterations).  `dist` is
    `torch.distributed` when initialised, else None (single process)."""
    if dist is None or not dist.is_initialized():
        return float(elapsed_s), float(newton_its)
    import torch
    tmax = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    itsum = torch.tensor([float(newton_its)], dtype=torch.float64, device=device)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(itsum, op=dist.ReduceOp.SUM)
    return float(tmax.item()), float(itsum.item())


def rate(elapsed_max_s: float, its_all: float) -> float:
    """Whole-job Newton iterations per second."""
    return its_all / elapsed_max_s if elapsed_max_s > 0 else 0.0
