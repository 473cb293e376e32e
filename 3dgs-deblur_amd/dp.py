"""Data-parallel sharding of the per-iteration camera batch.

The reference is single-GPU (/root/reference/train.py:114-122 passes no device /
world-size flags; /root/reference/README.md:7 "a recent NVidia RTX GPU"), so this is
new design mandated by north_star: one process per GPU, Gaussians replicated, the
camera batch partitioned across ranks, ONE collective per step — a sum all-reduce of
the flattened Gaussian gradients (59 fp32 per Gaussian at SH degree 3) over RCCL/xGMI.
Works on any torch.distributed backend ("nccl" == RCCL on ROCm; "gloo" in CPU tests).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_views(num_views: int, rank: int, world_size: int) -> List[int]:
    """Indices of the camera batch rendered by `rank` (round-robin, deterministic, covers each view once)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    return list(range(rank, num_views, world_size))


def flatten_grads(params: Sequence[torch.Tensor]) -> torch.Tensor:
    """One contiguous fp32 bucket holding every gradient (zeros where a grad is None)."""
    total = sum(p.numel() for p in params)
    ref = params[0]
    flat = torch.zeros(total, dtype=torch.float32, device=ref.device)
    off = 0
    for p in params:
        n = p.numel()
        if p.grad is not None:
            flat[off:off + n].copy_(p.grad.reshape(-1))
        off += n
    return flat


def unflatten_grads(flat: torch.Tensor, params: Sequence[torch.Tensor]) -> None:
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n


def allreduce_gradients(params: Iterable[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                        mode: str = "allreduce", average: bool = False) -> None:
    """Sum (or average) the gradients of `params` over all ranks with a single bucket.

    mode="allreduce": one dist.all_reduce (ring on RCCL, bound by one xGMI link).
    mode="rs_ag":     reduce_scatter + all_gather of the same bucket — every rank exchanges
                      1/world of the bucket with every peer, using all xGMI links at once
                      (SURVEY.md §5 estimates ~6x less time than the ring at 2M Gaussians).
    """
    params = [p for p in params if p.requires_grad]
    if not params or not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    flat = flatten_grads(params)
    if mode == "allreduce":
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    elif mode == "rs_ag":
        n = flat.numel()
        pad = (-n) % world
        if pad:
            flat = torch.cat([flat, flat.new_zeros(pad)])
        chunk = flat.numel() // world
        mine = torch.empty(chunk, dtype=flat.dtype, device=flat.device)
        if dist.get_backend(group) == "gloo":
            # gloo has no reduce_scatter: same result through all_reduce (CPU tests only)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM, group=group)
            dist.all_gather_into_tensor(flat, mine, group=group)
        flat = flat[:n]
    else:
        raise ValueError(f"unknown mode {mode!r}")
    if average:
        flat.div_(world)
    unflatten_grads(flat, params)
