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
    mode="sparse":    early termination leaves all but a few percent of the Gaussians without any
                      gradient for a given view, so each rank all-gathers only its non-zero gradient
                      ROWS (index + 59 floats) and every rank scatter-adds the union: ~13 MB per rank
                      instead of a 236 MB dense bucket at 1M Gaussians.  Falls back to the dense
                      all-reduce (same decision on every rank) when the rows are not sparse.
    """
    params = [p for p in params if p.requires_grad]
    if not params or not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    if mode == "sparse":
        if _allreduce_sparse_rows(params, group, world, average):
            return
        mode = "allreduce"
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


class _RowOps:
    """Row mask / pack / scatter-add over the gradient tensors.  CUDA tensors go through the HIP kernels of
    dp_exchange.hip (one launch per step for all tensors; raises if the library is missing); CPU tensors —
    the gloo tests of the exchange logic — through the equivalent torch ops."""

    def __init__(self, grads: Sequence[torch.Tensor]):
        self.grads = grads
        self.N = grads[0].shape[0]
        self.widths = [g[0].numel() if self.N else 1 for g in grads]
        self.wtot = sum(self.widths)
        self.dev = grads[0].device
        self.hip = self.dev.type == "cuda"
        if self.hip:
            import ctypes
            from . import _lib
            self._L = _lib.load()
            self._check = _lib.check
            n = len(grads)
            self._ptrs = (ctypes.c_void_p * n)(*[g.data_ptr() for g in grads])
            self._w = (ctypes.c_int * n)(*self.widths)
            self._stream = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
            self._vp = ctypes.c_void_p

    def row_mask(self) -> torch.Tensor:
        if self.hip:
            mask = torch.empty(self.N, dtype=torch.uint8, device=self.dev)
            self._check(self._L.gs_dp_row_mask(self.N, len(self.grads), self._ptrs, self._w,
                                               self._vp(mask.data_ptr()), self._stream), "dp_row_mask")
            return mask.bool()
        mask = torch.zeros(self.N, dtype=torch.bool, device=self.dev)
        for g in self.grads:
            mask |= (g.reshape(self.N, -1) != 0).any(dim=1)
        return mask

    def pack(self, idx: torch.Tensor, rows_padded: int) -> torch.Tensor:
        """[rows_padded, wtot+1] payload: the rows idx (int64) and, in the last column, their indices as int32
        bit patterns; rows beyond len(idx) are zero padding."""
        M = idx.numel()
        pay = torch.zeros(rows_padded, self.wtot + 1, dtype=torch.float32, device=self.dev)
        if M == 0:
            return pay
        if self.hip:
            self._check(self._L.gs_dp_pack_rows(M, self._vp(idx.data_ptr()), len(self.grads), self._ptrs, self._w,
                                                self._vp(pay.data_ptr()), self._stream), "dp_pack_rows")
        else:
            pay[:M, :self.wtot] = torch.cat([g.reshape(self.N, -1)[idx] for g in self.grads], dim=1)
            pay[:M, self.wtot] = idx.to(torch.int32).view(torch.float32)
        return pay

    def scatter_add(self, payload: torch.Tensor, M: int, scale: float) -> None:
        """grads[row] += scale * payload[:M]; the M row indices of one payload are unique."""
        if M == 0:
            return
        if self.hip:
            self._check(self._L.gs_dp_scatter_add_rows(M, self._vp(payload.data_ptr()), len(self.grads), self._ptrs,
                                                       self._w, float(scale), self._stream), "dp_scatter_add_rows")
            return
        rows = payload[:M, self.wtot].contiguous().view(torch.int32).to(torch.int64)
        off = 0
        for g, w in zip(self.grads, self.widths):
            g.reshape(self.N, -1).index_add_(0, rows, payload[:M, off:off + w] * scale)
            off += w


def _allreduce_sparse_rows(params: Sequence[torch.Tensor], group, world: int, average: bool,
                           dense_threshold: float = 0.25) -> bool:
    """Row-sparse gradient exchange.  All params must share the leading (per-Gaussian) dimension.
    Returns False (nothing changed) when the caller should use the dense path instead."""
    N = params[0].shape[0]
    if N == 0 or any(p.shape[0] != N for p in params):
        return False
    for p in params:            # a rank without a gradient for some tensor contributes zeros
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        elif not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
            p.grad = p.grad.contiguous().float()
    ops = _RowOps([p.grad for p in params])
    idx = ops.row_mask().nonzero(as_tuple=False).reshape(-1)
    m_local = torch.tensor([idx.numel()], dtype=torch.int64, device=ops.dev)
    counts_t = [torch.empty_like(m_local) for _ in range(world)]
    dist.all_gather(counts_t, m_local, group=group)
    counts = [int(c) for c in torch.stack(counts_t).reshape(-1).cpu()]
    Mmax = max(counts)
    if Mmax * world > dense_threshold * N:        # identical on every rank: the counts are global
        return False
    if Mmax == 0:
        return True
    # one collective: a payload row = the 59 gradient floats + the row index as a float32 bit pattern
    pay = ops.pack(idx, Mmax)
    pay_all = [torch.empty_like(pay) for _ in range(world)]
    dist.all_gather(pay_all, pay, group=group)
    # Replicas must stay BIT-identical, so the sum has one fixed order on every rank: start from zero, add
    # rank 0's rows, then rank 1's, ...  A rank's row indices are unique, so no add collides (a single
    # scatter-add over the concatenation would add in atomic, i.e. arbitrary, order).
    for p in params:
        p.grad.zero_()
    scale = 1.0 / world if average else 1.0
    for r in range(world):
        ops.scatter_add(pay_all[r], counts[r], scale)
    return True
