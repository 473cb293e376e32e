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


# GSD_DP_FORCE=1 (or force=True): run the exchange even at world size 1 — every collective then is RCCL's own
# single-rank kernel on the device buffers, so the pack -> collective -> scatter chain, its stream ordering and the
# pinned header path execute on a box with ONE GPU (tests/test_gpu_parity.py::test_gradient_exchange_over_rccl_world1,
# bench.py --force-exchange).  The gradients must come out unchanged.
FORCE = bool(int(__import__("os").environ.get("GSD_DP_FORCE", "0")))


def allreduce_gradients(params: Iterable[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                        mode: str = "allreduce", average: bool = False, sync_free: Optional[bool] = None,
                        force: Optional[bool] = None) -> None:
    """Sum (or average) the gradients of `params` over all ranks with a single bucket.

    mode="allreduce": one dist.all_reduce (ring on RCCL, bound by one xGMI link).
    mode="rs_ag":     reduce_scatter + all_gather of the same bucket — every rank exchanges
                      1/world of the bucket with every peer, using all xGMI links at once
                      (SURVEY.md §5 estimates ~6x less time than the ring at 2M Gaussians).
    mode="sparse":    early termination leaves all but a few percent of the Gaussians without any
                      gradient for a given view, so each rank all-gathers only its non-zero gradient
                      ROWS (index + 59 floats, fixed-capacity payload sized from the previous steps' counts,
                      see SparseExchangeState) and every rank scatter-adds the union in rank order.  Falls
                      back to the dense all-reduce (same decision on every rank) when the rows are not sparse
                      or — guarded mode, the default — when a payload turned out too small for this step.
                      sync_free=True (or GSD_DP_SYNC_FREE=1) skips the per-step look at the headers.
    """
    params = [p for p in params if p.requires_grad]
    if not params or not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1 and not (FORCE if force is None else force):
        return
    if mode == "sparse":
        if _allreduce_sparse_rows(params, group, world, average, sync_free):
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

    def pack_masked(self, cap: int) -> torch.Tensor:
        """[(cap + 1), wtot + 1] payload of the (at most cap) rows that hold a gradient; row 0 is the header
        {rows with a gradient, rows packed} as int32 bit patterns.  No host synchronisation on the GPU path."""
        stride = self.wtot + 1
        mask = self.row_mask()
        pay = torch.zeros(cap + 1, stride, dtype=torch.float32, device=self.dev)
        if self.hip:
            m8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
            pos = torch.cumsum(m8, 0, dtype=torch.int32) - m8.to(torch.int32)
            idx_ws = torch.empty(cap, dtype=torch.int32, device=self.dev)
            self._check(self._L.gs_dp_pack_masked_rows(self.N, self._vp(m8.data_ptr()), self._vp(pos.data_ptr()), cap,
                                                       self._vp(idx_ws.data_ptr()), len(self.grads), self._ptrs, self._w,
                                                       self._vp(pay.data_ptr()), self._stream), "dp_pack_masked_rows")
            return pay
        idx = mask.nonzero(as_tuple=False).reshape(-1)
        total = idx.numel()
        idx = idx[:cap]
        M = idx.numel()
        hdr = torch.tensor([total, M], dtype=torch.int32).view(torch.float32)
        pay[0, :2] = hdr
        if M:
            pay[1:1 + M, :self.wtot] = torch.cat([g.reshape(self.N, -1)[idx] for g in self.grads], dim=1)
            pay[1:1 + M, self.wtot] = idx.to(torch.int32).view(torch.float32)
        return pay

    def scatter_add_payload(self, payload: torch.Tensor, cap: int, scale: float) -> None:
        """grads[row] += scale * row for the rows of a pack_masked payload (count read from its header)"""
        if self.hip:
            payload = payload.contiguous()
            self._check(self._L.gs_dp_scatter_add_payload(cap, self._vp(payload.data_ptr()), len(self.grads), self._ptrs,
                                                          self._w, float(scale), self._stream), "dp_scatter_add_payload")
            return
        M = int(payload[0, 1].view(torch.int32))
        self.scatter_add(payload[1:], M, scale)

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


class SparseExchangeState:
    """Capacity bookkeeping of the row-sparse exchange.

    Every rank packs at most `cap` rows into a fixed-size payload whose header carries the true count, ONE
    all_gather_into_tensor moves the payloads, and the scatter-add kernels read the counts from the device.  Every
    rank sees the same gathered headers, hence takes the same decisions:
      * the capacity for the next steps = 2 x the largest count seen recently, rounded up to a power of two,
      * counts so large that a dense all-reduce moves fewer bytes -> `dense` until the counts fall again,
      * a count above the capacity that was used means a truncated payload.
    Two ways of looking at the headers:
      * guarded (default): the `world` header words are read back right after the all-gather, BEFORE any gradient is
        modified; on overflow the caller runs the dense bucket in the same step (never a truncated update) and the
        capacity follows the new counts.  One small device->host copy per step.
      * sync_free: the headers travel to pinned memory asynchronously and are digested when the NEXT step starts; an
        overflow can then only be reported one step late (RuntimeError) — for runs whose row density is known not to
        jump; `notify_regime_change()` (called by the densifier after an opacity reset / refinement) makes the next
        exchange take the synchronous first look again.
    The first exchange (no history) looks at the counts synchronously and sizes its payload from them."""

    def __init__(self, N: int, world: int, cap_min: int = 1024, dense_fraction: float = 0.25):
        self.N, self.world = N, world
        self.cap_max = max(1, int(N * dense_fraction / max(1, world)))
        self.cap_min = min(cap_min, self.cap_max)
        self.cap = self.cap_max
        self.dense = False
        self.pending = None          # (host tensor, event or None, capacity used)
        self.history = []
        self.overflows = 0           # exchanges that fell back to the dense bucket because a payload was too small

    def observe(self, totals) -> None:
        """Fold the per-rank row counts of one exchange into the capacity / dense decision of the next ones."""
        self.history = (self.history + [max(int(v) for v in totals)])[-8:]
        want = 2 * max(self.history)
        if want > self.cap_max:
            self.dense = True
        else:
            self.dense = False
            cap = self.cap_min
            while cap < want:
                cap *= 2
            self.cap = min(cap, self.cap_max)

    def forget(self) -> None:
        """Drop the count history (regime change): the next exchange looks at the counts synchronously first."""
        self.pending = None
        self.history = []
        self.dense = False
        self.cap = self.cap_max

    def settle(self) -> None:
        """Digest the counts of the previous exchange (blocks only if that exchange has not finished yet)."""
        if self.pending is None:
            return
        host, ev, cap_used = self.pending
        self.pending = None
        if ev is not None:
            ev.synchronize()
        totals = [int(v) for v in host.tolist()]
        if cap_used is not None and max(totals) > cap_used:
            self.forget()
            raise RuntimeError(f"row-sparse gradient exchange overflowed: a rank had {max(totals)} rows with a gradient, "
                               f"capacity was {cap_used}; the previous step's gradients are incomplete "
                               f"(use the guarded exchange, raise cap_min / dense_fraction or mode='allreduce')")
        self.observe(totals)


# one state per (world, group); it is re-created whenever N changes (every refinement does that), so no stale
# history / capacity / pending header of an earlier N is ever reused and nothing accumulates over a long run
_SPARSE_STATES = {}
# sync-free header handling (see SparseExchangeState); default: guarded
SYNC_FREE = bool(int(__import__("os").environ.get("GSD_DP_SYNC_FREE", "0")))


def _sparse_state(N: int, world: int, group) -> SparseExchangeState:
    key = (world, id(group))
    st = _SPARSE_STATES.get(key)
    if st is None or st.N != N:
        st = _SPARSE_STATES[key] = SparseExchangeState(N, world)
    return st


def reset_sparse_exchange_state() -> None:
    _SPARSE_STATES.clear()


def notify_regime_change() -> None:
    """The row density is about to jump (opacity reset, refinement): forget the count history of every exchange
    state so that the next exchange sizes its payload from a synchronous look at the counts.  The counts of the last
    exchange are digested FIRST: in sync-free mode an overflow of that exchange (its gradients were truncated and have
    been applied by now) must still be reported — forget() alone would drop the pending header unread."""
    for st in _SPARSE_STATES.values():
        try:
            st.settle()
        finally:
            st.forget()


def _allreduce_sparse_rows(params: Sequence[torch.Tensor], group, world: int, average: bool,
                           sync_free: Optional[bool] = None) -> bool:
    """Row-sparse gradient exchange.  All params must share the leading (per-Gaussian) dimension.  Returns False
    (gradients untouched) when the caller should use the dense path — rows not sparse, or (guarded mode) a payload
    that turned out too small for this step's rows."""
    N = params[0].shape[0]
    if N == 0 or any(p.shape[0] != N for p in params):
        return False
    sync_free = SYNC_FREE if sync_free is None else bool(sync_free)
    st = _sparse_state(N, world, group)
    st.settle()
    for p in params:            # a rank without a gradient for some tensor contributes zeros
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        elif not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
            p.grad = p.grad.contiguous().float()
    ops = _RowOps([p.grad for p in params])
    stride = ops.wtot + 1
    if not st.history:
        # first exchange for this (N, world) or after a regime change: no counts to size the payload from — ONE
        # synchronous look at them
        total = ops.row_mask().sum(dtype=torch.int32).reshape(1)
        totals = torch.empty(world, dtype=torch.int32, device=ops.dev)
        dist.all_gather_into_tensor(totals, total, group=group)
        st.observe(totals.cpu().tolist())
    if st.dense:
        # too many rows for the sparse form to pay: the caller runs the dense all-reduce; keep watching the counts
        # (one tiny all_gather) so that the exchange can return to the sparse form
        total = ops.row_mask().sum(dtype=torch.int32).reshape(1)
        totals = torch.empty(world, dtype=torch.int32, device=ops.dev)
        dist.all_gather_into_tensor(totals, total, group=group)
        st.pending = _async_to_host(totals) + (None,)
        return False
    cap = st.cap
    pay = ops.pack_masked(cap)                                   # [(cap + 1), stride], header in row 0
    pay_all = torch.empty(world * (cap + 1) * stride, dtype=torch.float32, device=ops.dev)
    dist.all_gather_into_tensor(pay_all, pay.reshape(-1), group=group)
    pay_all = pay_all.view(world, cap + 1, stride)
    headers = pay_all[:, 0, 0].contiguous().view(torch.int32)
    if sync_free:
        st.pending = _async_to_host(headers) + (cap,)
    else:
        # guarded: the gradients have not been modified yet — a payload that was too small costs this step the
        # dense bucket (same decision on every rank: they all hold the same headers), never a truncated update
        totals = headers.cpu().tolist()
        st.observe(totals)
        if max(totals) > cap:
            st.overflows += 1
            return False
    # Replicas must stay BIT-identical, so the sum has one fixed order on every rank: start from zero, add
    # rank 0's rows, then rank 1's, ...  A rank's row indices are unique, so no add collides (a single
    # scatter-add over the concatenation would add in atomic, i.e. arbitrary, order).
    for p in params:
        p.grad.zero_()
    scale = 1.0 / world if average else 1.0
    for r in range(world):
        ops.scatter_add_payload(pay_all[r], cap, scale)
    return True


def _async_to_host(t: torch.Tensor):
    """-> (host tensor, event or None): non-blocking device-to-host copy of a small tensor"""
    if t.device.type != "cuda":
        return t.clone(), None
    host = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    host.copy_(t, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return host, ev


def allreduce_dense_(tensors: Sequence[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                     average: bool = False, force: Optional[bool] = None) -> None:
    """In-place sum (or mean) of a few small tensors over the ranks through ONE flat bucket: the gradients of the
    parameters that are not per-Gaussian rows (learnable background, pose and velocity adjustments)."""
    tensors = [t for t in tensors if t is not None]
    if not tensors or not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1 and not (FORCE if force is None else force):
        return
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat.div_(world)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
