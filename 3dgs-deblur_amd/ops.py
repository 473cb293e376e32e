"""gsplat-compatible operator surface over the HIP C-ABI library.

Mirrors the (absent) SpectacularAI gsplat fork's Python layer — upstream gsplat
0.1.11 signatures, SURVEY.md §8b — so nerfstudio's splatfacto model can call
``project_gaussians`` / ``rasterize_gaussians`` / ``spherical_harmonics`` unchanged
(/root/reference/render_model.py:11-15,217; /root/reference/train.py:115-122), plus
the fused multi-sub-pose path (``render_subposes``) used by :mod:`model`.

Host code is Python on PyTorch-ROCm: torch owns device memory and streams, every
kernel is hand-written HIP behind ``include/gsdeblur.h``.  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib

import os
import threading

TILE = 16
REC = 16   # floats per rasterizer record (gs_math.h kRecFloats)
GRAD = 12  # floats per gradient record / gradient tuple (kGradFloats)
MAX_SUBPOSES = 256   # blur samples x rolling-shutter bands per frame (SliceDesc in csrc/binning.hip)
# depth slicing of the fused path: average tile-list length budget of the first slice (doubling per
# slice); 0 disables slicing (single pass over all intersections)
SLICE_BASE = int(os.environ.get("GSD_SLICE_BASE", "512"))
# the budget adapts across the frames of one scene (FrameHints below): a frame that issued two or more slices doubles it
# for the next ones, up to 8x (a scene whose tiles do not saturate pays ~0.15 ms per slice boundary for nothing:
# fitted-model-like bench scene, 512 / 1024 / 2048 / 4096 / 8192: 9.35 / 9.15 / 9.05 / 8.8 / 9.1 ms); a frame that stops
# after its first slice never grows it; forgotten every 256 frames.  Images do not depend on the slicing (bit for bit),
# gradients up to fp32 summation order.
SLICE_ADAPT = int(os.environ.get("GSD_SLICE_ADAPT", "1"))
# ... and only after a frame whose FIRST slice left at least this share of its tile lists open (FrameHints.feedback)
SLICE_GROW_OPEN = float(os.environ.get("GSD_SLICE_GROW_OPEN", "0.5"))
# guards every piece of frame-to-frame state of this module (hints, arena pool, caches).  Re-entrant: _ArenaLease.__del__
# takes it, and the cyclic GC may run a lease's finaliser on a thread that is already inside one of these blocks
_state_lock = threading.RLock()


class FrameHints:
    """What ONE scene's frames teach the next frame of that scene: the adaptive slice-budget multiplier and the arena
    size its frames needed.  The library is stateless; this is the binding's only frame-to-frame memory, and it has an
    owner: SplatfactoDeblurModel and bench.Workload hold their own instance (`hints=` of render_subposes /
    render_combined / render_step); callers that pass none share one per (device, frame shape, parameter storage) —
    so two scenes of one shape never fight over one hint.  Thread-safe."""

    __slots__ = ("mult", "age", "arena_bytes", "arena_retries", "frames", "box_share", "last_slices", "select_misses",
                 "select_cap", "select_overflows", "_recent", "_lock", "_views")

    def __init__(self):
        self.mult, self.age, self.arena_bytes, self.arena_retries, self.frames = 1, 0, 0, 0, 0
        self.box_share = None           # share of the last frame's bounding-box pairs its issued slices held (None: no frame yet)
        self.last_slices = None         # slices the last frame issued (None: no frame yet)
        self.select_misses = 0          # frames whose nearest-first selection had to be followed by the full sort
        self.select_cap = 0             # bound on a sub-pose's selected pairs handed to the next frame (0: none yet)
        self.select_overflows = 0       # frames whose selection outgrew it (sorted twice)
        self._recent = []               # (issued slices, budget multiplier, arena retries) of the last frames
        self._lock = threading.Lock()
        self._views = {}                # view(key): per-camera memories

    def slice_base(self) -> int:
        return SLICE_BASE * (self.mult if SLICE_ADAPT and SLICE_BASE > 0 else 1)

    def feedback(self, n_issued: int, retries: int = 0, box_share: Optional[float] = None, select_state: int = 0,
                 open_after_first: Optional[float] = None, max_selected: int = 0, select_overflow: int = 0) -> None:
        """open_after_first: share of the tile lists the frame's first slice left open (gs_frame_state; None: unknown).
        The budget only grows when that slice left at least SLICE_GROW_OPEN of them open — a frame whose tiles mostly do
        not saturate (a fitted model seen through faint splats).  A frame that needed more slices for a FEW tiles (a view
        that looks past the scene's edge: those tiles never stop, whatever the budget) keeps its budget: round 5 doubled
        it for every multi-slice frame, and a camera batch that held one such view drove the budget of ALL its views to 8x
        (bench.py view_sweep: 4.8 ms per view against 2.5 ms for the same views on their own)."""
        with self._lock:
            self.frames += 1
            self.last_slices = int(n_issued)
            if select_state == 2:
                self.select_misses += 1
            if max_selected > 0:
                # what the selective sort's tail passes are sized for next time: the largest selection seen lately, with
                # slack (a frame that outgrows it sorts twice and raises it)
                want = int(1.5 * max_selected) + 4096
                self.select_cap = want if (select_overflow or want > self.select_cap) else max(want, int(0.9 * self.select_cap))
            self.select_overflows += int(bool(select_overflow))
            if box_share is not None:
                self.box_share = float(box_share)
            self.arena_retries += retries
            self._recent = (self._recent + [(int(n_issued), self.mult, int(retries))])[-4:]
            if SLICE_ADAPT and SLICE_BASE > 0:
                grow = n_issued >= 2 and self.mult < 8 and (open_after_first is None or open_after_first < 0
                                                            or open_after_first >= SLICE_GROW_OPEN)
                if self.age >= 255:
                    self.mult, self.age = 1, 0
                else:
                    self.mult, self.age = (self.mult * 2 if grow else self.mult), self.age + 1

    @property
    def settled(self) -> bool:
        """the last two frames issued the same number of slices at the budget the NEXT frame will use, without an arena
        retry: what a caller that times frames waits for (bench.py)"""
        with self._lock:
            r = self._recent
            return (len(r) >= 2 and r[-1][0] == r[-2][0] and r[-1][1] == r[-2][1] == self.mult
                    and r[-1][2] == 0 and r[-2][2] == 0)

    def lazy_records(self) -> bool:
        """lazy records (LAZY_RECORDS) pay while a frame's slices hold a small share of its visible pairs: the scene's
        frames have stopped within the default first-slice budget so far, and the last frame's issued slices held less
        than a quarter of its bounding-box pairs (a scene that fits its first slice whole projects eagerly: every pair
        would be projected again, by gathers)"""
        return self.mult == 1 and (self.box_share is None or self.box_share < 0.25)

    def depth_select(self) -> bool:
        """nearest-first selection (DEPTH_SELECT) pays when the frame stops within its first slice and that slice is a
        small part of the frame: the last frame issued ONE slice that held under half of its bounding-box pairs"""
        return self.last_slices == 1 and self.box_share is not None and self.box_share < 0.5

    def view(self, key) -> "ViewHints":
        """The memory of ONE camera of this scene (round 6).  A training loop comes back to the same cameras epoch after
        epoch, and what a frame teaches the next one — did it stop in its first slice, what share of its pairs did that
        slice hold, how many pairs did it select — is a property of the CAMERA: a view that looks past the scene's edge
        needs five slices every time, its neighbour in the batch one.  With one last-frame memory for all cameras the
        cheap view sorted everything whenever it followed the expensive one (bench.py view_sweep: 1.13x its own
        fixed-view time).  The budget multiplier, the arena estimate and the counters stay with the scene (this object);
        a camera seen for the first time starts from the scene's last frame.  key: any hashable (the camera index)."""
        with self._lock:
            v = self._views.get(key)
            if v is None:
                if len(self._views) >= 16384:
                    self._views.clear()
                v = self._views[key] = ViewHints(self)
                v.box_share, v.last_slices, v.select_cap = self.box_share, self.last_slices, self.select_cap
            return v

    def reset(self) -> None:
        with self._lock:
            self.mult, self.age, self.arena_bytes, self._recent, self.box_share = 1, 0, 0, [], None
            self.last_slices, self.select_cap = None, 0
            self._views.clear()


class ViewHints:
    """FrameHints.view(key): what one camera's last frame learned (slices issued, share of the pairs they held, size of the
    selection); everything else — budget multiplier, arena estimate, counters — is read from and reported to the scene's
    FrameHints.  Takes the place of a FrameHints wherever one is accepted (`hints=`)."""

    __slots__ = ("scene", "box_share", "last_slices", "select_cap")

    def __init__(self, scene: "FrameHints"):
        self.scene = scene
        self.box_share, self.last_slices, self.select_cap = None, None, 0

    arena_bytes = property(lambda self: self.scene.arena_bytes,
                           lambda self, v: setattr(self.scene, "arena_bytes", v))
    mult = property(lambda self: self.scene.mult)
    frames = property(lambda self: self.scene.frames)
    arena_retries = property(lambda self: self.scene.arena_retries)
    select_misses = property(lambda self: self.scene.select_misses)
    select_overflows = property(lambda self: self.scene.select_overflows)
    settled = property(lambda self: self.scene.settled)

    def slice_base(self) -> int:
        return self.scene.slice_base()

    def lazy_records(self) -> bool:
        return self.scene.mult == 1 and (self.box_share is None or self.box_share < 0.25)

    def depth_select(self) -> bool:
        return self.last_slices == 1 and self.box_share is not None and self.box_share < 0.5

    def view(self, key) -> "ViewHints":
        return self.scene.view(key)

    def feedback(self, n_issued: int, retries: int = 0, box_share: Optional[float] = None, select_state: int = 0,
                 open_after_first: Optional[float] = None, max_selected: int = 0, select_overflow: int = 0) -> None:
        sc = self.scene
        with sc._lock:
            # (the scene's feedback below sizes ITS promise from its own history; this camera's comes from this camera's)
            self.last_slices = int(n_issued)
            if box_share is not None:
                self.box_share = float(box_share)
            if max_selected > 0:
                want = int(1.5 * max_selected) + 4096
                self.select_cap = want if (select_overflow or want > self.select_cap) else max(want, int(0.9 * self.select_cap))
        sc.feedback(n_issued, retries, box_share, select_state, open_after_first, max_selected, select_overflow)


_hints = {}          # default owner: (device, N, P, S, H, W, shared, storage of means3d) -> FrameHints


def hints_for(key) -> FrameHints:
    with _state_lock:
        h = _hints.get(key)
        if h is None:
            if len(_hints) >= 64:
                _hints.clear()
            h = _hints[key] = FrameHints()
        return h


# gs_frame_forward only: a slice that leaves at least this fraction of its open tiles open makes the next issued slice
# span twice as many planned ones (a frame whose tiles do not saturate pays ~0.15 ms per slice boundary for nothing);
# 0 = every planned slice on its own, which is what the Python orchestration does.  Images are the same bit for bit.
SLICE_MERGE = float(os.environ.get("GSD_SLICE_MERGE", "0.75"))
# gs_frame_forward only: 1 = its two read-backs (slice plan, open-tile count) are written into pinned host memory by a
# one-block kernel and the host polls a sequence word; 0 = hipMemcpyAsync + hipStreamSynchronize
FRAME_POLL = int(os.environ.get("GSD_FRAME_POLL", "1"))


def _host_cores() -> int:
    try:
        return len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return os.cpu_count() or 1


def frame_poll() -> int:
    """1 = polled read-backs.  A polling rank spins one host core while it waits (twice per frame, tens of microseconds):
    with one process per GPU that is LOCAL_WORLD_SIZE spinning cores, so the default only polls when the affinity mask
    holds at least two cores per local rank (one to spin, one for everything else); otherwise the read-backs go through
    hipStreamSynchronize (the rank sleeps in the runtime).  GSD_FRAME_POLL set explicitly wins."""
    if "GSD_FRAME_POLL" in os.environ or not FRAME_POLL:
        return int(FRAME_POLL)
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
    return 1 if _host_cores() >= 2 * local_world else 0


def readback_mode() -> dict:
    """how this process reads the slice plan / open-tile word back, and why (bench.py prints it per rank)"""
    return {"poll": bool(frame_poll()), "host_cores": _host_cores(),
            # one process per GPU: the arena pool and the pinned read-back buffers of this module belong to THIS process,
            # i.e. to LOCAL_RANK's device (keys: (device, stream) and (device, host thread, stream)); nothing is shared
            # between the ranks of a host
            "local_rank": int(os.environ.get("LOCAL_RANK", "0") or 0),
            "local_world": int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1),
            "forced": "GSD_FRAME_POLL" in os.environ}
# widest radix digit of the compacting depth pre-sort: 8 -> 4 passes of 8 bits (11 -> 3 passes of 11/10/10 bits was measured
# slower at this pipeline's sizes and lost its environment switch in round 6; the tests still drive both through here)
DEPTH_SORT_DIGIT = 8
# Gradient conventions (SURVEY App. A "Backward"), bit mask: three places where upstream gsplat 0.1.11's
# backward, as recollected, is not the derivative of its forward, each a straight-through rule: 1 = back-propagate
# through the fov clamp of x/z, y/z as if inactive; 2 = quaternion gradient w.r.t. the (assumed unit) quaternion, without
# the projection through q/|q|; 4 = let the gradient pass the alpha = min(0.999, .) clamp.  0 = the true derivatives.
# Default 6 (round 6): bits 2 and 4 follow the reference as recollected; bit 1 stays OPT-IN
# (GSD_UPSTREAM_GRADS=7) until it can be checked against the fork's source — it is the one rule whose effect was measured
# end to end, and it pushes one of four perturbed cameras AWAY in the pose-optimizer test (1.3 -> 3.8 cm) where the true
# derivative of the clamp recovers all four.  The oracle has the same switch with the same default (its UP_* constants,
# RenderConfig.upstream_grads) and the `-m gpu` suite runs the fused path under 7, 6 and 0.
# Bit 2 only exists on the compat op (project_gaussians): the FUSED ops (render_subposes / render_combined / render_step)
# take splatfacto's raw quaternions, i.e. they stand for `quats / quats.norm()` + the kernel, and the reference's
# end-to-end gradient of that pair is J_norm^T g — exactly the gradient through the normalisation the fused kernels return.
UPSTREAM_GRADS = int(os.environ.get("GSD_UPSTREAM_GRADS", "6"))


# 1 (default): rolling-shutter bands — the projection culls (band, Gaussian) pairs outside their band's tile rows (A/B: 0)
BAND_AWARE = int(os.environ.get("GSD_BAND_AWARE", "1"))
# Lazy records (round 5).  The fused projection writes a 64-byte record for every visible (sub-pose, Gaussian) pair — 4 M
# of them, 256 MB of the kernel's 314 MB of traffic, on the benchmark scene — and a frame that stops within its first
# depth slice reads 55 k of them.  1 (default): a scene whose frames have not needed a larger first-slice budget and
# whose last frame's slices held less than a quarter of its bounding-box pairs (FrameHints.lazy_records: its tiles
# saturate early) projects keys, tile counts and radii only, and every issued slice projects the records of its own pairs
# (gs_slice_project_records: same arithmetic, bit-identical rows); a scene whose budget has grown, or that fits its first
# slice whole, keeps the eager projection.  0: always eager; 2: always lazy.
LAZY_RECORDS = int(os.environ.get("GSD_LAZY_RECORDS", "1"))
# Nearest-first selection (round 6).  The depth pre-sort ordered every visible (sub-pose, Gaussian) pair — 3.95 M on the
# benchmark scene, 0.21 ms + a 5 M-rank count scan — and that scene's one depth slice holds 55 k of them.  1 (default): a
# scene whose LAST frame stopped within its first slice, that slice holding under half of the frame's bounding-box pairs
# (FrameHints.depth_select), ranks only the pairs its first-slice budget reaches (gs_depth_select: a two-level radix
# select over the depth keys, then the compacting sort over the selection); should that slice leave a tile open after
# all, the pairs behind the selection are sorted and planned then (one more sort: the price of a wrong guess, and the
# frame's feedback switches the selection off for the next one).  0: never; 2: always.  Images bit for bit the same.
DEPTH_SELECT = int(os.environ.get("GSD_DEPTH_SELECT", "1"))
# 1 (default): Gaussians whose scales differ by more than 8x get the covariance part of their projection backward
# (v_conic -> cov2d -> cov3d -> scale / quaternion / mean) recomputed in double (project_needle_hp_kernel): in fp32 that
# chain is percent-level wrong along a needle's long axis.  0: fp32 everywhere (A/B, tests).
NEEDLE_HP = int(os.environ.get("GSD_NEEDLE_HP", "1"))


def _proj_grad_flags() -> int:
    """gradient flags of the FUSED projection backward: bit 2 (raw quaternion gradient) never applies there"""
    return (UPSTREAM_GRADS & 1) | (0 if NEEDLE_HP else 8)


# measurement only (round 6): 1 = the compositing backward in its splat-parallel formulation
# (csrc/raster_bwd.hip raster_bwd_splat_kernel: lane = list entry, pixels streamed through the wave) instead of the
# tile-per-wave kernel.  Same gradients up to summation order; 3-4x slower (profiles/r06_bwd_splat_parallel.md).
BWD_SPLAT = int(os.environ.get("GSD_BWD_SPLAT", "0"))


def _bwd_variant() -> int:
    return (256 if (UPSTREAM_GRADS & 4) else 0) | (1024 if BWD_SPLAT else 0)
# per-slice emitted intersection counts of the last frame: ints, or 1-element device tensors that are only read back
# when somebody asks (module attribute `last_slice_intersects`, see __getattr__ below) — the frame itself never waits
_slice_totals = []


def __getattr__(name):
    if name == "last_slice_intersects":
        for i, v in enumerate(_slice_totals):
            if isinstance(v, torch.Tensor):
                _slice_totals[i] = int(v.item()) & 0xFFFFFFFF
        return list(_slice_totals)
    raise AttributeError(name)


from ._profile import FRAME_STAGES, StageProfiler, _NULL  # noqa: E402

profiler: Optional[StageProfiler] = None
last_num_intersects: int = 0
last_depth_select: int = 0      # gs_frame_state.depth_select of the last frame (0 full sort, 1 selection, 2 selection + rest)


def _stage(name: str):
    return _NULL if profiler is None else profiler.stage(name)


# --------------------------------------------------------------------------- #
# helpers
# --------------------------------------------------------------------------- #
def _L():
    return _lib.load()


def _ptr(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA(HIP) tensor: the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _tiles(h: int, w: int) -> Tuple[int, int]:
    return (w + TILE - 1) // TILE, (h + TILE - 1) // TILE


IDS_PAD = 8   # the scalar-cache compositors read their tile lists in aligned groups of four: id arrays are over-allocated


def _padded_i32(n: int, dev) -> Tensor:
    """int32 [n] view of an [n + IDS_PAD] allocation (the tail only has to be readable, its values are clamped)"""
    return torch.empty(n + IDS_PAD, dtype=torch.int32, device=dev)[:n]


def _bits(n: int) -> int:
    return max(1, int(math.ceil(math.log2(max(2, n)))))


# GSD_SYNC_CHECK=1: synchronise after every C-ABI call so that an ASYNCHRONOUS HIP fault (illegal address, ...) is
# reported with the stage that caused it instead of at the next unrelated synchronisation (debug aid; slow)
SYNC_CHECK = int(os.environ.get("GSD_SYNC_CHECK", "0"))


def _check(st: int, what: str):
    _lib.check(st, what)
    if SYNC_CHECK:
        try:
            torch.cuda.synchronize()
        except RuntimeError as e:
            raise _lib.HipLibraryError(f"asynchronous HIP fault surfaced right after stage '{what}': {e}") from e


def _pose_scratch(N: int, P: int, with_touched: bool, dev) -> Tensor:
    """scratch of the ordered camera-gradient reduction of a projection backward (include/gsdeblur.h: pose_scratch)"""
    return torch.empty(int(_L().gs_project_pose_scratch_bytes(int(N), int(P), int(bool(with_touched)))), dtype=torch.uint8,
                       device=dev)


def _viewmat16(viewmat: Tensor) -> Tensor:
    v = _f32(viewmat, "viewmat")
    if v.shape[-2:] == (3, 4):
        pad = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=v.device, dtype=v.dtype)
        v = torch.cat([v, pad.expand(v.shape[:-2] + (1, 4))], dim=-2).contiguous()
    if v.shape[-2:] != (4, 4):
        raise ValueError("viewmat must be [4,4] (or [3,4])")
    return v


# --------------------------------------------------------------------------- #
# scan / sort primitives (device-side; thin wrappers used by the binning code and tests)
# --------------------------------------------------------------------------- #
def exclusive_scan_u32(x: Tensor) -> Tuple[Tensor, Tensor]:
    """x int32 [n] (values >= 0) -> (exclusive prefix int32 [n], total int32 [1])"""
    assert x.dtype == torch.int32 and x.is_cuda and x.is_contiguous()
    n = x.numel()
    out = torch.empty_like(x)
    if n == 0:
        return out, torch.zeros(1, dtype=torch.int32, device=x.device)
    total = torch.empty(1, dtype=torch.int32, device=x.device)        # written by the kernel
    ws_bytes = _L().gs_scan_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    _check(_L().gs_exclusive_scan_u32(n, _ptr(x), _ptr(out), _ptr(total), _ptr(ws), ws_bytes, _stream()), "scan")
    return out, total


def radix_sort_pairs(keys: Tensor, vals: Optional[Tensor], begin_bit: int, end_bit: int,
                     gather_src: Optional[Tensor] = None, n_dev: Optional[Tensor] = None,
                     carry: Optional[Tensor] = None):
    """Stable ascending sort of (key, int32 value) pairs over key bits [begin_bit, end_bit).
    keys int32 (treated as u32) or int64 (u64); vals None => iota.  Inputs are clobbered.
    gather_src (int32 keys only): the final pass also returns gather_src[sorted values] as a third tensor.
    carry (int32 keys only; carry[i] belongs to input element i, left intact): a second payload sorted along, returned
    as a third tensor."""
    assert keys.is_cuda and keys.is_contiguous() and keys.dtype in (torch.int32, torch.int64)
    n = keys.numel()
    dev = keys.device
    if vals is None:
        v0 = _padded_i32(n, dev)
        iota = 1
    else:
        assert vals.dtype == torch.int32 and vals.is_contiguous() and vals.numel() == n
        v0, iota = vals, 0
    if n == 0:
        return keys, v0
    k1 = torch.empty_like(keys)
    v1 = _padded_i32(n, dev)
    L = _L()
    ws_bytes = L.gs_radix_sort_workspace_bytes(n, begin_bit, end_bit)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    res = ctypes.c_int(0)
    if carry is not None:
        assert keys.dtype == torch.int32 and carry.dtype == torch.int32 and carry.is_contiguous() and gather_src is None
        assert carry.numel() >= n
        pa, pb = _padded_i32(n, dev), _padded_i32(n, dev)
        res2 = ctypes.c_int(0)
        _check(L.gs_radix_sort_pairs_carry_u32(n, _ptr(keys), _ptr(v0), _ptr(k1), _ptr(v1), iota, begin_bit, end_bit,
                                               _ptr(ws), ws_bytes, ctypes.byref(res), _ptr(carry), _ptr(pa), _ptr(pb),
                                               ctypes.byref(res2), _ptr(n_dev), _stream()), "radix sort + carry")
        return ((k1, v1) if res.value == 1 else (keys, v0)) + (pb if res2.value == 1 else pa,)
    if gather_src is not None:
        assert keys.dtype == torch.int32 and gather_src.dtype == torch.int32 and gather_src.is_contiguous()
        gathered = _padded_i32(n, dev)
        _check(L.gs_radix_sort_pairs_gather_u32(n, _ptr(keys), _ptr(v0), _ptr(k1), _ptr(v1), iota, begin_bit, end_bit,
                                                _ptr(ws), ws_bytes, ctypes.byref(res), _ptr(gather_src), _ptr(gathered),
                                                _ptr(n_dev), _stream()), "radix sort + gather")
        return ((k1, v1) if res.value == 1 else (keys, v0)) + (gathered,)
    fn = L.gs_radix_sort_pairs_u32 if keys.dtype == torch.int32 else L.gs_radix_sort_pairs_u64
    _check(fn(n, _ptr(keys), _ptr(v0), _ptr(k1), _ptr(v1), iota, begin_bit, end_bit, _ptr(ws), ws_bytes,
              ctypes.byref(res), _stream()), "radix sort")
    return (k1, v1) if res.value == 1 else (keys, v0)


# --------------------------------------------------------------------------- #
# binning of rasterizer records (fast path shared by compat + fused ops)
# --------------------------------------------------------------------------- #
def bin_and_sort_records(records: Tensor, depth_keys: Tensor, num_tiles_hit: Tensor, P: int, N: int,
                         img_height: int, img_width: int, with_emission: bool = False):
    """-> (sorted_vals int32 [I], tile_bins int32 [P*T,2], n_isect, sorted_keys int32 [I]).

    Depth pre-sort (P*N keys) -> emission in depth order -> stable tile sort: the same total order
    as upstream's 64-bit (tile<<32|depth) sort at a fraction of the HBM traffic (see binning.hip).
    with_emission=True appends a dict for the gradient-tuple backward (gs_reduce_grad_tuples): `eids` = emission index of
    every sorted entry, `sorted_gi` / `counts` / `cum` = record index, tile count and exclusive offset of every depth rank."""
    L = _L()
    dev = records.device
    n = P * N
    tx, ty = _tiles(img_height, img_width)
    T = tx * ty
    global last_num_intersects
    with _stage("depth_sort"):
        keys64 = torch.empty(n, dtype=torch.int64, device=dev)
        _check(L.gs_make_depth_keys64(n, N, _ptr(depth_keys), _ptr(keys64), _stream()), "depth keys")
        end_bit = 32 + (_bits(P) if P > 1 else 0)
        _, sorted_gi = radix_sort_pairs(keys64, None, 0, end_bit)
    with _stage("count_scan"):
        counts = torch.empty(n, dtype=torch.int32, device=dev)
        _check(L.gs_gather_counts(n, _ptr(sorted_gi), _ptr(num_tiles_hit), _ptr(counts), _stream()), "gather counts")
        cum, total = exclusive_scan_u32(counts)
    n_isect = int(total.item())  # host sync, as upstream's cum_tiles_hit[-1].item()
    last_num_intersects = n_isect
    bins = torch.empty(P * T, 2, dtype=torch.int32, device=dev)
    if n_isect == 0:
        bins.zero_()
        z = torch.zeros(1, dtype=torch.int32, device=dev)
        return (z, bins, 0, z.clone()) + ((None,) if with_emission else ())
    if n_isect < 0:
        raise OverflowError("more than 2^31-1 tile intersections; chunk the sub-poses")
    with _stage("emit"):
        keys = torch.empty(n_isect, dtype=torch.int32, device=dev)
        vals = _padded_i32(n_isect, dev)
        _check(L.gs_emit_intersects(n, N, img_height, img_width, _ptr(sorted_gi), _ptr(cum), _ptr(records), n_isect,
                                    _ptr(keys), _ptr(vals), 0, _stream()), "emit intersects")
    extra = None
    with _stage("tile_sort"):
        if with_emission:
            # payload = emission index (iota); the record index travels as a second payload
            skeys, eids, svals = radix_sort_pairs(keys, None, 0, _bits(P * T), carry=vals)
            extra = dict(eids=eids, sorted_gi=sorted_gi, counts=counts, cum=cum)
        else:
            skeys, svals = radix_sort_pairs(keys, vals, 0, _bits(P * T))
    with _stage("bin_edges"):
        _check(L.gs_tile_bin_edges_u32(n_isect, _ptr(skeys), P * T, _ptr(bins), None, _stream()), "bin edges")
    return (svals, bins, n_isect, skeys) + ((extra,) if with_emission else ())


_band_edge_cache = {}


def _band_edges(img_height: int, rs_bands: int, device) -> Tensor:
    """tile-row edges of the rolling-shutter bands, on the device.  Cached: building it anew is a pageable
    host->device copy, i.e. a full stream synchronisation right after the projection launch of EVERY frame (the
    host then runs just in time behind the GPU for the rest of the step instead of ahead of it)."""
    key = (int(img_height), max(1, int(rs_bands)), str(device))
    t = _band_edge_cache.get(key)
    if t is None:
        _, ty = _tiles(img_height, 1)
        R = key[1]
        t = torch.tensor([(r * ty) // R for r in range(R + 1)], dtype=torch.int32, device=device)
        if len(_band_edge_cache) > 64:
            _band_edge_cache.clear()
        _band_edge_cache[key] = t
    return t


_placeholder_cache = {}


def _placeholder_i32(dev) -> Tensor:
    """a one-element int32 tensor per device for save_for_backward slots that hold nothing (a fresh torch.zeros is a
    fill launch per frame, issued on the host's critical path behind the last read-back)"""
    t = _placeholder_cache.get(str(dev))
    if t is None:
        t = _placeholder_cache[str(dev)] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


_band_done_cache = {}


def _band_tile_done(S: int, R: int, ty: int, tx: int, dev) -> Tensor:
    """u8 [S*R*ty*tx]: 1 for every tile of sub-pose (s, r) outside band r's tile rows — the initial tile_done of a
    rolling-shutter frame.  Cached (a dozen small torch ops per frame otherwise, behind the plan read-back)."""
    key = (S, R, ty, tx, str(dev))
    t = _band_done_cache.get(key)
    if t is None:
        e = [(r * ty) // R for r in range(R + 1)]       # same formula as _band_edges
        rows = torch.arange(ty, device=dev)
        band_open = torch.stack([(rows >= e[r]) & (rows < e[r + 1]) for r in range(R)])          # [R, ty]
        t = (~band_open).to(torch.uint8)[None, :, :, None].expand(S, R, ty, tx).reshape(-1).contiguous()
        if len(_band_done_cache) > 16:
            _band_done_cache.clear()
        _band_done_cache[key] = t
    return t


def _background(background: Optional[Tensor], device) -> Tensor:
    if background is None:
        return torch.zeros(3, dtype=torch.float32, device=device)
    bg = background.detach().to(device=device, dtype=torch.float32).contiguous()
    if bg.numel() != 3:
        raise ValueError("background must have 3 channels")
    return bg


# The depth-sliced pipeline of a frame is issued by the library itself (csrc/frame.hip: gs_frame_forward /
# gs_frame_backward, one caller-owned arena, ~45 launches per frame from C++).  `frame_backend` (default None) lets TEST
# infrastructure substitute its own orchestration of the same C-ABI kernels (tests/python_frame_path.py: the round-1..3
# Python slice loop with its A/B switches — one slice without culling, fp32 atomics, ...; an independent route to the
# same images that the equivalence tests compare the product path with).  The product never sets it.
frame_backend = None


class _FrameDesc(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int) for k in ("N", "P", "S", "R", "H", "W", "slice_base", "depth_sort_digit",
                                             "fwd_variant", "reserve_backward")] + [("merge_open_fraction", ctypes.c_float),
                                                                                     ("rolling_shutter_time", ctypes.c_float),
                                                                                     ("poll_readback", ctypes.c_int),
                                                                                     ("shared_list", ctypes.c_int),
                                                                                     ("combine_gamma", ctypes.c_float),
                                                                                     ("combine_min_level", ctypes.c_float),
                                                                                     ("band_clipped", ctypes.c_int),
                                                                                     ("depth_select", ctypes.c_int),
                                                                                     ("select_cap", ctypes.c_int),
                                                                                     ("sweep_t_min", ctypes.c_float),
                                                                                     ("sweep_t_max", ctypes.c_float),
                                                                                     ("lazy_records", ctypes.c_void_p)]


class _ProjectInputs(ctypes.Structure):
    """gs_project_inputs (include/gsdeblur.h): what gs_project_fused_fwd was called with, for the lazy records"""
    _fields_ = ([(k, ctypes.c_void_p) for k in ("means", "scales", "quats", "opacities", "viewmats")] +
                [(k, ctypes.c_float) for k in ("glob_scale", "fx", "fy", "cx", "cy", "clip_thresh")] +
                [(k, ctypes.c_int) for k in ("antialiased", "defer_color", "param_flags")])


class _FrameSlice(ctypes.Structure):
    _fields_ = ([("I", ctypes.c_longlong)] + [(k, ctypes.c_int) for k in ("n", "wave_per_gaussian", "first", "last")] +
                [(k, ctypes.c_longlong) for k in ("svals", "bins", "fidx", "gi_of_e", "sorted_ids", "slice_gi", "counts",
                                                  "cum", "tile_hot", "n_emitted_dev")])


class _FrameState(ctypes.Structure):
    _fields_ = ([(k, ctypes.c_int) for k in ("n_slices", "P", "N", "S", "R", "H", "W")] +
                [("rolling_shutter_time", ctypes.c_float), ("shared_list", ctypes.c_int), ("depth_select", ctypes.c_int),
                 ("select_overflow", ctypes.c_int), ("open_after_first", ctypes.c_float)] +
                [(k, ctypes.c_longlong) for k in ("n_total", "max_selected", "arena_used", "arena_required")] +
                [("slice", _FrameSlice * 16)])


def _box_share(state) -> Optional[float]:
    """share of a frame's bounding-box pairs that its issued depth slices held (gs_frame_state: the slices' list
    capacities over n_total); None for a frame without pairs.  What FrameHints.lazy_records() goes by."""
    n_total = int(state.n_total)
    if n_total <= 0:
        return None
    held = sum(int(state.slice[i].I) for i in range(int(state.n_slices)))
    return min(1.0, held / n_total)


class _ArenaTooSmall(Exception):
    pass


_ARENA_ATTEMPTS = 16    # first frame of a new scene: projections + partial frames until the arena holds every slice
# Frame arenas are PERSISTENT (a fresh multi-GB torch.empty per frame, and one more per retry,
# put hipMallocs inside timed loops): one pool per (device, stream); a frame leases the largest free arena and hands it
# back when its autograd node (or render_step's context) dies, so back-to-back frames — of any scene — run in the SAME
# memory, which only ever grows (on GS_ERR_WORKSPACE).  Frames alive at the same time get an arena each; at most two
# free ones are kept per pool.  release_arenas() gives the memory back.
_arena_pool = {}


class _Arena:
    """a pooled frame arena: the memory and the number of the lease that last took it.  Lease numbers come from ONE
    process-wide counter and live on this object, not in a table keyed by address (a dropped arena's
    address can be handed out again by the allocator, and a per-address generation restarted at 1 let a stale frame pass
    the recycled check)."""
    __slots__ = ("tensor", "gen")

    def __init__(self, tensor):
        self.tensor, self.gen = tensor, 0

    def numel(self):
        return self.tensor.numel()

    def data_ptr(self):
        return self.tensor.data_ptr()


_lease_counter = 0       # guarded by _state_lock


class _ArenaLease:
    __slots__ = ("arena", "key", "gen")

    def __init__(self, arena: _Arena, key):
        global _lease_counter
        self.arena, self.key = arena, key
        with _state_lock:
            _lease_counter += 1
            self.gen = arena.gen = _lease_counter

    @property
    def tensor(self):
        a = self.arena
        return None if a is None else a.tensor

    def release(self):
        """hand the arena back to the pool (idempotent); the frame keeps its own reference to the _Arena and can tell
        through its lease number whether a later frame has leased it since"""
        try:
            with _state_lock:
                a, self.arena = self.arena, None
                if a is None:
                    return
                free = _arena_pool.setdefault(self.key, [])
                free.append(a)
                if len(free) > 2:
                    free.remove(min(free, key=lambda x: x.numel()))
        except Exception:       # interpreter shutdown
            pass

    __del__ = release


def _arena_acquire(dev, want_bytes: int) -> _ArenaLease:
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    with _state_lock:
        free = _arena_pool.setdefault(key, [])
        free[:] = [x if isinstance(x, _Arena) else _Arena(x) for x in free]     # (tests seed the pool with bare tensors)
        a = max(free, key=lambda x: x.numel()) if free else None
        if a is not None:
            free.remove(a)
    if a is None or a.numel() < want_bytes:
        a = None                # the short one goes back to torch's allocator before the larger one is requested
        a = _Arena(torch.empty(int(want_bytes), dtype=torch.uint8, device=dev))
    return _ArenaLease(a, key)


def _arena_relend(frame) -> bool:
    """a frame whose lease was handed back (its first backward ran) wants its arena once more (a second backward under
    retain_graph=True): take the _Arena out of the free pool for the duration, so that no other thread can lease it
    meanwhile.  Raises when a later frame has leased it since (its memory is no longer this frame's)."""
    a = frame["arena_obj"]
    with _state_lock:
        free = _arena_pool.get(frame["lease"].key, [])
        if a.gen != frame["gen"] or a not in free:
            raise RuntimeError("this frame's arena was handed back after its first backward and a later frame has "
                               "used it since: a second backward of one frame (retain_graph=True) must run before "
                               "the next frame's forward")
        free.remove(a)
    return True


def _arena_return(frame) -> None:
    with _state_lock:
        free = _arena_pool.setdefault(frame["lease"].key, [])
        free.append(frame["arena_obj"])
        if len(free) > 2:
            free.remove(min(free, key=lambda x: x.numel()))


def release_arenas() -> None:
    """drop every pooled frame arena (they return to torch's caching allocator)"""
    with _state_lock:
        _arena_pool.clear()


_pinned_cache = {}


def _profile_mask() -> int:
    if profiler is None:
        return 0
    names = FRAME_STAGES if profiler.only is None else [n for n in FRAME_STAGES if n in profiler.only]
    return sum(1 << FRAME_STAGES.index(n) for n in names)


def native_frame_forward(records: Tensor, depth_keys: Tensor, num_tiles_hit: Tensor, P: int, N: int, S: int, R: int,
                         H: int, W: int, bg: Tensor, edges: Tensor, slice_base: int, color=None,
                         out_depth: Optional[Tensor] = None, reserve_backward: bool = True, rs=None, combine=None,
                         hints: Optional[FrameHints] = None, band_clipped: bool = False, lazy=None,
                         depth_select: Optional[bool] = None):
    """combine = (gamma, min_level, out [H,W,3]): the library launches the gamma-space average of the sample images itself,
    behind every slice's compositor (it overlaps the open-tile read-back).  rs = (pix_vel [N,2], rolling_shutter_time[, sample_times [S]]) or None; with sample_times the frame runs in the
    shared-list mode (P == 1: one record set and one tile list for the S samples).  gs_frame_forward: -> (out_img [S,H,W,3], out_T [S,H,W], frame) ; frame = dict(arena, state) for
    native_frame_backward.  Raises _ArenaTooSmall (after recording a larger size) when the arena did not hold the frame:
    the caller projects again (the depth keys were consumed) and calls once more."""
    global last_num_intersects, _slice_totals, last_depth_select
    L = _L()
    dev = records.device
    tx, ty = _tiles(H, W)
    shared = rs is not None and len(rs) > 2 and rs[2] is not None
    if hints is None:
        hints = hints_for((str(dev), N, P, S, H, W, shared))
    if depth_select is None:
        depth_select = bool(DEPTH_SELECT == 2 or (DEPTH_SELECT and hints.depth_select()))
    n = P * N
    nbytes = hints.arena_bytes
    if not nbytes:
        # depth pre-sort + plan + one slice of the default budget; the library prices the real plan and says so if this
        # is short (one retry per new high-water mark)
        I0 = max(1, slice_base) * tx * ty * P
        nbytes = 44 * n + 16 * S * H * W + (80 + (56 * S if shared else 0)) * I0 + (64 << 20)
    lease = _arena_acquire(dev, nbytes)
    arena = lease.tensor
    nbytes = arena.numel()
    # the read-back buffer is written by the GPU while gs_frame_forward polls it (ctypes releases the GIL for the call):
    # one per device, host thread and stream, so that concurrent frames never share it
    pin_key = (str(dev), threading.get_ident(), torch.cuda.current_stream().cuda_stream)
    pin = _pinned_cache.get(pin_key)
    need_pin = 4 * (2 * P * 16 + 2 * P + 2) + 64
    if pin is None or pin.numel() < need_pin:
        pin = _pinned_cache[pin_key] = torch.empty(max(8192, need_pin), dtype=torch.uint8, pin_memory=True)
    desc = _FrameDesc(N, P, S, R, H, W, int(slice_base), DEPTH_SORT_DIGIT, 0, int(reserve_backward),
                      float(SLICE_MERGE), float(rs[1]) if rs is not None else 0.0, frame_poll(), int(shared),
                      float(combine[0]) if combine is not None else 1.0, float(combine[1]) if combine is not None else 0.0,
                      int(bool(band_clipped)), int(bool(depth_select and slice_base > 0)),
                      int(min(hints.select_cap, 2 ** 31 - 1)) if depth_select else 0,
                      float(rs[3][0]) if shared and len(rs) > 3 else 0.0, float(rs[3][1]) if shared and len(rs) > 3 else 0.0,
                      ctypes.addressof(lazy) if lazy is not None else None)
    state = _FrameState()
    out_img = torch.empty(S, H, W, 3, device=dev)
    out_T = torch.empty(S, H, W, device=dev)
    band_done = _band_tile_done(S, R, ty, tx, dev) if R > 1 else None
    c_means = c_sh = c_rest = c_V = None
    c_K = c_deg = 0
    if color is not None:
        c_means, c_sh, c_rest, c_K, c_deg, c_V = color
    L.gs_frame_profile_enable(_profile_mask())
    st = L.gs_frame_forward(ctypes.byref(desc), _ptr(records), _ptr(depth_keys), _ptr(num_tiles_hit), _ptr(bg), _ptr(edges),
                            _ptr(band_done), _ptr(c_means), _ptr(c_sh), _ptr(c_rest), int(c_K), int(c_deg), _ptr(c_V),
                            _ptr(rs[0]) if rs is not None else None, _ptr(rs[2]) if shared else None, _ptr(out_img),
                            _ptr(out_T), _ptr(out_depth), _ptr(combine[2]) if combine is not None else None,
                            _ptr(arena), arena.numel(), ctypes.c_void_p(pin.data_ptr()), pin.numel(), ctypes.byref(state),
                            _stream())
    if st == 3:
        # the library prices one slice ahead; a frame that needs many (ever larger) slices is settled in several steps,
        # each at least half as large again as the last, so the number of retries is logarithmic in the final size
        hints.arena_bytes = max(int(state.arena_required * 1.15) + (32 << 20), int(nbytes * 1.5))
        raise _ArenaTooSmall()
    _check(st, "frame_forward")
    hints.arena_bytes = max(int(hints.arena_bytes),
                            int((state.arena_used + L.gs_frame_backward_bytes(ctypes.byref(state))) * 1.15))
    last_num_intersects = int(state.n_total)
    last_depth_select = int(state.depth_select)
    _slice_totals = [arena[sl.n_emitted_dev:sl.n_emitted_dev + 4].view(torch.int32)
                     for sl in (state.slice[i] for i in range(state.n_slices))]
    return out_img, out_T, dict(arena=arena, arena_obj=lease.arena, lease=lease, gen=lease.gen, state=state,
                                pix_vel=rs[0] if rs is not None else None,
                                sample_times=rs[2] if shared else None)


def native_frame_backward(frame, records: Tensor, bg: Tensor, edges: Tensor, out_T: Tensor, v_img: Tensor,
                          v_alpha: Optional[Tensor], v_records: Tensor, touched: Tensor, combine=None):
    L = _L()
    cmb = combine if combine is not None else (None, 1.0, 0.0)
    arena, state = frame["arena"], frame["state"]
    lease = frame["lease"]
    relent = lease.arena is None and _arena_relend(frame)
    try:
        _native_frame_backward(L, frame, arena, state, records, bg, edges, out_T, v_img, v_alpha, v_records, touched, cmb)
    finally:
        if relent:
            _arena_return(frame)
        else:
            # the compositing backward is done with the arena: the next frame's forward may lease it (everything is
            # ordered on one stream), whether or not this frame's autograd node outlives the backward
            lease.release()


def _native_frame_backward(L, frame, arena, state, records, bg, edges, out_T, v_img, v_alpha, v_records, touched, cmb):
    L.gs_frame_profile_enable(_profile_mask())
    st = L.gs_frame_backward(ctypes.byref(state), _ptr(records), _ptr(bg), _ptr(edges), _ptr(out_T), _ptr(v_img),
                             _ptr(v_alpha), _ptr(cmb[0]), float(cmb[1]), float(cmb[2]), _bwd_variant(), _ptr(v_records),
                             _ptr(touched), _ptr(frame.get("pix_vel")), _ptr(frame.get("sample_times")), _ptr(arena),
                             arena.numel(), _stream())
    if st == 3:
        raise _lib.HipLibraryError("frame_backward: the forward's arena cannot hold the backward's buffers "
                                   "(call native_frame_forward with reserve_backward=True)")
    _check(st, "frame_backward")


# --------------------------------------------------------------------------- #
# gsplat.project_gaussians
# --------------------------------------------------------------------------- #
class _ProjectGaussians(Function):
    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height, img_width,
                block_width, clip_thresh):
        if block_width != TILE:
            raise ValueError("only block_width=16 is supported")
        means3d, scales, quats = _f32(means3d, "means3d"), _f32(scales, "scales"), _f32(quats, "quats")
        V = _viewmat16(viewmat)
        N = means3d.shape[0]
        dev = means3d.device
        xys = torch.empty(N, 2, device=dev)
        depths = torch.empty(N, device=dev)
        radii = torch.empty(N, dtype=torch.int32, device=dev)
        conics = torch.empty(N, 3, device=dev)
        comp = torch.empty(N, device=dev)
        ntiles = torch.empty(N, dtype=torch.int32, device=dev)
        cov3d = torch.empty(N, 6, device=dev)
        _check(_L().gs_project_fwd(N, _ptr(means3d), _ptr(scales), float(glob_scale), _ptr(quats), _ptr(V),
                                   float(fx), float(fy), float(cx), float(cy), int(img_height), int(img_width),
                                   float(clip_thresh), _ptr(xys), _ptr(depths), _ptr(radii), _ptr(conics),
                                   _ptr(comp), _ptr(ntiles), _ptr(cov3d), None, _stream()), "project_fwd")
        ctx.save_for_backward(means3d, scales, quats, V)
        ctx.args = (float(glob_scale), float(fx), float(fy), float(cx), float(cy), int(img_height),
                    int(img_width), float(clip_thresh))
        ctx.mark_non_differentiable(radii, ntiles)
        return xys, depths, radii, conics, comp, ntiles, cov3d

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_comp, v_ntiles, v_cov3d):
        means3d, scales, quats, V = ctx.saved_tensors
        glob, fx, fy, cx, cy, H, W, clip = ctx.args
        N = means3d.shape[0]
        dev = means3d.device

        def z(g, shape):
            return torch.zeros(shape, device=dev) if g is None else g.contiguous().float()
        v_xys, v_depths, v_conics, v_comp = z(v_xys, (N, 2)), z(v_depths, (N,)), z(v_conics, (N, 3)), z(v_comp, (N,))
        v_means = torch.empty(N, 3, device=dev)
        v_scales = torch.empty(N, 3, device=dev)
        v_quats = torch.empty(N, 4, device=dev)
        need_v = ctx.needs_input_grad[4]
        v_V = torch.zeros(4, 4, device=dev) if need_v else None
        scratch = _pose_scratch(N, 1, False, dev) if need_v else None
        _check(_L().gs_project_bwd(N, _ptr(means3d), _ptr(scales), glob, _ptr(quats), _ptr(V), fx, fy, cx, cy, H, W,
                                   clip, _ptr(v_xys), _ptr(v_depths), _ptr(v_conics), _ptr(v_comp), _ptr(v_means),
                                   _ptr(v_scales), _ptr(v_quats), _ptr(v_V), UPSTREAM_GRADS & 3, _ptr(scratch),
                                   0 if scratch is None else scratch.numel(), _stream()),
               "project_bwd")
        return (v_means, v_scales, None, v_quats, v_V, None, None, None, None, None, None, None, None)


def project_gaussians(means3d: Tensor, scales: Tensor, glob_scale: float, quats: Tensor, viewmat: Tensor,
                      fx: float, fy: float, cx: float, cy: float, img_height: int, img_width: int,
                      block_width: int = TILE, clip_thresh: float = 0.01, *, lin_vel: Optional[Tensor] = None,
                      ang_vel: Optional[Tensor] = None, exposure_time: float = 0.0, rolling_shutter_time: float = 0.0,
                      blur_samples: int = 0):
    """gsplat.project_gaussians (0.1.11 positional signature).
    -> (xys, depths, radii, conics, compensation, num_tiles_hit, cov3d)
    Fork-style trailing keywords (the SpectacularAI fork's rasterizer takes the camera velocities,
    /root/reference/README.md:196-200, train.py:46-70; their exact names are unknown, the defaults are today's static
    behaviour bit for bit): with lin_vel / ang_vel [3] (OpenCV camera frame) the call is the pixel-velocity model's ONE
    projection and returns an 8th tensor pix_vels [N,2] (differentiable; gradients reach viewmat and the velocities);
    radii / num_tiles_hit then describe the tile boxes SWEPT over the blur_samples sample times of exposure_time plus the
    rolling_shutter_time readout, and Gaussians whose static box misses the image keep their centre (they may move in).
    Hand the same keywords and pix_vels to rasterize_gaussians."""
    if lin_vel is None and ang_vel is None:
        return _ProjectGaussians.apply(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height,
                                       img_width, block_width, clip_thresh)
    if lin_vel is None or ang_vel is None:
        raise ValueError("lin_vel and ang_vel go together")
    if block_width != TILE:
        raise ValueError("only block_width=16 is supported")
    from . import velocity_ops as VO
    _, _, span = VO.sample_span(blur_samples, exposure_time, rolling_shutter_time)
    return VO.ProjectGaussiansPixvel.apply(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_height,
                                           img_width, clip_thresh, lin_vel, ang_vel, span)


# --------------------------------------------------------------------------- #
# gsplat.spherical_harmonics
# --------------------------------------------------------------------------- #
class _SphericalHarmonics(Function):
    @staticmethod
    def forward(ctx, degrees_to_use, viewdirs, coeffs):
        viewdirs, coeffs = _f32(viewdirs, "viewdirs"), _f32(coeffs, "coeffs")
        N, K = coeffs.shape[0], coeffs.shape[1]
        colors = torch.empty(N, 3, device=coeffs.device)
        _check(_L().gs_sh_fwd(N, K, int(degrees_to_use), _ptr(viewdirs), _ptr(coeffs), _ptr(colors), _stream()),
               "sh_fwd")
        ctx.save_for_backward(viewdirs)
        ctx.K, ctx.deg = K, int(degrees_to_use)
        return colors

    @staticmethod
    def backward(ctx, v_colors):
        (viewdirs,) = ctx.saved_tensors
        N = viewdirs.shape[0]
        v_coeffs = torch.empty(N, ctx.K, 3, device=viewdirs.device)
        _check(_L().gs_sh_bwd(N, ctx.K, ctx.deg, _ptr(viewdirs), _ptr(v_colors.contiguous().float()),
                              _ptr(v_coeffs), _stream()), "sh_bwd")
        return None, None, v_coeffs


def spherical_harmonics(degrees_to_use: int, viewdirs: Tensor, coeffs: Tensor) -> Tensor:
    """gsplat.spherical_harmonics: coeffs [N,K,3], viewdirs [N,3] -> colors [N,3] (no grad to dirs)."""
    return _SphericalHarmonics.apply(degrees_to_use, viewdirs, coeffs)


# --------------------------------------------------------------------------- #
# gsplat.rasterize_gaussians
# --------------------------------------------------------------------------- #
class _RasterizeGaussians(Function):
    @staticmethod
    def forward(ctx, xys, depths, radii, conics, num_tiles_hit, colors, opacity, img_height, img_width,
                block_width, background, return_alpha):
        if block_width != TILE:
            raise ValueError("only block_width=16 is supported")
        assert colors.shape[-1] == 3          # rasterize_gaussians splits other channel counts into passes of three
        xys, depths, conics = _f32(xys, "xys"), _f32(depths, "depths"), _f32(conics, "conics")
        colors, opacity = _f32(colors, "colors"), _f32(opacity, "opacity").reshape(-1)
        radii = radii.to(torch.int32).contiguous()
        N = xys.shape[0]
        dev = xys.device
        H, W = int(img_height), int(img_width)
        L = _L()
        records = torch.empty(N, REC, device=dev)
        dkeys = torch.empty(N, dtype=torch.int32, device=dev)
        ntiles = torch.empty(N, dtype=torch.int32, device=dev)
        _check(L.gs_pack_records(N, _ptr(xys), _ptr(depths), _ptr(radii), _ptr(conics), _ptr(colors), _ptr(opacity),
                                 H, W, _ptr(records), _ptr(dkeys), _ptr(ntiles), _stream()), "pack_records")
        svals, bins, n_isect, _ = bin_and_sort_records(records, dkeys, ntiles, 1, N, H, W)
        bg = _background(background, dev)
        edges = _band_edges(H, 1, dev)
        out_img = torch.empty(1, H, W, 3, device=dev)
        out_T = torch.empty(1, H, W, device=dev)
        fidx = torch.empty(1, H, W, dtype=torch.int32, device=dev)
        _check(L.gs_rasterize_fwd(_ptr(records), _ptr(svals), _ptr(bins), _ptr(edges), _ptr(bg), 1, 1, H, W,
                                  _ptr(out_img), _ptr(out_T), _ptr(fidx), N if n_isect > 0 else 0, 0,
                                  _stream()), "rasterize_fwd")
        ctx.save_for_backward(records, svals, bins, edges, bg, out_T, fidx)
        ctx.dims = (N, H, W)
        ctx.bg_grad = background is not None and ctx.needs_input_grad[10]
        out = out_img[0]
        if return_alpha:
            return out, 1.0 - out_T[0]
        return out

    @staticmethod
    def backward(ctx, v_img, v_alpha=None):
        records, svals, bins, edges, bg, out_T, fidx = ctx.saved_tensors
        N, H, W = ctx.dims
        dev = records.device
        L = _L()
        v_img = v_img.contiguous().float()
        v_al = None if v_alpha is None else v_alpha.contiguous().float()
        v_records = torch.zeros(N, GRAD, device=dev)
        _check(L.gs_rasterize_bwd(_ptr(records), _ptr(svals), _ptr(bins), _ptr(edges), _ptr(bg), 1, 1, H, W,
                                  _ptr(out_T), _ptr(fidx), _ptr(v_img), _ptr(v_al), _ptr(v_records), N, _bwd_variant(),
                                  _stream()), "rasterize_bwd")
        v_xys = torch.empty(N, 2, device=dev)
        v_conics = torch.empty(N, 3, device=dev)
        v_colors = torch.empty(N, 3, device=dev)
        v_opacity = torch.empty(N, 1, device=dev)
        _check(L.gs_unpack_record_grads(N, _ptr(v_records), _ptr(v_xys), _ptr(v_conics), _ptr(v_colors),
                                        _ptr(v_opacity), _stream()), "unpack grads")
        v_bg = (out_T[0][..., None] * v_img).sum(dim=(0, 1)) if ctx.bg_grad else None
        return (v_xys, None, None, v_conics, None, v_colors, v_opacity, None, None, None, v_bg, None)


def rasterize_gaussians(xys: Tensor, depths: Tensor, radii: Tensor, conics: Tensor, num_tiles_hit: Tensor,
                        colors: Tensor, opacity: Tensor, img_height: int, img_width: int, block_width: int = TILE,
                        background: Optional[Tensor] = None, return_alpha: bool = False, *,
                        pix_vels: Optional[Tensor] = None, exposure_time: float = 0.0, rolling_shutter_time: float = 0.0,
                        blur_samples: int = 0, return_samples: bool = False):
    """gsplat.rasterize_gaussians (0.1.11 positional signature) -> out_img [H,W,C] (, out_alpha [H,W]).
    C = 3 is one pass of the compositor; any other channel count (upstream's nd_rasterize path: depth, features, ...)
    is composited three channels at a time over the same geometry — every channel sees exactly the weights of the
    RGB path, the shared inputs' gradients accumulate over the passes through autograd.
    Fork-style trailing keywords (defaults = the static behaviour bit for bit): pix_vels [N,2] from
    project_gaussians(lin_vel=..., ang_vel=...) with the SAME exposure_time / rolling_shutter_time / blur_samples renders
    the paper's model (/root/reference/README.md:196-200): one binning of the swept boxes, max(1,
    blur_samples) sample times over the exposure plus the per-row readout time inside the compositor.  The result is the
    plain mean of the sample images (linear colour; alpha likewise) or, with return_samples=True, the samples themselves
    [S,H,W,3] (, [S,H,W]) for a gamma-space average (combine_samples).  C must be 3 in this mode."""
    if pix_vels is not None:
        if block_width != TILE:
            raise ValueError("only block_width=16 is supported")
        if colors.shape[-1] != 3:
            raise ValueError("rasterize_gaussians with pix_vels composites 3 channels")
        from . import velocity_ops as VO
        _, times, span = VO.sample_span(blur_samples, exposure_time, rolling_shutter_time)
        img, alpha = VO.RasterizeGaussiansPixvel.apply(xys, depths, radii, conics, colors, opacity.reshape(-1, 1), pix_vels,
                                                       int(img_height), int(img_width), background, times,
                                                       float(rolling_shutter_time), span)
        if not return_samples:
            img, alpha = img.mean(dim=0), alpha.mean(dim=0)
        return (img, alpha) if return_alpha else img
    C = colors.shape[-1]
    if C == 3:
        return _RasterizeGaussians.apply(xys, depths, radii, conics, num_tiles_hit, colors, opacity.reshape(-1, 1),
                                         img_height, img_width, block_width, background, return_alpha)
    if background is not None and background.numel() != C:
        raise ValueError(f"background must have {C} channels")
    imgs, alpha = [], None
    for c0 in range(0, C, 3):
        n = min(3, C - c0)
        col = colors[:, c0:c0 + n]
        bg = None if background is None else background.reshape(-1)[c0:c0 + n]
        if n < 3:
            col = torch.cat([col, col.new_zeros(col.shape[0], 3 - n)], dim=1)
            bg = None if bg is None else torch.cat([bg, bg.new_zeros(3 - n)])
        res = _RasterizeGaussians.apply(xys, depths, radii, conics, num_tiles_hit, col, opacity.reshape(-1, 1),
                                        img_height, img_width, block_width, bg, return_alpha and alpha is None)
        if return_alpha and alpha is None:
            res, alpha = res
        imgs.append(res[..., :n])
    out = torch.cat(imgs, dim=-1)
    return (out, alpha) if return_alpha else out


# --------------------------------------------------------------------------- #
# gsplat utility ops (API parity; 64-bit intersection ids)
# --------------------------------------------------------------------------- #
def compute_cumulative_intersects(num_tiles_hit: Tensor) -> Tuple[int, Tensor]:
    """gsplat.utils.compute_cumulative_intersects -> (num_intersects, INCLUSIVE cumsum int32)"""
    nt = num_tiles_hit.to(torch.int32).contiguous()
    ex, total = exclusive_scan_u32(nt)
    return int(total.item()), ex + nt


def map_gaussian_to_intersects(num_points: int, num_intersects: int, xys: Tensor, depths: Tensor, radii: Tensor,
                               cum_tiles_hit: Tensor, tile_bounds, block_width: int = TILE):
    """gsplat.utils.map_gaussian_to_intersects -> (isect_ids int64 [I], gaussian_ids int32 [I]).
    tile_bounds = (tiles_x, tiles_y, 1) like upstream."""
    dev = xys.device
    isect = torch.zeros(max(1, num_intersects), dtype=torch.int64, device=dev)
    gids = torch.zeros(max(1, num_intersects), dtype=torch.int32, device=dev)
    W, H = int(tile_bounds[0]) * TILE, int(tile_bounds[1]) * TILE
    _check(_L().gs_map_gaussian_to_intersects(int(num_points), _ptr(_f32(xys, "xys")), _ptr(_f32(depths, "depths")),
                                              _ptr(radii.to(torch.int32).contiguous()),
                                              _ptr(cum_tiles_hit.to(torch.int32).contiguous()), H, W, _ptr(isect),
                                              _ptr(gids), _stream()), "map_gaussian_to_intersects")
    return isect[:num_intersects], gids[:num_intersects]


def get_tile_bin_edges(num_intersects: int, isect_ids_sorted: Tensor, tile_bounds) -> Tensor:
    """gsplat.utils.get_tile_bin_edges -> int32 [T,2]"""
    T = int(tile_bounds[0]) * int(tile_bounds[1])
    bins = torch.empty(T, 2, dtype=torch.int32, device=isect_ids_sorted.device)
    _check(_L().gs_tile_bin_edges_u64(int(num_intersects), _ptr(isect_ids_sorted.contiguous()), T, _ptr(bins),
                                      _stream()), "tile_bin_edges")
    return bins


def bin_and_sort_gaussians(num_points: int, num_intersects: int, xys: Tensor, depths: Tensor, radii: Tensor,
                           cum_tiles_hit: Tensor, tile_bounds, block_width: int = TILE):
    """gsplat.utils.bin_and_sort_gaussians ->
    (isect_ids_unsorted, gaussian_ids_unsorted, isect_ids_sorted, gaussian_ids_sorted, tile_bins)"""
    isect, gids = map_gaussian_to_intersects(num_points, num_intersects, xys, depths, radii, cum_tiles_hit,
                                             tile_bounds, block_width)
    T = int(tile_bounds[0]) * int(tile_bounds[1])
    if num_intersects == 0:
        return isect, gids, isect, gids, torch.zeros(T, 2, dtype=torch.int32, device=xys.device)
    ks, vs = radix_sort_pairs(isect.clone(), gids.clone(), 0, 32 + _bits(T))
    bins = get_tile_bin_edges(num_intersects, ks, tile_bounds)
    return isect, gids, ks, vs, bins


# --------------------------------------------------------------------------- #
# sub-pose viewmats (SE(3) screw interpolation)
# --------------------------------------------------------------------------- #
class _SubposeViewmats(Function):
    @staticmethod
    def forward(ctx, viewmat, lin_vel, ang_vel, times):
        V = _viewmat16(viewmat)
        lin, ang, times = _f32(lin_vel, "lin_vel"), _f32(ang_vel, "ang_vel"), _f32(times, "times")
        P = times.numel()
        out = torch.empty(P, 4, 4, device=V.device)
        _check(_L().gs_subpose_viewmats_fwd(P, _ptr(V), _ptr(lin), _ptr(ang), _ptr(times), _ptr(out), _stream()),
               "subpose_viewmats_fwd")
        ctx.save_for_backward(V, lin, ang, times)
        return out

    @staticmethod
    def backward(ctx, v_out):
        V, lin, ang, times = ctx.saved_tensors
        P = times.numel()
        dev = V.device
        acc = torch.zeros(22, device=dev)                  # one fill for the three accumulators
        v_V, v_lin, v_ang = acc[:16].view(4, 4), acc[16:19], acc[19:22]
        _check(_L().gs_subpose_viewmats_bwd(P, _ptr(V), _ptr(lin), _ptr(ang), _ptr(times),
                                            _ptr(v_out.contiguous().float()), _ptr(v_V), _ptr(v_lin), _ptr(v_ang),
                                            _stream()), "subpose_viewmats_bwd")
        return v_V, v_lin, v_ang, None


def subpose_viewmats(viewmat: Tensor, lin_vel: Tensor, ang_vel: Tensor, times: Tensor) -> Tensor:
    """viewmat(t) = Exp(-t [lin_vel; ang_vel]) @ viewmat for every t in times -> [P,4,4] (differentiable)."""
    return _SubposeViewmats.apply(viewmat, lin_vel, ang_vel, times)


def subpose_schedule(blur_samples: int, exposure_time: float, rs_bands: int, rolling_shutter_time: float):
    """Host-side schedule: (times [P], sample index [P], band index [P]); p = s*R + r.
    t = ((s+.5)/S - .5) * exposure + ((r+.5)/R - .5) * readout."""
    S, R = max(1, int(blur_samples)), max(1, int(rs_bands))
    times, samp, band = [], [], []
    for s in range(S):
        ts = ((s + 0.5) / S - 0.5) * exposure_time if S > 1 else 0.0
        for r in range(R):
            tr = ((r + 0.5) / R - 0.5) * rolling_shutter_time if R > 1 else 0.0
            times.append(ts + tr)
            samp.append(s)
            band.append(r)
    return times, samp, band


# --------------------------------------------------------------------------- #
# fused multi-sub-pose render
# --------------------------------------------------------------------------- #
class _RenderSubposes(Function):
    @staticmethod
    def forward(ctx, means3d, scales, quats, opacities, sh, viewmats, background, S, R, fx, fy, cx, cy,
                img_height, img_width, sh_degree, antialiased, glob_scale, clip_thresh, xy_grad_out, return_alpha,
                gamma, min_rgb_level, lin_vel=None, ang_vel=None, times=None, return_depth=False, rs_time=0.0,
                sh_rest=None, param_flags=0, shared_list=False, hints=None):
        # an output the loss does not use arrives as None in backward instead of a materialised zero tensor
        ctx.set_materialize_grads(False)
        means3d, scales, quats = _f32(means3d, "means3d"), _f32(scales, "scales"), _f32(quats, "quats")
        opacities, sh = _f32(opacities, "opacities").reshape(-1), _f32(sh, "sh")
        # raw splatfacto parameters (param_flags bit 0: log-scales, bit 1: opacity logits; sh_rest: features_rest beside
        # sh = features_dc): activations, their backward and the SH concatenation happen inside the projection kernels
        param_flags = int(param_flags)
        if sh_rest is not None:
            sh_rest = _f32(sh_rest, "sh_rest")
            if sh.reshape(sh.shape[0], -1).shape[1] != 3 or sh_rest.dim() != 3 or sh_rest.shape[0] != sh.shape[0]:
                raise ValueError("with sh_rest [N,K-1,3], sh must be features_dc [N,3] (or [N,1,3])")
        if xy_grad_out is not None:
            if (xy_grad_out.shape != (means3d.shape[0], 2) or xy_grad_out.dtype != torch.float32
                    or not xy_grad_out.is_contiguous() or xy_grad_out.device != means3d.device):
                raise ValueError("xy_grad_out must be a contiguous float32 [N,2] tensor on the Gaussians' device")
        ctx.xy_grad_out = xy_grad_out
        N, K = means3d.shape[0], (sh.shape[1] if sh_rest is None else 1 + sh_rest.shape[1])
        P = S * R
        if P > MAX_SUBPOSES:
            raise ValueError(f"{S} blur samples x {R} row bands = {P} sub-poses per frame; the slice descriptors travel "
                             f"in kernel arguments and hold at most {MAX_SUBPOSES} (kMaxSubposes, csrc/binning.hip)")
        # pixel-velocity model: ONE mid-exposure viewmat + the camera twist + the P sub-pose times
        pixvel = times is not None
        rs_time = float(rs_time or 0.0)
        if rs_time != 0.0 and (not pixvel or R != 1):
            raise ValueError("exact rolling shutter (rolling_shutter_time != 0) needs the pixel-velocity model "
                             "(times / lin_vel / ang_vel) and rs_bands == 1")
        # shared list (pixel-velocity model, rs_bands == 1): ONE record set, depth sort and tile list for the frame —
        # the splats at the centre of the sampled time span, tile boxes swept over the whole span (+ readout); the
        # compositor evaluates sample s at xy + (times[s] - centre + tau(y)) * pixel_velocity
        shared = None
        if shared_list:
            if not pixvel or R != 1:
                raise ValueError("shared_list needs the pixel-velocity model (times / lin_vel / ang_vel) and rs_bands == 1")
            tl = [float(t) for t in (times.reshape(-1).tolist() if isinstance(times, Tensor) else times)]
            if len(tl) != S:
                raise ValueError(f"times must hold {S} sample times")
            t_c = 0.5 * (min(tl) + max(tl))
            shared = (torch.tensor([t - t_c for t in tl], dtype=torch.float32, device=means3d.device),
                      (max(tl) - min(tl)) + abs(rs_time), (min(tl) - t_c, max(tl) - t_c))
            times = torch.tensor([t_c], dtype=torch.float32, device=means3d.device)
            P = 1
        if pixvel:
            V = _viewmat16(viewmats).reshape(4, 4)
            twist = torch.cat([_f32(lin_vel, "lin_vel").reshape(3), _f32(ang_vel, "ang_vel").reshape(3)]).contiguous()
            times = _f32(times, "times").reshape(-1)
            if times.numel() != P:
                raise ValueError(f"times must hold {P} sub-pose times")
        else:
            V = _f32(viewmats, "viewmats")
            twist = None
            if V.shape != (P, 4, 4):
                raise ValueError(f"viewmats must be [{P},4,4]")
        H, W = int(img_height), int(img_width)
        dev = means3d.device
        L = _L()
        records = torch.empty(P * N, REC, device=dev)
        dkeys = torch.empty(P * N, dtype=torch.int32, device=dev)
        ntiles = torch.empty(P * N, dtype=torch.int32, device=dev)
        radii = torch.empty(P, N, dtype=torch.int32, device=dev)
        args = (N, P, float(glob_scale), K, int(sh_degree), float(fx), float(fy), float(cx), float(cy), H, W,
                float(clip_thresh), int(bool(antialiased)))
        # bit 1: culled (Gaussian, sub-pose) pairs get no record at all.  Safe when nothing downstream looks at them:
        # the compacting pre-sort never ranks them and the tuple backward only visits rows the compositor touched
        # (the atomics backward of the pixel-velocity model tells "covers no tile" by an all-zero record)
        # bit 0: SH colour deferred to the depth slices (gs_slice_colors colours only what a slice emits)
        backend = frame_backend if (frame_backend is not None and not frame_backend.native_ok()) else None
        defer_flags = 3 if backend is None else backend.defer_flags()
        # bit 2 + R in bits 8..23: band-aware projection (a (band, Gaussian) pair whose tile rows miss the band is culled
        # before the depth pre-sort instead of being keyed, sorted, scanned and planned for nothing)
        if backend is None and R > 1 and BAND_AWARE:
            defer_flags |= 4 | (R << 8)
        if backend is None and hints is None:
            # default owner of the frame-to-frame hints: the scene's shape AND its parameter storage, so that two
            # scenes of one shape do not share a budget / arena estimate
            hints = hints_for((str(dev), N, P, S, H, W, shared is not None, means3d.untyped_storage().data_ptr()))
        # lazy records (see LAZY_RECORDS): SE(3) sub-poses through the library's frame path, planned slices, and a scene
        # whose frames have so far stopped within the default budget
        lazy = None
        if (backend is None and not pixvel and LAZY_RECORDS and hints.slice_base() > 0
                and (LAZY_RECORDS == 2 or hints.lazy_records())):
            defer_flags |= 16
        if shared is not None and backend is not None:
            raise ValueError("the shared-list mode runs through the library's frame path only")
        pix_vel = torch.empty(N, 2, device=dev) if (rs_time != 0.0 or shared is not None) else None
        rs = None if pix_vel is None else (pix_vel, rs_time) + ((shared[0], shared[2]) if shared is not None else ())
        box_sweep = shared[1] if shared is not None else rs_time      # what the projection widens the tile boxes by
        ctx.rs = rs

        def _project():
            if pixvel:
                _check(L.gs_project_pixvel_fwd(N, P, _ptr(means3d), _ptr(scales), args[2], _ptr(quats), _ptr(opacities),
                                               _ptr(sh), K, args[4], _ptr(V), _ptr(twist), _ptr(times), args[5], args[6],
                                               args[7], args[8], H, W, args[11], args[12], defer_flags,
                                               _ptr(records), _ptr(dkeys), _ptr(ntiles), _ptr(radii), box_sweep,
                                               _ptr(pix_vel), _ptr(sh_rest), param_flags, _stream()),
                       "project_pixvel_fwd")
            else:
                _check(L.gs_project_fused_fwd(N, P, _ptr(means3d), _ptr(scales), args[2], _ptr(quats), _ptr(opacities),
                                              _ptr(sh), K, args[4], _ptr(V), args[5], args[6], args[7], args[8], H, W,
                                              args[11], args[12], defer_flags, _ptr(records), _ptr(dkeys),
                                              _ptr(ntiles), _ptr(radii), _ptr(sh_rest), param_flags, _stream()),
                       "project_fused_fwd")

        with _stage("project_fwd"):
            _project()
        if defer_flags & 16:
            lazy = _ProjectInputs(means3d.data_ptr(), scales.data_ptr(), quats.data_ptr(), opacities.data_ptr(),
                                  V.data_ptr(), args[2], args[5], args[6], args[7], args[8], args[11], args[12],
                                  defer_flags & ~16, param_flags)
        bg = _background(background, dev)
        edges = _band_edges(H, R, dev)
        # deferred colour: the view direction of every sub-pose (pixel-velocity model: the mid-exposure pose for all)
        V_col = V.reshape(1, 16).expand(P, 16).contiguous() if pixvel else V
        color = (means3d, sh, sh_rest, K, args[4], V_col) if (defer_flags & 1) else None
        # optional fourth channel: sum of weight * camera-space depth per sample image (forward only)
        depth_acc = torch.zeros(S, H, W, device=dev) if return_depth else None
        ctx.prealloc = None
        ctx.frame = None
        ctx.backend = backend
        if backend is None:
            if rs is not None and R != 1:
                raise ValueError("exact rolling shutter renders with rs_bands == 1")
            # the backward's frame-sized buffers (and the one fill among them) are issued BEFORE the frame: behind the
            # frame's last read-back nothing but the averaging and the backward's own launches are left for the host
            if any(ctx.needs_input_grad):
                # ONE zero fill for what this node's backward accumulates into: touched flags [P*N] u8 | 16 P + 12 floats
                # of view-matrix / twist gradients
                t_len = (P * N + 15) // 16 * 16
                zbuf = torch.zeros(t_len + 4 * (16 * P + 12), dtype=torch.uint8, device=dev)
                zf = zbuf[t_len:].view(torch.float32)
                ctx.prealloc = {"touched": zbuf[:P * N], "v_records": torch.empty(P * N, GRAD, device=dev),
                                "v_V": zf[:16 * P], "v_tw": zf[16 * P:16 * P + 12],
                                "pose_scratch": _pose_scratch(N, P, True, dev)}
            # fused sub-frame averaging: the library launches it behind the last compositor (below: `averaged`)
            averaged = None
            if gamma is not None:
                averaged = (float(gamma), float(min_rgb_level) / 255.0, torch.empty(H, W, 3, device=dev))
            retries = 0
            for attempt in range(_ARENA_ATTEMPTS):
                try:
                    out_img, out_T, ctx.frame = native_frame_forward(records, dkeys, ntiles, P, N, S, R, H, W, bg, edges,
                                                                     hints.slice_base(), color, depth_acc,
                                                                     any(ctx.needs_input_grad), rs, averaged, hints,
                                                                     bool(defer_flags & 4), lazy)
                    hints.feedback(int(ctx.frame["state"].n_slices), retries, _box_share(ctx.frame["state"]),
                                   int(ctx.frame["state"].depth_select), float(ctx.frame["state"].open_after_first),
                                   int(ctx.frame["state"].max_selected), int(ctx.frame["state"].select_overflow))
                    break
                except _ArenaTooSmall:
                    retries += 1
                    if attempt == _ARENA_ATTEMPTS - 1:
                        raise _lib.HipLibraryError("frame_forward: the arena estimate did not converge")
                    # the depth keys were consumed by the pre-sort: project again, then retry with the larger arena
                    if depth_acc is not None:
                        depth_acc.zero_()
                    with _stage("project_fwd"):
                        _project()
            slices = []
        else:
            ctx.prealloc = {} if any(ctx.needs_input_grad) else None
            out_img, out_T, slices = backend.sliced_forward(records, dkeys, ntiles, P, N, S, R, H, W, bg, edges, SLICE_BASE,
                                                            color, depth_acc, ctx.prealloc, rs)
        ctx.slices = slices
        svals = bins = fidx = _placeholder_i32(dev)       # nothing to keep: the slices hold their own lists
        n_isect = last_num_intersects
        ctx.combine = None
        first = cmb_samples = cmb_rgb = out_img
        if gamma is not None:
            # fused sub-frame averaging: output 0 is the averaged image; backward never materialises the
            # per-sample gradients (the compositor's backward derives them per pixel)
            m = float(min_rgb_level) / 255.0
            if backend is None:
                first = cmb_rgb = averaged[2]
            else:
                first = cmb_rgb = torch.empty(H, W, 3, device=dev)
                with _stage("combine"):
                    _check(L.gs_combine_fwd(S, H * W * 3, _ptr(out_img), float(gamma), m, _ptr(first), _stream()),
                           "combine_fwd")
            ctx.combine = (float(gamma), m)
        else:
            cmb_samples = cmb_rgb = svals          # placeholders: nothing to keep
        ctx.pixvel = (twist, times) if pixvel else None
        ctx.param_flags = param_flags
        ctx.sh_rest = sh_rest
        ctx.save_for_backward(means3d, scales, quats, opacities, sh, V, records, svals, bins, edges, bg, out_T, fidx,
                              cmb_samples, cmb_rgb)
        ctx.args = args
        ctx.SR = (S, R)
        ctx.n_isect = n_isect
        ctx.bg_grad = background is not None and ctx.needs_input_grad[6]
        ctx.mark_non_differentiable(radii)
        if depth_acc is not None:
            ctx.mark_non_differentiable(depth_acc)
        ctx.img_shape = (S, H, W, 3) if gamma is None else (H, W, 3)
        return first, (1.0 - out_T) if return_alpha else None, radii, depth_acc

    @staticmethod
    def backward(ctx, v_img, v_alpha, _v_radii, _v_depth=None):
        (means3d, scales, quats, opacities, sh, V, records, svals, bins, edges, bg, out_T, fidx, cmb_samples,
         cmb_rgb) = ctx.saved_tensors
        N, P, glob, K, deg, fx, fy, cx, cy, H, W, clip, aa = ctx.args
        S, R = ctx.SR
        dev = means3d.device
        L = _L()
        if v_img is None and v_alpha is None:
            return (None,) * 32
        v_img = torch.zeros(ctx.img_shape, device=dev) if v_img is None else v_img.contiguous().float()
        v_al = None if v_alpha is None else v_alpha.contiguous().float()
        combine = None
        if ctx.combine is not None:
            samples, rgb = cmb_samples, cmb_rgb
            gamma, m = ctx.combine
            if ctx.bg_grad:
                # a learnable background needs the per-sample gradients themselves: two-step backward
                v_samples = torch.empty_like(samples)
                _check(L.gs_combine_bwd(S, rgb.numel(), _ptr(samples), gamma, m, _ptr(rgb), _ptr(v_img),
                                        _ptr(v_samples), _stream()), "combine_bwd")
                v_img = v_samples
            else:
                scale = torch.empty_like(rgb)
                with _stage("combine"):
                    _check(L.gs_combine_bwd_scale(S, rgb.numel(), gamma, _ptr(rgb), _ptr(v_img), _ptr(scale),
                                                  _stream()), "combine_bwd_scale")
                combine = (scale, gamma, m)
                v_img = samples
        # atomic-free path: only Gaussians the compositor touched get a gradient record (plain stores) and a
        # `touched` flag; the projection backward skips everything else, so v_records needs no 240 MB memset
        all_tuples = ctx.frame is not None or all(sl["gi_of_e"] is not None for sl in ctx.slices)
        pre = ctx.prealloc if ctx.prealloc else {}
        ctx.prealloc = None
        if all_tuples and "touched" in pre:
            v_records, touched = pre["v_records"], pre["touched"]
        elif all_tuples:
            v_records = torch.empty(P * N, GRAD, device=dev)
            touched = torch.zeros(P * N, dtype=torch.uint8, device=dev)
        else:
            v_records = torch.zeros(P * N, GRAD, device=dev)
            touched = None

        if ctx.frame is not None:
            native_frame_backward(ctx.frame, records, bg, edges, out_T, v_img, v_al, v_records, touched, combine)
        else:
            ctx.backend.sliced_backward(records, ctx.slices, S, R, H, W, bg, edges, out_T, v_img, v_al, v_records, touched,
                                        combine, ctx.rs)
        # the dense gradient outputs are carved out of ONE buffer; with touched flags the projection backward zero-fills
        # its own outputs (grad flag 32): no fill launches (round 3: one 236 MB fill per step)
        sh_rest = ctx.sh_rest
        sizes = [3 * N, 3 * N, 4 * N, N] + ([3 * K * N] if sh_rest is None else [3 * N, 3 * (K - 1) * N])
        shapes = [(N, 3), (N, 3), (N, 4), (N,)] + ([(N, K, 3)] if sh_rest is None else [tuple(sh.shape), (N, K - 1, 3)])
        flat = torch.empty(sum(sizes), device=dev)
        outs = [t.view(shape) for t, shape in zip(flat.split(sizes), shapes)]
        v_means, v_scales, v_quats, v_opac, v_sh = outs[:5]
        v_sh_rest = outs[5] if sh_rest is not None else None
        need_v = ctx.needs_input_grad[5]
        xy_out = ctx.xy_grad_out
        fill_flag = 32 if touched is not None else 0
        pf = ctx.param_flags
        v_lin = v_ang = None
        # scratch of the ordered camera-gradient reduction (the sparse form runs when touched flags exist)
        psc = pre.get("pose_scratch") if touched is not None else None
        if psc is None and (need_v or ctx.needs_input_grad[23] or ctx.needs_input_grad[24]):
            psc = _pose_scratch(N, P, touched is not None, dev)
        psc_n = 0 if psc is None else psc.numel()
        with _stage("project_bwd"):
            if ctx.pixvel is not None:
                twist, times = ctx.pixvel
                v_V = (pre["v_V"][:16].view(4, 4) if "v_V" in pre else torch.zeros(4, 4, device=dev)) if need_v else None
                need_tw = ctx.needs_input_grad[23] or ctx.needs_input_grad[24]
                v_tw = (pre["v_tw"] if "v_tw" in pre else torch.zeros(12, device=dev)) if need_tw else None
                _check(L.gs_project_pixvel_bwd(N, P, _ptr(means3d), _ptr(scales), glob, _ptr(quats), _ptr(opacities),
                                               _ptr(sh), K, deg, _ptr(V), _ptr(twist), _ptr(times), fx, fy, cx, cy, H, W,
                                               clip, aa, _ptr(records), _ptr(v_records), _ptr(v_means), _ptr(v_scales),
                                               _ptr(v_quats), _ptr(v_opac), _ptr(v_sh), _ptr(v_V), _ptr(v_tw),
                                               _ptr(touched), _ptr(xy_out),
                                               _proj_grad_flags() | (16 if ctx.rs is not None else 0) | fill_flag,
                                               _ptr(sh_rest), pf, _ptr(v_sh_rest), _ptr(psc), psc_n, _stream()),
                       "project_pixvel_bwd")
                if v_tw is not None:
                    v_lin, v_ang = v_tw[0:3], v_tw[3:6]
            else:
                v_V = (pre["v_V"].view(P, 4, 4) if "v_V" in pre else torch.zeros(P, 4, 4, device=dev)) if need_v else None
                _check(L.gs_project_fused_bwd(N, P, _ptr(means3d), _ptr(scales), glob, _ptr(quats), _ptr(opacities),
                                              _ptr(sh), K, deg, _ptr(V), fx, fy, cx, cy, H, W, clip, aa, _ptr(records),
                                              _ptr(v_records), _ptr(v_means), _ptr(v_scales), _ptr(v_quats),
                                              _ptr(v_opac), _ptr(v_sh), _ptr(v_V), _ptr(touched), _ptr(xy_out),
                                              _proj_grad_flags() | fill_flag, _ptr(sh_rest), pf, _ptr(v_sh_rest),
                                              _ptr(psc), psc_n, _stream()), "project_fused_bwd")
        v_bg = (out_T[..., None] * v_img).sum(dim=(0, 1, 2)) if ctx.bg_grad else None
        return ((v_means, v_scales, v_quats, v_opac, v_sh, v_V, v_bg) + (None,) * 16
                + (v_lin, v_ang, None, None, None, v_sh_rest, None, None, None))


def render_subposes(means3d: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, sh: Tensor,
                    viewmats: Tensor, background: Optional[Tensor], blur_samples: int, rs_bands: int,
                    fx: float, fy: float, cx: float, cy: float, img_height: int, img_width: int,
                    sh_degree: int = 3, antialiased: bool = True, glob_scale: float = 1.0,
                    clip_thresh: float = 0.01, xy_grad_out: Optional[Tensor] = None, return_alpha: bool = True,
                    lin_vel: Optional[Tensor] = None, ang_vel: Optional[Tensor] = None,
                    times: Optional[Tensor] = None, return_depth: bool = False, rolling_shutter_time: float = 0.0,
                    sh_rest: Optional[Tensor] = None, raw_params: bool = False, shared_list: bool = False,
                    hints: Optional[FrameHints] = None):
    """Fused hot path: project N Gaussians under P=S*R sub-pose viewmats, bin, sort, composite.
    -> (samples [S,H,W,3], alphas [S,H,W], radii int32 [P,N]).  scales/opacities are activated values — or, with
    raw_params=True, splatfacto's RAW parameters: log-scales and opacity logits (exp / sigmoid and their backward run
    inside the projection kernels); sh_rest [N,K-1,3] beside sh = features_dc [N,3] spares the concatenation.
    xy_grad_out (optional float32 [N,2]) is OVERWRITTEN during backward with the sum over the sub-poses of
    the screen-space centre gradient in pixels — what splatfacto's densification reads from ``xys.grad``.
    return_alpha=False returns None for alphas (as gsplat's rasterize_gaussians does by default).
    Pixel-velocity model (the paper's first-order blur / rolling-shutter model): pass `times` [P] together with
    lin_vel / ang_vel [3] (OpenCV camera frame) and ONE mid-exposure viewmat [4,4] as `viewmats`; every Gaussian is
    projected once and sub-pose p renders it at xy + times[p] * pixel_velocity (gradients reach viewmat and twist).
    return_depth=True appends a 4th result [S,H,W]: per sample the sum over the blended splats of weight *
    camera-space depth (no gradient); expected depth = that / alpha (splatfacto's outputs["depth"]).
    rolling_shutter_time != 0 (pixel-velocity model, rs_bands == 1, times = the S blur-sample times): EXACT per-row
    rolling shutter — pixel row y sees every splat at xy + (times[s] + tau(y)) * pixel_velocity with
    tau(y) = ((y + 0.5) / H - 0.5) * rolling_shutter_time; one projection / sort / list per blur sample, whatever H.
    shared_list=True (pixel-velocity model, rs_bands == 1, any rolling_shutter_time): ONE record set, sort and tile list
    for the whole frame (tile boxes swept over the sampled span); the S samples walk it.  radii is then [1,N].  The
    footprint of a splat is cut at the swept box instead of the per-sample box: contributions beyond 3 sigma inside
    the swept box (alpha between 1/255 and opacity * exp(-4.5)) are kept, so values differ from the per-sample lists
    by that fringe.
    hints: the caller's FrameHints (adaptive slice budget + arena estimate of ITS scene); None = one per (device, frame
    shape, storage of means3d)."""
    S, R = max(1, int(blur_samples)), max(1, int(rs_bands))
    out = _RenderSubposes.apply(means3d, scales, quats, opacities, sh, viewmats, background, S, R, fx, fy, cx, cy,
                                img_height, img_width, sh_degree, antialiased, glob_scale, clip_thresh, xy_grad_out,
                                bool(return_alpha), None, None, lin_vel, ang_vel, times, bool(return_depth),
                                float(rolling_shutter_time), sh_rest, 3 if raw_params else 0,
                                bool(shared_list), hints)
    return out if return_depth else out[:3]


def render_combined(means3d: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, sh: Tensor,
                    viewmats: Tensor, background: Optional[Tensor], blur_samples: int, rs_bands: int,
                    fx: float, fy: float, cx: float, cy: float, img_height: int, img_width: int,
                    gamma: float = 1.0, min_rgb_level: float = 0.0, sh_degree: int = 3, antialiased: bool = True,
                    glob_scale: float = 1.0, clip_thresh: float = 0.01, xy_grad_out: Optional[Tensor] = None,
                    return_alpha: bool = True, lin_vel: Optional[Tensor] = None, ang_vel: Optional[Tensor] = None,
                    times: Optional[Tensor] = None, return_depth: bool = False, rolling_shutter_time: float = 0.0,
                    sh_rest: Optional[Tensor] = None, raw_params: bool = False, shared_list: bool = False,
                    hints: Optional[FrameHints] = None):
    """render_subposes + combine_samples as ONE autograd node: -> (rgb [H,W,3], alphas [S,H,W] or None, radii).
    Same values as the two-step form; the backward skips the [S,H,W,3] per-sample gradient tensor — the
    compositor's backward derives every pixel's sample gradient from rgb and its gradient."""
    S, R = max(1, int(blur_samples)), max(1, int(rs_bands))
    out = _RenderSubposes.apply(means3d, scales, quats, opacities, sh, viewmats, background, S, R, fx, fy, cx, cy,
                                img_height, img_width, sh_degree, antialiased, glob_scale, clip_thresh, xy_grad_out,
                                bool(return_alpha), float(gamma), float(min_rgb_level), lin_vel, ang_vel, times,
                                bool(return_depth), float(rolling_shutter_time), sh_rest, 3 if raw_params else 0,
                                bool(shared_list), hints)
    return out if return_depth else out[:3]


# --------------------------------------------------------------------------- #
# sub-frame averaging
# --------------------------------------------------------------------------- #
class _CombineSamples(Function):
    @staticmethod
    def forward(ctx, samples, gamma, min_rgb_level):
        samples = _f32(samples, "samples")
        S = samples.shape[0]
        n = samples[0].numel()
        out = torch.empty_like(samples[0])
        m = float(min_rgb_level) / 255.0
        _check(_L().gs_combine_fwd(S, n, _ptr(samples), float(gamma), m, _ptr(out), _stream()), "combine_fwd")
        ctx.save_for_backward(samples, out)
        ctx.gm = (float(gamma), m)
        return out

    @staticmethod
    def backward(ctx, v_out):
        samples, out = ctx.saved_tensors
        gamma, m = ctx.gm
        S = samples.shape[0]
        n = samples[0].numel()
        v_samples = torch.empty_like(samples)
        _check(_L().gs_combine_bwd(S, n, _ptr(samples), gamma, m, _ptr(out), _ptr(v_out.contiguous().float()),
                                   _ptr(v_samples), _stream()), "combine_bwd")
        return v_samples, None, None


def combine_samples(samples: Tensor, gamma: float = 1.0, min_rgb_level: float = 0.0) -> Tensor:
    """[S,...] per-sample composites -> ( mean_k max(C_k, min/255)^gamma )^(1/gamma)"""
    return _CombineSamples.apply(samples, gamma, min_rgb_level)
