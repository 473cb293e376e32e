"""Per-stage timing of the frame path with HIP events (enabled by bench.py; off by default, costs nothing then)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib

# stage names of gs_frame_forward / gs_frame_backward's own HIP events, in the library's order (include/gsdeblur.h)
FRAME_STAGES = ("depth_sort", "count_scan", "slice_plan", "slice_count", "emit", "tile_sort", "bin_edges", "raster_fwd",
                "slice_sat", "raster_bwd", "grad_reduce")


class StageProfiler:
    """Optional per-stage timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  Enabled by bench.py; `None` (default) costs nothing."""

    def __init__(self, only=None):
        self.events = {}
        self.native = {}                                     # stage -> [ms] drained from the library's own events
        self.only = None if only is None else set(only)     # restrict to these stage names (others cost nothing)
        if _lib._lib is not None:
            _lib._lib.gs_frame_profile_read(0, None, None)   # forget event pairs of an earlier profiler

    class _Ctx:
        def __init__(self, prof, name):
            self.prof, self.name = prof, name

        def __enter__(self):
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
            return self

        def __exit__(self, *exc):
            self.b.record()
            self.prof.events.setdefault(self.name, []).append((self.a, self.b))
            return False

    def stage(self, name):
        if self.only is not None and name not in self.only:
            return _NULL
        return StageProfiler._Ctx(self, name)

    def summary_ms(self):
        torch.cuda.synchronize()
        out = {k: [a.elapsed_time(b) for a, b in v] for k, v in self.events.items()}
        # stages issued by the library itself (gs_frame_forward / gs_frame_backward record their own HIP events)
        L = _lib.load()
        cap = 1 << 16
        ids, ms = (ctypes.c_int * cap)(), (ctypes.c_float * cap)()
        n = L.gs_frame_profile_read(cap, ids, ms)
        for i in range(n):
            self.native.setdefault(FRAME_STAGES[ids[i]], []).append(float(ms[i]))
        for k, v in self.native.items():
            out.setdefault(k, []).extend(v)
        return out


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _Null()
