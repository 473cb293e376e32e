"""ctypes binding of the C-ABI HIP library (include/gsdeblur.h).

There is no CPU fallback: if the library cannot be loaded every op raises.
"""
from __future__ import annotations

import ctypes
from pathlib import Path

from ._build import LIB_PATH, build_library

_P, _I, _F, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_longlong

# name -> argtypes (restype int unless noted); mirrors include/gsdeblur.h
_SIGS = {
    "gs_subpose_viewmats_fwd": [_I, _P, _P, _P, _P, _P, _P],
    "gs_subpose_viewmats_bwd": [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "gs_project_fwd": [_I, _P, _P, _F, _P, _P, _F, _F, _F, _F, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "gs_project_bwd": [_I, _P, _P, _F, _P, _P, _F, _F, _F, _F, _I, _I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _L, _P],
    "gs_sh_fwd": [_I, _I, _I, _P, _P, _P, _P],
    "gs_sh_bwd": [_I, _I, _I, _P, _P, _P, _P],
    "gs_project_fused_fwd": [_I, _I, _P, _P, _F, _P, _P, _P, _I, _I, _P, _F, _F, _F, _F, _I, _I, _F, _I, _I,
                             _P, _P, _P, _P, _P, _I, _P],
    "gs_slice_colors": [_I, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P],
    "gs_project_fused_bwd": [_I, _I, _P, _P, _F, _P, _P, _P, _I, _I, _P, _F, _F, _F, _F, _I, _I, _F, _I,
                             _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _L, _P],
    "gs_project_pixvel_fwd": [_I, _I, _P, _P, _F, _P, _P, _P, _I, _I, _P, _P, _P, _F, _F, _F, _F, _I, _I, _F, _I, _I,
                              _P, _P, _P, _P, _F, _P, _P, _I, _P],
    "gs_rasterize_fwd_rs_slice": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P, _I, _P, _P, _P, _I, _F, _P,
                                  _P],
    "gs_rasterize_bwd_rs_slice": [_P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _F, _F,
                                  _P, _I, _F, _P, _P],
    "gs_project_pixvel_bwd": [_I, _I, _P, _P, _F, _P, _P, _P, _I, _I, _P, _P, _P, _F, _F, _F, _F, _I, _I, _F, _I,
                              _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _L, _P],
    "gs_pack_records": [_I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "gs_unpack_record_grads": [_I, _P, _P, _P, _P, _P, _P],
    "gs_exclusive_scan_u32": [_L, _P, _P, _P, _P, _L, _P],
    "gs_radix_sort_pairs_u32": [_L, _P, _P, _P, _P, _I, _I, _I, _P, _L, ctypes.POINTER(_I), _P],
    "gs_radix_sort_pairs_u64": [_L, _P, _P, _P, _P, _I, _I, _I, _P, _L, ctypes.POINTER(_I), _P],
    "gs_radix_sort_pairs_gather_u32": [_L, _P, _P, _P, _P, _I, _I, _I, _P, _L, ctypes.POINTER(_I), _P, _P, _P, _P],
    "gs_radix_sort_pairs_carry_u32": [_L, _P, _P, _P, _P, _I, _I, _I, _P, _L, ctypes.POINTER(_I), _P, _P, _P,
                                      ctypes.POINTER(_I), _P, _P],
    "gs_segmented_sort_pairs_u32": [_L, _L, _P, _P, _P, _P, _I, _I, _I, _P, _L, ctypes.POINTER(_I), _P],
    "gs_segmented_sort_compact_u32": [_L, _L, _P, _P, _P, _P, _I, _I, _I, ctypes.c_uint, _P, _P, _P, _P, _L,
                                      ctypes.POINTER(_I), _P],
    "gs_segmented_sort_select_u32": [_L, _L, _P, _P, _P, _P, _P, _I, _I, _I, ctypes.c_uint, _P, _P, _P, _P, _P, _P, _L,
                                     ctypes.POINTER(_I), _L, _P],
    "gs_depth_select": [_L, _L, _P, _P, _L, _P, _P, _P, _L, _P],
    "gs_exclusive_scan_segments_u32": [_L, _L, _P, _P, _P, _P, _P, _L, _P],
    "gs_make_depth_keys64": [_L, _I, _P, _P, _P],
    "gs_gather_counts": [_L, _P, _P, _P, _P],
    "gs_emit_intersects": [_L, _I, _I, _I, _P, _P, _P, _L, _P, _P, ctypes.c_uint, _P],
    "gs_tile_bin_edges_u32": [_L, _P, _I, _P, _P, _P],
    "gs_tile_bin_edges_u64": [_L, _P, _I, _P, _P],
    "gs_tile_bin_edges_ids_u32": [_L, _P, _I, _P, _P, _P, _P, _P],
    "gs_map_gaussian_to_intersects": [_I, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "gs_rasterize_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P],
    "gs_rasterize_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P],
    "gs_slice_plan": [_I, _I, _I, _P, _P, _L, _P, _P, _P, _P, _P, _P],
    "gs_slice_plan_select": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "gs_tile_open_sat": [_I, _I, _I, _P, _P, _P, _P, _P],
    "gs_slice_counts": [_I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "gs_emit_open_intersects": [_I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, ctypes.c_uint, _I, _I, _P, _P, _P, _P],
    "gs_slice_counts_exact": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P],
    "gs_slice_counts_exact_swept": [_I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _F, _F, _F,
                                    _P],
    "gs_rasterize_fwd_slice": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P,
                               _I, _P],
    "gs_rasterize_bwd_slice": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P,
                               _I, _P, _F, _F, _P],
    "gs_rasterize_fwd_slice_stats": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "gs_reduce_grad_tuples": [_I, _P, _P, _P, _P, _P, _P, _P, _L, _P, _I, _P],
    "gs_combine_fwd": [_I, _L, _P, _F, _F, _P, _P],
    "gs_combine_bwd": [_I, _L, _P, _F, _F, _P, _P, _P, _P],
    "gs_combine_bwd_scale": [_I, _L, _F, _P, _P, _P, _P],
    # host arrays (pointer table, widths) are passed as ctypes arrays -> plain pointers
    "gs_dp_row_mask": [_I, _I, _P, _P, _P, _P],
    "gs_dp_pack_rows": [_L, _P, _I, _P, _P, _P, _P],
    "gs_dp_scatter_add_rows": [_L, _P, _I, _P, _P, _F, _P],
    "gs_dp_pack_masked_rows": [_I, _P, _P, _I, _P, _I, _P, _P, _P, _P],
    "gs_dp_scatter_add_payload": [_I, _P, _I, _P, _P, _F, _P],
    "gs_slice_project_records": [_I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P],
    "gs_project_records": [_I, _I, _P, _I, _I, _P, _P],
    "gs_frame_forward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _L, _P, _P],
    "gs_frame_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _L, _P],
    "gs_frame_profile_enable": [ctypes.c_uint],
    "gs_frame_profile_read": [_I, _P, _P],
    "gs_image_loss_fwd_bwd": [_I, _I, _P, _P, _F, _P, _P, _P, _L, _P],
    "gs_adam_step": [_I, _P, _P, _P, _P, _P, _P, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _P],
}
_SIGS_LL = {
    "gs_scan_workspace_bytes": [_L],
    "gs_radix_sort_workspace_bytes": [_L, _I, _I],
    "gs_segmented_sort_workspace_bytes": [_L, _L, _I, _I],
    "gs_segmented_sort_compact_workspace_bytes": [_L, _L, _I, _I, _I],
    "gs_depth_select_workspace_bytes": [_I],
    "gs_project_pose_scratch_bytes": [_I, _I, _I],
    "gs_image_loss_workspace_bytes": [_I, _I],
    "gs_frame_backward_bytes": [_P],
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load libgsdeblur_hip.so (building it in-tree first if absent). Raises loudly on failure."""
    global _lib
    if _lib is not None:
        return _lib
    import os
    import shutil
    override = os.environ.get("GSD_LIB_PATH")               # A/B builds of the same sources: used as they are
    path = Path(override) if override else LIB_PATH
    if not override and build_if_missing and shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        # the in-tree library is rebuilt when the sources on disk are not the ones it was built from (content hash),
        # so an edited kernel never runs against a stale binary with the same export table.  The common case — the
        # library is current — touches nothing: no lock file, no hipcc.  A prebuilt library WITHOUT the hash sidecar
        # (an install made elsewhere) is used as it is unless a source is newer than it.
        from ._build import is_current, HASH_PATH, CSRC
        stale = not is_current()
        if stale and path.exists() and not HASH_PATH.exists():
            t = path.stat().st_mtime
            stale = any(f.stat().st_mtime > t for f in list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")))
        if stale:
            # one process per GPU: on a fresh checkout every rank gets here at once — exactly one may run hipcc
            import fcntl
            try:
                lock = open(str(path) + ".lock", "w")
            except OSError as e:
                # read-only install: nothing can be rebuilt here; an existing library is loaded with a warning
                if not path.exists():
                    raise HipLibraryError(f"{path} is missing and {path.parent} is not writable: {e}") from e
                import warnings
                warnings.warn(f"{path} may be older than its sources and {path.parent} is not writable; loading it as is")
            else:
                with lock:
                    fcntl.flock(lock, fcntl.LOCK_EX)
                    try:
                        build_library()          # re-checks the hash under the lock: only the first rank compiles
                    finally:
                        fcntl.flock(lock, fcntl.LOCK_UN)
    if not path.exists():
        raise HipLibraryError(f"{path} is missing; run __graft_entry__.build()")
    try:
        lib = ctypes.CDLL(str(path))
    except OSError as e:  # pragma: no cover
        raise HipLibraryError(f"cannot load HIP library {path}: {e}") from e
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _I
    for name, args in _SIGS_LL.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _L
    lib.gs_version.restype = ctypes.c_char_p
    lib.gs_version.argtypes = []
    _lib = lib
    return lib


def exported_names():
    return sorted(list(_SIGS) + list(_SIGS_LL) + ["gs_version"])


def check(status: int, what: str) -> None:
    if status != 0:
        kind = {1: "invalid argument", 3: "workspace too small"}.get(status, None)
        if kind is None and status >= 1000:
            kind = f"hipError_t {status - 1000}"
        raise HipLibraryError(f"{what} failed: {kind or status}")
