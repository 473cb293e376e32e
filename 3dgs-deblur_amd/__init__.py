"""3dgs-deblur_amd — MI355X-native differentiable 3DGS rasterizer with motion-blur /
rolling-shutter sub-frame averaging (the one hot path of SpectacularAI/3dgs-deblur).

The directory name is not a valid Python identifier; import it as ``gsdeblur_amd``
(the repo-root shim ``gsdeblur_amd.py`` registers this package under that name).
"""
from . import _lib  # noqa: F401
from .ops import (  # noqa: F401
    project_gaussians,
    rasterize_gaussians,
    spherical_harmonics,
    map_gaussian_to_intersects,
    get_tile_bin_edges,
    compute_cumulative_intersects,
    bin_and_sort_gaussians,
    bin_and_sort_records,
    render_subposes,
    render_combined,
    subpose_viewmats,
    subpose_schedule,
    combine_samples,
    exclusive_scan_u32,
    radix_sort_pairs,
)
from .model import Camera, SplatfactoDeblurConfig, SplatfactoDeblurModel  # noqa: F401
from . import dp  # noqa: F401
from . import data, densify, fused, step, train_step as training  # noqa: F401
from .step import render_step  # noqa: F401
from .data import load_transforms, load_seed_points_ply  # noqa: F401

__version__ = "0.1.0"
