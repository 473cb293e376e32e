"""Forward + backward of one frame as ONE host call, without the autograd engine (round 4).

`ops.render_combined` is an autograd node: a training iteration then crosses Python's autograd machinery between the
forward and the backward compositor — the engine's worker-thread hand-off, the loss's own nodes, allocator calls —
while the GPU queue is EMPTY (the forward ends with a host read-back).  On the 1M-Gaussian benchmark frame that window
was ~0.1 ms of a 2.2 ms step.  `render_step` issues the same C-ABI calls (the very same forward / backward bodies of
ops._RenderSubposes and ops._SubposeViewmats, driven through a plain context object) back to back: sub-pose viewmats ->
projection -> frame forward -> sub-frame average -> [d loss / d image] -> frame backward -> projection backward ->
sub-pose backward.  Results are bit-identical to the autograd route (tests/test_gpu_parity.py::
test_render_step_equals_autograd_route); what the loss contributes is either a fixed gradient image or a callable
(e.g. the HIP L1 + SSIM kernel of fused.py, whose forward already produces the gradient).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple, Union

import torch
from torch import Tensor

from . import ops


class _Ctx:
    """what torch.autograd.Function hands its static methods, for running them without autograd"""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


@torch.no_grad()
def render_step(means: Tensor, scales: Tensor, quats: Tensor, opacities: Tensor, sh: Tensor, viewmat: Tensor,
                lin_vel: Tensor, ang_vel: Tensor, times: Tensor, background: Optional[Tensor], blur_samples: int,
                rs_bands: int, fx: float, fy: float, cx: float, cy: float, img_height: int, img_width: int,
                grad_image: Union[Tensor, Callable[[Tensor], Tensor]], gamma: float = 1.0, min_rgb_level: float = 0.0,
                sh_degree: int = 3, antialiased: bool = True, sh_rest: Optional[Tensor] = None, raw_params: bool = True,
                motion_model: str = "se3", xy_grad_out: Optional[Tensor] = None, camera_grads: bool = True,
                background_grad: bool = False, glob_scale: float = 1.0, clip_thresh: float = 0.01,
                rolling_shutter_time: float = 0.0, shared_list: bool = False, hints: Optional["ops.FrameHints"] = None
                ) -> Tuple[Tensor, Dict[str, Optional[Tensor]], Tensor]:
    """One frame, forward and backward.  Arguments as ops.render_combined (raw_params: log-scales / opacity logits;
    sh_rest: features_rest beside sh = features_dc), but the camera comes as ONE mid-exposure `viewmat` [4,4] + body
    twist + the P sub-pose `times` for both motion models.  grad_image: d loss / d rgb [H,W,3], or a callable
    rgb -> d loss / d rgb that is invoked between the two halves.
    -> (rgb [H,W,3], gradients {means, scales, quats, opacities, sh, sh_rest, viewmat, lin_vel, ang_vel, background},
        radii [P,N] ([1,N] with shared_list, see ops.render_subposes)) — gradients of exactly the tensors handed in (raw parameters with raw_params=True)."""
    S, R = max(1, int(blur_samples)), max(1, int(rs_bands))
    pixvel = motion_model == "pixel_velocity"
    if not pixvel and motion_model != "se3":
        raise ValueError(f"unknown motion_model {motion_model!r}")
    sub_ctx = None
    if pixvel:
        vms = viewmat
    else:
        sub_ctx = _Ctx((camera_grads, camera_grads, camera_grads, False))
        vms = ops._SubposeViewmats.forward(sub_ctx, viewmat, lin_vel, ang_vel, times)
    needs = [True, True, True, True, True, camera_grads, background is not None and background_grad] + [False] * 16 + \
            [camera_grads and pixvel, camera_grads and pixvel, False, False, False, sh_rest is not None, False, False]
    ctx = _Ctx(needs)
    rgb, _alpha, radii, _depth = ops._RenderSubposes.forward(
        ctx, means, scales, quats, opacities, sh, vms, background, S, R, fx, fy, cx, cy, img_height, img_width,
        sh_degree, antialiased, glob_scale, clip_thresh, xy_grad_out, False, float(gamma), float(min_rgb_level),
        lin_vel if pixvel else None, ang_vel if pixvel else None, times if pixvel else None, False,
        float(rolling_shutter_time), sh_rest, 3 if raw_params else 0, bool(shared_list), hints)
    if callable(grad_image):
        # the callable may use torch.autograd itself (rgb.requires_grad_() + autograd.grad): it sees a detached leaf and
        # runs with gradient recording on (under this function's no_grad it silently got None)
        with torch.enable_grad():
            v_rgb = grad_image(rgb.detach())
        if v_rgb is None:
            raise ValueError("grad_image(rgb) returned None: it must return d loss / d rgb [H,W,3]")
        v_rgb = v_rgb.detach()
    else:
        v_rgb = grad_image
    g = ops._RenderSubposes.backward(ctx, v_rgb, None, None, None)
    grads = {"means": g[0], "scales": g[1], "quats": g[2], "opacities": g[3], "sh": g[4], "background": g[6],
             "sh_rest": g[28], "viewmat": None, "lin_vel": None, "ang_vel": None}
    if pixvel:
        grads["viewmat"], grads["lin_vel"], grads["ang_vel"] = g[5], g[23], g[24]
    elif camera_grads and g[5] is not None:
        grads["viewmat"], grads["lin_vel"], grads["ang_vel"], _ = ops._SubposeViewmats.backward(sub_ctx, g[5])
    return rgb, grads, radii
