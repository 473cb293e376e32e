"""Splatfacto-style model surface over the HIP hot path.

Mirrors what the reference needs from the nerfstudio fork's ``SplatfactoModel``
(absent submodule, /root/reference/.gitmodules:1-3):
  * config fields set by /root/reference/train.py:14-22,40,46-70,119-120
    (rasterize_mode, blur_samples, rolling_shutter_compensation, gamma, min_rgb_level,
    camera_optimizer.mode, camera_velocity_optimizer.*, background_color, ...)
  * ``get_outputs_for_camera(camera=...)`` returning ``rgb`` and ``depth`` and honouring
    ``camera.metadata['cam_idx']``                      (/root/reference/render_model.py:216-219)
  * per-frame ``camera_linear_velocity`` / ``camera_angular_velocity`` in the (OpenGL) camera
    frame plus scene-level ``exposure_time`` / ``rolling_shutter_time``
    (/root/reference/process_synthetic_inputs.py:113-129,157-176).
The forward/backward rendering surface; the densification step that follows it lives in densify.py
(SURVEY §8 f3); the trainer, data managers and the viewer are out of scope.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
from torch import Tensor, nn

from . import ops


@dataclass
class CameraOptimizerConfig:
    mode: str = "off"  # "off" | "SO3xR3"   (train.py:40)


@dataclass
class CameraVelocityOptimizerConfig:
    enabled: bool = False                 # train.py:66
    zero_initial_velocities: bool = False  # train.py:70


@dataclass
class SplatfactoDeblurConfig:
    sh_degree: int = 3
    sh_degree_interval: int = 0              # splatfacto raises the active SH degree every 1000 steps; 0 = all bands at once
    rasterize_mode: str = "antialiased"      # train.py:119 ("classic" disables the compensation factor)
    use_scale_regularization: bool = False   # train.py:120 (loss-side; carried for CLI parity)
    blur_samples: int = 5                    # train.py:46,51 ; 0 disables motion-blur sampling
    rolling_shutter_compensation: bool = True  # train.py:56
    rs_bands: int = 10                       # row bands per frame when rolling-shutter compensation is on
    gamma: float = 2.2                       # train.py:62 (gamma=1 when gamma correction is off)
    min_rgb_level: float = 0.0               # train.py:60 (10 with gamma correction)
    background_color: str = "black"          # train.py:17 ("auto" = learnable)
    num_downscales: int = 0                  # train.py:14; splatfacto's default is 2: training starts at 1/2^n resolution
    resolution_schedule: int = 3000          # ... and doubles every this many steps (nerfstudio 1.1.0 default)
    cull_scale_thresh: float = 0.5           # train.py:18 (densification; carried for CLI parity)
    optimize_eval_velocities: bool = True    # train.py:20
    output_depth_during_training: bool = False
    # how a sub-pose moves the splats: "se3" re-projects every Gaussian under the screw-interpolated pose
    # (north_star), "pixel_velocity" is the paper's first-order model — one projection, centres shifted by
    # t * pixel velocity, depth order / covariance / colour of the mid-exposure pose
    motion_model: str = "se3"
    # rolling shutter (with rolling_shutter_compensation): "bands" = rs_bands tile-row bands, each its own sub-pose
    # (both motion models); "exact" (pixel_velocity only) = continuous per-row time inside the compositor, ONE
    # projection / sort per blur sample whatever the band count (SURVEY App. A "row time (y/H - 1/2) * T_ro")
    rolling_shutter_mode: str = "bands"
    # pixel_velocity only: "per_sample" = one projection / sort / tile list per sub-pose; "shared" = ONE for the frame
    # (tile boxes swept over the exposure + readout, every blur sample walks the same list — the form the paper
    # describes; rolling shutter then only in its "exact" mode)
    pixel_velocity_lists: str = "per_sample"
    camera_optimizer: CameraOptimizerConfig = field(default_factory=CameraOptimizerConfig)
    camera_velocity_optimizer: CameraVelocityOptimizerConfig = field(default_factory=CameraVelocityOptimizerConfig)


@dataclass
class Camera:
    """Minimal stand-in for nerfstudio's Cameras (one camera).  camera_to_world uses the dataset's
    OpenGL convention (-z forward, +y up; process_synthetic_inputs.py:230-238)."""
    camera_to_world: Tensor  # [3,4] or [4,4]
    fx: float
    fy: float
    cx: float
    cy: float
    width: int
    height: int
    metadata: Dict = field(default_factory=dict)  # cam_idx, camera_linear_velocity, camera_angular_velocity,
    #                                               exposure_time, rolling_shutter_time

    def rescaled(self, d: int) -> "Camera":
        """the same camera at 1/d of the resolution (nerfstudio's rescale_output_resolution(1/d), floor rounding):
        what splatfacto renders while its resolution schedule is active"""
        s = 1.0 / float(d)
        return Camera(self.camera_to_world, self.fx * s, self.fy * s, self.cx * s, self.cy * s,
                      int(self.width * s), int(self.height * s), self.metadata)


def _skew(w: Tensor) -> Tensor:
    z = torch.zeros((), device=w.device, dtype=w.dtype)
    return torch.stack([torch.stack([z, -w[2], w[1]]), torch.stack([w[2], z, -w[0]]), torch.stack([-w[1], w[0], z])])


def _so3_exp(w: Tensor) -> Tensor:
    th2 = (w * w).sum()
    K = _skew(w)
    eye = torch.eye(3, device=w.device, dtype=w.dtype)
    small = th2 < 1e-10
    th = torch.sqrt(torch.clamp(th2, min=1e-20))
    A = torch.where(small, 1.0 - th2 / 6.0, torch.sin(th) / th)
    B = torch.where(small, 0.5 - th2 / 24.0, (1.0 - torch.cos(th)) / torch.clamp(th2, min=1e-20))
    return eye + A * K + B * (K @ K)


class SplatfactoDeblurModel(nn.Module):
    """Gaussian parameters + ``get_outputs`` through the fused HIP path."""

    def __init__(self, config: SplatfactoDeblurConfig, means: Tensor, log_scales: Tensor, quats: Tensor,
                 opacity_logits: Tensor, features_dc: Tensor, features_rest: Tensor, num_cameras: int = 1):
        super().__init__()
        self.config = config
        # parameter names follow splatfacto's gauss_params
        self.means = nn.Parameter(means.float())
        self.scales = nn.Parameter(log_scales.float())
        self.quats = nn.Parameter(quats.float())
        self.opacities = nn.Parameter(opacity_logits.float().reshape(-1, 1))
        self.features_dc = nn.Parameter(features_dc.float())
        self.features_rest = nn.Parameter(features_rest.float())
        if config.background_color == "auto":
            self.background_param = nn.Parameter(torch.zeros(3))
        else:
            self.background_param = None
        self.num_cameras = num_cameras
        if config.camera_optimizer.mode == "SO3xR3":
            self.pose_adjustment = nn.Parameter(torch.zeros(num_cameras, 6))
        elif config.camera_optimizer.mode == "off":
            self.pose_adjustment = None
        else:
            raise ValueError(f"unknown camera_optimizer.mode {config.camera_optimizer.mode!r}")
        if config.camera_velocity_optimizer.enabled:
            self.velocity_adjustment = nn.Parameter(torch.zeros(num_cameras, 6))
        else:
            self.velocity_adjustment = None
        self.step = 0                        # training iteration (advanced by train_step)
        self.radii: Optional[Tensor] = None
        # densification statistics (densify.py): when enabled, every training render leaves the summed
        # screen-space centre gradient of its backward pass in self.xy_grad [N,2] (pixels)
        self.collect_densify_stats = False
        self.xy_grad: Optional[Tensor] = None
        self.last_size = (0, 0)
        # frame-to-frame memory of THIS model's frames (adaptive slice budget, arena estimate): owned here, not by the
        # binding's module state, so two models of one shape never share it
        self.frame_hints = ops.FrameHints()

    def _hints_of(self, camera):
        """a camera that names itself (metadata['cam_idx'], as the datamanager's training cameras do:
        /root/reference/render_model.py:216-219) has its own memory inside this model's FrameHints (ops.FrameHints.view): a
        training loop revisits the same cameras, and what a frame teaches the next one is a property of the camera; a
        camera without an index (a novel view) goes through the scene-level memory"""
        idx = camera.metadata.get("cam_idx") if camera.metadata else None
        return self.frame_hints if idx is None else self.frame_hints.view(int(idx))

    # -- splatfacto-style accessors ------------------------------------------------
    @property
    def num_points(self) -> int:
        return self.means.shape[0]

    def active_sh_degree(self) -> int:
        """splatfacto's progressive SH schedule: degree min(step // sh_degree_interval, sh_degree) while training"""
        cfg = self.config
        if not self.training or cfg.sh_degree_interval <= 0:
            return cfg.sh_degree
        return min(self.step // cfg.sh_degree_interval, cfg.sh_degree)

    def gauss_params(self) -> Dict[str, nn.Parameter]:
        return {"means": self.means, "scales": self.scales, "quats": self.quats, "opacities": self.opacities,
                "features_dc": self.features_dc, "features_rest": self.features_rest}

    def _background(self, device) -> Tensor:
        c = self.config.background_color
        if c == "auto":
            return torch.sigmoid(self.background_param).to(device)
        if c == "random" and self.training:
            return torch.rand(3, device=device)
        if c == "white":
            return torch.ones(3, device=device)
        return torch.zeros(3, device=device)

    # -- camera handling -------------------------------------------------------------
    def _const(self, values) -> Tensor:
        """small constant on the parameters' device, uploaded once (a fresh torch.tensor(list, device=...) is a
        pageable host->device copy, i.e. a stream synchronisation, every frame)"""
        key = (tuple(float(v) for v in values), str(self.means.device))
        cache = self.__dict__.setdefault("_const_cache", {})
        t = cache.get(key)
        if t is None:
            if len(cache) > 256:
                cache.clear()
            t = cache[key] = torch.tensor(key[0], dtype=torch.float32, device=self.means.device)
        return t

    def _camera_inputs(self, camera: Camera):
        """the camera's pose and data velocities on the parameters' device, uploaded ONCE per camera: a pageable
        host->device copy of a 48-byte tensor is ordered behind everything queued on the stream, i.e. it blocks the host
        until the previous iteration's kernels have drained — three of them per frame cost a training loop its whole
        host run-ahead (round 5; the cache lives on the Camera object and follows its tensors' identity / values)"""
        dev = self.means.device
        md = camera.metadata
        lin_h = tuple(float(v) for v in md.get("camera_linear_velocity", (0.0, 0.0, 0.0)))
        ang_h = tuple(float(v) for v in md.get("camera_angular_velocity", (0.0, 0.0, 0.0)))
        key = (str(dev), id(camera.camera_to_world), camera.camera_to_world._version, lin_h, ang_h)
        hit = camera.__dict__.get("_gsd_device_inputs")
        if hit is None or hit[0] != key:
            c2w = camera.camera_to_world.to(device=dev, dtype=torch.float32)
            hit = (key, c2w, torch.tensor(lin_h, dtype=torch.float32, device=dev),
                   torch.tensor(ang_h, dtype=torch.float32, device=dev))
            camera.__dict__["_gsd_device_inputs"] = hit
        return hit[1], hit[2], hit[3]

    def _viewmat_and_velocity(self, camera: Camera):
        dev = self.means.device
        c2w, lin_data, ang_data = self._camera_inputs(camera)
        static = self.pose_adjustment is None and self.velocity_adjustment is None
        if static:
            # no camera-side parameters: (viewmat, lin, ang) is a pure function of the camera — a dozen tiny launches
            # per frame otherwise
            tag = (camera.__dict__["_gsd_device_inputs"][0], bool(self.config.camera_velocity_optimizer.zero_initial_velocities))
            hit = camera.__dict__.get("_gsd_view")
            if hit is not None and hit[0] == tag:
                return hit[1]
        R_gl, t = c2w[:3, :3], c2w[:3, 3]
        cam_idx = int(camera.metadata.get("cam_idx", 0))
        if self.pose_adjustment is not None and 0 <= cam_idx < self.num_cameras:
            # nerfstudio's camera optimizer composes in the CAMERA frame: c2w @ exp_map_SO3xR3(adj)
            adj = self.pose_adjustment[cam_idx]
            t = t + R_gl @ adj[:3]
            R_gl = R_gl @ _so3_exp(adj[3:])
        flip = self._const([1.0, -1.0, -1.0])
        R_cv = R_gl * flip[None, :]                # OpenGL -> OpenCV camera axes (x, -y, -z)
        R_wc = R_cv.T
        t_wc = -(R_wc @ t)
        viewmat = torch.eye(4, device=dev)
        viewmat = torch.cat([torch.cat([R_wc, t_wc[:, None]], dim=1), viewmat[3:4]], dim=0)
        md = camera.metadata
        zero3 = self._const([0.0, 0.0, 0.0])
        use_data_vel = not self.config.camera_velocity_optimizer.zero_initial_velocities
        lin, ang = lin_data, ang_data
        if not use_data_vel:
            lin, ang = zero3, zero3
        # velocities are given in the OpenGL camera frame (process_synthetic_inputs.py:163-165)
        lin, ang = lin * flip, ang * flip
        if self.velocity_adjustment is not None and 0 <= cam_idx < self.num_cameras:
            is_eval = bool(md.get("is_eval", False))
            adj = self.velocity_adjustment[cam_idx]
            if is_eval and not self.config.optimize_eval_velocities:
                adj = adj.detach() * 0.0
            lin, ang = lin + adj[:3], ang + adj[3:]
        if static:
            camera.__dict__["_gsd_view"] = (tag, (viewmat, lin, ang))
        return viewmat, lin, ang

    def _schedule(self, camera: Camera):
        cfg = self.config
        S = cfg.blur_samples if cfg.blur_samples > 0 else 1
        R = cfg.rs_bands if cfg.rolling_shutter_compensation else 1
        exposure = float(camera.metadata.get("exposure_time", 0.0))
        readout = float(camera.metadata.get("rolling_shutter_time", 0.0))
        if exposure == 0.0:
            S = 1
        if readout == 0.0:
            R = 1
        if cfg.rolling_shutter_mode == "exact":
            if cfg.motion_model != "pixel_velocity":
                raise ValueError("rolling_shutter_mode='exact' needs motion_model='pixel_velocity'")
            R = 1                                   # the row time lives in the compositor (see _rs_time)
        elif cfg.rolling_shutter_mode != "bands":
            raise ValueError(f"unknown rolling_shutter_mode {cfg.rolling_shutter_mode!r}")
        if cfg.pixel_velocity_lists == "shared":
            if cfg.motion_model != "pixel_velocity" or R != 1:
                raise ValueError("pixel_velocity_lists='shared' needs motion_model='pixel_velocity' and, with a rolling "
                                 "shutter, rolling_shutter_mode='exact'")
        elif cfg.pixel_velocity_lists != "per_sample":
            raise ValueError(f"unknown pixel_velocity_lists {cfg.pixel_velocity_lists!r}")
        times, _, _ = ops.subpose_schedule(S, exposure, R, readout)
        return S, R, times

    def _rs_time(self, camera: Camera) -> float:
        """readout time handed to the exact rolling-shutter compositors (0: off)"""
        cfg = self.config
        if cfg.rolling_shutter_mode != "exact" or not cfg.rolling_shutter_compensation:
            return 0.0
        return float(camera.metadata.get("rolling_shutter_time", 0.0))

    # -- rendering ---------------------------------------------------------------------
    def get_outputs(self, camera: Camera, detach_gaussians: bool = False) -> Dict[str, Tensor]:
        """detach_gaussians=True renders with the Gaussians as constants: only the camera-side parameters (pose /
        velocity adjustment, background) receive a gradient — what the fork's `--optimize-eval-cameras`
        (/root/reference/train.py:180-183, README.md:197) needs for the evaluation frames."""
        cfg = self.config
        dev = self.means.device
        d = self.downscale_factor()
        if d > 1:
            camera = camera.rescaled(d)          # splatfacto: camera.rescale_output_resolution(1 / d) while training
        viewmat, lin, ang = self._viewmat_and_velocity(camera)
        S, R, times = self._schedule(camera)
        times_t = self._const(times)
        pixvel = cfg.motion_model == "pixel_velocity"
        if not pixvel and cfg.motion_model != "se3":
            raise ValueError(f"unknown motion_model {cfg.motion_model!r}")
        viewmats = viewmat if pixvel else ops.subpose_viewmats(viewmat, lin, ang, times_t)
        shared = pixvel and cfg.pixel_velocity_lists == "shared"
        # the RAW parameters go to the kernels as they are: log-scales, opacity logits, features_dc / features_rest as two
        # pointers (no exp / sigmoid / cat launches and none of their backward; ops.render_combined raw_params / sh_rest)
        gp = (self.means, self.scales, self.quats, self.opacities, self.features_dc, self.features_rest)
        if detach_gaussians:
            gp = tuple(t.detach() for t in gp)
        means_, scales_, quats_, opac_, dc_, rest_ = gp
        bg = self._background(dev)
        use_gamma = cfg.blur_samples > 0
        self.xy_grad = None
        if self.training and self.collect_densify_stats and not detach_gaussians:
            self.xy_grad = torch.zeros(self.num_points, 2, device=dev)
        gamma = cfg.gamma if use_gamma else 1.0
        min_level = cfg.min_rgb_level if use_gamma else 0.0
        # one autograd node for composite + gamma-space average: no [S,H,W,3] sample-gradient tensor in backward
        want_depth = cfg.output_depth_during_training or not self.training
        res = ops.render_combined(
            means_, scales_, quats_, opac_.reshape(-1), dc_,
            viewmats, bg, S, R, camera.fx, camera.fy, camera.cx, camera.cy, camera.height, camera.width,
            gamma=gamma, min_rgb_level=min_level, sh_degree=self.active_sh_degree(),
            antialiased=(cfg.rasterize_mode == "antialiased"), xy_grad_out=self.xy_grad,
            lin_vel=lin if pixvel else None, ang_vel=ang if pixvel else None,
            times=(list(times) if shared else times_t) if pixvel else None,
            return_depth=want_depth, rolling_shutter_time=self._rs_time(camera) if pixvel else 0.0,
            sh_rest=rest_, raw_params=True, shared_list=shared, hints=self._hints_of(camera))
        rgb, alphas, radii = res[:3]
        depth_acc = res[3] if want_depth else None
        self.radii = radii
        self.last_size = (camera.width, camera.height)
        accumulation = alphas.mean(dim=0)[..., None]
        out = {"rgb": torch.clamp(rgb, max=1.0),      # splatfacto clamps in training too
               "accumulation": accumulation, "background": bg}
        if depth_acc is not None:
            # expected depth of the blended splats, from the SAME depth-sliced pass as the colour (a fourth
            # forward-only channel of the compositor): mean over the samples of sum(weight * depth), over alpha
            d = depth_acc.mean(dim=0)[..., None].detach()
            a = accumulation.detach()
            # zero-alpha pixels: splatfacto 1.1.0 fills with depth_im.detach().max(), the UN-normalised accumulated
            # maximum (what render_model.py:219's colour maps are normalised against)
            far = d.max()
            out["depth"] = torch.where(a > 0, d / torch.clamp(a, min=1e-10), far)
        else:
            out["depth"] = None
        return out

    def render_and_backward(self, camera: Camera, grad_image) -> Tensor:
        """One TRAINING frame, forward and backward in one host call (step.render_step: the same C-ABI calls as
        get_outputs + Tensor.backward, without the autograd engine between the two compositors — the entry bench.py
        times).  grad_image: callable rgb [H,W,3] -> d loss / d rgb, where rgb is what get_outputs()["rgb"] would hold
        (clamped at 1).  Gradients ACCUMULATE into .grad of the Gaussian parameters; pose / velocity adjustments and a
        learnable background get theirs through the small torch graph of _viewmat_and_velocity / _background.
        Returns rgb (detached).  Same values as the autograd route (tests: test_train_step_routes_agree)."""
        from .step import render_step
        cfg = self.config
        dev = self.means.device
        d = self.downscale_factor()
        if d > 1:
            camera = camera.rescaled(d)
        viewmat, lin, ang = self._viewmat_and_velocity(camera)
        S, R, times = self._schedule(camera)
        pixvel = cfg.motion_model == "pixel_velocity"
        if not pixvel and cfg.motion_model != "se3":
            raise ValueError(f"unknown motion_model {cfg.motion_model!r}")
        shared = pixvel and cfg.pixel_velocity_lists == "shared"
        bg = self._background(dev)
        use_gamma = cfg.blur_samples > 0
        self.xy_grad = None
        if self.training and self.collect_densify_stats:
            self.xy_grad = torch.zeros(self.num_points, 2, device=dev)
        cam_leaves = [t for t in (viewmat, lin, ang) if t.requires_grad]

        def v_rgb(rgb):
            # get_outputs clamps rgb at 1 before the loss sees it; the clamp's backward is the mask
            v = grad_image(torch.clamp(rgb, max=1.0))
            return v * (rgb <= 1.0)

        rgb, g, radii = render_step(
            self.means, self.scales, self.quats, self.opacities.reshape(-1), self.features_dc, viewmat.detach(),
            lin.detach(), ang.detach(), list(times) if shared else self._const(times), bg.detach(), S, R,
            camera.fx, camera.fy, camera.cx, camera.cy, camera.height, camera.width, v_rgb,
            gamma=cfg.gamma if use_gamma else 1.0, min_rgb_level=cfg.min_rgb_level if use_gamma else 0.0,
            sh_degree=self.active_sh_degree(), antialiased=(cfg.rasterize_mode == "antialiased"),
            sh_rest=self.features_rest, raw_params=True, motion_model=cfg.motion_model, xy_grad_out=self.xy_grad,
            camera_grads=bool(cam_leaves), background_grad=bg.requires_grad,
            rolling_shutter_time=self._rs_time(camera) if pixvel else 0.0, shared_list=shared, hints=self._hints_of(camera))
        for p, gr in ((self.means, g["means"]), (self.scales, g["scales"]), (self.quats, g["quats"]),
                      (self.opacities, g["opacities"]), (self.features_dc, g["sh"]), (self.features_rest, g["sh_rest"])):
            gr = gr.view_as(p)
            p.grad = gr if p.grad is None else p.grad + gr
        roots, seeds = [], []
        for t, key in ((viewmat, "viewmat"), (lin, "lin_vel"), (ang, "ang_vel")):
            if t.requires_grad and g[key] is not None:
                roots.append(t)
                seeds.append(g[key].view_as(t))
        if bg.requires_grad and g["background"] is not None:
            roots.append(bg)
            seeds.append(g["background"].view_as(bg))
        if roots:
            torch.autograd.backward(roots, seeds)
        self.radii = radii
        self.last_size = (camera.width, camera.height)
        return torch.clamp(rgb, max=1.0)

    def downscale_factor(self) -> int:
        """splatfacto's `_get_downscale_factor` (nerfstudio 1.1.0): while training, render (and compare) at
        1 / 2^max(num_downscales - step // resolution_schedule, 0) of the camera's resolution; full size otherwise
        (/root/reference/train.py:14 sets num-downscales 0 for the low-resolution synthetic sets)."""
        if not self.training:
            return 1
        return 2 ** max(int(self.config.num_downscales) - int(self.step) // max(1, int(self.config.resolution_schedule)), 0)

    @torch.no_grad()
    def get_outputs_for_camera(self, camera: Camera) -> Dict[str, Tensor]:
        """Eval entry point used by /root/reference/render_model.py:217."""
        was = self.training
        self.eval()
        try:
            return self.get_outputs(camera)
        finally:
            self.train(was)

    @staticmethod
    def from_scene(config: SplatfactoDeblurConfig, scene: Dict, device, num_cameras: int = 1):
        """Build from a dict with means/log_scales/quats/opacity_logits/sh (e.g. a synthetic scene)."""
        sh = scene["sh"]
        m = SplatfactoDeblurModel(config, scene["means"], scene["log_scales"], scene["quats"],
                                  scene["opacity_logits"], sh[:, 0, :], sh[:, 1:, :], num_cameras)
        return m.to(device)
