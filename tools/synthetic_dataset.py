"""Self-generated motion-blur / rolling-shutter dataset in the reference's wire format.

The reference's benchmark scenes (synthetic-mb/cozyroom etc.) are Zenodo downloads
(/root/reference/download_data.py:21-33) and there is no network here, so the end-to-end deblurring check runs on
a stand-in built the way /root/reference/process_synthetic_inputs.py builds its own (:44-201): a ground-truth scene
is rendered along a camera trajectory, every TRAINING frame as the average (in linear light) of many sharp renders
across its exposure — here 64 dense SE(3) sub-poses through this package's own HIP renderer — and every 8th frame
as a SHARP, zero-velocity evaluation frame (:287-293).  Output:

    <root>/transforms.json        fields exactly as :113-129 / :171-176 (velocities in the OpenGL camera frame,
                                  `R_w2c @ v_world`, :157-165)
    <root>/images/000.png ...     8-bit frames
    <root>/sparse_pc.ply          seed cloud (ASCII x y z r g b, :203-219): noisy subsample of the GT means

The ground truth is a set of Gaussians (a textured "room": walls of small splats plus a few objects), so the
generator and the trainer share nothing but the wire format and the renderer.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from gsdeblur_amd import data as _data  # noqa: E402
from gsdeblur_amd.model import Camera, SplatfactoDeblurConfig, SplatfactoDeblurModel  # noqa: E402

SH_C0 = 0.28209479177387814


def make_gt_scene(n: int = 6000, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Gaussians on the walls of a 4 x 3 x 4 m room around the origin plus three blobs: high-frequency colour
    (checker + noise) so that motion blur destroys visible detail."""
    g = torch.Generator().manual_seed(seed)
    n_wall = int(n * 0.85)
    face = torch.randint(0, 5, (n_wall,), generator=g)                  # 4 walls + floor
    u = torch.rand(n_wall, generator=g) * 2 - 1
    v = torch.rand(n_wall, generator=g) * 2 - 1
    hx, hy, hz = 2.0, 1.5, 2.0
    pts = torch.zeros(n_wall, 3)
    pts[face == 0] = torch.stack([u * hx, v * hy, torch.full_like(u, -hz)], -1)[face == 0]
    pts[face == 1] = torch.stack([u * hx, v * hy, torch.full_like(u, hz)], -1)[face == 1]
    pts[face == 2] = torch.stack([torch.full_like(u, -hx), v * hy, u * hz], -1)[face == 2]
    pts[face == 3] = torch.stack([torch.full_like(u, hx), v * hy, u * hz], -1)[face == 3]
    pts[face == 4] = torch.stack([u * hx, torch.full_like(u, -hy), v * hz], -1)[face == 4]
    checker = ((torch.floor((u + 1) * 6) + torch.floor((v + 1) * 6)) % 2)
    base = torch.stack([0.25 + 0.5 * checker, 0.3 + 0.4 * (face.float() / 4.0), 0.8 - 0.5 * checker], -1)
    col_w = (base + 0.15 * torch.randn(n_wall, 3, generator=g)).clamp(0.02, 0.98)
    n_obj = n - n_wall
    centres = torch.tensor([[0.6, -0.8, -0.4], [-0.7, -0.5, 0.5], [0.0, 0.2, -1.0]])
    which = torch.randint(0, 3, (n_obj,), generator=g)
    pts_o = centres[which] + 0.25 * torch.randn(n_obj, 3, generator=g)
    col_o = (torch.tensor([[0.9, 0.2, 0.1], [0.1, 0.8, 0.2], [0.95, 0.85, 0.1]])[which]
             + 0.2 * torch.randn(n_obj, 3, generator=g)).clamp(0.02, 0.98)
    means = torch.cat([pts, pts_o])
    rgb = torch.cat([col_w, col_o])
    log_scales = math.log(0.035) + 0.3 * torch.randn(n, 3, generator=g)
    quats = torch.randn(n, 4, generator=g)
    quats = quats / quats.norm(dim=-1, keepdim=True)
    opacity_logits = 2.5 + 0.5 * torch.randn(n, generator=g)
    sh = torch.zeros(n, 16, 3)
    sh[:, 0, :] = (rgb - 0.5) / SH_C0
    return dict(means=means, log_scales=log_scales, quats=quats, opacity_logits=opacity_logits, sh=sh, rgb=rgb)


def _look_at_gl(eye: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """camera-to-world [4,4], OpenGL axes (-z forward, +y up)"""
    f = target - eye
    f = f / f.norm()
    up = torch.tensor([0.0, 1.0, 0.0])
    r = torch.linalg.cross(f, up)
    r = r / r.norm()
    u = torch.linalg.cross(r, f)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = r, u, -f, eye
    return c2w


def trajectory(n_frames: int, speed: float = 1.0, seed: int = 0) -> List[Dict]:
    """An orbit inside the room; per frame the world-frame velocity of the camera and its angular velocity, turned
    into the camera frame as /root/reference/process_synthetic_inputs.py:157-165 does (R_w2c @ v)."""
    g = torch.Generator().manual_seed(seed + 77)
    frames = []
    for i in range(n_frames):
        a = 2 * math.pi * i / n_frames
        eye = torch.tensor([0.9 * math.cos(a), 0.1 * math.sin(2 * a), 0.9 * math.sin(a)])
        target = torch.tensor([1.9 * math.cos(a + 0.6), -0.2, 1.9 * math.sin(a + 0.6)])
        c2w = _look_at_gl(eye, target)
        R_w2c = c2w[:3, :3].T
        v_w = speed * (torch.rand(3, generator=g) * 2 - 1) * torch.tensor([1.2, 0.4, 1.2])
        w_w = speed * (torch.rand(3, generator=g) * 2 - 1) * torch.tensor([0.5, 1.5, 0.5])
        frames.append(dict(c2w=c2w, lin=R_w2c @ v_w, ang=R_w2c @ w_w))
    return frames


@torch.no_grad()
def render_rolling_shutter_frame(model, cam, exposure_time: float, readout_time: float, gamma: float,
                                 n_times: int = 192) -> torch.Tensor:
    """Ground truth of a rolling-shutter frame, independent of the renderer's own rolling-shutter machinery: n_times
    SHARP full frames along the SE(3) screw motion over [-(e + T_ro)/2, (e + T_ro)/2]; pixel row y integrates (in linear
    light) the frames whose time lies in its own exposure window [tau(y) - e/2, tau(y) + e/2], tau(y) = ((y + 0.5)/H -
    0.5) * T_ro — the continuous row time of a real sensor, no row bands."""
    from gsdeblur_amd import ops
    viewmat, lin, ang = model._viewmat_and_velocity(cam)
    span = exposure_time + readout_time
    times = torch.tensor([((k + 0.5) / n_times - 0.5) * span for k in range(n_times)], device=viewmat.device)
    vms = ops.subpose_viewmats(viewmat, lin, ang, times)
    sh = torch.cat([model.features_dc[:, None, :], model.features_rest], dim=1)
    bg = model._background(viewmat.device)
    samples, _, _ = ops.render_subposes(model.means, torch.exp(model.scales), model.quats,
                                        torch.sigmoid(model.opacities).reshape(-1), sh, vms, bg, n_times, 1, cam.fx,
                                        cam.fy, cam.cx, cam.cy, cam.height, cam.width, sh_degree=model.active_sh_degree(),
                                        antialiased=True, return_alpha=False)
    lin_img = samples.clamp(min=0.0) ** gamma                                        # [K,H,W,3]
    tau = ((torch.arange(cam.height, device=times.device) + 0.5) / cam.height - 0.5) * readout_time
    w = ((times[:, None] - tau[None, :]).abs() <= 0.5 * exposure_time + 1e-9).float()     # [K,H]
    img = (lin_img * w[:, :, None, None]).sum(0) / w.sum(0).clamp(min=1.0)[:, None, None]
    return torch.clamp(img ** (1.0 / gamma), max=1.0)


@torch.no_grad()
def generate(root: str, device, width: int = 160, height: int = 120, n_frames: int = 16, n_gaussians: int = 6000,
             exposure_time: float = 1.0 / 15.0, rolling_shutter_time: float = 0.0, dense_samples: int = 64,
             speed: float = 1.0, seed: int = 0, eval_interval: int = 8, seed_points: int = 3000,
             image_ext: str = "png") -> Dict:
    """Render and write the dataset; returns {'scene': GT parameters, 'frames': ..., 'root': root}."""
    gt = make_gt_scene(n_gaussians, seed)
    fx = fy = 0.75 * width
    cx, cy = width / 2.0, height / 2.0
    rs_bands = min(8, (height + 15) // 16)
    cfg = SplatfactoDeblurConfig(sh_degree=3, blur_samples=dense_samples,
                                 rolling_shutter_compensation=rolling_shutter_time > 0,
                                 rs_bands=rs_bands, gamma=2.2, min_rgb_level=0.0,
                                 background_color="black")
    model = SplatfactoDeblurModel.from_scene(cfg, gt, device).eval()
    traj = trajectory(n_frames, speed, seed)
    frames_json = []
    for i, fr in enumerate(traj):
        is_eval = i % eval_interval == 0
        lin = torch.zeros(3) if is_eval else fr["lin"]          # eval frames are sharp (process_synthetic_inputs.py:287-293)
        ang = torch.zeros(3) if is_eval else fr["ang"]
        cam = Camera(fr["c2w"][:3], fx, fy, cx, cy, width, height,
                     metadata=dict(cam_idx=0, camera_linear_velocity=lin.tolist(), camera_angular_velocity=ang.tolist(),
                                   exposure_time=exposure_time, rolling_shutter_time=rolling_shutter_time))
        if rolling_shutter_time > 0:
            rgb = render_rolling_shutter_frame(model, cam, exposure_time, rolling_shutter_time, cfg.gamma)
        else:
            rgb = model.get_outputs(cam)["rgb"]
        name = f"images/{i:03d}.{image_ext}"
        _data.save_image(os.path.join(root, name), rgb)
        frames_json.append(dict(file_path=f"./{name}", transform_matrix=fr["c2w"].tolist(),
                                camera_linear_velocity=lin.tolist(), camera_angular_velocity=ang.tolist()))
    g = torch.Generator().manual_seed(seed + 5)
    pick = torch.randperm(n_gaussians, generator=g)[:seed_points]
    xyz = gt["means"][pick] + 0.01 * torch.randn(len(pick), 3, generator=g)
    _data.write_seed_points_ply(os.path.join(root, "sparse_pc.ply"), xyz, gt["rgb"][pick])
    _data.write_transforms(root, width, height, fx, fy, cx, cy, exposure_time, rolling_shutter_time, frames_json,
                           ply_file_path="./sparse_pc.ply")
    return dict(scene=gt, frames=frames_json, root=root)


def init_from_seed_points(config: SplatfactoDeblurConfig, xyz: torch.Tensor, rgb: torch.Tensor, device,
                          num_cameras: int, seed: int = 0) -> SplatfactoDeblurModel:
    """splatfacto's initialisation from a seed cloud: isotropic scale = mean distance to the 3 nearest neighbours,
    random rotations, opacity 0.1, colour -> SH dc."""
    n = xyz.shape[0]
    g = torch.Generator().manual_seed(seed)
    d = torch.cdist(xyz, xyz)
    d.fill_diagonal_(float("inf"))
    knn = d.topk(3, largest=False).values.mean(dim=1).clamp(min=1e-4)
    log_scales = torch.log(knn)[:, None].repeat(1, 3)
    quats = torch.randn(n, 4, generator=g)
    quats = quats / quats.norm(dim=-1, keepdim=True)
    K = (config.sh_degree + 1) ** 2
    sh = torch.zeros(n, K, 3)
    sh[:, 0, :] = (rgb - 0.5) / SH_C0
    sc = dict(means=xyz.clone(), log_scales=log_scales, quats=quats,
              opacity_logits=torch.full((n,), math.log(0.1 / 0.9)), sh=sh)
    return SplatfactoDeblurModel.from_scene(config, sc, device, num_cameras=num_cameras)
