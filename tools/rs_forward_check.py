"""Forward-model accuracy of the rolling-shutter modes, no training involved: the ground-truth Gaussians rendered (a) as
tools/synthetic_dataset.render_rolling_shutter_frame does (192 sharp frames, per-row exposure windows) and (b) by the
model under each rolling-shutter mode; PSNR of (b) against (a) per moving frame.
usage: python tools/rs_forward_check.py [readout_time] [speed]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import gsdeblur_amd as gs          # noqa: E402
import synthetic_dataset as SD     # noqa: E402
from gsdeblur_amd.model import Camera   # noqa: E402

t_ro = float(sys.argv[1]) if len(sys.argv) > 1 else 1 / 15
speed = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
H, W = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (120, 160)
dev = torch.device("cuda", 0)
gt = SD.make_gt_scene(4000, 0)
traj = SD.trajectory(9, speed, 0)
# a first-order pixel-motion model cannot hold for a splat whose depth is comparable to the camera's displacement during
# the frame (the generator's camera flies THROUGH the Gaussian cloud: 10 % of the visible ones are nearer than 0.2 m):
# keep the Gaussians that stay at least `near` metres away from every camera position of the trajectory
near = float(sys.argv[5]) if len(sys.argv) > 5 else 0.6
cams = torch.stack([fr["c2w"][:3, 3] for fr in traj])
keep = (gt["means"][:, None, :] - cams[None, :, :]).norm(dim=-1).min(dim=1).values > near
gt = {k: (v[keep] if isinstance(v, torch.Tensor) and v.shape[:1] == keep.shape else v) for k, v in gt.items()}
print(f"{int(keep.sum())} of {keep.numel()} ground-truth Gaussians kept (nearest camera distance > {near} m)")
exposure = 1 / 15
variants = [("no compensation, se3", "se3", False, "bands"), ("8 row bands, se3", "se3", True, "bands"),
            ("8 row bands, pixel velocity", "pixel_velocity", True, "bands"),
            ("exact rows, pixel velocity", "pixel_velocity", True, "exact")]
models = {}
for name, mm, comp, mode in variants:
    cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=10, gamma=2.2, min_rgb_level=0.0, background_color="black",
                                    rolling_shutter_compensation=comp, rs_bands=8, rolling_shutter_mode=mode, motion_model=mm)
    models[name] = gs.SplatfactoDeblurModel.from_scene(cfg, gt, dev).eval()
ref_model = models[variants[0][0]]
acc = {n: [] for n, *_ in variants}
with torch.no_grad():
    for i, fr in enumerate(traj):
        if i % 8 == 0:
            continue
        cam = Camera(fr["c2w"][:3], 0.75 * W, 0.75 * W, W / 2.0, H / 2.0, W, H,
                     metadata=dict(cam_idx=0, camera_linear_velocity=fr["lin"].tolist(),
                                   camera_angular_velocity=fr["ang"].tolist(), exposure_time=exposure,
                                   rolling_shutter_time=t_ro))
        ref = SD.render_rolling_shutter_frame(ref_model, cam, exposure, t_ro, 2.2)
        for n, *_ in variants:
            img = models[n].get_outputs(cam)["rgb"]
            acc[n].append(gs.training.psnr(img, ref))
for n, v in acc.items():
    print(f"{n:32s} PSNR vs per-row ground truth: mean {sum(v) / len(v):6.2f} dB   per frame " + " ".join(f"{x:5.1f}" for x in v))
