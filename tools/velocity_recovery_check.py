"""End-to-end check of the camera-velocity optimizer (/root/reference/train.py:66,70: --camera-velocity-optimizer.enabled,
.zero-initial-velocities): ground-truth Gaussians, frames rendered with rolling shutter AND motion blur from known
velocities (per-pixel-row ground truth, tools/synthetic_dataset.render_rolling_shutter_frame), a model that starts
from ZERO velocities and learns one 6-vector per frame through the rasterizer's twist gradients, Gaussians constant.
Motion blur alone does not reveal the sign of a velocity (the exposure window is symmetric); the rolling shutter does.
Prints, per moving frame, the cosine and the norm ratio between the learned and the true velocity (OpenCV camera
frame), linear and angular parts apart.   usage: python tools/velocity_recovery_check.py [motion_model] [iterations]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import gsdeblur_amd as gs          # noqa: E402
import synthetic_dataset as SD     # noqa: E402
from gsdeblur_amd.model import Camera   # noqa: E402

mm = sys.argv[1] if len(sys.argv) > 1 else "se3"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
H, W = 120, 160
exposure, t_ro = 1 / 15, 1 / 15
dev = torch.device("cuda", 0)
gt = SD.make_gt_scene(4000, 0)
traj = SD.trajectory(17, 1.5, 0)
frames = [i for i in range(len(traj)) if i % 8 != 0][:6]
ref_cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=1, gamma=2.2, min_rgb_level=0.0, background_color="black",
                                    rolling_shutter_compensation=False)
ref_model = gs.SplatfactoDeblurModel.from_scene(ref_cfg, gt, dev).eval()
cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=5, gamma=2.2, min_rgb_level=0.0, background_color="black",
                                rolling_shutter_compensation=True, rs_bands=8, motion_model=mm,
                                rolling_shutter_mode="exact" if mm == "pixel_velocity" else "bands")
cfg.camera_velocity_optimizer.enabled = True
cfg.camera_velocity_optimizer.zero_initial_velocities = True
model = gs.SplatfactoDeblurModel.from_scene(cfg, gt, dev, num_cameras=len(traj))
opts = gs.training.make_optimizers(model)
assert "camera_velocity_opt" in opts, list(opts)
cams, imgs = {}, {}
with torch.no_grad():
    for i in frames:
        fr = traj[i]
        md = dict(cam_idx=i, camera_linear_velocity=fr["lin"].tolist(), camera_angular_velocity=fr["ang"].tolist(),
                  exposure_time=exposure, rolling_shutter_time=t_ro)
        cams[i] = Camera(fr["c2w"][:3], 0.75 * W, 0.75 * W, W / 2.0, H / 2.0, W, H, metadata=md)
        imgs[i] = SD.render_rolling_shutter_frame(ref_model, cams[i], exposure, t_ro, 2.2)
first = {i: None for i in frames}
for it in range(iters):
    for i in frames:
        loss = gs.training.eval_camera_step(model, opts, cams[i], imgs[i])
        if first[i] is None:
            first[i] = loss
flip = torch.tensor([1.0, -1.0, -1.0])
cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm() + 1e-12))     # noqa: E731
ok = 0
for i in frames:
    adj = model.velocity_adjustment[i].detach().cpu()
    lin_t, ang_t = traj[i]["lin"] * flip, traj[i]["ang"] * flip
    last = gs.training.eval_camera_step(model, opts, cams[i], imgs[i])
    c_l, c_a = cos(adj[:3], lin_t), cos(adj[3:], ang_t)
    ok += int(c_a > 0.9)
    print(f"frame {i:2d}: loss {first[i]:.4f} -> {last:.4f}   lin cos {c_l:+.3f} |learned|/|true| {float(adj[:3].norm() / lin_t.norm()):.2f}"
          f"   ang cos {c_a:+.3f} |learned|/|true| {float(adj[3:].norm() / ang_t.norm()):.2f}")
print(f"{mm}: angular velocity recovered (cos > 0.9) in {ok} of {len(frames)} frames")
