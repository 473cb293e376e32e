"""diagnostic (test infrastructure): where do the pixel-velocity render and the SE(3) render of the MODEL part ways on a
real camera pose?  Centres from gs.project_gaussians under the screw-interpolated viewmat vs centre(0) + t * pv with pv
from the float64 oracle, and the two model renders against the float64 oracle's renders of the same inputs."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tools", ROOT / "oracle"):
    sys.path.insert(0, str(p))
import gsdeblur_amd as gs
import gs_oracle as O
import synthetic_dataset as SD
from gsdeblur_amd.model import Camera
dev = torch.device("cuda", 0)
H, W, S = 60, 80, 5
gt = SD.make_gt_scene(1500, 0)
traj = SD.trajectory(9, 1.5, 0)
fr = traj[1]
cam = Camera(fr["c2w"][:3], 0.75 * W, 0.75 * W, W / 2.0, H / 2.0, W, H,
             metadata=dict(cam_idx=0, camera_linear_velocity=fr["lin"].tolist(), camera_angular_velocity=fr["ang"].tolist(),
                           exposure_time=1 / 15, rolling_shutter_time=0.0))
ms = {}
for mm in ("se3", "pixel_velocity"):
    cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=S, gamma=2.2, min_rgb_level=0.0, background_color="black",
                                    rolling_shutter_compensation=False, motion_model=mm)
    ms[mm] = gs.SplatfactoDeblurModel.from_scene(cfg, gt, dev).eval()
with torch.no_grad():
    a = ms["se3"].get_outputs(cam)["rgb"].cpu().double()
    b = ms["pixel_velocity"].get_outputs(cam)["rgb"].cpu().double()
    V, lin, ang = ms["se3"]._viewmat_and_velocity(cam)
m = ms["se3"]
Vd, lind, angd = V.cpu().double(), lin.cpu().double(), ang.cpu().double()
sh = torch.cat([m.features_dc[:, None, :], m.features_rest], dim=1).detach().cpu().double()
base = (m.means.detach().cpu().double(), m.scales.detach().cpu().double().exp(), m.quats.detach().cpu().double(),
        torch.sigmoid(m.opacities.detach().cpu().double()).reshape(-1), sh, Vd)
kw = dict(blur_samples=S, exposure_time=1 / 15, gamma=2.2, min_rgb_level=0.0)
oa, _ = O.render(O.RenderConfig(H, W, cam.fx, cam.fy, cam.cx, cam.cy, motion_model="se3", **kw), *base, lind, angd,
                 background=torch.zeros(3, dtype=torch.float64))
ob, _ = O.render(O.RenderConfig(H, W, cam.fx, cam.fy, cam.cx, cam.cy, motion_model="pixel_velocity", **kw), *base, lind, angd,
                 background=torch.zeros(3, dtype=torch.float64))
ps = gs.training.psnr
print("viewmat\n", Vd, "\nlin", lind.tolist(), "ang", angd.tolist())
print("model se3 vs model pixvel      %.1f dB" % ps(a, b))
print("oracle se3 vs oracle pixvel    %.1f dB" % ps(oa, ob))
print("model se3 vs oracle se3        %.1f dB" % ps(a, oa))
print("model pixvel vs oracle pixvel  %.1f dB" % ps(b, ob))
