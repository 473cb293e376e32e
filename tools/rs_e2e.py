"""End-to-end check of the rolling-shutter modes: a dataset rendered WITH rolling shutter (and motion blur), a perturbed
ground-truth start (as tests/test_data_and_training.py::test_end_to_end_deblurring_on_a_self_generated_dataset), the
same model trained without rolling-shutter compensation, with row bands (both motion models) and with the exact
per-row mode of the pixel-velocity model; scored on the sharp, static evaluation frames.
usage: python tools/rs_e2e.py [readout_time] [speed] [iterations]"""
import json
import sys
import tempfile
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
import gsdeblur_amd as gs          # noqa: E402
import synthetic_dataset as SD     # noqa: E402

t_ro = float(sys.argv[1]) if len(sys.argv) > 1 else 1 / 15
speed = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 700
dev = torch.device("cuda", 0)
root = tempfile.mkdtemp()
info = SD.generate(root, dev, width=160, height=120, n_frames=17, n_gaussians=4000, speed=speed, dense_samples=32,
                   rolling_shutter_time=t_ro)
scene = gs.load_transforms(root)
images = [gs.data.load_image(p, dev) for p in scene.image_paths]
gt = info["scene"]
g = torch.Generator().manual_seed(1)
start = dict(gt)
start["sh"] = gt["sh"] + 0.15 * torch.randn(gt["sh"].shape, generator=g) * (torch.arange(16) == 0)[None, :, None]
start["means"] = gt["means"] + 0.004 * torch.randn(gt["means"].shape, generator=g)
start["log_scales"] = gt["log_scales"] + 0.1
res = {}
for name, mm, comp, mode in (("no_rs_compensation_se3", "se3", False, "bands"),
                             ("row_bands_8_se3", "se3", True, "bands"),
                             ("no_rs_compensation_pixvel", "pixel_velocity", False, "bands"),
                             ("row_bands_8_pixvel", "pixel_velocity", True, "bands"),
                             ("exact_rows_pixvel", "pixel_velocity", True, "exact")):
    cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=5, gamma=2.2, min_rgb_level=0.0,
                                    rolling_shutter_compensation=comp, rs_bands=8, rolling_shutter_mode=mode,
                                    motion_model=mm)
    model = gs.SplatfactoDeblurModel.from_scene(cfg, start, dev, num_cameras=len(scene.cameras))
    r = gs.training.train_scene(model, scene, images, iterations=iters, lr_scale=1.0)
    res[name] = {"psnr": round(r["results"]["psnr"], 3), "ssim": round(r["results"]["ssim"], 4),
                 "seconds": round(r["wall_clock_time_seconds"], 2)}
    print(name, json.dumps(res[name]), flush=True)
print(json.dumps({"readout_time": t_ro, "speed": speed, "iterations": iters, "results": res}))
