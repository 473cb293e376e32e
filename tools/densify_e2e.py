"""End-to-end check of densification: the same seed-cloud start trained with and without the refinement
schedule on a blurred dataset; sharp-frame PSNR / SSIM and the number of Gaussians.  usage: python tools/densify_e2e.py [iterations]"""
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
from gsdeblur_amd import densify as D   # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda", 0)
root = tempfile.mkdtemp()
SD.generate(root, dev, width=240, height=160, n_frames=24, n_gaussians=8000, speed=1.0, dense_samples=32, seed_points=1500)
scene = gs.load_transforms(root)
images = gs.data.load_scene_images(scene, dev)
xyz, rgb = gs.load_seed_points_ply(scene.ply_file_path)
res = {}
for name, dcfg in (("no_densification", None),
                   ("densification", D.DensifyConfig(warmup_length=200, refine_every=100, reset_alpha_every=8,
                                                     stop_split_at=int(0.7 * iters), stop_screen_size_at=int(0.3 * iters)))):
    cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=5, gamma=2.2, min_rgb_level=0.0,
                                    rolling_shutter_compensation=False, use_scale_regularization=True)
    model = SD.init_from_seed_points(cfg, xyz, rgb, dev, num_cameras=len(scene.cameras))
    n0 = model.num_points
    r = gs.training.train_scene(model, scene, images, iters, densify=dcfg)
    res[name] = {"psnr": round(r["results"]["psnr"], 3), "ssim": round(r["results"]["ssim"], 4), "gaussians": [n0, model.num_points],
                 "seconds": round(r["wall_clock_time_seconds"], 2)}
    print(name, json.dumps(res[name]), flush=True)
print(json.dumps({"iterations": iters, "results": res}))
