#!/usr/bin/env python
"""End-to-end deblurring run on a transforms.json dataset, scored like the reference scores its own
(/root/reference/train.py:78-109: metrics.json with results{psnr, ssim} + wall_clock_time_seconds).

  python tools/train_deblur.py --generate /tmp/ds            # write the self-generated dataset, then train on it
  python tools/train_deblur.py --data /tmp/ds --blur-samples 0 5 10 --iterations 1500 --out gpurun_out/deblur

Variants follow /root/reference/train.py:29-76: blur_samples 0 = no motion-blur compensation (the baseline),
5 (the default, train.py:46) and 10 (synthetic sets, train.py:22); --motion-model picks the SE(3) re-projection
(north_star) or the paper's pixel-velocity model; --optimize-eval-cameras refines the evaluation poses without
touching the Gaussians (train.py:180-183)."""
import argparse
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import gsdeblur_amd as gs  # noqa: E402
sys.path.insert(0, str(Path(__file__).resolve().parent))
import synthetic_dataset as SD  # noqa: E402  (test / demo data generation: not part of the product package)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default=None)
    ap.add_argument("--generate", default=None, help="write the synthetic dataset here first (and use it)")
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--gaussians", type=int, default=20000)
    ap.add_argument("--speed", type=float, default=1.0)
    ap.add_argument("--rolling-shutter-time", type=float, default=0.0)
    ap.add_argument("--blur-samples", type=int, nargs="+", default=[0, 5, 10])
    ap.add_argument("--motion-model", default="se3", choices=["se3", "pixel_velocity"])
    ap.add_argument("--rolling-shutter-mode", default="bands", choices=["bands", "exact", "off"],
                    help="bands: R row bands; exact: per pixel row (pixel_velocity model only); off: no compensation")
    ap.add_argument("--iterations", type=int, default=1500)
    ap.add_argument("--optimize-eval-cameras", action="store_true")
    ap.add_argument("--pose-noise", type=float, default=0.0, help="std (m / rad) of noise on the evaluation poses")
    ap.add_argument("--out", default="gpurun_out/deblur")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    root = args.data
    if args.generate:
        root = args.generate
        SD.generate(root, dev, args.width, args.height, args.frames, args.gaussians, speed=args.speed,
                    rolling_shutter_time=args.rolling_shutter_time)
    scene = gs.load_transforms(root)
    images = gs.data.load_scene_images(scene, dev)          # undistorts when the scene carries lens coefficients
    xyz, rgb = gs.load_seed_points_ply(scene.ply_file_path)
    if args.pose_noise > 0:
        g = torch.Generator().manual_seed(3)
        for i in scene.eval_indices:
            scene.cameras[i].camera_to_world[:, 3] += args.pose_noise * torch.randn(3, generator=g)
    os.makedirs(args.out, exist_ok=True)
    table = {}
    for bs in args.blur_samples:
        cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=bs, gamma=2.2 if bs > 0 else 1.0,
                                        min_rgb_level=0.0,
                                        rolling_shutter_compensation=(args.rolling_shutter_time > 0 and
                                                                      args.rolling_shutter_mode != "off"),
                                        rolling_shutter_mode="exact" if args.rolling_shutter_mode == "exact" else "bands",
                                        rs_bands=min(8, (scene.cameras[0].height + 15) // 16),
                                        motion_model=args.motion_model, use_scale_regularization=True)
        if args.optimize_eval_cameras:
            cfg.camera_optimizer.mode = "SO3xR3"
        model = SD.init_from_seed_points(cfg, xyz, rgb, dev, num_cameras=len(scene.cameras))
        res = gs.training.train_scene(model, scene, images, args.iterations,
                                      optimize_eval_cameras=args.optimize_eval_cameras, log_every=100)
        name = (f"blur_samples_{bs}" + ("_pixvel" if args.motion_model == "pixel_velocity" else "") +
                ("" if args.rolling_shutter_time <= 0 else f"_rs_{args.rolling_shutter_mode}"))
        with open(os.path.join(args.out, f"metrics_{name}.json"), "wt") as f:
            json.dump({"results": res["results"], "wall_clock_time_seconds": res["wall_clock_time_seconds"]}, f)
        table[name] = res["results"] | {"time": round(res["wall_clock_time_seconds"], 1)}
        print(name, json.dumps(table[name]), flush=True)
    print(json.dumps(table))


if __name__ == "__main__":
    main()
