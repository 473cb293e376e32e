"""Which of the oracle's threshold-fragile pixels really flip on the HIP path, and how deep inside the band they sit
(VERDICT round 5 item 6).  GPU tool: python tools/fragile_histogram.py > profiles/r06_fragile_histogram.txt

A pixel is "fragile" when one of its decisions (alpha >= 1/255, T > 1e-4) lies within the fp32 rounding band of its
threshold in the float64 oracle (oracle/gs_oracle.py: FRAGILE_ALPHA_BAND on alpha / (1/255), FRAGILE_T_FLOOR +
FRAGILE_T_GAIN * rss(alpha / (1 - alpha)) on T / 1e-4); such pixels are left out of strict comparisons on both sides.
For the reduced-size BASELINE configs 4 and 5 (tests/test_gpu_parity.py: the scenes with the largest shares) this prints,
per criterion and per decile of the band, how many (sub-pose, pixel) decisions are flagged and how many of them FLIP —
the HIP sample differs from the oracle's by more than the comparison's tolerance there (one blend weight, not rounding).
If flips reach the outer deciles the band is as tight as it can be; where they stop it could shrink."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle"):
    sys.path.insert(0, str(p))
import gs_oracle as O          # noqa: E402  (checker; this is a measurement tool, not product code)
import gsdeblur_amd as gs      # noqa: E402

IMG_ATOL = 2e-4
dev = torch.device("cuda:0")
CASES = [("config5: 10 motion-blur sub-poses", 10, 1, 160, 96, 2800, 5.0), ("config4: 5 samples x 2 bands", 5, 2, 176, 112, 3600, 5.0),
         ("config2: 5 motion-blur sub-poses", 5, 1, 240, 136, 6000, 5.0)]
for tag, S, R, W, H, n, mult in CASES:
    sc = O.synthetic_scene(n, W, H, seed=300 + S * 10 + R, scale_mult=mult)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    sc["opacity_logits"] = sc["opacity_logits"].clone()
    sc["opacity_logits"][::10] += 9.0
    et, rt, gamma, mlevel = 1 / 60, 1 / 30, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=R, exposure_time=et,
                         rolling_shutter_time=rt, gamma=gamma, min_rgb_level=mlevel)
    with torch.no_grad():
        ref, _, ref_samples, frag, parts, _ = O.render(cfg, sc["means"].double(), sc["log_scales"].double().exp(), sc["quats"].double(),
                                                       torch.sigmoid(sc["opacity_logits"].double()), sc["sh"].double(),
                                                       sc["viewmat"].double(), sc["lin_vel"].double(), sc["ang_vel"].double(),
                                                       background=bg.double(), return_parts=True)
        q = {k: sc[k].float().to(dev) for k in ("means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat")}
        times, samp, band = gs.subpose_schedule(S, et, R, rt)
        vms = gs.subpose_viewmats(q["viewmat"], q["lin_vel"], q["ang_vel"], torch.tensor(times, device=dev))
        samples, _, _ = gs.render_subposes(q["means"], q["log_scales"].exp(), q["quats"], torch.sigmoid(q["opacity_logits"]), q["sh"],
                                           vms, bg.to(dev), S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, sh_degree=3)
    err = (samples.cpu().double() - ref_samples).abs().max(dim=-1).values          # [S,H,W]
    print(f"== {tag}: {W}x{H}, {n} Gaussians; frame-level fragile share {float(frag.float().mean()):.4f} "
          f"(union over {S * R} sub-poses); per sub-pose {np.mean([float(p[4].fragile.float().mean()) for p in parts]):.4f}")
    bins = np.linspace(0.0, 1.0, 11)
    for row, name in ((0, "alpha = 1/255"), (1, "T = 1e-4")):
        flagged = np.zeros(10, int)
        flipped = np.zeros(10, int)
        for p, part in enumerate(parts):
            m = part[4].margin[row].numpy()
            e = err[samp[p]].numpy()
            sel = m < 1.0
            idx = np.minimum((m[sel] * 10).astype(int), 9)
            np.add.at(flagged, idx, 1)
            np.add.at(flipped, idx[e[sel] > IMG_ATOL], 1)
        print(f"  {name}: decisions inside the band, by depth (fraction of the band) -> flagged / flipped on the HIP path")
        print("    " + "  ".join(f"{bins[i]:.1f}-{bins[i + 1]:.1f}: {flagged[i]}/{flipped[i]}" for i in range(10)))
        print(f"    total {flagged.sum()} flagged, {flipped.sum()} flipped")
    unflagged_bad = int(((err > IMG_ATOL) & ~frag[None]).sum())
    print(f"  pixels OUTSIDE every band whose HIP sample differs by more than {IMG_ATOL}: {unflagged_bad} (must be 0)")
