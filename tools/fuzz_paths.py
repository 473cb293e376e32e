"""Randomised equivalence check of the default path (depth slices, exact culling, compact emission from hit
masks, deferred colour, gradient tuples) against the plainest one (one slice, no culling, atomics) over random
sizes / sub-pose layouts / slice budgets.  Images must be bit-identical, gradients equal up to summation order.
usage: python tools/fuzz_paths.py [trials] [seed]"""
import random
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import gsdeblur_amd as gs  # noqa: E402
from gsdeblur_amd import ops  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = random.Random(seed)
dev = torch.device("cuda", 0)
KNOBS = ("SLICE_BASE", "EXACT_TILE_CULL", "COMPACT_EMIT", "HIT_MASKS", "GRAD_TUPLES", "DEFER_COLOR")
saved = {k: getattr(ops, k) for k in KNOBS}
bad = 0
t0 = time.time()
for trial in range(trials):
    n = rng.choice([1, 2, 7, 64, 300, 2000, 8000, 30000, 120000])
    W, H = rng.randint(17, 900), rng.randint(17, 600)
    S, R = rng.choice([1, 2, 3]), rng.choice([1, 1, 2, 4])
    mult = rng.choice([1.0, 3.0, 6.0, 12.0])
    base = rng.choice([1, 4, 16, 64, 512])
    sc = gs.data.synthetic_scene(n, W, H, seed=1000 + trial, scale_mult=mult)
    sc = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
    times, _, _ = gs.subpose_schedule(S, 1 / 60, R, 1 / 30)
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(trial)).to(dev)
    res = []
    try:
        for plain in (False, True):
            for k in KNOBS:
                setattr(ops, k, 0 if plain else saved[k])
            if not plain:
                ops.SLICE_BASE = base
            p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
            vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"] * 20, sc["ang_vel"] * 10,
                                      torch.tensor(times, device=dev))
            rgb, alphas, radii = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"],
                                                    torch.sigmoid(p["opacity_logits"]), p["sh"], vms,
                                                    torch.tensor([0.1, 0.2, 0.3], device=dev), S, R, sc["fx"], sc["fy"],
                                                    sc["cx"], sc["cy"], H, W, gamma=2.2, min_rgb_level=10.0)
            ((rgb * wt).sum() + 0.5 * alphas.sum()).backward()
            res.append((rgb.detach().clone(), alphas.detach().clone(), {k: v.grad.clone() for k, v in p.items()},
                        len(ops.last_slice_intersects)))
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
    (img_f, al_f, g_f, nsl), (img_p, al_p, g_p, _) = res
    ok = torch.equal(img_f, img_p) and torch.equal(al_f, al_p)
    worst = 0.0
    for k in g_f:
        a, b = g_f[k].double().cpu().numpy(), g_p[k].double().cpu().numpy()
        worst = max(worst, float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)))
    ok = ok and worst < 3e-3 and all(torch.isfinite(v).all() for v in g_f.values())
    bad += 0 if ok else 1
    print(f"trial {trial:3d} n={n:6d} {W}x{H} S={S} R={R} mult={mult} base={base} slices={nsl} "
          f"img_equal={torch.equal(img_f, img_p)} grad_rel={worst:.1e} {'ok' if ok else 'FAIL'}", flush=True)
print(f"fuzz: {trials - bad}/{trials} trials ok in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
