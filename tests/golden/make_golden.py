"""Generates tests/golden/*.npz with the float64 oracle (oracle/gs_oracle.py).

SELF-GOLDEN, NOT REFERENCE-GOLDEN: the reference's implementation of this path (the
SpectacularAI gsplat/nerfstudio forks) is not vendored under /root/reference and cannot be
imported, so no reference outputs exist to record.  These fixtures freeze the
oracle's answers so that (a) the oracle cannot drift silently and (b) the HIP path is checked
against committed numbers on the GPU box.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import gs_oracle as O  # noqa: E402

OUT = Path(__file__).resolve().parent


def scene_arrays(sc):
    return {k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in sc.items()}


def make_case(name, n, W, H, cfg_kw, seed, sh_degree=3, scale_mult=16.0):
    sc = O.synthetic_scene(n, W, H, sh_degree=sh_degree, seed=seed, scale_mult=scale_mult)
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], sh_degree=sh_degree, **cfg_kw)
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel"]
    ps = {k: sc[k].double().requires_grad_(True) for k in names}
    V = sc["viewmat"].double().requires_grad_(True)
    bg = torch.tensor([0.1, 0.2, 0.3], dtype=torch.float64)
    out, alpha, samples, frag, parts, vms = O.render(
        cfg, ps["means"], ps["log_scales"].exp(), ps["quats"], torch.sigmoid(ps["opacity_logits"]), ps["sh"], V,
        ps["lin_vel"], ps["ang_vel"], background=bg, return_parts=True)
    g = torch.Generator().manual_seed(seed + 1)
    wt = torch.rand(H, W, 3, generator=g, dtype=torch.float64) * (~frag)[..., None]
    (out * wt).sum().backward()
    # float32 integer parity data for sub-pose 0 (same viewmat in float32)
    pr32 = O.project_gaussians(sc["means"], sc["log_scales"].exp(), 1.0, sc["quats"], vms[0].detach().float(),
                               sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W)
    keys, gids = O.map_gaussian_to_intersects(pr32, W)
    skeys, sgids = O.sort_intersects(keys, gids)
    d = scene_arrays(sc)
    d.update(
        cfg=np.array([H, W, cfg.blur_samples, cfg.rs_bands, sh_degree], dtype=np.int64),
        cfg_f=np.array([cfg.exposure_time, cfg.rolling_shutter_time, cfg.gamma, cfg.min_rgb_level], dtype=np.float64),
        background=bg.numpy(), weights=wt.numpy(), fragile=frag.numpy(),
        out=out.detach().numpy(), alpha=alpha.detach().numpy(), samples=samples.detach().numpy(),
        viewmats=vms.detach().numpy(),
        g_means=ps["means"].grad.numpy(), g_log_scales=ps["log_scales"].grad.numpy(),
        g_quats=ps["quats"].grad.numpy(), g_opacity_logits=ps["opacity_logits"].grad.numpy(),
        g_sh=ps["sh"].grad.numpy(), g_lin_vel=ps["lin_vel"].grad.numpy(), g_ang_vel=ps["ang_vel"].grad.numpy(),
        g_viewmat=V.grad.numpy(),
        p0_radii=pr32.radii.numpy(), p0_num_tiles_hit=pr32.num_tiles_hit.numpy(),
        p0_sorted_isect_ids=skeys, p0_sorted_gaussian_ids=sgids,
    )
    np.savez_compressed(OUT / f"{name}.npz", **d)
    print(name, "I0 =", len(skeys), "out mean", float(out.detach().mean()), "alpha mean", float(alpha.detach().mean()),
          "fragile px", int(frag.sum()))


if __name__ == "__main__":
    torch.set_num_threads(8)
    make_case("static_small", 600, 64, 48, dict(blur_samples=1, rs_bands=1), seed=11, scale_mult=5.0)
    make_case("blur_rs_small", 500, 80, 64,
              dict(blur_samples=3, rs_bands=2, exposure_time=1 / 60, rolling_shutter_time=1 / 30, gamma=2.2,
                   min_rgb_level=10.0), seed=12)
