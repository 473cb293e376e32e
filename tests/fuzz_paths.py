"""Randomised equivalence check of the default path (depth slices, exact culling, compact emission from hit
masks, deferred colour, gradient tuples) against the plainest one (one slice, no culling, atomics) over random
sizes / sub-pose layouts / slice budgets.  Images must be bit-identical, gradients equal up to summation order.
With `oracle` as third argument the default path is ALSO held against the float64 CPU oracle (tiny sizes only;
test infrastructure, never part of the product path).
With `pixvel` as an argument the trials render the paper's pixel-velocity model instead of the SE(3) re-projection, a
third of them with EXACT per-row rolling shutter (round 3); the exact mode has no atomics path, so its other side is the
Python orchestration rendering ONE slice.
usage: python tests/fuzz_paths.py [trials] [seed] [oracle] [pixvel] [only:<trial>]   (only: replay the draws, run that trial alone)"""
import random
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import gsdeblur_amd as gs  # noqa: E402
from gsdeblur_amd import ops  # noqa: E402
sys.path.insert(0, str(Path(__file__).resolve().parent))
import python_frame_path  # noqa: E402    (the Python orchestration twin and its switches: test infrastructure)
python_frame_path.install()

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
with_oracle = len(sys.argv) > 3 and sys.argv[3] == "oracle"
pixvel = "pixvel" in sys.argv[1:]
only = next((int(a[5:]) for a in sys.argv[1:] if a.startswith("only:")), None)
dump = next((a[5:] for a in sys.argv[1:] if a.startswith("dump:")), None)     # with only: save the trial's inputs + gradients
if with_oracle:
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "oracle"))
    import gs_oracle as O  # noqa: E402
rng = random.Random(seed)
dev = torch.device("cuda", 0)
KNOBS = ("SLICE_BASE", "EXACT_TILE_CULL", "COMPACT_EMIT", "HIT_MASKS", "GRAD_TUPLES", "DEFER_COLOR")
# route switches that must not change anything: drawn at random for the default-path side of every trial
ROUTES = ("SPECULATE", "TILE_SORT_CARRY", "DEPTH_SORT_COMPACT", "DEVICE_SIZES", "PREALLOC_BWD")
# round 3: half of the trials go through the C++ frame orchestration (gs_frame_forward / gs_frame_backward; it needs the
# default routes) with slice merging and the polled read-backs drawn at random; the radix passes run in either form
FRAME = ("NATIVE_FRAME", "SLICE_MERGE", "FRAME_POLL", "LAZY_RECORDS", "DEPTH_SELECT")
saved = {k: getattr(ops, k) for k in KNOBS + ROUTES + FRAME}
from gsdeblur_amd import _lib  # noqa: E402
_L = _lib.load()
bad = 0
t0 = time.time()
for trial in range(trials):
    n = rng.choice([1, 2, 7, 64, 300, 2000, 8000, 30000, 120000])
    W, H = rng.randint(17, 900), rng.randint(17, 600)
    if with_oracle:
        n, W, H = rng.choice([1, 5, 40, 200, 600]), rng.randint(17, 80), rng.randint(17, 64)
    deg = rng.choice([0, 1, 2, 3, 3])
    aa = rng.choice([True, True, False])
    gamma, mlevel = rng.choice([(2.2, 10.0), (2.2, 0.0), (1.0, 0.0)])
    bg = rng.choice([None, torch.tensor([0.1, 0.2, 0.3]), torch.tensor([1.0, 1.0, 1.0])])
    S, R = rng.choice([1, 2, 3]), rng.choice([1, 1, 2, 4])
    mult = rng.choice([1.0, 3.0, 6.0, 12.0])
    base = rng.choice([1, 4, 16, 64, 512])
    sc = gs.data.synthetic_scene(n, W, H, sh_degree=deg, seed=1000 + trial, scale_mult=mult,
                                 profile=rng.choice(["survey", "survey", "trained"]))
    if rng.random() < 0.3:            # a few opacities above the 0.999 alpha clamp (tile_hot / clamping loop version)
        sc["opacity_logits"] = sc["opacity_logits"].clone()
        # 7.5: sigmoid = 0.99945 (above the clamp) while torch's own fp32 sigmoid backward, y*(1-y), still resolves
        # 1-y to 1e-4 relative (at logit 14 it does not, and the oracle comparison of that gradient measures torch)
        sc["opacity_logits"][::rng.choice([3, 17, 101])] = 7.5
    routes = {k: rng.choice([0, 1]) for k in ROUTES}
    # half of the trials see the scene from a random rigid pose (world = R^T (camera - t), viewmat = [R | t]): from the
    # identity pose a pose applied from the wrong side or a camera centre read from the wrong column cannot show
    if rng.random() < 0.5:
        import math
        ax, ang_ = [rng.uniform(-1, 1) for _ in range(3)], rng.uniform(0.2, 2.5)
        nrm = math.sqrt(sum(a * a for a in ax)) + 1e-9
        kx, ky, kz = (a / nrm for a in ax)
        Kc = torch.tensor([[0.0, -kz, ky], [kz, 0.0, -kx], [-ky, kx, 0.0]])
        Rm = torch.eye(3) + math.sin(ang_) * Kc + (1 - math.cos(ang_)) * (Kc @ Kc)
        tv = torch.tensor([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1)])
        sc = dict(sc)
        sc["means"] = (sc["means"] - tv) @ Rm
        Vp = torch.eye(4)
        Vp[:3, :3], Vp[:3, 3] = Rm, tv
        sc["viewmat"] = Vp
    native = rng.random() < 0.5
    frame = {"NATIVE_FRAME": int(native), "SLICE_MERGE": rng.choice([0.0, 0.3, 0.75]), "FRAME_POLL": rng.choice([0, 1]),
             # round 5: lazy records (never / always), drawn from a generator of its own so that the older draws — and
             # with them every trial of an earlier seed — stay what they were
             "LAZY_RECORDS": random.Random(seed * 7919 + trial).choice([0, 2]),
             # round 6: nearest-first selection (never / always), its own generator again
             "DEPTH_SELECT": random.Random(seed * 104729 + trial).choice([0, 2])}
    single_pass = rng.choice([0, 1])      # (round 6: the single-pass sort is gone; the draw stays so that earlier seeds replay)
    rs_time = 0.0
    if pixvel:
        rs_time = rng.choice([0.0, 0.0, 1 / 30])
        if rs_time:
            R = 1
            routes = {k: saved[k] for k in ROUTES}       # (the A/B routes are not wired for the exact mode)
    if native:
        routes = {k: saved[k] for k in ROUTES}
    sc_cpu = sc
    if only is not None and trial != only:
        continue
    sc = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}
    times, _, _ = gs.subpose_schedule(S, 1 / 60, R, 1 / 30)
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(trial)).to(dev)
    res = []
    twice_any = False
    try:
        for plain in (False, True):
            for k in KNOBS:
                setattr(ops, k, 0 if plain else saved[k])
            for k in ROUTES:
                setattr(ops, k, saved[k] if plain else routes[k])
            for k in FRAME:
                setattr(ops, k, (0 if k in ("NATIVE_FRAME", "DEPTH_SELECT") else saved[k]) if plain else frame[k])
            if plain and rs_time:
                # exact rolling shutter needs the tuple backward and box lists: its "plain" side is the Python
                # orchestration with every planned slice folded into one
                for k in KNOBS:
                    setattr(ops, k, saved[k])
                ops.SLICE_BASE = 0
            if not plain:
                ops.SLICE_BASE = base
            # round 6: half of the selecting trials render the frame TWICE through one FrameHints and compare the second
            # frame — its sort runs with the first frame's selection size as a promise (the one-block tail of the
            # selective sort, the overflow re-sort when the promise is too small)
            hints = ops.FrameHints()
            twice = (not plain) and frame["DEPTH_SELECT"] == 2 and random.Random(seed * 7919 + trial).random() < 0.5
            twice_any = twice_any or twice
            for _warm in range(2 if twice else 1):
              p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
              if pixvel:
                  rgb, alphas, radii = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"],
                                                          torch.sigmoid(p["opacity_logits"]), p["sh"], sc["viewmat"],
                                                          None if bg is None else bg.to(dev), S, R, sc["fx"], sc["fy"],
                                                          sc["cx"], sc["cy"], H, W, gamma=gamma, min_rgb_level=mlevel,
                                                          sh_degree=deg, antialiased=aa, lin_vel=sc["lin_vel"] * 20,
                                                          ang_vel=sc["ang_vel"] * 10, times=torch.tensor(times, device=dev),
                                                          rolling_shutter_time=rs_time, hints=hints)
              else:
                  vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"] * 20, sc["ang_vel"] * 10,
                                            torch.tensor(times, device=dev))
                  rgb, alphas, radii = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"],
                                                          torch.sigmoid(p["opacity_logits"]), p["sh"], vms,
                                                          None if bg is None else bg.to(dev), S, R, sc["fx"], sc["fy"],
                                                          sc["cx"], sc["cy"], H, W, gamma=gamma, min_rgb_level=mlevel,
                                                          sh_degree=deg, antialiased=aa, hints=hints)
              ((rgb * wt).sum() + 0.5 * alphas.sum()).backward()
            res.append((rgb.detach().clone(), alphas.detach().clone(), {k: v.grad.clone() for k, v in p.items()},
                        len(ops.last_slice_intersects)))
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
    (img_f, al_f, g_f, nsl), (img_p, al_p, g_p, _) = res
    ok = torch.equal(img_f, img_p) and torch.equal(al_f, al_p)
    worst, worst_key = 0.0, ""
    for k in g_f:
        a, b = g_f[k].double().cpu().numpy(), g_p[k].double().cpu().numpy()
        d = float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
        if d > worst:
            worst, worst_key = d, k
        if only is not None:
            i = int(np.abs(a - b).reshape(a.shape[0], -1).max(1).argmax())
            print(f"   {k}: rel {d:.2e}  max|b| {np.abs(b).max():.3e}  worst row {i}: fast {a.reshape(a.shape[0], -1)[i][:4]} plain "
                  f"{b.reshape(b.shape[0], -1)[i][:4]}  logit {float(sc_cpu['opacity_logits'][i]):.2f}")
    ok = ok and worst < 3e-3 and all(torch.isfinite(v).all() for v in g_f.values())
    if dump and only is not None:
        np.savez_compressed(dump, W=W, H=H, S=S, R=R, deg=deg, aa=int(aa), gamma=gamma, mlevel=mlevel, base=base,
                            bg=np.zeros(0) if bg is None else bg.numpy(), wt=wt.cpu().numpy(),
                            **{"in_" + k: v.numpy() for k, v in sc_cpu.items() if isinstance(v, torch.Tensor)},
                            **{"sc_" + k: float(v) for k, v in sc_cpu.items() if not isinstance(v, torch.Tensor)},
                            **{"fast_" + k: v.cpu().numpy() for k, v in g_f.items()},
                            **{"plain_" + k: v.cpu().numpy() for k, v in g_p.items()})
    extra = ""
    if with_oracle:
        cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=R,
                             exposure_time=1 / 60, rolling_shutter_time=1 / 30, gamma=gamma, min_rgb_level=mlevel,
                             sh_degree=deg, antialiased=aa, motion_model="pixel_velocity" if pixvel else "se3",
                             rs_exact=bool(rs_time))
        q = {k: sc_cpu[k].double().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
        ref, ref_a, _, frag, _, _ = O.render(cfg, q["means"], q["log_scales"].exp(), q["quats"],
                                             torch.sigmoid(q["opacity_logits"]), q["sh"], sc_cpu["viewmat"].double(),
                                             (sc_cpu["lin_vel"] * 20).double(), (sc_cpu["ang_vel"] * 10).double(),
                                             background=None if bg is None else bg.double(), return_parts=True)
        loss_ref = (ref * wt.cpu().double()).sum() + 0.5 * S * ref_a.sum()         # sum_s alpha_s = S * mean
        if loss_ref.requires_grad:
            loss_ref.backward()
        for k in q:                                     # nothing on screen: the frame does not depend on the scene
            if q[k].grad is None:
                q[k].grad = torch.zeros_like(q[k])
        good = ~frag
        d_img = float((img_f.cpu().double() - ref)[good].abs().max()) if good.any() else 0.0
        d_grad = 0.0
        if float(frag.float().mean()) < 0.02:          # gradients are only comparable when no threshold decision is at risk
            for k in g_f:
                a, b = g_f[k].double().cpu().numpy(), q[k].grad.numpy()
                d_grad = max(d_grad, float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30)))
        ok = ok and d_img < 5e-4 and d_grad < 3e-3
        extra = f" oracle: img {d_img:.1e} grad {d_grad:.1e} fragile {float(frag.float().mean()):.3f}"
    bad += 0 if ok else 1
    print(f"trial {trial:3d} n={n:6d} {W}x{H} S={S} R={R} mult={mult} base={base} slices={nsl} "
          f"deg={deg} aa={int(aa)} gamma={gamma} routes={''.join(str(routes[k]) for k in ROUTES)} "
          f"frame={int(native)}/{frame['SLICE_MERGE']}/{frame['FRAME_POLL']}/lazy{frame['LAZY_RECORDS']}/sel{frame['DEPTH_SELECT']}{'x2' if twice_any else ''} "
          f"{'pixvel rs=%.3f ' % rs_time if pixvel else ''}img_equal={torch.equal(img_f, img_p)} grad_rel={worst:.1e} ({worst_key}){extra} "
          f"{'ok' if ok else 'FAIL'}", flush=True)
print(f"fuzz: {trials - bad}/{trials} trials ok in {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
