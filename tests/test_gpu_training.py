"""GPU parity of the training step's HIP kernels (csrc/train.hip) against their torch restatements:
the image loss (0.8 L1 + 0.2 (1 - SSIM)) forward + backward against float64 torch autograd of the same formula,
the multi-tensor Adam against torch.optim.Adam(eps=1e-15) over many steps, and a whole train_step through both."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _images(H, W, seed, flat=False):
    """a smooth-ish prediction / ground-truth pair in [0, 1] with saturated and perfectly flat regions (the SSIM
    variances cancel to ~0 there: the badly conditioned spot of the fp32 formula)"""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 6.28, H), torch.linspace(0, 9.42, W), indexing="ij")
    gt = torch.stack([0.5 + 0.4 * torch.sin(xx + c) * torch.cos(yy * (1 + 0.3 * c)) for c in range(3)], -1)
    gt = (gt + 0.05 * torch.randn(H, W, 3, generator=g)).clamp(0, 1)
    pred = (gt + 0.1 * torch.randn(H, W, 3, generator=g) + 0.05).clamp(0, 1)
    if flat and H > 24 and W > 24:
        gt[: H // 3, : W // 3] = 0.25
        pred[: H // 3, : W // 3] = 0.25          # identical flat patch: |gt - pred| = 0 (the L1 kink), variances 0
        pred[-H // 4:, -W // 4:] = 1.0           # saturated prediction
    return pred, gt


@pytest.mark.parametrize("H,W,lam,flat", [(11, 11, 0.2, False), (37, 53, 0.2, True), (120, 161, 0.2, True),
                                          (64, 48, 1.0, False), (9, 7, 0.0, False), (270, 480, 0.2, True)])
def test_image_loss_fwd_bwd_vs_float64_torch(gs, dev, H, W, lam, flat):
    """loss value, its L1 / SSIM parts and d loss / d pred against float64 autograd of train_step.image_loss_torch
    (pytorch_msssim-style SSIM: 11x11 Gaussian, sigma 1.5, valid convolution).  Ragged sizes (tile edges at 32 x 8),
    the minimum 11x11 image (one window), lambda 0 (L1 only, any size) and 1 (SSIM only)."""
    pred, gt = _images(H, W, seed=H * 1000 + W, flat=flat)
    p64 = pred.double().requires_grad_(True)
    ref = gs.training.image_loss_torch(p64, gt.double(), lam)
    ref.backward()
    pg = pred.to(dev).requires_grad_(True)
    loss, parts = gs.fused.image_loss(pg, gt.to(dev), lam, return_parts=True)
    (3.0 * loss).backward()                                                   # the upstream gradient scales through
    assert abs(float(loss.detach()) - float(ref.detach())) < 3e-6            # fp32 moments: E[xx] - mu^2 cancels in flat windows
    l1_ref = float((gt.double() - pred.double()).abs().mean())
    assert abs(float(parts[0]) - l1_ref) < 1e-6
    if lam > 0:
        assert abs(float(parts[1]) - float(gs.training.ssim(pred.double(), gt.double()))) < 1e-5
    got = pg.grad.cpu().double().numpy() / 3.0
    want = p64.grad.numpy()
    scale = np.abs(want).max()
    err = np.abs(got - want)
    # fp32 moments: sigma_xx = E[xx] - mu^2 loses ~1e-7 absolute, the SSIM denominators are >= C2 = 9e-4, so the
    # per-pixel derivative carries up to ~1e-3 relative error where the window is flat; everywhere else ~1e-5
    assert err.max() <= 2e-3 * scale, (err.max(), scale)
    assert err.mean() <= 2e-5 * scale
    # the L1 kink: identical pixels get NO L1 gradient (torch's sign(0) = 0)
    if flat and lam == 0.2 and H > 24:
        same = (pred == gt).numpy()
        assert same.any()


def test_image_loss_matches_the_torch_gpu_formulation_and_small_images_fall_back_to_l1(gs, dev):
    pred, gt = _images(96, 128, seed=5, flat=True)
    pg, gg = pred.to(dev), gt.to(dev)
    a = float(gs.fused.image_loss(pg, gg, 0.2))
    b = float(gs.training.image_loss_torch(pg, gg, 0.2))                       # conv2d on the GPU, fp32
    assert abs(a - b) < 5e-6
    # a frame smaller than the 11x11 SSIM window (only while the resolution schedule has it downscaled): L1 only
    small = float(gs.fused.image_loss(pg[:10], gg[:10], 0.2))
    assert abs(small - float((pg[:10] - gg[:10]).abs().mean())) < 1e-6
    with pytest.raises(ValueError, match="no CPU fallback"):
        gs.fused.image_loss(pred, gt, 0.2)


def test_hip_adam_equals_torch_adam_over_many_steps(gs, dev):
    """HipAdam / adam_step_all against torch.optim.Adam(eps=1e-15) on the six Gaussian tensors' shapes (sizes that
    are not multiples of four, a 3-element tensor, a missing gradient), different learning rates, 25 steps with
    sparse gradients (most rows zero, as after the rasterizer's backward): parameters and both moments agree to fp32
    rounding of the update rule."""
    N = 1237
    shapes = [(N, 3), (N, 3), (N, 4), (N, 1), (N, 3), (N, 15, 3), (3,), (1, 6)]
    lrs = [1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3, 1.25e-4, 1e-3, 1e-4]
    g = torch.Generator().manual_seed(3)
    init = [torch.randn(s, generator=g) for s in shapes]
    pa = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    pb = [torch.nn.Parameter(t.clone().to(dev)) for t in init]
    oa = [gs.fused.HipAdam([p], lr=lr, eps=1e-15) for p, lr in zip(pa, lrs)]
    ob = [torch.optim.Adam([p], lr=lr, eps=1e-15) for p, lr in zip(pb, lrs)]
    for step in range(25):
        rows = (torch.rand(N, generator=g) < 0.1)
        for i, s in enumerate(shapes):
            if i == 6 and step % 3 == 0:
                pa[i].grad = pb[i].grad = None                                # a step without a gradient: skipped
                continue
            gr = torch.randn(s, generator=g) * (10.0 ** float(torch.randint(-4, 2, (1,), generator=g)))
            if s[0] == N:
                gr = gr * rows.view(-1, *([1] * (len(s) - 1)))
            pa[i].grad = gr.to(dev)
            pb[i].grad = gr.to(dev).clone()
        gs.fused.adam_step_all(oa)
        for o in ob:
            o.step()
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.allclose(a.data, b.data, rtol=2e-5, atol=1e-6), i
        sa, sb = oa[i].state[a], ob[i].state[b]
        assert int(sa["step"]) == int(sb["step"])
        # (elements of exp_avg that pass through zero carry the rounding of the terms that cancelled: absolute floor)
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-6 * float(sb["exp_avg"].abs().max())), i
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5,
                              atol=1e-7 * float(sb["exp_avg_sq"].abs().max())), i
    # a single optimizer's own .step() is the same launch
    solo = gs.fused.HipAdam([torch.nn.Parameter(init[0].clone().to(dev))], lr=1e-3, eps=1e-15)
    p = solo.param_groups[0]["params"][0]
    p.grad = torch.ones_like(p)
    solo.step()
    assert torch.allclose(p.data, init[0].to(dev) - 1e-3, atol=1e-6)
    with pytest.raises(ValueError, match="no CPU fallback"):
        bad = gs.fused.HipAdam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3)
        bad.param_groups[0]["params"][0].grad = torch.ones(4)
        bad.step()


def test_train_step_fused_equals_torch_loss_and_optimizers(gs, dev, oracle):
    """three training iterations of the same model through (HIP loss + HipAdam) and through (torch conv2d SSIM loss +
    torch.optim.Adam): the render is the same HIP path in both, so parameters must agree to the kernels' tolerance"""
    H, W, n = 96, 128, 4000
    sc = oracle.synthetic_scene(n, W, H, seed=21, scale_mult=5.0)
    cfg = gs.SplatfactoDeblurConfig(blur_samples=3, rolling_shutter_compensation=False, use_scale_regularization=True)
    c2w = torch.eye(4)[:3].clone()
    c2w[:, 1] *= -1
    c2w[:, 2] *= -1                        # OpenGL camera looking down the oracle scene's +z
    cam = gs.Camera(c2w, sc["fx"], sc["fy"], sc["cx"], sc["cy"], W, H,
                    metadata=dict(cam_idx=0, camera_linear_velocity=[0.3, 0.1, 0.0],
                                  camera_angular_velocity=[0.0, 0.2, 0.1], exposure_time=1 / 60,
                                  rolling_shutter_time=0.0))
    target, _ = _images(H, W, seed=9)
    target = target.to(dev)
    models = [gs.SplatfactoDeblurModel.from_scene(cfg, sc, dev) for _ in range(2)]
    opts = [gs.training.make_optimizers(models[0]), gs.training.make_optimizers(models[1], fused=False)]
    assert type(opts[0]["means"]).__name__ == "HipAdam" and type(opts[1]["means"]).__name__ == "Adam"
    for it in range(3):
        gs.training.train_step(models[0], opts[0], cam, target, 0.2)
        # the torch twin: same render, torch loss, torch Adam
        m = models[1]
        m.train()
        for o in opts[1].values():
            o.zero_grad(set_to_none=True)
        out = m.get_outputs(cam)
        loss = gs.training.image_loss_torch(out["rgb"], target, 0.2) + gs.training.scale_regularization(m.scales)
        loss.backward()
        for o in opts[1].values():
            o.step()
    for (k, a), b in zip(models[0].gauss_params().items(), models[1].gauss_params().values()):
        d = (a.data - b.data).abs().max().item()
        assert d <= 2e-5 * max(1.0, b.data.abs().max().item()), (k, d)
