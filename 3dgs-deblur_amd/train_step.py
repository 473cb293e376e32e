"""Loss + optimizer step around the hot path: what the nerfstudio fork's trainer
does right after `get_outputs` — splatfacto's `0.8*L1 + 0.2*(1-SSIM)` image loss, the scale
regularisation switched on by /root/reference/train.py:120 (`use-scale-regularization`), and one Adam
step per parameter group.  On the GPU the loss (forward + backward) and the optimizer step are HIP kernels
(:mod:`fused`, csrc/train.hip): two tile kernels instead of nine conv2d launches plus their autograd graph, one
multi-tensor Adam launch instead of six foreach passes.  The torch formulations below are the same math for CPU
tensors (the gloo host-logic tests drive `train_step` with a CPU stand-in for the render) and the oracle the HIP
kernels are tested against; a CUDA tensor never takes them silently — `image_loss` / `make_optimizers` choose by
device and the HIP side raises when the library is missing.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from .model import Camera, SplatfactoDeblurModel


def _gauss_window(size: int = 11, sigma: float = 1.5, device=None, dtype=torch.float32) -> Tensor:
    x = torch.arange(size, dtype=torch.float32, device=device) - (size - 1) / 2.0
    g = torch.exp(-(x * x) / (2 * sigma * sigma))
    g = (g / g.sum()).to(dtype)          # the window is built in float32 (as pytorch_msssim does), whatever the images
    return (g[:, None] * g[None, :])[None, None]


def ssim(img: Tensor, ref: Tensor, window: int = 11) -> Tensor:
    """Mean SSIM of two [H,W,3] images in [0,1] (Gaussian 11x11 window, sigma 1.5, valid padding)."""
    x = img.permute(2, 0, 1)[None]
    y = ref.permute(2, 0, 1)[None]
    w = _gauss_window(window, 1.5, img.device, img.dtype).expand(3, 1, window, window)
    mu_x, mu_y = F.conv2d(x, w, groups=3), F.conv2d(y, w, groups=3)
    sxx = F.conv2d(x * x, w, groups=3) - mu_x * mu_x
    syy = F.conv2d(y * y, w, groups=3) - mu_y * mu_y
    sxy = F.conv2d(x * y, w, groups=3) - mu_x * mu_y
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    s = ((2 * mu_x * mu_y + c1) * (2 * sxy + c2)) / ((mu_x * mu_x + mu_y * mu_y + c1) * (sxx + syy + c2))
    return s.mean()


def psnr(img: Tensor, ref: Tensor) -> float:
    mse = torch.mean((img.clamp(0, 1) - ref.clamp(0, 1)) ** 2).item()
    return float("inf") if mse == 0 else -10.0 * math.log10(mse)


# A/B switch: GSD_TORCH_TRAIN=1 runs the loss as torch ops and the optimizers as torch.optim.Adam on the GPU too
TORCH_TRAIN = int(os.environ.get("GSD_TORCH_TRAIN", "0"))


def image_loss(pred: Tensor, gt: Tensor, ssim_lambda: float = 0.2) -> Tensor:
    if pred.is_cuda and not TORCH_TRAIN:
        from . import fused
        return fused.image_loss(pred, gt, ssim_lambda)
    return image_loss_torch(pred, gt, ssim_lambda)


def image_loss_torch(pred: Tensor, gt: Tensor, ssim_lambda: float = 0.2) -> Tensor:
    """the same loss in plain torch ops (CPU tensors; reference for the HIP kernels' tests)"""
    l1 = torch.abs(gt - pred).mean()
    if ssim_lambda <= 0:
        return l1
    return (1.0 - ssim_lambda) * l1 + ssim_lambda * (1.0 - ssim(pred, gt))


def downscale_image(img: Tensor, d: int) -> Tensor:
    """[H,W,3] -> [H//d, W//d, 3] by averaging d x d blocks: splatfacto's `_downscale_if_required` / `resize_image`
    (a stride-d convolution with uniform weights) applied to the ground truth while the resolution schedule is active"""
    if d <= 1:
        return img
    return F.avg_pool2d(img.permute(2, 0, 1)[None], kernel_size=d, stride=d)[0].permute(1, 2, 0).contiguous()


def scale_regularization(log_scales: Tensor, max_gauss_ratio: float = 10.0) -> Tensor:
    """Penalise needle-like Gaussians (PhysGaussian-style, as in splatfacto): mean(max(s_max/s_min, r) - r)."""
    s = torch.exp(log_scales)
    ratio = s.amax(dim=-1) / s.amin(dim=-1)
    r = torch.tensor(max_gauss_ratio, device=s.device, dtype=s.dtype)
    return 0.1 * (torch.maximum(ratio, r) - r).mean()


def make_optimizers(model: SplatfactoDeblurModel, lr_scale: float = 1.0,
                    fused: Optional[bool] = None) -> Dict[str, torch.optim.Optimizer]:
    """One Adam per parameter group with splatfacto's default learning rates.  fused (default: the parameters live on
    a GPU): HipAdam — same update rule and state layout as torch.optim.Adam, stepped by ONE multi-tensor HIP launch
    (`optimizers_step`); False: torch.optim.Adam (CPU tensors, A/B)."""
    lrs = {"means": 1.6e-4, "scales": 5e-3, "quats": 1e-3, "opacities": 5e-2, "features_dc": 2.5e-3,
           "features_rest": 2.5e-3 / 20}
    if fused is None:
        fused = model.means.is_cuda and not TORCH_TRAIN
    if fused:
        from .fused import HipAdam as Adam
    else:
        Adam = torch.optim.Adam
    opts = {k: Adam([p], lr=lrs[k] * lr_scale, eps=1e-15) for k, p in model.gauss_params().items()}
    if model.pose_adjustment is not None:
        opts["camera_opt"] = Adam([model.pose_adjustment], lr=1e-4 * lr_scale, eps=1e-15)
    if model.velocity_adjustment is not None:
        opts["camera_velocity_opt"] = Adam([model.velocity_adjustment], lr=1e-3 * lr_scale, eps=1e-15)
    if model.background_param is not None:
        opts["background"] = Adam([model.background_param], lr=1e-3 * lr_scale, eps=1e-15)
    return opts


def optimizers_step(optimizers) -> None:
    """step every optimizer of the iteration; HipAdam instances share one multi-tensor launch"""
    opts = list(optimizers)
    if any(type(o).__name__ == "HipAdam" for o in opts):
        from .fused import adam_step_all
        adam_step_all(opts)
    else:
        for o in opts:
            o.step()


# GSD_TRAIN_AUTOGRAD=1: train_step goes through get_outputs + torch.autograd instead of the one-call route (A/B, tests)
TRAIN_AUTOGRAD = int(os.environ.get("GSD_TRAIN_AUTOGRAD", "0"))


def one_call_route(model: SplatfactoDeblurModel) -> bool:
    """train_step renders through model.render_and_backward (step.render_step) unless the model lives on the CPU (host
    logic tests with a stand-in render), the torch loss / autograd A/B switches are set, or training wants the depth
    output (a forward-only extra the one-call route does not produce)"""
    return (model.means.is_cuda and not TORCH_TRAIN and not TRAIN_AUTOGRAD
            and not model.config.output_depth_during_training and "get_outputs" not in model.__dict__
            and type(model).get_outputs is SplatfactoDeblurModel.get_outputs)


def train_step(model: SplatfactoDeblurModel, optimizers: Dict[str, torch.optim.Optimizer], camera: Camera,
               gt_image: Tensor, ssim_lambda: float = 0.2, allreduce: Optional[str] = None) -> Dict[str, float]:
    """One training iteration: render (HIP) -> loss -> backward (HIP) -> [DP gradient all-reduce] -> Adam."""
    model.train()
    for o in optimizers.values():
        o.zero_grad(set_to_none=True)
    gt_image = downscale_image(gt_image, model.downscale_factor())     # num_downscales resolution schedule
    if one_call_route(model):
        # forward + backward of the frame as ONE host call (model.render_and_backward -> step.render_step): the HIP loss
        # kernel's forward already produces d loss / d rgb, so it sits between the two halves as a plain callable
        from . import fused
        box = {}

        def grad_image(rgb):
            box["loss"], v, _ = fused.image_loss_with_grad(rgb, gt_image, ssim_lambda)
            return v
        rgb = model.render_and_backward(camera, grad_image)
        loss = box["loss"]
        if model.config.use_scale_regularization:
            reg = scale_regularization(model.scales)
            reg.backward()
            loss = loss + reg.detach()
    else:
        out = model.get_outputs(camera)
        rgb = out["rgb"].detach()
        loss = image_loss(out["rgb"], gt_image, ssim_lambda)
        if model.config.use_scale_regularization:
            loss = loss + scale_regularization(model.scales)
        loss.backward()
    if allreduce is not None:
        from . import dp
        dp.allreduce_gradients(list(model.gauss_params().values()), mode=allreduce, average=True)
        # the parameters that are not per-Gaussian rows (learnable background, pose / velocity adjustments) see
        # only this rank's views too: one small dense bucket, or the replicas drift apart silently
        small = [p for p in (model.background_param, model.pose_adjustment, model.velocity_adjustment)
                 if p is not None]
        for p in small:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        dp.allreduce_dense_([p.grad for p in small], average=True)
    optimizers_step(optimizers.values())
    model.step += 1
    # ONE read-back for the step's two log values (each .item() is a stream synchronisation)
    mse = F.mse_loss(rgb.clamp(0, 1), gt_image.clamp(0, 1))
    loss_v, mse_v = torch.stack([loss.detach().reshape(()).float(), mse.float()]).tolist()
    return {"loss": float(loss_v), "psnr": float("inf") if mse_v == 0 else -10.0 * math.log10(mse_v)}


# --------------------------------------------------------------------------- #
# whole-scene training / evaluation on a transforms.json dataset (the trainer-side callers of the hot path that the
# end-to-end deblurring check needs; /root/reference/train.py:78-109 scores runs the same way: PSNR / SSIM of the
# evaluation frames plus wall-clock time, stored as metrics.json)
# --------------------------------------------------------------------------- #
def eval_camera_step(model: SplatfactoDeblurModel, optimizers: Dict[str, torch.optim.Optimizer], camera: Camera,
                     gt_image: Tensor, ssim_lambda: float = 0.2) -> float:
    """`--optimize-eval-cameras` (/root/reference/train.py:180-183, README.md:197): one step on an EVALUATION frame
    in which only its pose / velocity adjustment is updated — the Gaussians are constants (no gradient reaches them,
    their optimizers do not step)."""
    model.train()
    cam_opts = [optimizers[k] for k in ("camera_opt", "camera_velocity_opt") if k in optimizers]
    if not cam_opts:
        return float("nan")
    for o in cam_opts:
        o.zero_grad(set_to_none=True)
    out = model.get_outputs(camera, detach_gaussians=True)
    gt_image = downscale_image(gt_image, model.downscale_factor())
    loss = image_loss(out["rgb"], gt_image, ssim_lambda)
    loss.backward()
    optimizers_step(cam_opts)
    return float(loss.item())


@torch.no_grad()
def evaluate(model: SplatfactoDeblurModel, cameras, images, indices) -> Dict[str, float]:
    """mean PSNR / SSIM of the model's renders of `indices` against their images (sharp frames in the synthetic sets)"""
    ps, ss = [], []
    for i in indices:
        rgb = model.get_outputs_for_camera(cameras[i])["rgb"]
        ps.append(psnr(rgb, images[i]))
        ss.append(float(ssim(rgb.clamp(0, 1), images[i]).item()))
    return {"psnr": sum(ps) / max(1, len(ps)), "ssim": sum(ss) / max(1, len(ss))}


def train_scene(model: SplatfactoDeblurModel, scene, images, iterations: int, lr_scale: float = 1.0,
                ssim_lambda: float = 0.2, optimize_eval_cameras: bool = False, eval_camera_every: int = 4,
                densify=None, log_every: int = 0, seed: int = 0) -> Dict:
    """Train on scene.train_indices (one view per step, seeded shuffle), optionally refining the evaluation cameras
    in between; returns {'results': {psnr, ssim}, 'wall_clock_time_seconds', 'history'} like the reference's
    metrics.json (/root/reference/train.py:87-100, parse_outputs.py:58)."""
    import time
    optimizers = make_optimizers(model, lr_scale)
    g = torch.Generator().manual_seed(seed)
    order = []
    history = []
    t0 = time.time()
    state = None
    if densify is not None:
        from . import densify as D
        model.collect_densify_stats = True
        state = D.DensifyState(model.num_points, model.means.device)
        if densify.num_train_data <= 0:
            # upstream's post-reset guard counts in passes over the training images (nerfstudio sets num_train_data
            # from the datamanager); the in-tree trainer knows the number right here
            import dataclasses
            densify = dataclasses.replace(densify, num_train_data=len(scene.train_indices))
    ev_pos = 0
    for it in range(1, iterations + 1):
        if not order:
            order = [scene.train_indices[j] for j in torch.randperm(len(scene.train_indices), generator=g).tolist()]
        i = order.pop()
        h = train_step(model, optimizers, scene.cameras[i], images[i], ssim_lambda)
        if densify is not None:
            from . import densify as D
            D.step_callback(model, optimizers, state, it, densify)
        if optimize_eval_cameras and scene.eval_indices and it % eval_camera_every == 0:
            e = scene.eval_indices[ev_pos % len(scene.eval_indices)]
            ev_pos += 1
            eval_camera_step(model, optimizers, scene.cameras[e], images[e], ssim_lambda)
        if log_every and it % log_every == 0:
            history.append({"step": it, **h})
    if model.means.is_cuda:
        torch.cuda.synchronize()
    wall = time.time() - t0
    res = evaluate(model, scene.cameras, images, scene.eval_indices)
    return {"results": res, "wall_clock_time_seconds": wall, "history": history}
