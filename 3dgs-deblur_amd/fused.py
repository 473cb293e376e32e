"""HIP loss + optimizer step of the training iteration (csrc/train.hip).

`image_loss` = splatfacto's ``(1 - l) * L1 + l * (1 - SSIM)`` (nerfstudio 1.1.0 ``get_loss_dict``, reached from
/root/reference/train.py:115-122) as ONE autograd node whose forward already produced d loss / d pred;
`HipAdam` = torch.optim.Adam(eps=1e-15) semantics, state laid out like torch's (``exp_avg`` / ``exp_avg_sq`` /
``step`` per parameter, so the densifier's row surgery works on either), and `adam_step_all` runs every HipAdam of a
step in ONE multi-tensor launch.  GPU tensors only: these raise on CPU tensors — the CPU stand-ins of the host-logic
tests use the torch implementations explicitly (train_step.image_loss / make_optimizers pick by device).
"""
from __future__ import annotations

import ctypes
from typing import Dict, Iterable, List

import torch
from torch import Tensor
from torch.autograd import Function

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA(HIP) tensor: the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise ValueError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def image_loss_with_grad(pred: Tensor, gt: Tensor, ssim_lambda: float = 0.2):
    """-> (loss [], d loss / d pred [H,W,3], parts [2] = {mean |gt - pred|, mean SSIM}): ONE pass of the two tile kernels
    produces the value and the gradient (what step.render_step's grad_image callable returns)."""
    pred, gt = _need_cuda(pred, "pred"), _need_cuda(gt, "gt")
    if pred.dim() != 3 or pred.shape[-1] != 3 or gt.shape != pred.shape:
        raise ValueError("pred and gt must both be [H,W,3]")
    H, W = int(pred.shape[0]), int(pred.shape[1])
    lam = float(ssim_lambda)
    if lam != 0.0 and min(H, W) < 11:
        # the 11x11 SSIM window does not fit (frames this small only occur while splatfacto's resolution schedule
        # has them downscaled): L1 only, as the window-less limit of the loss
        lam = 0.0
    L = _lib.load()
    dev = pred.device
    if gt.device != dev:
        raise ValueError("pred and gt must live on the same device")
    with torch.cuda.device(dev):          # launch on pred's device and ITS current stream, whatever is current
        ws_bytes = L.gs_image_loss_workspace_bytes(H, W)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        v_pred = torch.empty_like(pred)
        out = torch.empty(3, device=dev)
        _lib.check(L.gs_image_loss_fwd_bwd(H, W, ctypes.c_void_p(pred.data_ptr()), ctypes.c_void_p(gt.data_ptr()), lam,
                                           ctypes.c_void_p(v_pred.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                           ctypes.c_void_p(ws.data_ptr()), ws_bytes, _stream()), "image_loss_fwd_bwd")
    return out[0], v_pred, out[1:]


class _ImageLoss(Function):
    @staticmethod
    def forward(ctx, pred, gt, ssim_lambda):
        loss, v_pred, parts = image_loss_with_grad(pred, gt, ssim_lambda)
        ctx.save_for_backward(v_pred)
        parts = parts.clone()
        ctx.mark_non_differentiable(parts)
        return loss.clone(), parts

    @staticmethod
    def backward(ctx, v_loss, _v_parts):
        (v_pred,) = ctx.saved_tensors
        return v_pred * v_loss, None, None


def image_loss(pred: Tensor, gt: Tensor, ssim_lambda: float = 0.2, return_parts: bool = False):
    """(1 - ssim_lambda) * mean|gt - pred| + ssim_lambda * (1 - SSIM(pred, gt)); gradient to `pred` only.
    return_parts=True also returns a [2] tensor {mean |gt - pred|, mean SSIM} (no gradient)."""
    loss, parts = _ImageLoss.apply(pred, gt.detach(), float(ssim_lambda))
    return (loss, parts) if return_parts else loss


# --------------------------------------------------------------------------- #
class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam (no weight decay, no amsgrad) through gs_adam_step; state keys as torch's."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def _gather(self, items: List[tuple]) -> None:
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                _need_cuda(p.data, "parameter")
                if not p.data.is_contiguous():
                    raise ValueError("HipAdam needs contiguous parameters")
                g = _need_cuda(p.grad, "gradient")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p.data)
                    st["exp_avg_sq"] = torch.zeros_like(p.data)
                st["step"] = int(st["step"]) + 1
                items.append((p.data, g, st["exp_avg"], st["exp_avg_sq"], float(group["lr"]), tuple(group["betas"]),
                              float(group["eps"]), st["step"]))

    @torch.no_grad()
    def step(self, closure=None):
        items: List[tuple] = []
        self._gather(items)
        _launch(items)


def _launch(items: List[tuple]) -> None:
    """tensors that share (betas, eps, step) go out in one launch of up to 8"""
    if not items:
        return
    L = _lib.load()
    groups: Dict[tuple, List[tuple]] = {}
    for it in items:
        groups.setdefault((it[5], it[6], it[7], it[0].device), []).append(it)
    for (betas, eps, step, dev), its in groups.items():
        with torch.cuda.device(dev):
            for i in range(0, len(its), 8):
                chunk = its[i:i + 8]
                n = len(chunk)
                vp = ctypes.c_void_p
                P = (vp * n)(*[c[0].data_ptr() for c in chunk])
                G = (vp * n)(*[c[1].data_ptr() for c in chunk])
                M = (vp * n)(*[c[2].data_ptr() for c in chunk])
                V = (vp * n)(*[c[3].data_ptr() for c in chunk])
                NE = (ctypes.c_longlong * n)(*[c[0].numel() for c in chunk])
                LR = (ctypes.c_float * n)(*[c[4] for c in chunk])
                _lib.check(L.gs_adam_step(n, P, G, M, V, NE, LR, float(betas[0]), float(betas[1]), float(eps), int(step),
                                          _stream()), "adam_step")


@torch.no_grad()
def adam_step_all(optimizers: Iterable[torch.optim.Optimizer]) -> None:
    """One step of every optimizer: all HipAdam instances together in one multi-tensor launch (per distinct
    betas / eps / step count), anything else through its own .step()."""
    items: List[tuple] = []
    for o in optimizers:
        if isinstance(o, HipAdam):
            o._gather(items)
        else:
            o.step()
    _launch(items)
