"""Densify / cull — the step that follows the hot path in a splatfacto training loop.

The reference configures it only through ``--pipeline.model.cull-scale-thresh`` (/root/reference/train.py:18)
and otherwise inherits nerfstudio v1.1.0's splatfacto refinement (absent submodule,
/root/reference/.gitmodules:1-3; defaults below are recollected, SURVEY ⚠R):

  * after every training backward: accumulate, for the Gaussians visible in the view, the norm of the
    screen-space centre gradient, a visibility count and the largest screen radius;
  * every ``refine_every`` steps (after ``warmup_length``): Gaussians whose average gradient norm exceeds
    ``densify_grad_thresh`` are SPLIT (large ones: ``n_split_samples`` children sampled inside the parent,
    scales / 1.6, parent removed) or DUPLICATED (small ones); then low-opacity, over-sized
    (``cull_scale_thresh``) and over-large-on-screen Gaussians are culled; every
    ``reset_alpha_every * refine_every`` steps opacities are clamped down;
  * the Adam state follows the parameters (new rows start at zero, culled rows are dropped).

With motion-blur sub-poses the screen-space gradient of a Gaussian is the SUM over the sub-poses of its
per-sub-pose centre gradients (``render_subposes(..., xy_grad_out=)`` — written by the HIP projection
backward, no extra pass) and its radius the largest over the sub-poses.

Everything here is torch tensor surgery on whatever device the model lives on — no kernel of its own.
Data parallel (dp.py): the statistics are summed / max-ed over the ranks before the decision and the split
noise comes from a generator seeded by the step, so every rank takes the identical decision and the
replicated Gaussians stay bit-identical without broadcasting parameters.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
from torch import Tensor, nn

from .model import SplatfactoDeblurModel


@dataclass
class DensifyConfig:
    warmup_length: int = 500
    refine_every: int = 100
    cull_alpha_thresh: float = 0.1
    cull_scale_thresh: float = 0.5          # train.py:18 overrides this per dataset
    continue_cull_post_densification: bool = True
    reset_alpha_every: int = 30
    densify_grad_thresh: float = 0.0008
    densify_size_thresh: float = 0.01
    n_split_samples: int = 2
    cull_screen_size: float = 0.15
    split_screen_size: float = 0.05
    stop_screen_size_at: int = 4000
    stop_split_at: int = 15000
    num_train_data: int = 0                 # training images per pass (upstream's post-reset guard)
    seed: int = 0


class DensifyState:
    """Per-Gaussian accumulators between two refinements."""

    def __init__(self, num_points: int, device):
        self.xys_grad_norm = torch.zeros(num_points, device=device)
        self.vis_counts = torch.zeros(num_points, device=device)
        self.max_2Dsize = torch.zeros(num_points, device=device)
        self.size = (1, 1)

    @torch.no_grad()
    def after_backward(self, radii: Tensor, xy_grad: Tensor, width: int, height: int) -> None:
        """radii int32 [P,N] (0 = culled in that sub-pose), xy_grad float32 [N,2] in pixels."""
        radii = radii.reshape(-1, radii.shape[-1])
        visible = (radii > 0).any(dim=0)
        self.xys_grad_norm += torch.where(visible, xy_grad.norm(dim=-1), torch.zeros_like(self.xys_grad_norm))
        self.vis_counts += visible.to(self.vis_counts.dtype)
        rel = radii.max(dim=0).values.to(torch.float32) / float(max(width, height))
        self.max_2Dsize = torch.where(visible, torch.maximum(self.max_2Dsize, rel), self.max_2Dsize)
        self.size = (width, height)

    def allreduce(self, group=None) -> None:
        """Data parallel: ranks saw different views; combine before deciding (SURVEY §8e last sentence)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        dist.all_reduce(self.xys_grad_norm, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(self.vis_counts, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(self.max_2Dsize, op=dist.ReduceOp.MAX, group=group)


def _quat_to_rotmat(q: Tensor) -> Tensor:
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


def _swap_parameter(model: SplatfactoDeblurModel, optimizers: Dict[str, torch.optim.Optimizer], name: str,
                    new_value: Tensor, keep: Optional[Tensor], n_new: int) -> None:
    """Replace parameter `name` by new_value; carry the Adam moments: old rows filtered by `keep`
    (bool over the OLD rows, None = keep all), then n_new zero rows appended."""
    old = getattr(model, name)
    new_p = nn.Parameter(new_value.contiguous())
    setattr(model, name, new_p)
    opt = optimizers.get(name)
    if opt is None:
        return
    st = opt.state.pop(old, None)
    opt.param_groups[0]["params"] = [new_p]
    if st:
        for key in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            if key in st:
                v = st[key] if keep is None else st[key][keep]
                if n_new:
                    v = torch.cat([v, torch.zeros((n_new,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)])
                st[key] = v.contiguous()
        opt.state[new_p] = st


@torch.no_grad()
def reset_opacities(model: SplatfactoDeblurModel, optimizers: Dict[str, torch.optim.Optimizer],
                    cfg: DensifyConfig) -> None:
    """Clamp opacities down to 2*cull_alpha_thresh and forget their Adam moments."""
    reset_value = cfg.cull_alpha_thresh * 2.0
    logit = math.log(reset_value / (1.0 - reset_value))
    model.opacities.data = torch.clamp(model.opacities.data, max=logit)
    opt = optimizers.get("opacities")
    if opt is not None and model.opacities in opt.state:
        st = opt.state[model.opacities]
        for key in ("exp_avg", "exp_avg_sq"):
            if key in st:
                st[key].zero_()


@torch.no_grad()
def refine(model: SplatfactoDeblurModel, optimizers: Dict[str, torch.optim.Optimizer], state: DensifyState,
           step: int, cfg: DensifyConfig) -> Dict[str, int]:
    """One refinement (split / duplicate / cull) with nerfstudio v1.1.0 splatfacto's `refinement_after` rules
    (recollected, SURVEY ⚠R):

      * densify only while ``step < stop_split_at`` AND every training image has been seen since the last
        opacity reset: ``step % reset_interval > num_train_data + refine_every``;
      * ``splits = (big | on-screen-large [before stop_screen_size_at]) & high_grads``, ``dups = ~big & high_grads``
        (a small Gaussian that is large on screen is split AND duplicated, as upstream);
      * the cull runs over [old rows | split children | duplicates] — a child or duplicate that is itself
        transparent or over-sized goes at once — and only when densification ran, or after ``stop_split_at``
        with ``continue_cull_post_densification``; between an opacity reset and the next full pass over the
        training images nothing is pruned.

    Returns the counts and leaves `state` reset for the new N."""
    dev = model.means.device
    N = model.num_points
    names = list(model.gauss_params().keys())
    W, H = state.size
    reset_interval = cfg.refine_every * cfg.reset_alpha_every
    do_densify = step < cfg.stop_split_at and step % reset_interval > cfg.num_train_data + cfg.refine_every
    do_cull = do_densify or (step >= cfg.stop_split_at and cfg.continue_cull_post_densification)
    n_split = n_dup = n_new = 0
    split_mask = torch.zeros(N, dtype=torch.bool, device=dev)
    extra: Dict[str, Tensor] = {}
    if do_densify:
        avg_grad = state.xys_grad_norm / torch.clamp(state.vis_counts, min=1.0) * 0.5 * float(max(W, H))
        high = avg_grad > cfg.densify_grad_thresh
        big = torch.exp(model.scales).max(dim=-1).values > cfg.densify_size_thresh
        split_mask = big.clone()
        if step < cfg.stop_screen_size_at:
            split_mask |= state.max_2Dsize > cfg.split_screen_size
        split_mask &= high
        dup_mask = high & ~big
        n_split, n_dup = int(split_mask.sum()), int(dup_mask.sum())
        k = cfg.n_split_samples
        # children of the split Gaussians: positions sampled inside the parent, scales / 1.6
        gen = torch.Generator(device=dev)
        gen.manual_seed(cfg.seed * 1000003 + step)
        noise = torch.randn(k * n_split, 3, device=dev, generator=gen)
        sc = torch.exp(model.scales[split_mask]).repeat(k, 1)
        Rm = _quat_to_rotmat(model.quats[split_mask]).repeat(k, 1, 1)
        child_means = torch.bmm(Rm, (sc * noise)[..., None]).squeeze(-1) + model.means[split_mask].repeat(k, 1)
        for name, p in model.gauss_params().items():
            src_split = p[split_mask]
            rep = src_split.repeat((k,) + (1,) * (p.dim() - 1))
            if name == "means":
                rep = child_means
            elif name == "scales":
                rep = torch.log(torch.exp(src_split) / 1.6).repeat(k, 1)
            extra[name] = torch.cat([rep, p[dup_mask]])
        n_new = k * n_split + n_dup

    # cull over [old rows | new rows]; split parents always go
    n_low = n_big = 0
    cull_all = torch.zeros(N + n_new, dtype=torch.bool, device=dev)
    if do_cull:
        opac_all = model.opacities.data
        scales_all = model.scales.data
        if n_new:
            opac_all = torch.cat([opac_all, extra["opacities"]])
            scales_all = torch.cat([scales_all, extra["scales"]])
        cull_all = torch.sigmoid(opac_all).reshape(-1) < cfg.cull_alpha_thresh
        n_low = int(cull_all.sum())
        cull_all[:N] |= split_mask
        if step > reset_interval:
            too_big = torch.exp(scales_all).max(dim=-1).values > cfg.cull_scale_thresh
            if step < cfg.stop_screen_size_at:
                size2d = torch.cat([state.max_2Dsize, state.max_2Dsize.new_zeros(n_new)])   # new rows: not seen yet
                too_big |= size2d > cfg.cull_screen_size
            n_big = int((too_big & ~cull_all).sum())
            cull_all |= too_big
    keep_all = ~cull_all
    keep_old, keep_new = keep_all[:N], keep_all[N:]
    n_kept_new = int(keep_new.sum()) if n_new else 0
    if do_cull or n_new:
        for name in names:
            p = getattr(model, name)
            new_value = p.data[keep_old]
            if n_kept_new:
                new_value = torch.cat([new_value, extra[name][keep_new]])
            _swap_parameter(model, optimizers, name, new_value, keep_old, n_kept_new)
    n_after = model.num_points
    fresh = DensifyState(n_after, dev)
    fresh.size = state.size
    state.__dict__.update(fresh.__dict__)
    return {"split": n_split, "duplicated": n_dup, "culled_low_opacity": n_low, "culled_too_big": n_big,
            "before": N, "after": n_after}


def step_callback(model: SplatfactoDeblurModel, optimizers: Dict[str, torch.optim.Optimizer], state: DensifyState,
                  step: int, cfg: DensifyConfig, group=None) -> Optional[Dict[str, int]]:
    """Call once per training step AFTER backward + optimizer step (what splatfacto registers as its
    AFTER_TRAIN_ITERATION callbacks): accumulates the statistics of the step's render and refines on schedule."""
    # upstream's after_train returns early once densification has stopped: no statistics are gathered any more
    if step < cfg.stop_split_at and model.xy_grad is not None and model.radii is not None:
        state.after_backward(model.radii, model.xy_grad, *model.last_size)
    result = None
    if step > cfg.warmup_length and step % cfg.refine_every == 0:
        state.allreduce(group)
        result = refine(model, optimizers, state, step, cfg)
        reset_interval = cfg.refine_every * cfg.reset_alpha_every
        if step < cfg.stop_split_at and step % reset_interval == cfg.refine_every:
            reset_opacities(model, optimizers, cfg)
        # N changed and / or every opacity dropped to ~0.02: the share of Gaussians with a gradient is about to jump
        # (early termination stops), so the row-sparse gradient exchange must not size its payload from old counts
        from . import dp
        dp.notify_regime_change()
    return result
