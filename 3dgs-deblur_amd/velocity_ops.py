"""Fork-style velocity keywords of the two gsplat-compatible ops (round 5).

`ops.project_gaussians(..., lin_vel=, ang_vel=, exposure_time=, rolling_shutter_time=, blur_samples=)` and
`ops.rasterize_gaussians(..., pix_vels=, ...)` dispatch here when the keywords are given: the paper's pixel-velocity
model (/root/reference/README.md:196-200) through the SAME kernels as `render_subposes(shared_list=True)` —
`gs_project_pixvel_fwd/bwd` with ONE sub-pose at the mid-exposure pose, unit opacity (the record's opacity slot then IS the
compensation factor) and deferred colour; records with swept tile boxes assembled in torch; one binning;
`gs_rasterize_fwd/bwd_rs_slice` (csrc/raster_rs.hip, the sample loop and the per-row readout time inside the
compositor) + `gs_reduce_grad_tuples`.  No kernel of its own, no CPU fallback.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops
from .ops import (GRAD, REC, TILE, _L, _background, _band_edges, _check, _f32, _ptr, _stream, _tiles, _viewmat16,
                  bin_and_sort_records, subpose_schedule)


def sample_span(blur_samples: int, exposure_time: float, rolling_shutter_time: float):
    """sample times of the fork-style compat calls (subpose_schedule's blur schedule, symmetric about the mid-exposure
    pose, so the shared list's centre time is 0) and the time span the tile boxes are swept over"""
    S = max(1, int(blur_samples))
    times, _, _ = subpose_schedule(S, float(exposure_time), 1, 0.0)
    return S, times, (max(times) - min(times)) + abs(float(rolling_shutter_time))


class ProjectGaussiansPixvel(Function):
    """project_gaussians with the fork's velocity keywords: the pixel-velocity model's ONE projection (shared-list form,
    gs_project_pixvel_fwd with P = 1 at the mid-exposure pose): centres / conics / compensation of every geometrically
    valid Gaussian, radii / num_tiles_hit of the tile boxes SWEPT over the sampled span, and the pixel velocities."""

    @staticmethod
    def forward(ctx, means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, H, W, clip, lin_vel, ang_vel, span):
        ctx.set_materialize_grads(False)        # an output nobody differentiates arrives as None in backward
        means3d, scales, quats = _f32(means3d, "means3d"), _f32(scales, "scales"), _f32(quats, "quats")
        V = _viewmat16(viewmat).reshape(4, 4)
        N, dev, L = means3d.shape[0], means3d.device, _L()
        twist = torch.cat([_f32(lin_vel, "lin_vel").reshape(3), _f32(ang_vel, "ang_vel").reshape(3)]).contiguous()
        ones = torch.ones(N, device=dev)
        sh0 = torch.zeros(N, 1, 3, device=dev)
        t0 = torch.zeros(1, device=dev)
        records = torch.empty(N, REC, device=dev)
        dkeys = torch.empty(N, dtype=torch.int32, device=dev)
        ntiles = torch.empty(N, dtype=torch.int32, device=dev)
        radii = torch.empty(N, dtype=torch.int32, device=dev)
        pix_vel = torch.empty(N, 2, device=dev)
        # opacity 1 + antialiasing: the record's opacity slot is the compensation factor; colour deferred (never made)
        _check(L.gs_project_pixvel_fwd(N, 1, _ptr(means3d), _ptr(scales), float(glob_scale), _ptr(quats), _ptr(ones),
                                       _ptr(sh0), 1, 0, _ptr(V), _ptr(twist), _ptr(t0), float(fx), float(fy), float(cx),
                                       float(cy), int(H), int(W), float(clip), 1, 1, _ptr(records), _ptr(dkeys),
                                       _ptr(ntiles), _ptr(radii), float(span), _ptr(pix_vel), None, 0, _stream()),
               "project_pixvel_fwd")
        cov3d = torch.empty(N, 6, device=dev)
        scratch = [torch.empty(N, 2, device=dev), torch.empty(N, device=dev), torch.empty(N, dtype=torch.int32, device=dev),
                   torch.empty(N, 3, device=dev), torch.empty(N, device=dev), torch.empty(N, dtype=torch.int32, device=dev)]
        _check(L.gs_project_fwd(N, _ptr(means3d), _ptr(scales), float(glob_scale), _ptr(quats), _ptr(V), float(fx),
                                float(fy), float(cx), float(cy), int(H), int(W), float(clip), *(_ptr(t) for t in scratch),
                                _ptr(cov3d), None, _stream()), "project_fwd")
        depths = scratch[1]                                   # camera-space z of every Gaussian (culled ones too)
        live = (radii > 0)[:, None]
        xys = torch.where(live, records[:, 0:2], torch.zeros_like(pix_vel))
        conics = torch.where(live, records[:, 2:5], torch.zeros(N, 3, device=dev))
        comp = torch.where(live[:, 0], records[:, 5], torch.zeros(N, device=dev))
        pix_vel = torch.where(live, pix_vel, torch.zeros_like(pix_vel))
        ctx.save_for_backward(means3d, scales, quats, V, twist, records, ones, sh0, t0)
        ctx.args = (float(glob_scale), float(fx), float(fy), float(cx), float(cy), int(H), int(W), float(clip))
        ctx.mark_non_differentiable(radii, ntiles)
        return xys.contiguous(), depths, radii, conics.contiguous(), comp.contiguous(), ntiles, cov3d, pix_vel

    @staticmethod
    def backward(ctx, v_xys, v_depths, v_radii, v_conics, v_comp, v_ntiles, v_cov3d, v_pv):
        means3d, scales, quats, V, twist, records, ones, sh0, t0 = ctx.saved_tensors
        glob, fx, fy, cx, cy, H, W, clip = ctx.args
        N, dev, L = means3d.shape[0], means3d.device, _L()
        if v_depths is not None or v_cov3d is not None:
            raise NotImplementedError("project_gaussians with velocity keywords: no gradient through depths / cov3d "
                                      "(rasterize_gaussians produces none)")
        v_rec = torch.zeros(N, GRAD, device=dev)
        for g, sl in ((v_xys, slice(0, 2)), (v_conics, slice(2, 5)), (v_pv, slice(9, 11))):
            if g is not None:
                v_rec[:, sl] = g
        if v_comp is not None:
            v_rec[:, 5] = v_comp
        v_means, v_scales = torch.empty(N, 3, device=dev), torch.empty(N, 3, device=dev)
        v_quats, v_op, v_sh = torch.empty(N, 4, device=dev), torch.empty(N, device=dev), torch.empty(N, 1, 3, device=dev)
        acc = torch.zeros(16 + 12, device=dev)
        v_V, v_tw = acc[:16].view(4, 4), acc[16:]
        psc = ops._pose_scratch(N, 1, False, dev)
        _check(L.gs_project_pixvel_bwd(N, 1, _ptr(means3d), _ptr(scales), glob, _ptr(quats), _ptr(ones), _ptr(sh0), 1, 0,
                                       _ptr(V), _ptr(twist), _ptr(t0), fx, fy, cx, cy, H, W, clip, 1, _ptr(records),
                                       _ptr(v_rec), _ptr(v_means), _ptr(v_scales), _ptr(v_quats), _ptr(v_op), _ptr(v_sh),
                                       _ptr(v_V), _ptr(v_tw), None, None, (ops.UPSTREAM_GRADS & 3) | 16 | (0 if ops.NEEDLE_HP else 8),
                                       None, 0, None, _ptr(psc), psc.numel(), _stream()), "project_pixvel_bwd")
        return (v_means, v_scales, None, v_quats, v_V, None, None, None, None, None, None, None, v_tw[0:3], v_tw[3:6], None)


class RasterizeGaussiansPixvel(Function):
    """rasterize_gaussians with the fork's velocity keywords: ONE swept-box binning, the S blur samples (and the per-row
    rolling-shutter time) inside the compositor — csrc/raster_rs.hip in its shared-list form, one slice."""

    @staticmethod
    def forward(ctx, xys, depths, radii, conics, colors, opacity, pix_vels, H, W, background, times, rs_time, span):
        ctx.set_materialize_grads(False)
        xys, depths, conics = _f32(xys, "xys"), _f32(depths, "depths"), _f32(conics, "conics")
        colors, opacity, pv = _f32(colors, "colors"), _f32(opacity, "opacity").reshape(-1), _f32(pix_vels, "pix_vels")
        radii = radii.to(torch.int32).contiguous()
        N, dev, L = xys.shape[0], xys.device, _L()
        S = len(times)
        tx, ty = _tiles(H, W)
        # records [N,16] with the SWEPT tile box (gs_math.h::tile_bounds_swept, float32 op for op: the rolling-shutter
        # compositors read floats 0..9, the packed box and the constants 12..15)
        half = torch.tensor(0.5 * span, dtype=torch.float32, device=dev)
        inv_tile = torch.tensor(1.0 / TILE, dtype=torch.float32, device=dev)
        xa, xb = xys[:, 0] - half * pv[:, 0], xys[:, 0] + half * pv[:, 0]
        ya, yb = xys[:, 1] - half * pv[:, 1], xys[:, 1] + half * pv[:, 1]
        tr = radii.float() * inv_tile
        x0 = torch.trunc(torch.minimum(xa, xb) * inv_tile - tr).clamp(0, tx).int()
        x1 = torch.trunc((torch.maximum(xa, xb) * inv_tile + tr) + 1.0).clamp(0, tx).int()
        y0 = torch.trunc(torch.minimum(ya, yb) * inv_tile - tr).clamp(0, ty).int()
        y1 = torch.trunc((torch.maximum(ya, yb) * inv_tile + tr) + 1.0).clamp(0, ty).int()
        area = (x1 - x0) * (y1 - y0)
        ok = (radii > 0) & (area > 0)
        records = torch.zeros(N, REC, device=dev)
        records[:, 0:2], records[:, 2:5], records[:, 5], records[:, 6:9], records[:, 9] = xys, conics, opacity, colors, depths
        records[:, 10:12] = torch.stack([x0 | (y0 << 16), x1 | (y1 << 16)], dim=1).view(torch.float32)
        # floats 12..15: the compositors' per-entry constants (csrc/gs_math.h::rec_aux): nmid = log2(255 op) / 2,
        # kmul = op 2^-nmid (alpha = kmul 2^u, u = nmid - log2(e) sigma; |u| <= nmid <=> sigma >= 0 and alpha >= 1/255),
        # and the conic's diagonal pre-scaled by -log2(e) / 2
        blend = opacity >= (1.0 / 255.0)
        nmid = torch.where(blend, 0.5 * torch.log2(255.0 * opacity.clamp_min(1e-30)), torch.full_like(opacity, -1.0))
        records[:, 12] = nmid
        records[:, 13] = torch.where(blend, opacity * torch.exp2(-nmid), torch.zeros_like(opacity))
        records[:, 14], records[:, 15] = conics[:, 0] * (-0.5 * 1.4426950408889634), conics[:, 2] * (-0.5 * 1.4426950408889634)
        records = torch.where(ok[:, None], records, torch.zeros_like(records)).contiguous()
        ntiles = torch.where(ok, area, torch.zeros_like(area)).contiguous()
        dkeys = torch.where(ok, depths.view(torch.int32), torch.full_like(area, -1)).contiguous()
        sorted_ids, bins, n_isect, _, em = bin_and_sort_records(records, dkeys, ntiles, 1, N, H, W, with_emission=True)
        bg = _background(background, dev)
        edges = _band_edges(H, 1, dev)
        times_t = torch.tensor(times, dtype=torch.float32, device=dev)
        out_img = torch.empty(S, H, W, 3, device=dev)
        out_T = torch.empty(S, H, W, device=dev)
        fidx = torch.empty(S, H, W, dtype=torch.int32, device=dev)
        if n_isect > 0:
            _check(L.gs_rasterize_fwd_rs_slice(_ptr(records), _ptr(bins), _ptr(edges), _ptr(bg), S, H, W, _ptr(out_img),
                                               _ptr(out_T), None, _ptr(fidx), None, 1, 1, _ptr(sorted_ids), N, None, None,
                                               _ptr(pv), N, float(rs_time), _ptr(times_t), _stream()), "rasterize_fwd_rs")
        else:
            out_img[:] = bg
            out_T.fill_(1.0)
        ctx.save_for_backward(records, sorted_ids, bins, edges, bg, out_T, fidx, pv, times_t)
        ctx.em, ctx.dims, ctx.n_isect = em, (N, S, H, W, float(rs_time)), n_isect
        ctx.bg_grad = background is not None and ctx.needs_input_grad[9]
        return out_img, 1.0 - out_T

    @staticmethod
    def backward(ctx, v_img, v_alpha):
        records, sorted_ids, bins, edges, bg, out_T, fidx, pv, times_t = ctx.saved_tensors
        N, S, H, W, rs_time = ctx.dims
        dev, L, em, I = records.device, _L(), ctx.em, ctx.n_isect
        if v_img is None and v_alpha is None:
            return (None,) * 13
        v_img = torch.zeros(S, H, W, 3, device=dev) if v_img is None else v_img.contiguous().float()
        # (out_alpha = 1 - out_T: the compositor's v_alpha input is d loss / d alpha)
        v_al = None if v_alpha is None else v_alpha.contiguous().float()
        v_records = torch.zeros(N, GRAD, device=dev)
        if I > 0:
            tuples = torch.empty(I * S, GRAD, device=dev)
            flags = torch.zeros(I * S, dtype=torch.uint8, device=dev)
            _check(L.gs_rasterize_bwd_rs_slice(_ptr(records), _ptr(em["eids"]), _ptr(bins), _ptr(edges), _ptr(bg), S, H, W,
                                               _ptr(out_T), _ptr(fidx), _ptr(v_img), _ptr(v_al), None, None, _ptr(tuples),
                                               _ptr(flags), _ptr(sorted_ids), N, ops._bwd_variant(), None, 1.0, 0.0, _ptr(pv),
                                               N, rs_time, _ptr(times_t), _stream()), "rasterize_bwd_rs")
            _check(L.gs_reduce_grad_tuples(N, _ptr(em["sorted_gi"]), _ptr(em["counts"]), _ptr(em["cum"]), _ptr(tuples),
                                           _ptr(flags), _ptr(v_records), None, I, _ptr(records), S, _stream()),
                   "reduce_grad_tuples")
        v_xys, v_conics = torch.empty(N, 2, device=dev), torch.empty(N, 3, device=dev)
        v_colors, v_opacity = torch.empty(N, 3, device=dev), torch.empty(N, 1, device=dev)
        _check(L.gs_unpack_record_grads(N, _ptr(v_records), _ptr(v_xys), _ptr(v_conics), _ptr(v_colors), _ptr(v_opacity),
                                        _stream()), "unpack grads")
        v_bg = (out_T[..., None] * v_img).sum(dim=(0, 1, 2)) if ctx.bg_grad else None
        return (v_xys, None, None, v_conics, v_colors, v_opacity, v_records[:, 9:11].contiguous(), None, None, v_bg, None,
                None, None)
