"""Second, INDEPENDENT CPU restatement of the compositor — a literal per-pixel loop.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see gs_oracle.py's header: the fork's sources are not under /root/reference).

Why a second one: `gs_oracle.rasterize_sorted` is vectorised per tile (cumprod over the whole list, masks instead of
an early `break`) and its float32 path was written to the op order of csrc/gs_math.h — the checker is shaped like the
thing it checks ("Missing 6").  This file is written from SURVEY.md App. A "Blend" / "Backward"
alone, the way upstream gsplat 0.1.11's `_torch_impl.rasterize_forward` is written (BASELINE.json config 1 names that
role): one pixel at a time, one Gaussian at a time, plain Python floats (IEEE double), a real `break` at the
transmittance stop, and a hand-derived reverse loop for the gradients instead of autograd.  It shares no code with
gs_oracle.py; `tests/test_oracle.py` cross-checks the two on small images (values, final indices, gradients).

Only for small cases: O(pixels x list length) Python iterations.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# SURVEY.md App. A "Blend" — recollected constants, restated here on purpose (not imported)
_ALPHA_CAP = 0.999
_ALPHA_SKIP = 1.0 / 255.0
_T_STOP = 1e-4
_BLOCK = 16


def tile_lists_by_brute_force(tile_min: np.ndarray, tile_max: np.ndarray, depths: np.ndarray, visible: np.ndarray,
                              img_height: int, img_width: int) -> Tuple[np.ndarray, np.ndarray]:
    """Per tile: every visible Gaussian whose tile box [tile_min, tile_max) contains the tile, ordered by
    (float32 depth bits, Gaussian id) — App. A "Keys" says sort by (tile, depth bits); the id tie-break is this
    repo's documented convention.  No key packing, no global sort: a direct statement of what the binning must
    produce.  -> (ids_sorted int32 [I], tile_bins int32 [T,2])"""
    tiles_x = (img_width + _BLOCK - 1) // _BLOCK
    tiles_y = (img_height + _BLOCK - 1) // _BLOCK
    bits = np.asarray(depths, dtype=np.float32).view(np.int32)
    ids: List[int] = []
    bins = np.zeros((tiles_x * tiles_y, 2), dtype=np.int32)
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            members = [g for g in range(len(bits)) if visible[g]
                       and tile_min[g, 0] <= tx < tile_max[g, 0] and tile_min[g, 1] <= ty < tile_max[g, 1]]
            members.sort(key=lambda g: (int(bits[g]), g))
            if members:
                bins[ty * tiles_x + tx] = (len(ids), len(ids) + len(members))
                ids.extend(members)
    return np.asarray(ids, dtype=np.int32), bins


def composite_pixel_loop(xys: np.ndarray, conics: np.ndarray, colors: np.ndarray, opacities: np.ndarray,
                         ids_sorted: np.ndarray, tile_bins: np.ndarray, img_height: int, img_width: int,
                         background: Optional[Sequence[float]] = None, pix_vel: Optional[np.ndarray] = None,
                         row_time: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    """App. A "Blend", one pixel at a time.  pix_vel [N,2] + row_time [H] (both or neither): the exact per-row rolling
    shutter of the paper's model, App. A "plus row time" — pixel row i sees every splat at xy + row_time[i] * pix_vel.  -> {'img' [H,W,3], 'final_T' [H,W], 'final_idx' [H,W] int32,
    'stops': pixels that hit the transmittance stop}.
    final_idx is ONE PAST the last blended list position (the tile's list start when nothing was blended)."""
    H, W = img_height, img_width
    tiles_x = (W + _BLOCK - 1) // _BLOCK
    bg = (0.0, 0.0, 0.0) if background is None else tuple(float(b) for b in background)
    img = np.zeros((H, W, 3))
    final_T = np.ones((H, W))
    final_idx = np.zeros((H, W), dtype=np.int32)
    stops = 0
    for i in range(H):
        for j in range(W):
            start, end = (int(v) for v in tile_bins[(i // _BLOCK) * tiles_x + (j // _BLOCK)])
            px, py = j + 0.5, i + 0.5
            T = 1.0
            C = [0.0, 0.0, 0.0]
            last = start
            tau = 0.0 if row_time is None else float(row_time[i])
            for k in range(start, end):
                g = int(ids_sorted[k])
                dx = float(xys[g, 0]) - px
                dy = float(xys[g, 1]) - py
                if pix_vel is not None:
                    dx += tau * float(pix_vel[g, 0])
                    dy += tau * float(pix_vel[g, 1])
                sigma = (0.5 * (float(conics[g, 0]) * dx * dx + float(conics[g, 2]) * dy * dy)
                         + float(conics[g, 1]) * dx * dy)
                if sigma < 0.0:
                    continue
                alpha = min(_ALPHA_CAP, float(opacities[g]) * math.exp(-sigma))
                if alpha < _ALPHA_SKIP:
                    continue
                next_T = T * (1.0 - alpha)
                if next_T <= _T_STOP:
                    stops += 1
                    break                      # this Gaussian is NOT blended; the pixel is finished
                vis = alpha * T
                for c in range(3):
                    C[c] += vis * float(colors[g, c])
                T = next_T
                last = k + 1
            for c in range(3):
                img[i, j, c] = C[c] + T * bg[c]
            final_T[i, j] = T
            final_idx[i, j] = last
    return {"img": img, "final_T": final_T, "final_idx": final_idx, "stops": stops}


def composite_backward_pixel_loop(xys: np.ndarray, conics: np.ndarray, colors: np.ndarray, opacities: np.ndarray,
                                  ids_sorted: np.ndarray, tile_bins: np.ndarray, img_height: int, img_width: int,
                                  fwd: Dict[str, np.ndarray], v_img: np.ndarray,
                                  v_alpha: Optional[np.ndarray] = None,
                                  background: Optional[Sequence[float]] = None,
                                  clamp_blocks_gradient: bool = True, pix_vel: Optional[np.ndarray] = None,
                                  row_time: Optional[np.ndarray] = None) -> Dict[str, np.ndarray]:
    """App. A "Backward": every pixel walks its list from final_idx back to the tile start.
    With T = transmittance in front of the Gaussian, S = colour accumulated BEHIND it, ra = 1/(1-alpha):
        v_rgb += alpha*T * v_C
        v_alpha_i = sum_c (rgb_c*T - S_c*ra) * v_C,c  +  T_final*ra*(v_alpha_out - sum_c bg_c v_C,c)
        v_sigma = -o*e^{-sigma} * v_alpha_i,  v_o = e^{-sigma} * v_alpha_i   (zero through an ACTIVE 0.999 clamp when
        clamp_blocks_gradient, the true derivative — this repo's default; upstream lets it pass)
        v_conic = (0.5 v_sigma dx^2, v_sigma dx dy, 0.5 v_sigma dy^2),  v_xy = v_sigma * (cx dx + cy dy, cy dx + cz dy)
    With pix_vel / row_time (exact rolling shutter) the centre a pixel of row i sees is xy + row_time[i] * pix_vel, so
    v_pix_vel = sum over its pixels of row_time[i] * (that pixel's v_xy contribution): returned as 'v_pix_vel' [N,2].
    -> {'v_xy' [N,2], 'v_conic' [N,3], 'v_colors' [N,3], 'v_opacity' [N]} (image-plane alpha output = 1 - final_T)."""
    H, W = img_height, img_width
    tiles_x = (W + _BLOCK - 1) // _BLOCK
    bg = (0.0, 0.0, 0.0) if background is None else tuple(float(b) for b in background)
    N = xys.shape[0]
    v_xy = np.zeros((N, 2)); v_conic = np.zeros((N, 3)); v_colors = np.zeros((N, 3)); v_opacity = np.zeros(N)
    v_pv = np.zeros((N, 2))
    for i in range(H):
        for j in range(W):
            start, _ = (int(v) for v in tile_bins[(i // _BLOCK) * tiles_x + (j // _BLOCK)])
            px, py = j + 0.5, i + 0.5
            T_final = float(fwd["final_T"][i, j])
            vC = [float(v_img[i, j, c]) for c in range(3)]
            # the image's alpha is 1 - T_final: d loss / d T_final = -v_alpha_out ... folded below as in App. A
            va_out = 0.0 if v_alpha is None else float(v_alpha[i, j])
            T = T_final
            S = [0.0, 0.0, 0.0]
            tau = 0.0 if row_time is None else float(row_time[i])
            for k in range(int(fwd["final_idx"][i, j]) - 1, start - 1, -1):
                g = int(ids_sorted[k])
                dx = float(xys[g, 0]) - px
                dy = float(xys[g, 1]) - py
                if pix_vel is not None:
                    dx += tau * float(pix_vel[g, 0])
                    dy += tau * float(pix_vel[g, 1])
                cx_, cy_, cz_ = float(conics[g, 0]), float(conics[g, 1]), float(conics[g, 2])
                sigma = 0.5 * (cx_ * dx * dx + cz_ * dy * dy) + cy_ * dx * dy
                if sigma < 0.0:
                    continue
                e = math.exp(-sigma)
                raw = float(opacities[g]) * e
                alpha = min(_ALPHA_CAP, raw)
                if alpha < _ALPHA_SKIP:
                    continue
                ra = 1.0 / (1.0 - alpha)
                T *= ra                                    # transmittance in front of this Gaussian
                fac = alpha * T
                v_a = 0.0
                for c in range(3):
                    v_colors[g, c] += fac * vC[c]
                    v_a += (float(colors[g, c]) * T - S[c] * ra) * vC[c]
                    S[c] += float(colors[g, c]) * fac
                v_a += T_final * ra * (va_out - sum(bg[c] * vC[c] for c in range(3)))
                if clamp_blocks_gradient and raw > _ALPHA_CAP:
                    continue                               # alpha sits on the clamp: no gradient to sigma / opacity
                v_sigma = -raw * v_a
                v_opacity[g] += e * v_a
                v_conic[g, 0] += 0.5 * v_sigma * dx * dx
                v_conic[g, 1] += v_sigma * dx * dy
                v_conic[g, 2] += 0.5 * v_sigma * dy * dy
                gx, gy = v_sigma * (cx_ * dx + cy_ * dy), v_sigma * (cy_ * dx + cz_ * dy)
                v_xy[g, 0] += gx
                v_xy[g, 1] += gy
                v_pv[g, 0] += tau * gx
                v_pv[g, 1] += tau * gy
    return {"v_xy": v_xy, "v_conic": v_conic, "v_colors": v_colors, "v_opacity": v_opacity, "v_pix_vel": v_pv}
