"""Oracle fixtures of the BENCHMARK's own configurations at FULL size.

The `-m gpu` oracle comparisons run at sizes the float64 oracle finishes in seconds (<= 24 000 Gaussians); the
configuration bench.py times (BASELINE.json config 2's metric scene: 1M Gaussians, 1920x1080, 5 motion-blur
sub-poses, SURVEY §8d seed 1234) and config 3 (1M, 1080p, 10 rolling-shutter row bands) were only checked against this
repo's own plain path.  This script evaluates the ORACLE (oracle/gs_oracle.py) on those two scenes for a sample of
tiles:

  * float32 oracle projection of all 1M Gaussians under every sub-pose (integer outputs bit-exact by construction:
    tile boxes, depth-key bits) -> for each sampled (tile, sub-pose) the complete list of Gaussians whose box covers the
    tile, in (depth bits, id) order — what map_gaussian_to_intersects + sort_intersects + get_tile_bin_edges give for
    that tile, without materialising the frame's 238 M pairs;
  * float64 projection / SH colour / compositing of those Gaussians over the tile's 256 pixels (the arithmetic of
    rasterize_sorted, chunked with an early exit once every pixel has stopped): colour, final T, the list position at
    which each pixel stops (= list length when it never does), the oracle's own final index (one past the last blended
    entry) and its fragile mask.

Stored per (tile, sub-pose): the list's length, a checksum of the whole list, its first `n_keep` ids (everything any
pixel of the tile can reach, + 8), and the pixel arrays.  tests/test_gpu_parity.py::test_full_size_* renders the same
scenes through the HIP path at full size and compares.  SELF-GOLDEN (oracle-made), like tests/golden/*.npz: parity
with the absent fork stays unpinned.

    python tests/golden/make_full_size_fixtures.py          (~3 min on 8 cores, build container or any CPU box)
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import gs_oracle as O  # noqa: E402

HERE = Path(__file__).resolve().parent
TILE = O.TILE
N, W, H = 1_000_000, 1920, 1080
N_TILES = 64


def list_checksum(ids: np.ndarray) -> int:
    """order-sensitive checksum of an id list (mod 2^61 - 1)"""
    m = (1 << 61) - 1
    pos = np.arange(1, ids.size + 1, dtype=np.uint64)
    return int((np.sum((ids.astype(np.uint64) + 1) * ((pos * 2654435761) % 1000003) % m) % m))


def composite_tile(xy, conic, rgb, op, px, py, chunk=512):
    """float64 front-to-back compositing of one tile's sorted list (arrays in list order) over pixel centres px, py.
    Same arithmetic as gs_oracle.rasterize_sorted, in chunks, leaving once every pixel has stopped.
    -> colour [P,3] (no background), final T [P], stop position [P], oracle final index [P] (one past the last blended
    entry, relative to the list start), fragile [P], entries any pixel reached"""
    P = px.shape[0]
    T = torch.ones(P, dtype=torch.float64)
    C = torch.zeros(P, 3, dtype=torch.float64)
    stopped = torch.zeros(P, dtype=torch.bool)
    stop_pos = torch.full((P,), xy.shape[0], dtype=torch.int64)
    last = torch.zeros(P, dtype=torch.int64)
    frag = torch.zeros(P, dtype=torch.bool)
    amp2 = torch.zeros(P, dtype=torch.float64)        # running sum of (alpha / (1 - alpha))^2 of the blended entries
    n = xy.shape[0]
    reached = 0
    for c0 in range(0, n, chunk):
        if bool(stopped.all()):
            break
        c1 = min(n, c0 + chunk)
        dx = xy[c0:c1, 0:1] - px[None, :]
        dy = xy[c0:c1, 1:2] - py[None, :]
        sigma = 0.5 * (conic[c0:c1, 0:1] * dx * dx + conic[c0:c1, 2:3] * dy * dy) + conic[c0:c1, 1:2] * dx * dy
        ov = op[c0:c1, None] * torch.exp(-sigma)
        alpha = torch.clamp(ov, max=O.ALPHA_MAX)
        valid = (sigma >= 0) & (alpha >= O.ALPHA_MIN)
        a = torch.where(valid, alpha, torch.zeros_like(alpha))
        Tincl = T[None, :] * torch.cumprod(1.0 - a, dim=0)
        Texcl = torch.cat([T[None, :], Tincl[:-1]], dim=0)
        live = (Tincl > O.T_MIN) & ~stopped[None, :]
        # a pixel stops at its first valid entry that would push T to <= T_MIN; entries behind it are not blended
        first_dead = torch.where(~(Tincl > O.T_MIN), torch.arange(c0, c1)[:, None], torch.full_like(Tincl, n, dtype=torch.int64).long())
        fd = first_dead.min(dim=0).values
        newly = (~stopped) & (fd < n)
        k = torch.arange(c0, c1)[:, None]
        blended = valid & live & (k < torch.where(stopped, torch.zeros_like(fd), fd)[None, :])
        w = torch.where(blended, a * Texcl, torch.zeros_like(a))
        C = C + (w[:, :, None] * rgb[c0:c1, None, :]).sum(dim=0)
        last = torch.maximum(last, torch.where(blended, k + 1, torch.zeros_like(k)).max(dim=0).values)
        reach = (Texcl > O.T_MIN) & ~stopped[None, :] & (k <= torch.where(fd < n, fd, torch.full_like(fd, n))[None, :])
        f1 = (reach & ((alpha / O.ALPHA_MIN - 1.0).abs() < O.FRAGILE_ALPHA_BAND)).any(dim=0)
        # the oracle's rounding model of T (gs_oracle.rasterize_sorted): floor + gain * root-sum-square of the entries'
        # alpha / (1 - alpha) so far (an alpha ON the clamp is exact and adds nothing)
        on_clamp = ov > O.ALPHA_MAX * (1.0 + 2.0 * O.FRAGILE_ALPHA_BAND)
        amp = torch.where(on_clamp, torch.zeros_like(a), a / (1.0 - a))
        rss2 = amp2[None, :] + torch.cumsum(amp * amp, dim=0)
        band_T = O.FRAGILE_T_FLOOR + O.FRAGILE_T_GAIN * torch.sqrt(rss2)
        f2 = (valid & reach & ((Tincl / O.T_MIN - 1.0).abs() < band_T)).any(dim=0)
        amp2 = torch.where(stopped, amp2, rss2[-1])
        f3 = (reach & (sigma.abs() < 1e-7) & (sigma != 0)).any(dim=0)
        frag |= f1 | f2 | f3
        # T after the chunk: the product over blended entries only
        Tnew = T * torch.where(blended, 1.0 - a, torch.ones_like(a)).prod(dim=0)
        T = torch.where(stopped, T, Tnew)
        stop_pos = torch.where(newly, fd, stop_pos)
        stopped = stopped | newly
        reached = c1
    return C, T, stop_pos, last, frag, reached


def tile_weights(ti: int, p: int, hh: int, ww: int) -> torch.Tensor:
    """d loss / d rgb of one sampled (tile, sub-pose): seeded, float64 [hh,ww,3] in [0,1)"""
    g = torch.Generator().manual_seed(100003 * int(p) + int(ti) + 17)
    return torch.rand(hh, ww, 3, generator=g, dtype=torch.float64)


def tensor_hash(t: torch.Tensor) -> str:
    import hashlib
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def activate(sc):
    """exp / sigmoid through float64, rounded to float32: the same bits on every machine (torch's vectorised float32
    CPU exp is not correctly rounded and depends on the CPU's vector width)"""
    return sc["log_scales"].double().exp().float(), torch.sigmoid(sc["opacity_logits"].double()).float()


def make(tag, S, R, out_name):
    sc = O.synthetic_scene(N, W, H, seed=1234)
    et, rt = sc["exposure_time"], sc["rolling_shutter_time"]
    times, samp, band = O.subpose_times(S, et, R, rt)
    vms32 = O.subpose_viewmats(sc["viewmat"], sc["lin_vel"], sc["ang_vel"], times)
    vms64 = O.subpose_viewmats(sc["viewmat"].double(), sc["lin_vel"].double(), sc["ang_vel"].double(), times)
    rows = O.band_tile_rows(H, R)
    tiles_x, tiles_y = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    rng = np.random.default_rng(2024)
    tiles = np.sort(rng.choice(tiles_x * tiles_y, size=N_TILES, replace=False))
    scales32, op32 = activate(sc)
    means64, scales64, quats64 = sc["means"].double(), scales32.double(), sc["quats"].double()
    op64, sh64 = op32.double(), sc["sh"].double()
    out = {"tiles": tiles.astype(np.int32), "S": S, "R": R, "N": N, "W": W, "H": H, "seed": 1234,
           # the sub-pose viewmats the lists were made with (the test feeds exactly these to the HIP path) and a hash of
           # every scene tensor (the test regenerates the scene from the seed: equal hashes = the very same inputs)
           "viewmats": vms32.numpy().astype(np.float32),
           "scene_hashes": np.array([f"{k}:{tensor_hash(v)}" for k, v in
                                     (("means", sc["means"]), ("scales", scales32), ("quats", sc["quats"]),
                                      ("opacities", op32), ("sh", sc["sh"]))])}
    total_pairs = 0
    for p in range(S * R):
        pr = O.project_gaussians(sc["means"], scales32, 1.0, sc["quats"], vms32[p], sc["fx"], sc["fy"], sc["cx"], sc["cy"],
                                 H, W)
        total_pairs += int(pr.num_tiles_hit.long().sum())
        tmin, tmax = pr.tile_min.numpy(), pr.tile_max.numpy()
        dbits = pr.depths.to(torch.float32).numpy().view(np.int32).astype(np.int64)
        hit = pr.num_tiles_hit.numpy() > 0
        ty0, ty1 = rows[band[p]]
        for ti, t in enumerate(tiles):
            ty, tx = divmod(int(t), tiles_x)
            if not (ty0 <= ty < ty1):
                continue                                   # this sub-pose does not render the tile's row band
            sel = np.nonzero(hit & (tmin[:, 0] <= tx) & (tx < tmax[:, 0]) & (tmin[:, 1] <= ty) & (ty < tmax[:, 1]))[0]
            order = np.lexsort((sel, dbits[sel]))          # (depth bits, id): the (tile, depth, id) order within a tile
            ids = sel[order].astype(np.int64)
            # float64 values of the listed Gaussians only
            idt = torch.from_numpy(ids)
            pr64 = O.project_gaussians(means64[idt], scales64[idt], 1.0, quats64[idt], vms64[p], sc["fx"], sc["fy"], sc["cx"],
                                       sc["cy"], H, W, keep_offscreen=True)
            V = vms64[p]
            cam = -(V[:3, :3].T @ V[:3, 3])
            rgb = torch.clamp(O.spherical_harmonics(3, means64[idt] - cam[None, :], sh64[idt]) + 0.5, min=0.0)
            op = op64[idt] * pr64.compensation                   # antialiased mode
            y_lo, x_lo = ty * TILE, tx * TILE
            hh, ww = min(TILE, H - y_lo), min(TILE, W - x_lo)
            py = (torch.arange(y_lo, y_lo + hh, dtype=torch.float64) + 0.5)[:, None].expand(hh, ww).reshape(-1)
            px = (torch.arange(x_lo, x_lo + ww, dtype=torch.float64) + 0.5)[None, :].expand(hh, ww).reshape(-1)
            C, T, stop_pos, last, frag, reached = composite_tile(pr64.xys, pr64.conics, rgb, op, px, py)
            keep = min(ids.size, int(min(stop_pos.max().item(), ids.size - 1)) + 9)
            key = f"t{ti}_p{p}"
            out[key + "_n"] = np.int64(ids.size)
            out[key + "_sum"] = np.uint64(list_checksum(ids))
            out[key + "_ids"] = ids[:keep].astype(np.int32)
            out[key + "_rgb"] = C.reshape(hh, ww, 3).numpy()
            out[key + "_T"] = T.reshape(hh, ww).numpy()
            out[key + "_stop"] = stop_pos.reshape(hh, ww).numpy().astype(np.int32)
            out[key + "_last"] = last.reshape(hh, ww).numpy().astype(np.int32)
            out[key + "_frag"] = np.packbits(frag.numpy())
            # ---- round 5: GRADIENTS at the bench's own size.  d (sum_pixels w . rgb) / d (record
            # fields) of the tile's reachable entries — screen-space centre (2), conic (3), opacity (1), colour (3) — by
            # float64 autograd through gs_oracle.rasterize_sorted on the tile alone, under both gradient conventions of
            # the alpha clamp (_gu: the reference's = the product's default, _g: true derivatives).  w: seeded per
            # (tile, sub-pose), zero on the fragile pixels.  The test feeds the SAME w (zero outside the sampled tiles) to
            # the HIP backward of the full frame and compares the rows of these entries.
            wt = tile_weights(ti, p, hh, ww) * (~frag).reshape(hh, ww, 1)
            shift = torch.tensor([float(x_lo), float(y_lo)], dtype=torch.float64)
            bins1 = np.array([[0, keep]], dtype=np.int32)
            for conv, upflag in (("_gu", O.UP_ALPHA_CLAMP), ("_g", 0)):
                leaves = [t[:keep].detach().clone().requires_grad_(True) for t in (pr64.xys, pr64.conics, rgb, op)]
                r = O.rasterize_sorted(leaves[0] - shift[None, :], leaves[1], leaves[2], leaves[3],
                                       np.arange(keep, dtype=np.int32), bins1, hh, ww, None, upstream=upflag)
                if conv == "_gu":
                    assert (r.img.detach().reshape(-1, 3) - C).abs().max() < 1e-12, key
                (r.img * wt).sum().backward()
                gmat = torch.cat([leaves[0].grad, leaves[1].grad, leaves[3].grad[:, None], leaves[2].grad],
                                 dim=1).numpy().astype(np.float32)                    # [keep, 9]: xy, conic, opacity, rgb
                if conv == "_gu" or not np.array_equal(gmat, out[key + "_gu"]):
                    out[key + conv] = gmat        # _g is stored only where the clamp is reached (else: equal to _gu)
        print(f"{tag}: sub-pose {p + 1}/{S * R} done", flush=True)
    out["tile_intersections_per_step"] = np.int64(total_pairs)
    np.savez_compressed(HERE / out_name, **out)
    print(f"wrote {HERE / out_name}: {total_pairs} bounding-box pairs in the frame")


if __name__ == "__main__":
    torch.set_num_threads(8)
    make("headline (config 2 metric scene)", 5, 1, "full_size_headline.npz")
    make("config 3 (10 row bands)", 1, 10, "full_size_config3.npz")
