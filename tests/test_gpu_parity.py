"""GPU parity: the HIP path (through the C-ABI library) against the CPU oracle on the same seeded
inputs and against the committed golden fixtures.

Bars (north_star): tile counts / radii / sort keys / sorted order / bin edges BIT-EXACT; floats
within the fp32 tolerances written at each assert (fp32 HIP with fast exp vs float64 oracle).
Pixels whose threshold decisions (alpha >= 1/255, T <= 1e-4) sit within rounding of flipping are
flagged by the oracle (`fragile`) and excluded from strict comparisons; everything else is compared.
"""
import math
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

import python_frame_path      # test infrastructure: the Python orchestration twin + its switches (conftest installs it)

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"

IMG_ATOL = 2e-4       # rendered colour, absolute (values in [0,1])
GRAD_RTOL = 3e-3      # gradients, relative to the tensor's max |g| (sums of ~1e3-1e5 fp32 atomics)


def _fuzz_cases(tag, with_bg=False):
    """GSD_FUZZ_CASES=N appends N seeded random (n, W, H, scale_mult) shapes to the compat-op parity tests
    (ragged sizes, 50..20000 Gaussians, many empty tiles); 0 / unset keeps the suite at its committed size."""
    import os
    import random
    k = int(os.environ.get("GSD_FUZZ_CASES", "0"))
    rng = random.Random(sum(map(ord, tag)) + 17)          # stable across processes (str hashes are salted)
    out = []
    for _ in range(k):
        c = (rng.choice([50, 200, 700, 4000, 20000]), rng.randint(16, 500), rng.randint(16, 300),
             rng.choice([1.0, 2.0, 5.0, 12.0]))
        out.append(c + (rng.random() < 0.5,) if with_bg else c)
    return out


def rel_max(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


import contextlib  # noqa: E402

# the two gradient conventions the FIXTURES store (gu_* / g_*): 7 = the reference's, as recollected (opt-in since round
# 6), 0 = the true derivatives.  The default on both sides is 6 (ops.UPSTREAM_GRADS, oracle DEFAULT_GRADS: 7 without the
# fov rule): the live-oracle comparisons (baseline configs, convention switches, smoke) run under it as well
CONVENTIONS = [pytest.param(7, id="upstream-grads"), pytest.param(0, id="true-derivatives")]


@contextlib.contextmanager
def grad_convention(flags):
    from gsdeblur_amd import ops
    old = ops.UPSTREAM_GRADS
    ops.UPSTREAM_GRADS = int(flags)
    try:
        yield "gu_" if flags else "g_"
    finally:
        ops.UPSTREAM_GRADS = old


# Per-ELEMENT gradient bar against the float64 oracle (a bound relative to the tensor's max says
# nothing about small gradients): |got - ref| <= GRAD_EL_RTOL * |ref| + GRAD_EL_AFRAC * max|ref|.  The absolute
# part is the fp32 noise floor of sums of 1e3..1e5 terms that partly cancel (measured worst case ~4e-6 of the
# tensor's max, run r02_run6); the relative part binds every element above that floor.
GRAD_EL_RTOL = 1e-4
GRAD_EL_AFRAC = 1e-5
FRAGILE_MAX = 0.10    # default ceiling for a scene without an entry in FRAGILE_OBSERVED
# Share of the pixels the ORACLE flags as within rounding of a threshold decision (alpha = 1/255, T = 1e-4), per test
# scene: a property of the seeded scene and the oracle's bands, printed by every run ("[fragile <tag>]").  Each scene is
# held to 2x the share observed (bound it near what is observed instead of a flat 10 %).
FRAGILE_OBSERVED = {   # round 5 (the oracle's rounding-model bands), gpurun visit r5_v6 (profiles/r05_suite_prints.log;
                       # regenerate with tools/fragile_table.py <pytest -s log>)
    'baseline config2: 5 motion-blur sub-poses': 0.01728,
    'baseline config3: 10 rolling-shutter bands': 0.00354,
    'baseline config4: 5 samples x 2 bands': 0.02110,
    'baseline config5: 10 motion-blur sub-poses': 0.03730,
    'exact-rs S=1 144x128 n=2500 base=None': 0.00309,
    'exact-rs S=2 96x128 n=6000 base=8': 0.00814,
    'exact-rs S=3 128x160 n=3000 base=None': 0.01079,
    'full-size full_size_config3.npz': 0.00378,
    'full-size full_size_headline.npz': 0.00410,
    'fused S=1 R=1 160x96 n=3000': 0.00306,
    'fused S=1 R=6 96x200 n=2000': 0.00161,
    'fused S=2 R=3 112x80 n=1500': 0.00592,
    'fused S=5 R=1 128x128 n=2000': 0.01349,
    'golden blur_large': 0.01792,
    'needle pixel_velocity': 0.01042,
    'needle se3': 0.00897,
    'pixvel S=1 R=4 96x144 n=2000': 0.00188,
    'pixvel S=3 R=2 128x96 n=2500': 0.01123,
    'pixvel S=5 R=1 160x96 n=3000': 0.01706,
    'posed pixel_velocity S=3 R=2': 0.00991,
    'posed pixel_velocity S=5 R=1': 0.01830,
    'posed se3 S=3 R=2': 0.00991,
    'rasterize n=2000 48x40 mult=3.0': 0.00104,
    'rasterize n=3000 100x60 mult=12.0': 0.00183,
    'rasterize n=5000 256x256 mult=4.0': 0.00314,
    'shared-list S=2 96x128 n=6000 rt=0.0333 base=8': 0.00814,
    'shared-list S=3 144x128 n=2500 rt=0.0333 base=None': 0.00939,
    'shared-list S=5 128x160 n=3000 rt=0.0000 base=None': 0.01714,
}


def check_fragile(frag, tag):
    f = float(np.asarray(frag, dtype=np.float64).mean()) if not isinstance(frag, torch.Tensor) else float(frag.double().mean())
    bound = 2.0 * FRAGILE_OBSERVED[tag] + 1e-4 if tag in FRAGILE_OBSERVED else FRAGILE_MAX
    print(f"[fragile {tag}] {f:.5f} (bound {bound:.5f})")
    assert f <= bound, (tag, f, bound)
    return f


def grad_el_ratio(got, ref):
    """max over elements of |got-ref| / (rtol*|ref| + afrac*max|ref|)  (<= 1 passes).
    (Round 5 gave the camera-level tensors — view matrix, velocities: <= 16 elements — a 3x wider absolute floor because
    their last reduction was fp32 atomics whose order changed from run to run; since round 6 those sums are ordered
    (csrc/project.hip: reduce_vV / pose_reduce_kernel, subpose_bwd_kernel) and every tensor is held to the same bar;
    tests/test_gpu_round6.py asserts that two runs of a frame give them bit for bit.)"""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    tol = GRAD_EL_RTOL * np.abs(ref) + GRAD_EL_AFRAC * (np.abs(ref).max() + 1e-300)
    return float((np.abs(got - ref) / tol).max())


# Two fp32 compositor formulations (round 4: the scalar-cache kernels test `sigma >= 0 and alpha >= 1/255` with one
# compare on a shifted exponent; the round-1 kernels with two compares on alpha) round differently, so a pixel within
# an ulp of a threshold (alpha = 1/255, T = 1e-4) may take the other branch: such a pixel differs by up to one blend
# weight.  Compared like the oracle comparisons: values within IMG_ATOL, except a bounded fraction of threshold
# pixels.  The bound is 2x the largest fraction observed on the GPU (printed by every call; profiles/r04_*).
KERNEL_FRAGILE_FRAC = 1e-5


def images_close(a, b, atol, what, frac_max=KERNEL_FRAGILE_FRAC):
    d = (a.double() - b.double()).abs()
    frac = float((d > atol).double().mean())
    print(f"[kernel-vs-kernel {what}] max|d| = {float(d.max()):.3e}, values over {atol:g}: {frac:.3e} of {d.numel()}")
    assert torch.isfinite(a).all() and frac <= frac_max, (what, frac, float(d.max()))
    return frac


def to_dev(sc, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}


# --------------------------------------------------------------------------- #
# integer primitives
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("n", [1, 63, 2048, 2049, 100_003, 3_000_017])
def test_exclusive_scan(gs, dev, n):
    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 300, (n,), generator=g, dtype=torch.int32)
    ex, total = gs.exclusive_scan_u32(x.to(dev))
    ref = torch.cumsum(x.long(), 0) - x.long()
    assert torch.equal(ex.cpu().long(), ref)
    assert int(total.item()) == int(x.long().sum())


@pytest.mark.parametrize("n,bits,dtype", [(1, 8, torch.int32), (4097, 16, torch.int32), (300_001, 17, torch.int32),
                                          (1_000_003, 32, torch.int32), (200_003, 45, torch.int64),
                                          (50_001, 35, torch.int64), (70_000, 13, torch.int32)])
def test_radix_sort_stable(gs, dev, n, bits, dtype):
    """ascending, STABLE (ties keep input order), payload follows; keys limited to `bits` bits."""
    g = torch.Generator().manual_seed(bits * 1000 + n % 997)
    hi = 2 ** min(bits, 62)
    keys = torch.randint(0, hi, (n,), generator=g, dtype=torch.int64)
    if n > 100:
        keys[: n // 3] = keys[n // 3: 2 * (n // 3)]        # plenty of ties
    vals = torch.arange(n, dtype=torch.int32)
    kd = keys.to(dtype).to(dev) if dtype == torch.int64 else keys.to(torch.int32).to(dev)   # u32 bit pattern
    ks, vs = gs.radix_sort_pairs(kd.clone(), vals.to(dev).clone(), 0, bits)
    order = torch.sort(keys, stable=True).indices
    assert torch.equal(vs.cpu().long(), order)
    got = ks.cpu().long()
    if dtype == torch.int32:
        got = got & 0xFFFFFFFF
    assert torch.equal(got, keys[order])
    # iota payload path
    ks2, vs2 = gs.radix_sort_pairs(kd.clone(), None, 0, bits)
    assert torch.equal(vs2.cpu().long(), order)


@pytest.mark.parametrize("n,bits,cap", [(1, 8, 0), (4097, 16, 0), (300_001, 17, 0), (70_000, 16, 200_000),
                                        (1_000_003, 24, 0)])
def test_radix_sort_carries_a_second_payload(gs, dev, n, bits, cap):
    """the tile sort's form: payload = input index (iota), second payload carried along with the keys through every
    pass (instead of gathered by payload afterwards); optionally with the element count on the device"""
    g = torch.Generator().manual_seed(n + bits)
    keys = torch.randint(0, 2 ** bits, (n,), generator=g, dtype=torch.int64)
    if n > 100:
        keys[: n // 3] = keys[n // 3: 2 * (n // 3)]
    second = torch.randint(0, 2 ** 31 - 1, (max(n, cap) + 8,), generator=g, dtype=torch.int32)
    order = torch.sort(keys, stable=True).indices
    kd = keys.to(torch.int32).to(dev)
    n_dev = None
    if cap:
        kbuf = torch.full((cap,), 12345, dtype=torch.int32, device=dev)
        kbuf[:n] = kd
        kd, n_dev = kbuf, torch.tensor([n], dtype=torch.int32, device=dev)
    src = second.to(dev)
    ks, vs, ps = gs.radix_sort_pairs(kd, None, 0, bits, carry=src, n_dev=n_dev)
    assert torch.equal(src.cpu(), second)                                  # input payload left intact
    assert torch.equal(vs[:n].cpu().long(), order)
    assert torch.equal(ks[:n].cpu().long(), keys[order])
    assert torch.equal(ps[:n].cpu(), second[:n][order])


@pytest.mark.parametrize("P,N", [(1, 5000), (5, 4097), (3, 100_003), (10, 4096)])
def test_segmented_sort_stable(gs, dev, P, N):
    """every segment sorted independently, ascending, stable; payload = global index"""
    from gsdeblur_amd import ops
    g = torch.Generator().manual_seed(P * 7 + N)
    keys = torch.randint(0, 2 ** 32, (P * N,), generator=g, dtype=torch.int64)
    keys[: N // 2] = keys[N // 2: 2 * (N // 2)]                 # ties
    keys[-3:] = 0xFFFFFFFF                                      # "culled" marker sorts last
    ks, vs = python_frame_path.segmented_sort_pairs_u32(keys.to(torch.int32).to(dev), N)
    for p in range(P):
        seg = keys[p * N:(p + 1) * N]
        order = torch.sort(seg, stable=True).indices + p * N
        assert torch.equal(vs[p * N:(p + 1) * N].cpu().long(), order)
        assert torch.equal(ks[p * N:(p + 1) * N].cpu().long() & 0xFFFFFFFF, keys[order])


@pytest.mark.parametrize("digit", [8, 11])
@pytest.mark.parametrize("P,N,keep,max_tiles", [(1, 5000, 0.3, 50), (5, 4097, 0.25, 50), (3, 100_003, 0.27, 50),
                                                (10, 4096, 0.0, 50), (4, 9000, 1.0, 50), (6, 20_000, 0.01, 50),
                                                # 8.5 M keys = 24 index bits: the tile counts packed into the payload
                                                # (round 5, 8-bit digits) saturate at 255 and are looked up instead
                                                (5, 1_700_000, 0.25, 3000), (2, 300_000, 0.5, 9000)])
def test_depth_rank_compacting(gs, dev, P, N, keep, digit, max_tiles):
    """compacting depth pre-sort: culled keys dropped by the first pass, survivors sorted stably at the start of their
    segment, their tile counts gathered by the last pass, the segment-aware scan treats everything behind as zero;
    the result is the same ranking / prefix the full sort + gather + scan produce"""
    from gsdeblur_amd import ops
    g = torch.Generator().manual_seed(P * 11 + N)
    keys = torch.randint(0, 2 ** 31, (P * N,), generator=g, dtype=torch.int64)
    keys[: N // 2] = keys[N // 2: 2 * (N // 2)]                 # ties
    culled = torch.rand(P * N, generator=g) >= keep
    if P > 2:
        culled[N:2 * N] = True                                  # one sub-pose sees nothing at all
    keys[culled] = 0xFFFFFFFF
    ntiles = torch.randint(1, max_tiles, (P * N,), generator=g, dtype=torch.int32)
    if max_tiles > 50:
        # most Gaussians small, every value around the packed field's limit (2^(32 - index bits) - 1) present
        ntiles[torch.rand(P * N, generator=g) < 0.9] = 3
        cap = 2 ** (32 - max(1, (P * N - 1).bit_length())) - 1
        ntiles[:6] = torch.tensor([cap - 1, cap, cap + 1, 1, max_tiles, cap], dtype=torch.int32)
        culled[:6] = False
        keys[:6] = torch.randint(0, 2 ** 31, (6,), generator=g, dtype=torch.int64)
    ntiles[culled] = 0
    records = torch.zeros(1, device=dev)
    old = ops.DEPTH_SORT_COMPACT, ops.DEPTH_SORT_DIGIT
    try:
        ops.DEPTH_SORT_COMPACT, ops.DEPTH_SORT_DIGIT = 1, digit      # 4 passes of 8 bits / 3 passes of 11, 11, 9
        sgi, cum, total, n_live = python_frame_path._depth_rank(records, keys.to(torch.int32).to(dev), ntiles.to(dev), P, N)
        ops.DEPTH_SORT_COMPACT = 0
        sgi0, cum0, total0, none = python_frame_path._depth_rank(records, keys.to(torch.int32).to(dev), ntiles.to(dev), P, N)
    finally:
        ops.DEPTH_SORT_COMPACT, ops.DEPTH_SORT_DIGIT = old
    assert none is None
    live = n_live.cpu().long()
    assert torch.equal(live, (~culled).view(P, N).sum(1))
    assert int(total.item()) == int(total0.item()) == int(ntiles.sum())
    # the prefix and the ranking where they are defined: the live ranks and every segment's first rank
    for p in range(P):
        m = int(live[p])
        assert torch.equal(cum[p * N:p * N + max(m, 1)].cpu(), cum0[p * N:p * N + max(m, 1)].cpu())
        assert torch.equal(sgi[p * N:p * N + m].cpu(), sgi0[p * N:p * N + m].cpu())


@pytest.mark.parametrize("n,num_bins,cap", [(1, 700, 0), (5000, 40_801, 0), (200_000, 40_801, 0), (3000, 40_801, 9000),
                                            (0, 513, 64), (64, 64, 0), (70_000, 129, 100_000)])
def test_bin_edges_write_every_bin_without_a_prior_fill(gs, dev, n, num_bins, cap):
    """gs_tile_bin_edges_u32 (round 5: self-zeroing): [start, end) for every key that occurs, (0, 0) for every key that
    does not — the gap in front of the first key, between keys, behind the last one — with the bins POISONED beforehand;
    with the entry count on the host or on the device (n_dev, including a device-side count of zero)"""
    from gsdeblur_amd import _lib
    from gsdeblur_amd.ops import _ptr, _stream
    L = _lib.load()
    g = torch.Generator().manual_seed(n + num_bins)
    keys = torch.sort(torch.randint(0, num_bins, (n,), generator=g, dtype=torch.int64)).values
    if n > 100:
        # long key-less stretches, the first and the last bin empty / occupied
        keys = torch.sort(torch.where((keys > num_bins // 5) & (keys < num_bins // 2), keys[0], keys)).values
        keys[-1] = num_bins - 1 if n % 2 else keys[-1]
    kbuf = torch.full((max(n, cap, 1),), num_bins - 1, dtype=torch.int32, device=dev)
    kbuf[:n] = keys.to(torch.int32).to(dev)
    n_dev = torch.tensor([n], dtype=torch.int32, device=dev) if cap else None
    bins = torch.full((num_bins, 2), 0x7F7F7F7F, dtype=torch.int32, device=dev)
    _lib.check(L.gs_tile_bin_edges_u32(max(n, cap), _ptr(kbuf), num_bins, _ptr(bins), _ptr(n_dev), _stream()), "bin edges")
    lo = torch.searchsorted(keys, torch.arange(num_bins), right=False)
    hi = torch.searchsorted(keys, torch.arange(num_bins), right=True)
    want = torch.stack([lo, hi], 1)
    want[hi == lo] = 0
    assert torch.equal(bins.cpu().long(), want)


def test_combine_bwd_scale_takes_unaligned_tensors(gs, dev):
    """the averaging backward's per-pixel factor reads and writes 16 bytes per thread when its three tensors allow it;
    views at odd offsets (a gradient that is a slice of a larger tensor) give the same values"""
    from gsdeblur_amd import _lib
    from gsdeblur_amd.ops import _ptr, _stream
    L = _lib.load()
    n = 3 * 37 * 53 + 2
    g = torch.Generator().manual_seed(5)
    out, v = torch.rand(n + 8, generator=g).to(dev), torch.randn(n + 8, generator=g).to(dev)
    got = []
    for off_o, off_v, off_s in ((0, 0, 0), (1, 0, 0), (0, 3, 0), (0, 0, 2), (1, 2, 3)):
        o = torch.empty(n + 8, device=dev)[off_o:off_o + n].copy_(out[:n])
        vv = torch.empty(n + 8, device=dev)[off_v:off_v + n].copy_(v[:n])
        sc = torch.full((n + 8,), float("nan"), device=dev)[off_s:off_s + n]
        _lib.check(L.gs_combine_bwd_scale(5, n, 2.2, _ptr(o), _ptr(vv), _ptr(sc), _stream()), "combine scale")
        got.append(sc.cpu().clone())
    want = (out[:n].double().clamp_min(1e-12) ** (1.0 - 2.2) / 2.2 / 5.0 * v[:n].double()).cpu()
    assert torch.allclose(got[0].double(), want, rtol=2e-5, atol=1e-9)
    for x in got[1:]:
        assert torch.equal(x, got[0])


# --------------------------------------------------------------------------- #
# projection / SH / sub-poses
# --------------------------------------------------------------------------- #
def _scene(oracle, n, W, H, seed, scale_mult, dev):
    sc = oracle.synthetic_scene(n, W, H, seed=seed, scale_mult=scale_mult)
    means = sc["means"].clone()
    k = max(1, n // 100)
    means[:k, 2] = -1.0
    means[k:2 * k, 0] *= 5.0
    sc["means"] = means
    return sc


def test_subpose_viewmats_fwd_bwd(gs, oracle, dev):
    V0 = oracle.subpose_viewmats(torch.eye(4, dtype=torch.float64), torch.tensor([0.3, -0.2, 0.5], dtype=torch.float64),
                                 torch.tensor([0.2, 0.4, -0.1], dtype=torch.float64), [1.0])[0].float()
    lin, ang = torch.tensor([0.1, 0.05, -0.2]), torch.tensor([0.05, -0.08, 0.03])
    times = torch.tensor([-0.02, -0.01, 0.0, 0.01, 0.3])
    for a in (ang, torch.zeros(3)):
        Vd, ld, ad = (t.to(dev).requires_grad_(True) for t in (V0, lin, a))
        out = gs.subpose_viewmats(Vd, ld, ad, times.to(dev))
        V64, l64, a64 = (t.double().requires_grad_(True) for t in (V0, lin, a))
        ref = oracle.subpose_viewmats(V64, l64, a64, times.tolist())
        assert np.abs(out.detach().cpu().numpy() - ref.detach().numpy()).max() < 2e-6
        go = torch.randn(5, 4, 4, generator=torch.Generator().manual_seed(1))
        go[:, 3, :] = 0
        (out * go.to(dev)).sum().backward()
        (ref * go.double()).sum().backward()
        assert rel_max(Vd.grad.cpu()[:3], V64.grad[:3]) < 1e-5
        assert rel_max(ld.grad.cpu(), l64.grad) < 1e-5
        assert rel_max(ad.grad.cpu(), a64.grad) < 1e-5


@pytest.mark.parametrize("n,W,H,mult", [(5000, 256, 256, 4.0), (20000, 640, 360, 2.0), (7, 16, 16, 10.0)]
                         + _fuzz_cases("project"))
def test_project_gaussians_parity(gs, oracle, dev, n, W, H, mult):
    O = oracle
    sc = _scene(O, n, W, H, 7, mult, dev)
    V = O.subpose_viewmats(torch.eye(4, dtype=torch.float64), torch.tensor([0.1, 0.05, -0.2], dtype=torch.float64),
                           torch.tensor([0.05, -0.08, 0.03], dtype=torch.float64), [1.0])[0].float()
    scales, quats = sc["log_scales"].exp(), sc["quats"] * 1.7
    pr = O.project_gaussians(sc["means"], scales, 1.0, quats, V, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W)
    md, sd, qd, Vd = (t.to(dev).requires_grad_(True) for t in (sc["means"], scales, quats, V))
    xys, depths, radii, conics, comp, ntiles, cov3d = gs.project_gaussians(
        md, sd, 1.0, qd, Vd, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, 16)
    # integers and depth key bits: exact
    assert torch.equal(radii.cpu(), pr.radii)
    assert torch.equal(ntiles.cpu(), pr.num_tiles_hit)
    assert torch.equal(depths.cpu().view(torch.int32), pr.depths.view(torch.int32))
    ok = pr.radii > 0

    def ulps(a, b):
        return (a.cpu().contiguous().view(torch.int32).long() - b.contiguous().view(torch.int32).long()).abs()
    # floats: same IEEE op order, no fma contraction, correctly rounded div/sqrt on both sides ->
    # bit-equal to the float32 oracle (cov3d, xys, conics); comp goes through one more sqrt/div chain
    diag = {k: int(ulps(a[ok], b[ok]).max()) for k, (a, b) in dict(
        xys=(xys.detach(), pr.xys), conics=(conics.detach(), pr.conics), comp=(comp.detach(), pr.compensation),
        cov3d=(cov3d.detach(), pr.cov3d)).items()}
    print("project float ulp diffs vs float32 oracle:", diag)
    assert diag["xys"] == 0 and diag["cov3d"] == 0 and diag["conics"] == 0, diag
    assert diag["comp"] <= 2, diag
    assert (xys.detach().cpu()[~ok] == 0).all()
    # backward vs float64 autograd
    g = torch.Generator().manual_seed(1)
    vx, vd, vc, vcomp = (torch.randn(*s, generator=g) for s in ((n, 2), (n,), (n, 3), (n,)))
    loss = (xys * vx.to(dev)).sum() + (depths * vd.to(dev) * (radii > 0)).sum() + (conics * vc.to(dev)).sum() + \
        (comp * vcomp.to(dev)).sum()
    loss.backward()
    m64, s64, q64, V64 = (t.double().requires_grad_(True) for t in (sc["means"], scales, quats, V))
    prd = O.project_gaussians(m64, s64, 1.0, q64, V64, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W)
    l64 = (prd.xys * vx.double()).sum() + (prd.depths * vd.double() * (prd.radii > 0)).sum() + \
        (prd.conics * vc.double()).sum() + (prd.compensation * vcomp.double()).sum()
    l64.backward()
    assert rel_max(md.grad.cpu(), m64.grad) < 3e-5
    assert rel_max(sd.grad.cpu(), s64.grad) < 3e-5
    assert rel_max(qd.grad.cpu(), q64.grad) < 3e-5
    assert rel_max(Vd.grad.cpu()[:3], V64.grad[:3]) < 3e-4     # sum over N of fp32 terms (wave reduce + atomics)


@pytest.mark.parametrize("deg,K", [(0, 1), (1, 4), (2, 9), (3, 16), (4, 25), (2, 16)])
def test_spherical_harmonics_parity(gs, oracle, dev, deg, K):
    g = torch.Generator().manual_seed(deg)
    n = 3000
    dirs = torch.randn(n, 3, generator=g) * 3.0                   # not normalised on purpose
    coeffs = torch.randn(n, K, 3, generator=g)
    cd = coeffs.to(dev).requires_grad_(True)
    col = gs.spherical_harmonics(deg, dirs.to(dev), cd)
    c64 = coeffs.double().requires_grad_(True)
    ref = oracle.spherical_harmonics(deg, dirs.double(), c64)
    assert np.abs(col.detach().cpu().numpy() - ref.detach().numpy()).max() < 2e-5
    go = torch.randn(n, 3, generator=g)
    (col * go.to(dev)).sum().backward()
    (ref * go.double()).sum().backward()
    assert rel_max(cd.grad.cpu(), c64.grad) < 1e-5
    assert cd.grad.shape == (n, K, 3)


# --------------------------------------------------------------------------- #
# binning: 64-bit upstream-format route and the 2-stage route agree with the oracle exactly
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("n,W,H,mult", [(5000, 256, 256, 4.0), (30000, 640, 368, 2.0), (3, 16, 16, 10.0)]
                         + _fuzz_cases("binning"))
def test_binning_keys_and_order_bit_exact(gs, oracle, dev, n, W, H, mult):
    O = oracle
    sc = _scene(O, n, W, H, 13, mult, dev)
    if n >= 260:                      # force exact depth ties: identical Gaussians must order by id
        sc["means"][200:260] = sc["means"][140:200]
    scales = sc["log_scales"].exp()
    pr = O.project_gaussians(sc["means"], scales, 1.0, sc["quats"], sc["viewmat"], sc["fx"], sc["fy"], sc["cx"],
                             sc["cy"], H, W)
    keys, gids = O.map_gaussian_to_intersects(pr, W)
    skeys, sgids = O.sort_intersects(keys, gids)
    tx, ty = (W + 15) // 16, (H + 15) // 16
    bins_ref = O.get_tile_bin_edges(skeys, tx * ty)

    xys, depths, radii, conics, comp, ntiles, _ = gs.project_gaussians(
        sc["means"].to(dev), scales.to(dev), 1.0, sc["quats"].to(dev), sc["viewmat"].to(dev), sc["fx"], sc["fy"],
        sc["cx"], sc["cy"], H, W, 16)
    num_isect, cum = gs.compute_cumulative_intersects(ntiles)
    assert num_isect == len(keys)
    assert torch.equal(cum.cpu().long(), torch.cumsum(pr.num_tiles_hit.long(), 0))
    iu, gu, isrt, gsrt, bins = gs.bin_and_sort_gaussians(n, num_isect, xys, depths, radii, cum, (tx, ty, 1), 16)
    assert np.array_equal(iu.cpu().numpy(), keys) and np.array_equal(gu.cpu().numpy(), gids)
    assert np.array_equal(isrt.cpu().numpy(), skeys)                      # sort keys bit-exact
    assert np.array_equal(gsrt.cpu().numpy(), sgids)                      # incl. the tie-break order
    assert np.array_equal(bins.cpu().numpy(), bins_ref)

    # 2-stage route (depth pre-sort -> emit -> stable tile sort) yields the identical order
    from gsdeblur_amd import ops
    L = gs._lib.load()
    rec = torch.empty(n, ops.REC, device=dev)
    dk = torch.empty(n, dtype=torch.int32, device=dev)
    nt2 = torch.empty(n, dtype=torch.int32, device=dev)
    col = torch.rand(n, 3, device=dev)
    op = torch.rand(n, device=dev)
    gs._lib.check(L.gs_pack_records(n, ops._ptr(xys), ops._ptr(depths), ops._ptr(radii), ops._ptr(conics),
                                    ops._ptr(col), ops._ptr(op), H, W, ops._ptr(rec), ops._ptr(dk), ops._ptr(nt2),
                                    ops._stream()), "pack")
    assert torch.equal(nt2.cpu(), pr.num_tiles_hit)
    svals, bins2, I2, skeys32 = gs.bin_and_sort_records(rec, dk, nt2, 1, n, H, W)
    assert I2 == len(keys)
    if I2:
        assert np.array_equal(svals.cpu().numpy(), sgids)
        assert np.array_equal(skeys32.cpu().numpy().astype(np.int64), skeys >> 32)
    assert np.array_equal(bins2.cpu().numpy(), bins_ref)


# --------------------------------------------------------------------------- #
# rasterize_gaussians forward / backward
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("n,W,H,mult,bg", [(5000, 256, 256, 4.0, True), (3000, 100, 60, 12.0, False),
                                           (2000, 48, 40, 3.0, True)] + _fuzz_cases("raster", with_bg=True))
def test_rasterize_gaussians_parity(gs, oracle, dev, n, W, H, mult, bg):
    """config 1 of BASELINE.json (5k Gaussians, 256x256, 1 sub-pose) plus ragged image sizes."""
    O = oracle
    sc = _scene(O, n, W, H, 17, mult, dev)
    scales = sc["log_scales"].exp()
    pr = O.project_gaussians(sc["means"], scales, 1.0, sc["quats"], sc["viewmat"], sc["fx"], sc["fy"], sc["cx"],
                             sc["cy"], H, W)
    g = torch.Generator().manual_seed(2)
    colors = torch.rand(n, 3, generator=g)
    opac = torch.sigmoid(sc["opacity_logits"]) * pr.compensation
    background = torch.tensor([0.3, 0.6, 0.1]) if bg else None
    # oracle in float64 on the float32-projected inputs
    x64, c64, col64, o64 = (t.double().requires_grad_(True) for t in (pr.xys, pr.conics, colors, opac))
    img_ref, alpha_ref, r = O.rasterize_gaussians(x64, pr.depths, pr.radii, c64, pr.num_tiles_hit, col64, o64, H, W,
                                                  background=None if background is None else background.double(),
                                                  return_alpha=True, proj=pr)
    xd, cd, cold, od = (t.to(dev).requires_grad_(True) for t in (pr.xys, pr.conics, colors, opac))
    bgd = None if background is None else background.to(dev).requires_grad_(True)
    img, alpha = gs.rasterize_gaussians(xd, pr.depths.to(dev), pr.radii.to(dev), cd, pr.num_tiles_hit.to(dev), cold,
                                        od[:, None], H, W, 16, bgd, return_alpha=True)
    good = ~r.fragile
    check_fragile(r.fragile, f"rasterize n={n} {W}x{H} mult={mult}")
    d_img = (img.detach().cpu().double() - img_ref.detach()).abs()
    assert d_img[good].max().item() < IMG_ATOL
    assert (alpha.detach().cpu().double() - alpha_ref.detach()).abs()[good].max().item() < IMG_ATOL
    # fragile pixels may flip one threshold decision but stay bounded
    assert d_img.max().item() < 0.05
    # backward on the non-fragile pixels
    wt = torch.rand(H, W, 3, generator=g) * good[..., None]
    wa = torch.rand(H, W, generator=g) * good
    ((img * wt.to(dev)).sum() + (alpha * wa.to(dev)).sum()).backward()
    ((img_ref * wt.double()).sum() + (alpha_ref * wa.double()).sum()).backward()
    assert rel_max(xd.grad.cpu(), x64.grad) < GRAD_RTOL
    assert rel_max(cd.grad.cpu(), c64.grad) < GRAD_RTOL
    assert rel_max(cold.grad.cpu(), col64.grad) < GRAD_RTOL
    assert rel_max(od.grad.cpu(), o64.grad) < GRAD_RTOL
    if bg:
        vbg_ref = (r.final_T.detach()[..., None] * wt.double()).sum(dim=(0, 1))
        assert rel_max(bgd.grad.cpu(), vbg_ref) < 1e-4


def test_rasterize_empty_scene(gs, dev):
    """all Gaussians culled -> background image, zero alpha, zero gradients (edge case: I == 0)."""
    n, H, W = 10, 40, 50
    xys = torch.zeros(n, 2, device=dev, requires_grad=True)
    col = torch.rand(n, 3, device=dev, requires_grad=True)
    bg = torch.tensor([0.2, 0.4, 0.6], device=dev)
    img, alpha = gs.rasterize_gaussians(xys, torch.ones(n, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
                                        torch.ones(n, 3, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
                                        col, torch.ones(n, 1, device=dev), H, W, 16, bg, return_alpha=True)
    assert torch.allclose(img, bg.expand(H, W, 3)) and alpha.abs().max().item() == 0
    img.sum().backward()
    assert xys.grad.abs().max().item() == 0 and col.grad.abs().max().item() == 0


# --------------------------------------------------------------------------- #
# fused multi-sub-pose path vs the full oracle, and the committed golden fixtures
# --------------------------------------------------------------------------- #
def _run_full(gs, oracle, dev, sc, H, W, S, R, et, rt, gamma, mlevel, deg, background, weights):
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    p = {k: sc[k].float().to(dev).requires_grad_(True) for k in names}
    times, _, _ = gs.subpose_schedule(S, et, R, rt)
    vms = gs.subpose_viewmats(p["viewmat"], p["lin_vel"], p["ang_vel"], torch.tensor(times, device=dev))
    samples, alphas, radii = gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"],
                                                torch.sigmoid(p["opacity_logits"]), p["sh"], vms,
                                                background.float().to(dev), S, R, sc["fx"], sc["fy"], sc["cx"],
                                                sc["cy"], H, W, sh_degree=deg, antialiased=True)
    out = gs.combine_samples(samples, gamma, mlevel)
    (out * weights.float().to(dev)).sum().backward()
    return out, alphas.mean(0), samples, vms, p, radii


@pytest.mark.parametrize("up", CONVENTIONS)
@pytest.mark.parametrize("name", ["static_small", "blur_rs_small"])
def test_fused_path_matches_golden(gs, dev, name, up):
    with grad_convention(up) as gk:
        _fused_path_matches_golden(gs, dev, name, gk)


def _fused_path_matches_golden(gs, dev, name, gk):
    d = np.load(GOLD / f"{name}.npz")
    H, W, S, R, deg = (int(v) for v in d["cfg"])
    et, rt, gamma, mlevel = (float(v) for v in d["cfg_f"])
    sc = {k: torch.from_numpy(d[k]) for k in ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel",
                                              "ang_vel", "viewmat"]}
    sc.update(fx=float(d["fx"]), fy=float(d["fy"]), cx=float(d["cx"]), cy=float(d["cy"]))
    out, alpha, samples, vms, p, radii = _run_full(gs, None, dev, sc, H, W, S, R, et, rt, gamma, mlevel, deg,
                                                   torch.from_numpy(d["background"]), torch.from_numpy(d["weights"]))
    assert np.abs(vms.detach().cpu().numpy() - d["viewmats"]).max() < 2e-6
    good = ~d["fragile"]
    # tolerance: IMG_ATOL on the per-sample composites; the gamma chain (x^2.2 -> mean -> ^(1/2.2)) with the
    # fast pow amplifies it near the floor, hence 5e-4 on the combined image
    assert np.abs(samples.detach().cpu().numpy() - d["samples"])[:, good].max() < IMG_ATOL
    assert np.abs(out.detach().cpu().numpy() - d["out"])[good].max() < 5e-4
    assert np.abs(alpha.detach().cpu().numpy() - d["alpha"])[good].max() < IMG_ATOL
    if S * R == 1:      # t = 0: the sub-pose viewmat is the input viewmat exactly, so integers must match exactly
        assert np.array_equal(radii[0].cpu().numpy(), d["p0_radii"])
    else:               # HIP float32 closed-form SE(3) vs the fixture's float64 matrix_exp: a ceil() may move
        assert (radii[0].cpu().numpy() != d["p0_radii"]).mean() < 2e-3
    for k in ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel"]:
        assert rel_max(p[k].grad.cpu(), d[gk + k]) < GRAD_RTOL, k
    assert rel_max(p["viewmat"].grad.cpu()[:3], d[gk + "viewmat"][:3]) < GRAD_RTOL


@pytest.mark.parametrize("up", CONVENTIONS)
def test_pixel_velocity_exact_rolling_shutter_matches_golden(gs, dev, up):
    with grad_convention(up) as gk:
        _pixel_velocity_exact_rolling_shutter_matches_golden(gs, dev, gk)


def _pixel_velocity_exact_rolling_shutter_matches_golden(gs, dev, gk):
    """tests/golden/pixvel_exact_rs_posed_small.npz (round 3): the paper's pixel-velocity model with exact per-row
    rolling shutter, three blur samples, seen from a rotated and translated camera — committed float64 oracle numbers:
    sample images, averaged image, alpha and every gradient (Gaussians, viewmat, twist)."""
    d = np.load(GOLD / "pixvel_exact_rs_posed_small.npz")
    H, W, S, R, deg = (int(v) for v in d["cfg"])
    et, rt, gamma, mlevel = (float(v) for v in d["cfg_f"])
    assert [int(v) for v in d["cfg_model"]] == [1, 1] and R == 1
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    p = {k: torch.from_numpy(d[k]).float().to(dev).requires_grad_(True) for k in names}
    times, _, _ = gs.subpose_schedule(S, et, 1, 0.0)
    samples, alphas, radii = gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"],
                                                torch.sigmoid(p["opacity_logits"]), p["sh"], p["viewmat"],
                                                torch.from_numpy(d["background"]).float().to(dev), S, 1, float(d["fx"]),
                                                float(d["fy"]), float(d["cx"]), float(d["cy"]), H, W, sh_degree=deg,
                                                lin_vel=p["lin_vel"], ang_vel=p["ang_vel"],
                                                times=torch.tensor(times, device=dev), rolling_shutter_time=rt)
    out = gs.combine_samples(samples, gamma, mlevel)
    (out * torch.from_numpy(d["weights"]).float().to(dev)).sum().backward()
    good = ~d["fragile"]
    assert good.mean() > 0.9
    assert np.abs(samples.detach().cpu().numpy() - d["samples"])[:, good].max() < IMG_ATOL
    assert np.abs(out.detach().cpu().numpy() - d["out"])[good].max() < 5e-4
    assert np.abs(alphas.mean(0).detach().cpu().numpy() - d["alpha"])[good].max() < IMG_ATOL
    for k in names[:-1]:
        assert rel_max(p[k].grad.cpu(), d[gk + k]) < GRAD_RTOL, k
    assert rel_max(p["viewmat"].grad.cpu()[:3], d[gk + "viewmat"][:3]) < GRAD_RTOL
    assert float(np.abs(d["g_ang_vel"]).max()) > 0 and float(np.abs(d["g_viewmat"]).max()) > 0


@pytest.mark.parametrize("up", CONVENTIONS)
def test_fused_path_matches_large_golden(gs, dev, up):
    with grad_convention(up) as gk:
        _fused_path_matches_large_golden(gs, dev, gk)


def _fused_path_matches_large_golden(gs, dev, gk):
    """24k Gaussians, 640x368, 5 motion-blur sub-poses, SH degree 3, gamma 2.2 (tests/golden/blur_large.npz, float64
    oracle, generated one sub-pose at a time): image / alpha / first sample on the non-fragile pixels, every gradient
    element-wise.  The scene is rebuilt from the seeded generator the fixture names (inputs are not stored)."""
    d = np.load(GOLD / "blur_large.npz")
    n, W, H, seed, deg = (int(v) for v in d["scene"])
    mult, lv, av = (float(v) for v in d["scene_f"])
    Hc, Wc, S, R, _ = (int(v) for v in d["cfg"])
    et, rt, gamma, mlevel = (float(v) for v in d["cfg_f"])
    assert (Hc, Wc) == (H, W) and n >= 20000 and W >= 640 and H >= 360 and S == 5
    sc = gs.data.synthetic_scene(n, W, H, sh_degree=deg, seed=seed, scale_mult=mult)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * lv, sc["ang_vel"] * av
    frag = np.unpackbits(d["fragile"])[:H * W].reshape(H, W).astype(bool)
    check_fragile(frag, "golden blur_large")
    g = torch.Generator().manual_seed(int(d["weights_seed"][0]))
    wt = torch.rand(H, W, 3, generator=g, dtype=torch.float64) * torch.from_numpy(~frag)[..., None]
    out, alpha, samples, vms, p, radii = _run_full(gs, None, dev, sc, H, W, S, R, et, rt, gamma, mlevel, deg,
                                                   torch.from_numpy(d["background"]), wt)
    good = ~frag
    assert np.abs(samples[0].detach().cpu().numpy() - d["sample0"])[good].max() < IMG_ATOL
    assert np.abs(samples.detach().mean(dim=(1, 2)).cpu().numpy() - d["samples_mean"]).max() < 2e-5
    assert np.abs(out.detach().cpu().numpy() - d["out"])[good].max() < 5e-4
    assert np.abs(alpha.detach().cpu().numpy() - d["alpha"])[good].max() < IMG_ATOL
    assert (radii[0].cpu().numpy() != d["p0_radii"]).mean() < 2e-3      # fp32 SE(3) vs float64: a ceil() may move
    worst = {}
    for k in ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel"]:
        worst[k] = grad_el_ratio(p[k].grad.cpu().numpy(), d[gk + k])
    worst["viewmat"] = grad_el_ratio(p["viewmat"].grad.cpu().numpy()[:3], d[gk + "viewmat"][:3])
    print("blur_large: per-element gradient error / tolerance:", {k: round(v, 3) for k, v in worst.items()})
    for k, v in worst.items():
        assert v <= 1.0, (k, v)


@pytest.mark.parametrize("S,R,W,H,n", [(1, 1, 160, 96, 3000), (5, 1, 128, 128, 2000), (1, 6, 96, 200, 2000),
                                       (2, 3, 112, 80, 1500)])
def test_fused_path_vs_oracle_integers_and_image(gs, oracle, dev, S, R, W, H, n):
    """Per sub-pose: radii / tile counts / sorted order bit-exact vs the float32 oracle fed the SAME
    viewmats; image vs the float64 oracle."""
    O = oracle
    sc = O.synthetic_scene(n, W, H, seed=100 + S * 10 + R, scale_mult=6.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10     # visible motion at this size
    et, rt, gamma, mlevel = 1 / 60, 1 / 30, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5))
    out, alpha, samples, vms, p, radii = _run_full(gs, O, dev, sc, H, W, S, R, et, rt, gamma, mlevel, 3, bg, wt)
    vms_cpu = vms.detach().cpu()
    from gsdeblur_amd import ops
    # integer parity per sub-pose with identical viewmats
    P = S * R
    for pidx in range(P):
        pr = O.project_gaussians(sc["means"], sc["log_scales"].exp(), 1.0, sc["quats"], vms_cpu[pidx], sc["fx"],
                                 sc["fy"], sc["cx"], sc["cy"], H, W)
        assert np.array_equal(radii[pidx].cpu().numpy(), pr.radii.numpy())
    # full image vs float64 oracle driven by the same float32 viewmats (skip its own SE(3) exp)
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=R, exposure_time=et,
                         rolling_shutter_time=rt, gamma=gamma, min_rgb_level=mlevel)
    ref, ref_alpha, ref_samples, frag, parts, _ = O.render(
        cfg, sc["means"].double(), sc["log_scales"].double().exp(), sc["quats"].double(),
        torch.sigmoid(sc["opacity_logits"].double()), sc["sh"].double(), sc["viewmat"].double(),
        sc["lin_vel"].double(), sc["ang_vel"].double(), background=bg.double(), return_parts=True)
    good = ~frag
    check_fragile(frag, f"fused S={S} R={R} {W}x{H} n={n}")
    assert (samples.detach().cpu().double() - ref_samples)[:, good].abs().max().item() < IMG_ATOL
    assert (out.detach().cpu().double() - ref)[good].abs().max().item() < 5e-4


# (up = the gradient convention on both sides: 6 = the default since round 6, 7 = the reference's as recollected, 0 = true
#  derivatives; configs 3 and 4 run under 7 and 0, configs 2 and 5 — a minute of float64 oracle each — under the default;
#  the golden fixtures and the full-size fixtures compare under 7 and 0, test_upstream_gradient_convention_switches under
#  every mask)
@pytest.mark.parametrize("tag,S,R,W,H,n,mult,up", [
    ("config2: 5 motion-blur sub-poses", 5, 1, 240, 136, 6000, 5.0, 6),
    ("config3: 10 rolling-shutter bands", 1, 10, 160, 240, 5000, 5.0, 7),
    ("config3: 10 rolling-shutter bands", 1, 10, 160, 240, 5000, 5.0, 0),
    ("config4: 5 samples x 2 bands", 5, 2, 176, 112, 3600, 5.0, 7),
    ("config4: 5 samples x 2 bands", 5, 2, 176, 112, 3600, 5.0, 0),
    ("config5: 10 motion-blur sub-poses", 10, 1, 160, 96, 2800, 5.0, 6)])
def test_baseline_configs_vs_float64_oracle_image_and_per_element_gradients(gs, oracle, dev, tag, S, R, W, H, n, mult, up):
    """BASELINE.json configs 2-5 at reduced N / resolution with the SAME sub-pose structure (S, R), SH degree 3,
    gamma 2.2, min-rgb 10: per-sample composites and the averaged image against the float64 oracle, and every
    gradient ELEMENT-wise (|d| <= 1e-4 |g| + 1e-5 max|g|); at most 10 % of the pixels may be threshold-fragile.
    `up` = the gradient convention on both sides (6: the default; 7: the reference's as recollected; 0: true derivatives); a tenth
    of the opacities is raised to ~1 so that the alpha clamp the two differ on is really reached."""
    with grad_convention(up):
        _baseline_config_vs_oracle(gs, oracle, dev, tag, S, R, W, H, n, mult, up)


def _baseline_config_vs_oracle(gs, oracle, dev, tag, S, R, W, H, n, mult, up):
    O = oracle
    sc = O.synthetic_scene(n, W, H, seed=300 + S * 10 + R, scale_mult=mult)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10     # visible motion at this size
    et, rt, gamma, mlevel = 1 / 60, 1 / 30, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=R, exposure_time=et,
                         rolling_shutter_time=rt, gamma=gamma, min_rgb_level=mlevel, upstream_grads=up)
    sc["opacity_logits"] = sc["opacity_logits"].clone()
    sc["opacity_logits"][::10] += 9.0               # opacity ~ 1: alpha = 0.999 at these splats' centres (clamp active)
    q = {k: sc[k].double().requires_grad_(True) for k in names}
    ref, ref_alpha, ref_samples, frag, parts, _ = O.render(
        cfg, q["means"], q["log_scales"].exp(), q["quats"], torch.sigmoid(q["opacity_logits"]), q["sh"], q["viewmat"],
        q["lin_vel"], q["ang_vel"], background=bg.double(), return_parts=True)
    good = ~frag
    check_fragile(frag, f"baseline {tag}")
    # round 6: the frame-level mask is the UNION over the S * R sub-poses — a pixel of the averaged
    # image is fragile when any sample's is — which is what makes the small multi-sample scenes' shares 1.7-3.7 %; a
    # sample image is only compared where ITS OWN sub-poses are fragile (0.2-0.4 % each: profiles/r06_fragile_histogram.txt)
    _, samp_of, _ = O.subpose_times(S, et, R, rt)
    frag_s = torch.zeros(S, H, W, dtype=torch.bool)
    for pi, part in enumerate(parts):
        frag_s[samp_of[pi]] |= part[4].fragile
    per_sample = float(frag_s.float().mean())
    print(f"[fragile baseline {tag}] per sample image {per_sample:.5f} (frame-level union {float(frag.float().mean()):.5f})")
    assert per_sample < 0.01, per_sample
    # the loss ignores the fragile pixels on both sides (a flipped threshold there changes the gradient by O(1))
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)) * good[..., None]
    (ref * wt.double()).sum().backward()
    out, alpha, samples, vms, p, radii = _run_full(gs, O, dev, sc, H, W, S, R, et, rt, gamma, mlevel, 3, bg, wt)
    assert (samples.detach().cpu().double() - ref_samples)[~frag_s].abs().max().item() < IMG_ATOL
    assert (out.detach().cpu().double() - ref.detach())[good].abs().max().item() < 5e-4
    worst = {}
    for k in names:
        g_hip, g_ref = p[k].grad.cpu().numpy(), q[k].grad.numpy()
        if k == "viewmat":
            g_hip, g_ref = g_hip[:3], g_ref[:3]
        worst[k] = grad_el_ratio(g_hip, g_ref)
    print(f"{tag}: per-element gradient error / tolerance:", {k: round(v, 3) for k, v in worst.items()},
          f"fragile {frag.float().mean().item():.4f}")
    for k, v in worst.items():
        assert v <= 1.0, (tag, k, v)


@pytest.mark.parametrize("S,R,W,H,n", [(5, 1, 160, 96, 3000), (3, 2, 128, 96, 2500), (1, 4, 96, 144, 2000)])
def test_pixel_velocity_model_vs_oracle(gs, oracle, dev, S, R, W, H, n):
    """The paper's first-order model through the HIP path (gs_project_pixvel_fwd/bwd in front of the unchanged
    binning + compositors): per-sub-pose radii / tile counts against the float32 oracle (integers), image and every
    gradient — Gaussians, mid-exposure viewmat, linear and angular velocity — against the float64 oracle."""
    O = oracle
    sc = O.synthetic_scene(n, W, H, seed=500 + S * 10 + R, scale_mult=6.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    et, rt, gamma, mlevel = 1 / 60, 1 / 30, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=R, exposure_time=et,
                         rolling_shutter_time=rt, gamma=gamma, min_rgb_level=mlevel, motion_model="pixel_velocity")
    q = {k: sc[k].double().requires_grad_(True) for k in names}
    ref, ref_alpha, ref_samples, frag, parts, _ = O.render(
        cfg, q["means"], q["log_scales"].exp(), q["quats"], torch.sigmoid(q["opacity_logits"]), q["sh"], q["viewmat"],
        q["lin_vel"], q["ang_vel"], background=bg.double(), return_parts=True)
    good = ~frag
    check_fragile(frag, f"pixvel S={S} R={R} {W}x{H} n={n}")
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)) * good[..., None]
    (ref * wt.double()).sum().backward()
    p = {k: sc[k].float().to(dev).requires_grad_(True) for k in names}
    times, _, _ = gs.subpose_schedule(S, et, R, rt)
    times_t = torch.tensor(times, device=dev)
    samples, alphas, radii = gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"],
                                                torch.sigmoid(p["opacity_logits"]), p["sh"], p["viewmat"], bg.to(dev),
                                                S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, sh_degree=3,
                                                lin_vel=p["lin_vel"], ang_vel=p["ang_vel"], times=times_t)
    out = gs.combine_samples(samples, gamma, mlevel)
    (out * wt.to(dev)).sum().backward()
    # integers: the float32 oracle re-centres the same float32 projection
    pr0 = O.project_gaussians(sc["means"], sc["log_scales"].exp(), 1.0, sc["quats"], sc["viewmat"], sc["fx"], sc["fy"],
                              sc["cx"], sc["cy"], H, W, keep_offscreen=True)
    pv = O.pixel_velocity(sc["means"], sc["viewmat"], sc["fx"], sc["fy"], sc["lin_vel"], sc["ang_vel"], 0.01, W, H)
    geom = (pr0.radii > 0).float()[:, None]
    for pidx, tau in enumerate(times):
        prp = O._recentre(pr0, (pr0.xys + torch.tensor(tau, dtype=torch.float32) * pv) * geom, H, W)
        assert np.array_equal(radii[pidx].cpu().numpy(), prp.radii.numpy()), pidx
    assert (samples.detach().cpu().double() - ref_samples)[:, good].abs().max().item() < IMG_ATOL
    assert (out.detach().cpu().double() - ref.detach())[good].abs().max().item() < 5e-4
    worst = {}
    for k in names:
        g_hip, g_ref = p[k].grad.cpu().numpy(), q[k].grad.numpy()
        if k == "viewmat":
            g_hip, g_ref = g_hip[:3], g_ref[:3]
        worst[k] = grad_el_ratio(g_hip, g_ref)
    print(f"pixel velocity S={S} R={R}: per-element gradient error / tolerance:", {k: round(v, 3) for k, v in worst.items()})
    for k, v in worst.items():
        assert v <= 1.0, (k, v)


@pytest.mark.parametrize("model,S,R", [("se3", 3, 2), ("pixel_velocity", 3, 2), ("pixel_velocity", 5, 1)])
def test_real_camera_pose_and_grazing_gaussians_vs_oracle(gs, oracle, dev, model, S, R):
    """round 3: every other oracle comparison renders from the IDENTITY pose, where a pose applied from the wrong side,
    a camera centre taken from the wrong column or a world-frame / camera-frame mix-up cannot show.  Here the survey
    scene is seen from a rotated and translated camera (world = R^T (camera - t), viewmat = [R | t]) and forty large
    opaque Gaussians sit just in front of the camera plane at grazing angles (z = 0.02..0.05, x/z up to 100): never
    on screen in a true sub-pose; in the pixel-velocity model they move with the Jacobian of the fov guard band's edge
    (gs_math.h pixel_velocity) and every re-centred box misses the image (with the unclamped Jacobian they were dragged
    across it at 1e5 px/s: 11 dB against the SE(3) frame, tools/rs_forward_check.py; round 3 culled everything whose
    centre is out of band).  Four more — large, a metre away, centre beside the image OUTSIDE the band, footprint well
    inside it (a floor, a wall) — must stay in both models; their velocity gradient passes the clamp
    to z.  Image and every gradient, viewmat and twist included, against the float64 oracle."""
    import math
    O = oracle
    W, H, n = 144, 96, 2500
    sc = O.synthetic_scene(n, W, H, seed=640 + S, scale_mult=6.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10

    def rot(ax, a):
        c, s_ = math.cos(a), math.sin(a)
        Rm = torch.eye(3)
        i, j = [(1, 2), (0, 2), (0, 1)][ax]
        Rm[i, i] = c; Rm[j, j] = c; Rm[i, j] = -s_; Rm[j, i] = s_
        return Rm
    Rm = rot(1, 0.7) @ rot(0, -0.4) @ rot(2, 1.1)
    t = torch.tensor([0.3, -0.2, 0.5])
    g = torch.Generator().manual_seed(5)
    k = 40
    pc = torch.stack([(torch.rand(k, generator=g) * 2 + 1) * torch.sign(torch.rand(k, generator=g) - 0.5),
                      (torch.rand(k, generator=g) * 2 - 1) * 2, 0.02 + 0.03 * torch.rand(k, generator=g)], dim=1)
    k2 = 4                                    # x/z = +-1.0, y/z = +-0.75: outside the band (0.8125 x 0.8125 here)
    wall = torch.tensor([[1.0, 0.1, 1.0], [-1.0, -0.2, 1.0], [0.15, 0.75, 1.0], [-0.1, -0.75, 1.0]]) * \
        torch.tensor([1.0, 1.3, 0.9, 1.2])[:, None]
    cam_pts = torch.cat([sc["means"], pc, wall])
    sc["means"] = (cam_pts - t) @ Rm
    sc["log_scales"] = torch.cat([sc["log_scales"], torch.full((k, 3), -2.5), torch.full((k2, 3), -1.0)])
    sc["quats"] = torch.cat([sc["quats"], sc["quats"][:k + k2]])
    sc["opacity_logits"] = torch.cat([sc["opacity_logits"], torch.full((k,), 4.0), torch.full((k2,), -0.5)])
    sc["sh"] = torch.cat([sc["sh"], sc["sh"][:k + k2] + 0.5])
    V = torch.eye(4)
    V[:3, :3], V[:3, 3] = Rm, t
    sc["viewmat"] = V
    et, rt, gamma, mlevel = 1 / 60, 1 / 30, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=R, exposure_time=et,
                         rolling_shutter_time=rt, gamma=gamma, min_rgb_level=mlevel, motion_model=model)
    q = {k_: sc[k_].double().requires_grad_(True) for k_ in names}
    ref, ref_alpha, ref_samples, frag, parts, _ = O.render(
        cfg, q["means"], q["log_scales"].exp(), q["quats"], torch.sigmoid(q["opacity_logits"]), q["sh"], q["viewmat"],
        q["lin_vel"], q["ang_vel"], background=bg.double(), return_parts=True)
    good = ~frag
    check_fragile(frag, f"posed {model} S={S} R={R}")
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)) * good[..., None]
    (ref * wt.double()).sum().backward()
    p = {k_: sc[k_].float().to(dev).requires_grad_(True) for k_ in names}
    times, _, _ = gs.subpose_schedule(S, et, R, rt)
    times_t = torch.tensor(times, device=dev)
    common = (p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]), p["sh"])
    if model == "se3":
        vms = gs.subpose_viewmats(p["viewmat"], p["lin_vel"], p["ang_vel"], times_t)
        samples, alphas, radii = gs.render_subposes(*common, vms, bg.to(dev), S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"],
                                                    H, W, sh_degree=3)
    else:
        samples, alphas, radii = gs.render_subposes(*common, p["viewmat"], bg.to(dev), S, R, sc["fx"], sc["fy"], sc["cx"],
                                                    sc["cy"], H, W, sh_degree=3, lin_vel=p["lin_vel"], ang_vel=p["ang_vel"],
                                                    times=times_t)
        assert int(radii[:, n:n + k].abs().sum()) == 0           # the grazing Gaussians: no sub-pose reaches the image
    assert int((radii[:, n + k:] > 0).sum()) == radii.shape[0] * k2, radii[:, n + k:]     # the walls: every sub-pose
    out = gs.combine_samples(samples, gamma, mlevel)
    (out * wt.to(dev)).sum().backward()
    assert (samples.detach().cpu().double() - ref_samples)[:, good].abs().max().item() < IMG_ATOL
    assert (out.detach().cpu().double() - ref.detach())[good].abs().max().item() < 5e-4
    worst = {}
    for k_ in names:
        g_hip, g_ref = p[k_].grad.cpu().numpy(), q[k_].grad.numpy()
        if k_ == "viewmat":
            g_hip, g_ref = g_hip[:3], g_ref[:3]
        worst[k_] = grad_el_ratio(g_hip, g_ref)
    print(f"real pose, {model} S={S} R={R}: per-element gradient error / tolerance:", {k_: round(v, 3) for k_, v in worst.items()})
    for k_, v in worst.items():
        assert v <= 1.0, (k_, v)
    assert float(p["viewmat"].grad.abs().max()) > 0 and float(p["ang_vel"].grad.abs().max()) > 0


def test_upstream_gradient_convention_switches(gs, oracle, dev):
    """DESIGN.md §1.2, VERDICT round 4 item 2: the reference's three gradient conventions (recollected from gsplat
    0.1.11; ops.UPSTREAM_GRADS bit mask, default 6 = all but the fov rule) against the ORACLE's implementation of each
    (gs_oracle.UP_*, straight-through rules), on a scene built so that every one of them matters: a tenth of the
    Gaussians beyond the 1.3 tan(fov/2) guard band, non-unit quaternions, opacities of ~1 (alpha reaches the 0.999 clamp;
    no antialiasing compensation).  For every mask in {0, 1, 4, 5, 6, 7}: same image, every gradient of the FUSED path per
    element against the float64 oracle in that mode, and the modes really differ from each other.  Bit 2 (raw
    quaternion gradient) exists on the compat op only: there against the oracle's UP_QUAT_RAW; the fused path ignores it."""
    from gsdeblur_amd import ops
    O = oracle
    W, H, n = 96, 64, 400
    sc = O.synthetic_scene(n, W, H, seed=9, scale_mult=6.0)
    means = sc["means"].clone()
    means[:40, 0] = means[:40, 0].sign() * means[:40, 2] * (1.45 * 0.5 * W / sc["fx"])    # x/z just beyond 1.3 tan(fov/2) ...
    sc["log_scales"] = sc["log_scales"].clone()
    sc["log_scales"][:40] += 1.2                          # ... and large enough to reach back into the image
    quats = sc["quats"] * (0.5 + torch.rand(n, 1, generator=torch.Generator().manual_seed(1)))    # NOT unit
    op_logit = sc["opacity_logits"].clone()
    op_logit[40:80] = 12.0                                # opacity ~ 1: alpha clamps at 0.999 near the centre
    V = sc["viewmat"].to(dev)
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(2))
    cfg0 = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], antialiased=False, upstream_grads=0)
    with torch.no_grad():
        _, _, _, frag, _, _ = O.render(cfg0, means.double(), sc["log_scales"].double().exp(), quats.double(),
                                       torch.sigmoid(op_logit.double()), sc["sh"].double(), sc["viewmat"].double(),
                                       sc["lin_vel"].double() * 0, sc["ang_vel"].double() * 0, return_parts=True)
    wt = wt * (~frag)[..., None]

    def run(flags):
        with grad_convention(flags):
            p = {"means": means.to(dev).requires_grad_(True), "log_scales": sc["log_scales"].to(dev).requires_grad_(True),
                 "quats": quats.to(dev).requires_grad_(True), "op": op_logit.to(dev).requires_grad_(True),
                 "sh": sc["sh"].to(dev).requires_grad_(True)}
            s_, _, _ = gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["op"]), p["sh"],
                                          V[None], None, 1, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
                                          antialiased=False)
            (s_[0] * wt.to(dev)).sum().backward()
            return s_.detach(), {k: v.grad.detach().cpu().double() for k, v in p.items()}

    def run_oracle(flags):
        import dataclasses
        q = {"means": means.double().requires_grad_(True), "log_scales": sc["log_scales"].double().requires_grad_(True),
             "quats": quats.double().requires_grad_(True), "op": op_logit.double().requires_grad_(True),
             "sh": sc["sh"].double().requires_grad_(True)}
        out, _ = O.render(dataclasses.replace(cfg0, upstream_grads=flags), q["means"], q["log_scales"].exp(), q["quats"],
                          torch.sigmoid(q["op"]), q["sh"], sc["viewmat"].double(), sc["lin_vel"].double() * 0,
                          sc["ang_vel"].double() * 0)
        (out * wt.double()).sum().backward()
        return {k: v.grad.clone() for k, v in q.items()}

    hip, ref = {}, {}
    for flags in (0, 1, 4, 5, 6, 7):
        img, hip[flags] = run(flags)
        ref[flags] = run_oracle(flags)
        if flags:
            assert torch.equal(img, img0)                   # forward untouched by any switch
        else:
            img0 = img
        worst = {k: round(grad_el_ratio(hip[flags][k].numpy(), ref[flags][k].numpy()), 3) for k in hip[flags]}
        print(f"gradient conventions, mask {flags}: per-element error / tolerance vs the oracle in that mode:", worst)
        for k, v in worst.items():
            assert v <= 1.0, (flags, k, v)
    # the modes differ where they should (otherwise the comparisons above would not tell them apart) ...
    d_fov_ref = (ref[1]["means"] - ref[0]["means"]).abs().max() / ref[0]["means"].abs().max()
    d_fov_hip = (hip[1]["means"] - hip[0]["means"]).abs().max() / hip[0]["means"].abs().max()
    # the alpha clamp is only reached at the few pixels at an opaque splat's very centre: it moves those splats' centre
    # gradients by ~3e-4 of the tensor's max (27x the comparison's per-element floor) and their logit gradients by ~1 % of
    # THEIR max — which is 1e-8 of the tensor's max (sigmoid' is 6e-6 at a logit of 12, and 1 - sigmoid is an fp32
    # cancellation there), so the centres carry the evidence
    d_alpha_ref = (ref[4]["means"] - ref[0]["means"])[40:80].abs().max() / ref[0]["means"].abs().max()
    d_alpha_hip = (hip[4]["means"] - hip[0]["means"])[40:80].abs().max() / hip[0]["means"].abs().max()
    d_op_ref = (ref[4]["op"] - ref[0]["op"])[40:80].abs().max() / ref[0]["op"][40:80].abs().max()
    print(f"what the conventions change (relative to the tensor's max): fov clamp {float(d_fov_ref):.2e} (oracle) "
          f"{float(d_fov_hip):.2e} (HIP); alpha clamp {float(d_alpha_ref):.2e} (oracle) {float(d_alpha_hip):.2e} (HIP), "
          f"opaque splats' logits {float(d_op_ref):.2e} of their own max")
    # well above the comparisons' per-element tolerance (1e-5 of the max) — the modes are told apart
    assert d_fov_ref > 1e-4 and d_fov_hip > 1e-4 and d_alpha_ref > 1e-4 and d_alpha_hip > 1e-4 and d_op_ref > 3e-3
    # ... and only there: Gaussians inside the guard band / opacities away from the clamp are untouched bit for bit
    xz = (means[:, 0] / means[:, 2]).abs()
    yz = (means[:, 1] / means[:, 2]).abs()
    inside = (xz < 1.25 * 0.5 * W / sc["fx"]) & (yz < 1.25 * 0.5 * H / sc["fy"])
    assert torch.equal(hip[1]["means"][inside], hip[0]["means"][inside])
    assert (hip[4]["op"] - hip[0]["op"])[200:].abs().max().item() < 1e-6 * (hip[0]["op"].abs().max().item() + 1e-12)
    for k in hip[7]:
        assert torch.equal(hip[7][k], hip[5][k]), k        # bit 2 does not exist on the fused path
        assert torch.equal(hip[6][k], hip[4][k]), k        # (the default, 6, is the alpha rule alone there)
    # compat op: raw quaternion gradient (bit 2) and straight-through fov clamp (bit 1) against the oracle's modes
    wc = torch.rand(n, 3, generator=torch.Generator().manual_seed(3))
    wx = torch.rand(n, 2, generator=torch.Generator().manual_seed(4))
    for flags in (0, 1, 2, 3):
        with grad_convention(flags):
            pm, pq = means.to(dev).requires_grad_(True), quats.to(dev).requires_grad_(True)
            ps = sc["log_scales"].exp().to(dev).requires_grad_(True)
            xys, _, radii, conics, _, _, _ = gs.project_gaussians(pm, ps, 1.0, pq, V, sc["fx"], sc["fy"], sc["cx"], sc["cy"],
                                                                  H, W)
            ((conics * wc.to(dev)).sum() + (xys * wx.to(dev)).sum()).backward()
        qm, qq = means.double().requires_grad_(True), quats.double().requires_grad_(True)
        qs = sc["log_scales"].double().exp().requires_grad_(True)
        pr = O.project_gaussians(qm, qs, 1.0, qq, sc["viewmat"].double(), sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
                                 upstream=flags)
        ((pr.conics * wc.double()).sum() + (pr.xys * wx.double()).sum()).backward()
        worst = {"means": grad_el_ratio(pm.grad.cpu().numpy(), qm.grad.numpy()),
                 "scales": grad_el_ratio(ps.grad.cpu().numpy(), qs.grad.numpy()),
                 "quats": grad_el_ratio(pq.grad.cpu().numpy(), qq.grad.numpy())}
        print(f"project_gaussians, mask {flags}: per-element error / tolerance:", {k: round(v, 3) for k, v in worst.items()})
        for k, v in worst.items():
            assert v <= 1.0, (flags, k, v)
        if flags == 2:
            g = qq.grad
            qn = quats.double() / quats.double().norm(dim=1, keepdim=True)
            assert (qn * g).sum(1).abs().max() > 1e-3 * g.abs().max()     # raw: the radial component survives


@pytest.mark.parametrize("S,R,base", [(1, 1, 4), (3, 2, 16), (2, 1, 1)])
def test_depth_sliced_equals_single_pass(gs, oracle, dev, S, R, base):
    """Depth slicing only removes intersections the compositor would never reach: the forward image is
    BIT-IDENTICAL to the single-pass path and the gradients agree to atomic-reordering noise."""
    from gsdeblur_amd import ops
    O = oracle
    W, H, n = 208, 144, 6000
    sc = O.synthetic_scene(n, W, H, seed=77, scale_mult=7.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    bg = torch.tensor([0.2, 0.1, 0.4])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(9))
    res = {}
    old = ops.SLICE_BASE
    try:
        for mode, sb in (("single", 0), ("sliced", base)):
            ops.SLICE_BASE = sb
            out, alpha, samples, vms, p, radii = _run_full(gs, O, dev, sc, H, W, S, R, 1 / 60, 1 / 30, 2.2, 10.0, 3,
                                                           bg, wt)
            res[mode] = (samples.detach().clone(), alpha.detach().clone(),
                         {k: v.grad.detach().clone() for k, v in p.items()})
            if sb:
                assert len(ops.last_slice_intersects) >= 3                   # really multi-slice
                assert sum(ops.last_slice_intersects) <= ops.last_num_intersects
    finally:
        ops.SLICE_BASE = old
    assert torch.equal(res["single"][0], res["sliced"][0])
    assert torch.equal(res["single"][1], res["sliced"][1])
    for k in res["single"][2]:
        assert rel_max(res["sliced"][2][k].cpu(), res["single"][2][k].cpu()) < 1e-4, k


@pytest.mark.parametrize("knob,value", [("DEVICE_SIZES", 0), ("DEPTH_SORT_COMPACT", 0), ("DEPTH_SORT_SEGMENTED", 0),
                                        ("TILE_SORT_CARRY", 0), ("PREALLOC_BWD", 0), ("HIT_MASKS", 0),
                                        ("DEFER_COLOR", 0), ("DEPTH_SORT_DIGIT", 11)])
@pytest.mark.parametrize("S,R,base", [(2, 2, 8), (3, 1, 512)])
def test_every_runtime_knob_gives_the_default_result(gs, oracle, dev, knob, value, S, R, base):
    """the A/B switches of ops.py (environment variables GSD_*) select alternative routes through the same pipeline —
    read-backs instead of device-side counts, the full instead of the compacting depth pre-sort, the 64-bit sort
    route, the gathered instead of the carried record index, backward-side allocation, emission without hit masks,
    SH colour in the projection: every one of them must give the default path's images and gradients, on a
    multi-slice rolling-shutter frame and on a single-slice one"""
    from gsdeblur_amd import ops
    O = oracle
    W, H, n = 176, 144, 5000
    sc = O.synthetic_scene(n, W, H, seed=31 + S, scale_mult=7.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    bg = torch.tensor([0.1, 0.3, 0.2])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(4))
    res = {}
    old = (ops.SLICE_BASE, getattr(ops, knob))
    try:
        ops.SLICE_BASE = base
        for v in (old[1], value):
            setattr(ops, knob, v)
            out, alpha, samples, vms, p, radii = _run_full(gs, O, dev, sc, H, W, S, R, 1 / 60, 1 / 30, 2.2, 10.0, 3, bg, wt)
            res[v] = (samples.detach().clone(), alpha.detach().clone(), {k: g.grad.detach().clone() for k, g in p.items()})
    finally:
        ops.SLICE_BASE = old[0]
        setattr(ops, knob, old[1])
    a, b = res[old[1]], res[value]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        # images are bit-identical; gradients agree up to fp32 summation order (the tuple reduce picks its segment
        # strategy by the buffer size it is given — a capacity on the default path, the exact count with
        # DEVICE_SIZES=0 —, SH colour evaluated in the projection rounds differently, pose / velocity gradients end
        # in fp32 atomics)
        tol = 1e-4 if (knob == "DEFER_COLOR" or k in ("viewmat", "lin_vel", "ang_vel")) else 2e-5
        assert rel_max(a[2][k].cpu(), b[2][k].cpu()) < tol, k


@pytest.mark.parametrize("S,R,base", [(2, 1, 4), (3, 2, 16), (1, 1, 512)])
def test_speculative_slices_change_nothing(gs, oracle, dev, S, R, base):
    """GSD_SPECULATE=1: slices after the first are launched behind a device-side gate (the previous compositor's
    "a tile is still open" word) instead of after a host read-back, and the backward drops the ones whose gate was
    closed: images and gradients identical to the default path, whether the frame needs all its planned slices (small
    base), a few of them, or one"""
    from gsdeblur_amd import ops
    O = oracle
    W, H, n = 192, 128, 6000
    sc = O.synthetic_scene(n, W, H, seed=77 + S, scale_mult=7.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    bg = torch.tensor([0.3, 0.2, 0.1])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(3))
    res = {}
    old = (ops.SLICE_BASE, ops.SPECULATE, ops.SLICE_MERGE, ops.BAND_AWARE)
    try:
        ops.SLICE_BASE, ops.SLICE_MERGE = base, 0.0         # speculation runs the planned slices one by one
        ops.BAND_AWARE = 0          # (the twin's projection keys every (band, Gaussian) pair: same slice plan on both sides)
        for spec in (0, 1):
            ops.SPECULATE = spec
            out, alpha, samples, vms, p, radii = _run_full(gs, O, dev, sc, H, W, S, R, 1 / 60, 1 / 30, 2.2, 10.0, 3, bg, wt)
            res[spec] = (samples.detach().clone(), alpha.detach().clone(), {k: v.grad.detach().clone() for k, v in p.items()},
                         list(ops.last_slice_intersects))
    finally:
        ops.SLICE_BASE, ops.SPECULATE, ops.SLICE_MERGE, ops.BAND_AWARE = old
    a, b = res[0], res[1]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert [t for t in a[3] if t] == [t for t in b[3] if t]            # the same non-empty slices
    for k in a[2]:
        if k in ("viewmat", "lin_vel", "ang_vel"):                      # a dozen fp32 atomics per block
            assert rel_max(a[2][k].cpu(), b[2][k].cpu()) < 1e-4, k
        else:
            assert torch.equal(a[2][k], b[2][k]), k


@pytest.mark.parametrize("S,R,base,W,H,n,hot", [(2, 2, 8, 208, 144, 6000, False), (1, 1, 0, 131, 77, 900, False),
                                                 (3, 1, 512, 320, 200, 20000, False), (2, 1, 8, 176, 112, 5000, True),
                                                 (1, 2, 512, 131, 90, 2500, True)])
def test_scalar_cache_compositor_equals_readlane_compositor(gs, oracle, dev, S, R, base, W, H, n, hot):
    """The scalar-cache compositors (records through s_load; round 4: one-compare validity on a shifted exponent, stop
    handling in a rarely taken branch, final_idx = stop position) against the round-1 kernels (records broadcast with
    v_readlane, alpha / sigma / T tests as three compares, final_idx = last blended entry + 1).  Two formulations of
    the same blend in fp32: sample images and alphas agree within IMG_ATOL except threshold pixels (images_close),
    gradients up to fp32 summation order and those pixels.  Single- and multi-slice frames, ragged image sizes,
    rolling-shutter bands, tuple and atomics accumulation.  hot: a few large Gaussians have an opacity above the 0.999
    alpha clamp — the tiles they land on take the clamping loop version, the others the clamp-free one (tile_hot); the
    round-1 kernels always clamp."""
    from gsdeblur_amd import ops
    O = oracle
    sc = O.synthetic_scene(n, W, H, seed=21 + S, scale_mult=7.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    if hot:
        sc["opacity_logits"], sc["log_scales"] = sc["opacity_logits"].clone(), sc["log_scales"].clone()
        sc["opacity_logits"][::250] = 14.0
        sc["log_scales"][::250] += 1.0        # large enough for the anti-aliasing compensation to stay above 0.999
        pr = O.project_gaussians(sc["means"], sc["log_scales"].exp(), 1.0, sc["quats"], sc["viewmat"], sc["fx"],
                                 sc["fy"], sc["cx"], sc["cy"], H, W)
        n_hot = int(((torch.sigmoid(sc["opacity_logits"]) * pr.compensation > 0.999) & (pr.radii > 0)).sum())
        assert n_hot >= 1, n_hot
    bg = torch.tensor([0.2, 0.1, 0.4])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(9))
    res = {}
    old = (ops.SLICE_BASE, ops.RASTER_FWD_VARIANT, ops.RASTER_BWD_VARIANT, ops.GRAD_TUPLES)
    try:
        ops.SLICE_BASE = base
        for tuples in (1, 0):
            ops.GRAD_TUPLES = tuples        # 0: the fp32-atomics backward (order-dependent sums)
            for fv, bv in ((2, 2), (0, 0)):
                ops.RASTER_FWD_VARIANT, ops.RASTER_BWD_VARIANT = fv, bv
                out, alpha, samples, vms, p, radii = _run_full(gs, O, dev, sc, H, W, S, R, 1 / 60, 1 / 30, 2.2, 10.0,
                                                               3, bg, wt)
                res[(tuples, fv, bv)] = (samples.detach().clone(), alpha.detach().clone(),
                                         {k: v.grad.detach().clone() for k, v in p.items()})
    finally:
        ops.SLICE_BASE, ops.RASTER_FWD_VARIANT, ops.RASTER_BWD_VARIANT, ops.GRAD_TUPLES = old
    for tuples in (1, 0):
        b, c = res[(tuples, 2, 2)], res[(tuples, 0, 0)]
        # small frames: one threshold pixel of 30k is 3e-5
        images_close(c[0], b[0], IMG_ATOL, f"samples tuples={tuples}", frac_max=1e-4)
        images_close(c[1], b[1], IMG_ATOL, f"alpha tuples={tuples}", frac_max=1e-4)
        for k in b[2]:
            r = rel_max(c[2][k].cpu(), b[2][k].cpu())
            print(f"[kernel-vs-kernel grad {k} tuples={tuples}] rel_max = {r:.3e}")
            assert r < 5e-5, (k, r)              # observed: <= 5.6e-6 (visit r4_v2)
    # the two accumulation routes behind the SAME kernels: fp32 summation order only
    for k in res[(1, 0, 0)][2]:
        assert rel_max(res[(0, 0, 0)][2][k].cpu(), res[(1, 0, 0)][2][k].cpu()) < 1e-4, k
    assert torch.equal(res[(0, 0, 0)][0], res[(1, 0, 0)][0])


def test_tuple_backward_equals_atomic_backward(gs, oracle, dev):
    """The atomic-free backward (per-entry gradient tuples + segmented reduce) and the fp32-atomics backward
    compute the same sums; only the summation order differs."""
    from gsdeblur_amd import ops
    O = oracle
    W, H, n = 200, 152, 7000
    sc = O.synthetic_scene(n, W, H, seed=55, scale_mult=7.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    bg = torch.tensor([0.1, 0.3, 0.2])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(6))
    res = {}
    old = (ops.GRAD_TUPLES, ops.SLICE_BASE)
    try:
        for tup in (0, 1):
            ops.GRAD_TUPLES, ops.SLICE_BASE = tup, 8          # several slices, some with holes
            out, alpha, samples, vms, p, radii = _run_full(gs, O, dev, sc, H, W, 2, 2, 1 / 60, 1 / 30, 2.2, 10.0, 3, bg, wt)
            res[tup] = (samples.detach().clone(), {k: v.grad.detach().clone() for k, v in p.items()})
    finally:
        ops.GRAD_TUPLES, ops.SLICE_BASE = old
    assert torch.equal(res[0][0], res[1][0])
    for k in res[0][1]:
        assert rel_max(res[1][1][k].cpu(), res[0][1][k].cpu()) < 1e-4, k


def test_deferred_colour_and_compact_emission_change_nothing(gs, oracle, dev):
    """Deferred SH colouring (only emitted Gaussians are coloured) and compact emission (exact hit counts) are
    pure work-avoidance: images bit-identical, gradients equal up to summation order."""
    from gsdeblur_amd import ops
    O = oracle
    W, H, n = 216, 168, 7000
    sc = O.synthetic_scene(n, W, H, seed=321, scale_mult=6.5)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    bg = torch.tensor([0.15, 0.25, 0.05])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(8))
    res = {}
    old = (ops.DEFER_COLOR, ops.COMPACT_EMIT, ops.SLICE_BASE)
    try:
        for mode in ((0, 0), (1, 1), (1, 0), (0, 1)):
            ops.DEFER_COLOR, ops.COMPACT_EMIT = mode
            ops.SLICE_BASE = 16
            out, alpha, samples, vms, p, radii = _run_full(gs, O, dev, sc, H, W, 2, 2, 1 / 60, 1 / 30, 2.2, 10.0, 3, bg, wt)
            res[mode] = (samples.detach().clone(), {k: v.grad.detach().clone() for k, v in p.items()})
    finally:
        ops.DEFER_COLOR, ops.COMPACT_EMIT, ops.SLICE_BASE = old
    for mode in ((1, 1), (1, 0), (0, 1)):
        assert torch.equal(res[(0, 0)][0], res[mode][0]), mode
        for k in res[mode][1]:
            assert rel_max(res[mode][1][k].cpu(), res[(0, 0)][1][k].cpu()) < 1e-4, (mode, k)


def test_exact_tile_culling_changes_nothing(gs, oracle, dev):
    """Culling (Gaussian, tile) pairs whose pixel rectangle lies outside the alpha >= 1/255 ellipse must
    leave the image BIT-IDENTICAL (the pairs contributed exactly nothing) and the gradients equal up to
    atomic reordering."""
    from gsdeblur_amd import ops
    O = oracle
    W, H, n = 240, 176, 8000
    sc = O.synthetic_scene(n, W, H, seed=123, scale_mult=6.0)
    sc["log_scales"] = sc["log_scales"] + torch.tensor([0.9, -0.9, 0.0])        # needle-like Gaussians
    bg = torch.tensor([0.3, 0.2, 0.1])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(4))
    res = {}
    old = ops.EXACT_TILE_CULL
    try:
        for cull in (0, 1):
            ops.EXACT_TILE_CULL = cull
            out, alpha, samples, vms, p, radii = _run_full(gs, O, dev, sc, H, W, 2, 2, 1 / 60, 1 / 30, 2.2, 10.0, 3, bg, wt)
            res[cull] = (samples.detach().clone(), alpha.detach().clone(),
                         {k: v.grad.detach().clone() for k, v in p.items()})
    finally:
        ops.EXACT_TILE_CULL = old
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert rel_max(res[1][2][k].cpu(), res[0][2][k].cpu()) < 1e-4, k


# --------------------------------------------------------------------------- #
# size-independent properties at larger sizes (no oracle run needed)
# --------------------------------------------------------------------------- #
def test_properties_at_scale(gs, oracle, dev):
    """300k Gaussians, 1080p (BASELINE.json config 2 shape): sortedness of the tile keys, bin edges
    partition the list, depth order inside tiles, zero-velocity S-sample average == static render,
    gradient linearity."""
    O = oracle
    n, W, H = 300_000, 1920, 1080
    sc = to_dev(O.synthetic_scene(n, W, H, seed=1234), dev)
    from gsdeblur_amd import ops
    L = gs._lib.load()
    S, R = 2, 1
    sh = sc["sh"]
    vm1 = sc["viewmat"][None].contiguous()
    scales, opac = sc["log_scales"].exp(), torch.sigmoid(sc["opacity_logits"])
    s1, a1, rad1 = gs.render_subposes(sc["means"], scales, sc["quats"], opac, sh, vm1, None, 1, 1, sc["fx"],
                                      sc["fy"], sc["cx"], sc["cy"], H, W)
    vm2 = sc["viewmat"][None].repeat(2, 1, 1).contiguous()
    s2, a2, _ = gs.render_subposes(sc["means"], scales, sc["quats"], opac, sh, vm2, None, 2, 1, sc["fx"], sc["fy"],
                                   sc["cx"], sc["cy"], H, W)
    assert torch.equal(s2[0], s1[0]) and torch.equal(s2[1], s1[0])           # identical poses -> identical samples
    assert torch.allclose(gs.combine_samples(s2, 2.2, 10.0), gs.combine_samples(s1, 2.2, 10.0), atol=2e-6)
    assert torch.isfinite(s1).all() and s1.min() >= 0
    # binning invariants
    N = n
    rec = torch.empty(N, ops.REC, device=dev)
    dk = torch.empty(N, dtype=torch.int32, device=dev)
    nt = torch.empty(N, dtype=torch.int32, device=dev)
    gs._lib.check(L.gs_project_fused_fwd(N, 1, ops._ptr(sc["means"]), ops._ptr(scales.contiguous()), 1.0,
                                         ops._ptr(sc["quats"]), ops._ptr(opac.contiguous()), ops._ptr(sh), 16, 3,
                                         ops._ptr(vm1), sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, 0.01, 1, 0,
                                         ops._ptr(rec), ops._ptr(dk), ops._ptr(nt), None, None, 0, ops._stream()), "fused")
    svals, bins, I, skeys = gs.bin_and_sort_records(rec, dk, nt, 1, N, H, W)
    assert I == int(nt.long().sum().item())
    sk = skeys.long()
    assert bool((sk[1:] >= sk[:-1]).all())                                   # sortedness
    T = 120 * 68
    b = bins.long()
    assert int((b[:, 1] - b[:, 0]).sum().item()) == I                       # bins partition the list
    nonempty = b[:, 1] > b[:, 0]
    assert bool((sk[b[nonempty, 0]] == torch.arange(T, device=dev)[nonempty]).all())
    depth_sorted = rec[svals.long(), 9]
    same_tile = sk[1:] == sk[:-1]
    assert bool((depth_sorted[1:][same_tile] >= depth_sorted[:-1][same_tile]).all())   # front-to-back inside tiles
    # gradient linearity: d(2*loss) == 2*d(loss)
    m = sc["means"].clone().requires_grad_(True)
    w = torch.rand(1, H, W, 3, device=dev)
    g = []
    for scale in (1.0, 2.0):
        if m.grad is not None:
            m.grad = None
        s, _, _ = gs.render_subposes(m, scales, sc["quats"], opac, sh, vm1, None, 1, 1, sc["fx"], sc["fy"],
                                     sc["cx"], sc["cy"], H, W)
        (scale * (s * w).sum()).backward()
        g.append(m.grad.clone())
    assert torch.isfinite(g[0]).all()
    assert rel_max(g[1].cpu(), (2 * g[0]).cpu()) < 2e-3      # atomics reorder fp32 sums run to run


def test_model_get_outputs(gs, oracle, dev):
    """Model surface: dict keys / shapes / ranges that render_model.py:217-219 relies on, and backward."""
    O = oracle
    W, H, n = 160, 120, 4000
    sc = O.synthetic_scene(n, W, H, seed=31, scale_mult=6.0)
    cfg = gs.SplatfactoDeblurConfig(blur_samples=3, rs_bands=4, gamma=2.2, min_rgb_level=10.0,
                                    background_color="auto")
    cfg.camera_optimizer.mode = "SO3xR3"
    cfg.camera_velocity_optimizer.enabled = True
    model = gs.SplatfactoDeblurModel.from_scene(cfg, sc, dev, num_cameras=3)
    c2w = torch.eye(4)[:3].clone()
    c2w[:, 1] *= -1
    c2w[:, 2] *= -1                        # OpenGL camera looking down the oracle scene's +z
    cam = gs.Camera(c2w, sc["fx"], sc["fy"], sc["cx"], sc["cy"], W, H,
                    metadata=dict(cam_idx=1, camera_linear_velocity=[0.5, 0.1, 0.0],
                                  camera_angular_velocity=[0.0, 0.3, 0.1], exposure_time=1 / 60,
                                  rolling_shutter_time=1 / 30))
    model.train()
    out = model.get_outputs(cam)
    assert out["rgb"].shape == (H, W, 3) and out["accumulation"].shape == (H, W, 1)
    assert model.radii.shape == (12, n)
    out["rgb"].mean().backward()
    for name, prm in model.gauss_params().items():
        assert prm.grad is not None and torch.isfinite(prm.grad).all(), name
    assert model.pose_adjustment.grad[1].abs().sum() > 0 and model.pose_adjustment.grad[0].abs().sum() == 0
    assert model.velocity_adjustment.grad[1].abs().sum() > 0
    assert model.background_param.grad is not None
    ev = model.get_outputs_for_camera(cam)
    assert ev["depth"].shape == (H, W, 1) and torch.isfinite(ev["depth"]).all()
    assert 0.0 <= ev["rgb"].min().item() and ev["rgb"].max().item() <= 1.0
    assert (ev["depth"][ev["accumulation"] > 0.5] > 0.9).all()


@pytest.mark.parametrize("S,R,base", [(1, 1, 512), (3, 2, 8)])
def test_depth_channel_of_the_sliced_path_matches_oracle(gs, oracle, dev, S, R, base):
    """outputs["depth"] (/root/reference/render_model.py:219) comes out of the SAME depth-sliced pass as the colour:
    a fourth, forward-only channel sum(weight * camera-space depth).  Against the float64 oracle compositing the
    per-Gaussian depth as a colour, single- and multi-slice, with rolling-shutter bands."""
    from gsdeblur_amd import ops
    O = oracle
    W, H, n = 176, 112, 4000
    sc = O.synthetic_scene(n, W, H, seed=41, scale_mult=6.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    et, rt = 1 / 60, 1 / 30
    times, samp, band = gs.subpose_schedule(S, et, R, rt)
    p = {k: sc[k].to(dev) for k in ("means", "log_scales", "quats", "opacity_logits", "sh", "viewmat", "lin_vel", "ang_vel")}
    vms = gs.subpose_viewmats(p["viewmat"], p["lin_vel"], p["ang_vel"], torch.tensor(times, device=dev))
    old = ops.SLICE_BASE
    try:
        ops.SLICE_BASE = base
        samples, alphas, radii, depth_acc = gs.render_subposes(
            p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]), p["sh"], vms, None, S, R,
            sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, return_depth=True)
        if base < 512:
            assert sum(1 for x in ops.last_slice_intersects if x > 0) >= 2
    finally:
        ops.SLICE_BASE = old
    rows = O.band_tile_rows(H, R)
    ref = torch.zeros(S, H, W, dtype=torch.float64)
    frag = torch.zeros(H, W, dtype=torch.bool)
    vms64 = vms.cpu().double()
    for pi in range(S * R):
        pr = O.project_gaussians(sc["means"].double(), sc["log_scales"].double().exp(), 1.0, sc["quats"].double(),
                                 vms64[pi], sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W)
        op = torch.sigmoid(sc["opacity_logits"].double()) * pr.compensation
        keys, gids = O.sort_intersects(*O.map_gaussian_to_intersects(pr, W))
        bins = O.get_tile_bin_edges(keys, ((W + 15) // 16) * ((H + 15) // 16))
        r = O.rasterize_sorted(pr.xys, pr.conics, pr.depths[:, None].repeat(1, 3), op, gids, bins, H, W, None,
                               tile_rows=rows[band[pi]])
        ref[samp[pi]] += r.img[..., 0]
        frag |= r.fragile
    good = ~frag
    err = (depth_acc.cpu().double() - ref)[:, good].abs().max().item()
    assert err < 2e-4 * ref.max().item(), err


@pytest.mark.parametrize("C", [1, 2, 5])
def test_rasterize_gaussians_other_channel_counts(gs, oracle, dev, C):
    """upstream's nd_rasterize surface: colours with C != 3 channels (depth, features) composite with exactly the
    RGB path's weights, forward and backward, three channels per pass"""
    O = oracle
    n, W, H = 1500, 96, 80
    sc = _scene(O, n, W, H, 13, 8.0, dev)
    pr = O.project_gaussians(sc["means"], sc["log_scales"].exp(), 1.0, sc["quats"], torch.eye(4), sc["fx"], sc["fy"],
                             sc["cx"], sc["cy"], H, W)
    g = torch.Generator().manual_seed(C)
    colors = torch.rand(n, C, generator=g)
    opac = torch.sigmoid(sc["opacity_logits"])
    bg = torch.rand(C, generator=g)
    wt = torch.rand(H, W, C, generator=g)
    xd, cd, cold, od = (t.to(dev).requires_grad_(True) for t in (pr.xys, pr.conics, colors, opac))
    img, alpha = gs.rasterize_gaussians(xd, pr.depths.to(dev), pr.radii.to(dev), cd, pr.num_tiles_hit.to(dev), cold,
                                        od[:, None], H, W, 16, bg.to(dev), return_alpha=True)
    assert img.shape == (H, W, C)
    x64, c64, col64, o64 = (t.double().requires_grad_(True) for t in (pr.xys, pr.conics, colors, opac))
    ref = []
    frag = torch.zeros(H, W, dtype=torch.bool)
    for c in range(C):
        i3, r = O.rasterize_gaussians(x64, pr.depths, pr.radii, c64, pr.num_tiles_hit, col64[:, c:c + 1].repeat(1, 3),
                                      o64, H, W, background=bg[c].double().repeat(3), proj=pr)
        ref.append(i3[..., 0])
        frag |= r.fragile
    ref = torch.stack(ref, dim=-1)
    good = ~frag
    assert (img.detach().cpu().double() - ref.detach())[good].abs().max().item() < IMG_ATOL
    wt = wt * good[..., None]
    (img * wt.to(dev)).sum().backward()
    (ref * wt.double()).sum().backward()
    for got, want, name in ((xd.grad, x64.grad, "xys"), (cd.grad, c64.grad, "conics"), (cold.grad, col64.grad, "colors"),
                            (od.grad, o64.grad, "opacity")):
        assert rel_max(got.cpu(), want) < 1e-4, name


# --------------------------------------------------------------------------- #
# densification statistic out of the HIP projection backward, and the refine step in a loop
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("S,R", [(1, 1), (3, 2)])
def test_xy_grad_statistic_matches_oracle(gs, oracle, dev, S, R):
    """xy_grad_out == sum over the sub-poses of d loss / d xys (float64 oracle, autograd through its
    per-sub-pose projections), and Gaussians culled everywhere report exactly zero."""
    O = oracle
    W, H, n = 128, 96, 1500
    sc = O.synthetic_scene(n, W, H, seed=77, scale_mult=6.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    et, rt, gamma, mlevel = 1 / 60, 1 / 30, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(6))
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    p = {k: sc[k].float().to(dev).requires_grad_(True) for k in names}
    times, _, _ = gs.subpose_schedule(S, et, R, rt)
    vms = gs.subpose_viewmats(p["viewmat"], p["lin_vel"], p["ang_vel"], torch.tensor(times, device=dev))
    xy_grad = torch.full((n, 2), 123.0, device=dev)                   # must be overwritten, not accumulated
    samples, alphas, radii = gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"],
                                                torch.sigmoid(p["opacity_logits"]), p["sh"], vms, bg.to(dev), S, R,
                                                sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, sh_degree=3,
                                                xy_grad_out=xy_grad)
    out = gs.combine_samples(samples, gamma, mlevel)
    (out * wt.to(dev)).sum().backward()
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=R, exposure_time=et,
                         rolling_shutter_time=rt, gamma=gamma, min_rgb_level=mlevel)
    q = {k: sc[k].double().requires_grad_(True) for k in names}
    ref, _, _, frag, parts, _ = O.render(cfg, q["means"], q["log_scales"].exp(), q["quats"],
                                         torch.sigmoid(q["opacity_logits"]), q["sh"], q["viewmat"], q["lin_vel"],
                                         q["ang_vel"], background=bg.double(), return_parts=True)
    for part in parts:
        part[0].xys.retain_grad()
    (ref * wt.double()).sum().backward()
    want = sum(part[0].xys.grad for part in parts)
    assert rel_max(xy_grad.cpu(), want) < GRAD_RTOL
    invisible = (radii == 0).all(dim=0)
    assert invisible.any() and float(xy_grad[invisible].abs().sum()) == 0.0
    with pytest.raises(ValueError):
        gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]), p["sh"],
                           vms, bg.to(dev), S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
                           xy_grad_out=torch.zeros(n, 3, device=dev))


def test_training_loop_with_densification(gs, oracle, dev):
    """f2 + f3 end to end: fit a blurred target rendered from a reference scene, refining every few steps; N
    changes, Adam follows, the loss goes down and stays finite."""
    O = oracle
    W, H = 128, 96
    target_sc = O.synthetic_scene(1500, W, H, seed=5, scale_mult=6.0)
    cfg = gs.SplatfactoDeblurConfig(blur_samples=3, rolling_shutter_compensation=False, gamma=2.2, min_rgb_level=10.0)
    c2w = torch.eye(4)[:3].clone()
    c2w[:, 1] *= -1
    c2w[:, 2] *= -1
    cam = gs.Camera(c2w, target_sc["fx"], target_sc["fy"], target_sc["cx"], target_sc["cy"], W, H,
                    metadata=dict(cam_idx=0, camera_linear_velocity=[1.0, 0.2, 0.0],
                                  camera_angular_velocity=[0.0, 0.5, 0.2], exposure_time=1 / 60,
                                  rolling_shutter_time=0.0))
    target_model = gs.SplatfactoDeblurModel.from_scene(cfg, target_sc, dev)
    gt = target_model.get_outputs_for_camera(cam)["rgb"].detach()
    start = O.synthetic_scene(400, W, H, seed=6, scale_mult=8.0)
    model = gs.SplatfactoDeblurModel.from_scene(cfg, start, dev)
    model.collect_densify_stats = True
    opts = gs.training.make_optimizers(model, lr_scale=10.0)
    dcfg = gs.densify.DensifyConfig(warmup_length=5, refine_every=5, densify_grad_thresh=2e-5)
    state = gs.densify.DensifyState(model.num_points, dev)
    losses, sizes, results = [], [model.num_points], []
    for step in range(1, 31):
        st = gs.training.train_step(model, opts, cam, gt)
        assert math.isfinite(st["loss"])
        losses.append(st["loss"])
        assert model.xy_grad is not None and model.xy_grad.shape == (model.num_points, 2)
        r = gs.densify.step_callback(model, opts, state, step, dcfg)
        if r is not None:
            results.append(r)
            sizes.append(r["after"])
            for name, prm in model.gauss_params().items():
                assert prm.shape[0] == r["after"] and opts[name].param_groups[0]["params"][0] is prm
    assert len(results) == 5 and sum(r["split"] + r["duplicated"] for r in results) > 0
    assert sizes[-1] != sizes[0]
    assert min(losses[-10:]) < losses[0]


# --------------------------------------------------------------------------- #
# device side of the row-sparse gradient exchange (the collective itself is covered by the
# world-2 gloo tests on CPU; here the HIP row kernels are held against the torch ops they replace)
# --------------------------------------------------------------------------- #
def test_dp_row_kernels_match_torch(gs, dev):
    from gsdeblur_amd.dp import _RowOps
    N = 20011
    shapes = [(N, 3), (N, 3), (N, 4), (N, 1), (N, 3), (N, 15, 3)]
    world = 3
    gens = [torch.Generator().manual_seed(900 + r) for r in range(world)]
    rank_grads = []
    for r in range(world):
        touched = torch.rand(N, generator=gens[r]) < 0.03
        rank_grads.append([torch.randn(s, generator=gens[r]) * touched.view(-1, *([1] * (len(s) - 1))) for s in shapes])
    rank_grads[1][3].zero_()                                   # a tensor that is zero on one rank
    payloads, counts = [], []
    for r in range(world):
        cpu = _RowOps([g.clone() for g in rank_grads[r]])
        hip = _RowOps([g.clone().to(dev) for g in rank_grads[r]])
        m_cpu, m_hip = cpu.row_mask(), hip.row_mask()
        assert torch.equal(m_cpu, m_hip.cpu())
        idx = m_cpu.nonzero().reshape(-1)
        Mpad = idx.numel() + 7
        p_cpu, p_hip = cpu.pack(idx, Mpad), hip.pack(idx.to(dev), Mpad)
        assert torch.equal(p_cpu.view(torch.int32), p_hip.cpu().view(torch.int32))       # bit-exact incl. the index column
        payloads.append(p_cpu)
        counts.append(idx.numel())
    # every "rank" applies all payloads in rank order: HIP result == torch result, bit for bit
    acc_cpu = _RowOps([torch.zeros(s) for s in shapes])
    acc_hip = _RowOps([torch.zeros(s, device=dev) for s in shapes])
    for r in range(world):
        acc_cpu.scatter_add(payloads[r], counts[r], 1.0 / world)
        acc_hip.scatter_add(payloads[r].to(dev), counts[r], 1.0 / world)
    for a, b, i in zip(acc_cpu.grads, acc_hip.grads, range(len(shapes))):
        assert torch.equal(a, b.cpu()), i
        want = sum(rank_grads[r][i] for r in range(world)) / world
        assert torch.allclose(a, want, atol=1e-6)


def _nccl_worker(rank, world, port, q):
    """every exchange form of gsdeblur_amd.dp on the nccl (= RCCL) backend with device tensors.  world == 1: forced
    (dp.allreduce_gradients(force=True)) — RCCL's single-rank collectives, gradients must come out bit-identical."""
    import os
    import sys
    import torch.distributed as dist
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    import gsdeblur_amd as gs
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    dev = torch.device("cuda", rank)
    force = world == 1
    N = 200_000
    shapes = [(N, 3), (N, 3), (N, 4), (N, 1), (N, 3), (N, 15, 3)]
    ok = dist.get_backend() == "nccl"
    names = set()
    prof = None
    try:                                                    # which device kernels / copies the exchange issues
        from torch.profiler import ProfilerActivity, profile
        prof = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
        prof.__enter__()
    except Exception:
        prof = None
    plan = (("sparse", 0.005, None), ("sparse", 0.006, None), ("sparse", 0.4, None), ("sparse", 0.01, None),
            ("sparse", 0.012, True), ("sparse", 0.011, True), ("allreduce", 0.01, None), ("rs_ag", 0.01, None))
    for step, (mode, density, sync_free) in enumerate(plan):
        grads = []
        for r in range(world):
            g = torch.Generator().manual_seed(7 + 31 * step + r)
            touched = torch.rand(N, generator=g) < density
            grads.append([torch.randn(s, generator=g) * touched.view(-1, *([1] * (len(s) - 1))) for s in shapes])
        params = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in shapes]
        for p, g in zip(params, grads[rank]):
            p.grad = g.to(dev)
        gs.dp.allreduce_gradients(params, mode=mode, sync_free=sync_free, force=force)
        for i, p in enumerate(params):
            want = sum(grads[r][i] for r in range(world))
            if world == 1:
                ok &= bool(torch.equal(p.grad.cpu(), want))          # one rank: the exchange must change nothing
            else:
                ok &= bool(torch.allclose(p.grad.cpu(), want, atol=1e-5))
    small = [torch.full((3,), float(rank + 1), device=dev), torch.full((2, 6), 2.0 * (rank + 1), device=dev)]
    gs.dp.allreduce_dense_(small, average=True, force=force)
    ok &= bool(torch.allclose(small[0].cpu(), torch.full((3,), (world + 1) / 2.0)))
    st = gs.dp._sparse_state(N, world, None)
    st.settle()
    ok &= st.overflows == 1          # the one density jump (0.6 % -> 40 % of the rows): guarded fallback to the dense bucket
    if prof is not None:
        try:
            torch.cuda.synchronize()
            prof.__exit__(None, None, None)
            names = {e.key for e in prof.key_averages() if "nccl" in e.key.lower() or "rccl" in e.key.lower()
                     or "gs_dp" in e.key or "dp_" in e.key}
        except Exception:
            names = set()
    q.put((rank, ok, sorted(names)))
    dist.destroy_process_group()


def _run_nccl_workers(world):
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return res


def test_gradient_exchange_over_rccl_world1(gs, dev):
    """the RCCL path on the ONE GPU this build can reach.  A process group of world size 1 on the
    nccl backend; dp.allreduce_gradients(force=True) drives all four exchange forms (guarded row-sparse, sync-free
    row-sparse, reduce-scatter + all-gather, dense all-reduce) and the small dense bucket on device tensors: pack kernel
    -> RCCL collective -> scatter kernel in stream order, headers through pinned memory.  The gradients must come out
    bit-identical.  What RCCL itself launches for a single rank (a device kernel or a copy) is printed, not asserted:
    no scaling curve is claimed from this."""
    res = _run_nccl_workers(1)
    assert len(res) == 1 and res[0][0] == 0 and res[0][1] is True, res
    print(f"[rccl world 1] device-side names seen by the profiler: {res[0][2]}")


def test_gradient_exchange_over_rccl_world2(gs, dev):
    """the DP exchange (sync-free row-sparse form, its dense fallback, dense all-reduce, reduce-scatter + all-gather,
    the small dense bucket) on the nccl (= RCCL) backend, two GPUs of one node; skipped on a single-GPU box"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    res = _run_nccl_workers(2)
    assert [(r, ok) for r, ok, _ in res] == [(0, True), (1, True)], res


def test_dp_masked_pack_and_payload_scatter_match_torch(gs, dev):
    """sync-free exchange kernels on one GPU: fixed-capacity pack of the masked rows (header = true count, packed
    count) and the count-from-header scatter-add, against plain torch ops; capacity below and above the row count"""
    from gsdeblur_amd.dp import _RowOps
    N = 50_000
    g = torch.Generator().manual_seed(3)
    touched = torch.rand(N, generator=g) < 0.02
    shapes = [(N, 3), (N, 4), (N,), (N, 15, 3)]
    grads = [(torch.randn(s, generator=g) * touched.view(-1, *([1] * (len(s) - 1)))).to(dev) for s in shapes]
    total = int(touched.sum())
    ops_ = _RowOps(grads)
    ref_rows = torch.cat([x.reshape(N, -1) for x in grads], dim=1)[touched.to(dev)]
    for cap in (total + 100, total // 2):
        pay = ops_.pack_masked(cap)
        hdr = pay[0, :2].view(torch.int32).tolist()
        assert hdr == [total, min(total, cap)]
        M = hdr[1]
        assert torch.equal(pay[1:1 + M, :ops_.wtot], ref_rows[:M])
        assert torch.equal(pay[1:1 + M, ops_.wtot].contiguous().view(torch.int32).cpu().long(),
                           touched.nonzero().reshape(-1)[:M])
        acc = [torch.zeros_like(x) for x in grads]
        _RowOps(acc).scatter_add_payload(pay, cap, 0.5)
        idx = touched.nonzero().reshape(-1)[:M].to(dev)
        for a, x in zip(acc, grads):
            want = torch.zeros_like(x)
            want[idx] = x[idx] * 0.5
            assert torch.equal(a, want)


@pytest.mark.parametrize("S,R,base,learn_bg", [(3, 2, 16, False), (4, 1, 512, False), (2, 1, 4, True)])
def test_render_combined_equals_two_step(gs, oracle, dev, S, R, base, learn_bg):
    """render_combined (one autograd node, sample gradients derived inside the compositor's backward) ==
    render_subposes + combine_samples: same image bit for bit, same gradients; with a learnable background the
    fused node falls back to the two-step backward."""
    from gsdeblur_amd import ops
    O = oracle
    W, H, n = 144, 96, 2500
    sc = O.synthetic_scene(n, W, H, seed=41 + S, scale_mult=6.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    times, _, _ = gs.subpose_schedule(S, 1 / 60, R, 1 / 30)
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(8)).to(dev)
    wa = torch.rand(S, H, W, generator=torch.Generator().manual_seed(9)).to(dev)
    old_base = ops.SLICE_BASE
    ops.SLICE_BASE = base
    try:
        res = []
        for fused in (False, True):
            p = {k: sc[k].float().to(dev).requires_grad_(True) for k in names}
            bgp = torch.tensor([0.2, 0.1, 0.3], device=dev, requires_grad=learn_bg)
            vms = gs.subpose_viewmats(p["viewmat"], p["lin_vel"], p["ang_vel"], torch.tensor(times, device=dev))
            args = (p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]), p["sh"], vms,
                    bgp, S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W)
            if fused:
                rgb, alphas, radii = gs.render_combined(*args, gamma=2.2, min_rgb_level=10.0)
            else:
                samples, alphas, radii = gs.render_subposes(*args)
                rgb = gs.combine_samples(samples, 2.2, 10.0)
            ((rgb * wt).sum() + 0.3 * (alphas * wa).sum()).backward()
            res.append((rgb.detach(), alphas.detach(), {k: v.grad for k, v in p.items()}, bgp.grad))
    finally:
        ops.SLICE_BASE = old_base
    (rgb0, al0, g0, b0), (rgb1, al1, g1, b1) = res
    assert torch.equal(rgb0, rgb1) and torch.equal(al0, al1)
    for k in names:
        assert rel_max(g1[k].cpu(), g0[k].cpu()) < 1e-6, k
    if learn_bg:
        assert rel_max(b1.cpu(), b0.cpu()) < 1e-6


# --------------------------------------------------------------------------- #
# edge cases of the fused path (empty / degenerate inputs)
# --------------------------------------------------------------------------- #
def _render_simple(gs, dev, means, log_scales, quats, opac_logit, sh, S, R, W, H, bg, lin=None, ang=None):
    p = dict(means=means, log_scales=log_scales, quats=quats, opacity_logits=opac_logit, sh=sh)
    p = {k: v.float().to(dev).requires_grad_(True) for k, v in p.items()}
    times, _, _ = gs.subpose_schedule(S, 1 / 60, R, 1 / 30)
    vm = torch.eye(4, device=dev)
    lin = torch.zeros(3) if lin is None else lin
    ang = torch.zeros(3) if ang is None else ang
    vms = gs.subpose_viewmats(vm, lin.to(dev), ang.to(dev), torch.tensor(times, device=dev))
    rgb, alphas, radii = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"],
                                            torch.sigmoid(p["opacity_logits"]), p["sh"], vms, bg.to(dev), S, R,
                                            0.8 * W, 0.8 * W, W / 2, H / 2, H, W, gamma=2.2, min_rgb_level=0.0)
    return rgb, alphas, radii, p


@pytest.mark.parametrize("S,R", [(1, 1), (3, 2)])
def test_fused_path_nothing_visible(gs, dev, S, R):
    """every Gaussian behind the camera: image == background exactly, alpha == 0, all gradients exactly zero"""
    n, W, H = 300, 70, 50                                        # neither dimension a multiple of 16
    g = torch.Generator().manual_seed(3)
    means = torch.randn(n, 3, generator=g)
    means[:, 2] = -1.0 - torch.rand(n, generator=g)              # z < 0 in the camera frame
    bg = torch.tensor([0.25, 0.5, 0.75])
    rgb, alphas, radii, p = _render_simple(gs, dev, means, torch.full((n, 3), -3.0), torch.randn(n, 4, generator=g),
                                           torch.zeros(n), torch.rand(n, 16, 3, generator=g), S, R, W, H, bg)
    assert int(radii.abs().sum()) == 0
    assert torch.allclose(rgb.cpu(), bg.expand(H, W, 3), atol=1e-6) and float(alphas.abs().max()) == 0.0
    (rgb.sum() + alphas.sum()).backward()
    for k, v in p.items():
        assert v.grad is not None and float(v.grad.abs().max()) == 0.0, k


def test_fused_path_single_gaussian_and_full_screen_gaussian(gs, oracle, dev):
    """N = 1 (one tiny splat), and one opaque Gaussian larger than the whole image in front of many others:
    every tile terminates on its first entry, the rest of the scene gets no gradient at all."""
    O = oracle
    W, H = 90, 60
    bg = torch.tensor([0.1, 0.2, 0.3])
    one = dict(means=torch.tensor([[0.0, 0.0, 4.0]]), log_scales=torch.full((1, 3), math.log(0.05)),
               quats=torch.tensor([[1.0, 0.0, 0.0, 0.0]]), opac=torch.tensor([3.0]), sh=torch.zeros(1, 16, 3))
    one["sh"][0, 0] = torch.tensor([1.0, 0.5, -0.2])
    rgb, alphas, radii, p = _render_simple(gs, dev, one["means"], one["log_scales"], one["quats"], one["opac"],
                                           one["sh"], 1, 1, W, H, bg)
    cfg = O.RenderConfig(H, W, 0.8 * W, 0.8 * W, W / 2, H / 2, blur_samples=1, rs_bands=1, gamma=2.2)
    ref, ref_a = O.render(cfg, one["means"].double(), one["log_scales"].double().exp(), one["quats"].double(),
                          torch.sigmoid(one["opac"].double()), one["sh"].double(), torch.eye(4).double(),
                          torch.zeros(3).double(), torch.zeros(3).double(), background=bg.double())
    assert (rgb.detach().cpu().double() - ref).abs().max() < 5e-4 and int(radii[0, 0]) > 0
    rgb.sum().backward()
    assert float(p["means"].grad.abs().sum()) > 0

    sc = O.synthetic_scene(3000, W, H, seed=12, scale_mult=4.0)
    means = torch.cat([torch.tensor([[0.0, 0.0, 0.5]]), sc["means"]])          # in front of everything (z >= 1)
    log_scales = torch.cat([torch.full((1, 3), math.log(10.0)), sc["log_scales"]])
    quats = torch.cat([torch.tensor([[1.0, 0.0, 0.0, 0.0]]), sc["quats"]])
    opac = torch.cat([torch.tensor([20.0]), sc["opacity_logits"]])              # alpha clamps at 0.999
    sh = torch.cat([torch.zeros(1, 16, 3), sc["sh"]])
    rgb, alphas, radii, p = _render_simple(gs, dev, means, log_scales, quats, opac, sh, 2, 1, W, H, bg,
                                           lin=torch.tensor([0.5, 0.0, 0.0]))
    assert torch.isfinite(rgb).all() and float(alphas.min()) > 0.998
    rgb.sum().backward()
    g = p["means"].grad
    assert torch.isfinite(g).all()
    # T after the wall is 1e-3: a few Gaussians behind it still contribute until T <= 1e-4, but only a handful
    touched = int((g[1:].abs().sum(dim=1) > 0).sum())
    assert touched < 0.2 * (means.shape[0] - 1)


def test_full_size_fast_path_equals_plain_path(gs, oracle, dev):
    """BASELINE.json's metric configuration (1M Gaussians, 1920x1080, 5 sub-poses): the default path (depth
    slices, exact tile culling, compact emission from hit masks, deferred colour, gradient tuples) against the
    plainest one (ONE slice holding all 238 M bounding-box intersections, no culling, colour in the projection,
    fp32 atomics): identical images bit for bit, gradients equal up to the atomics' summation order."""
    from gsdeblur_amd import ops
    n, W, H, S = 1_000_000, 1920, 1080, 5
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=1234), dev)
    times, _, _ = gs.subpose_schedule(S, sc["exposure_time"], 1, sc["rolling_shutter_time"])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    knobs = ("SLICE_BASE", "EXACT_TILE_CULL", "COMPACT_EMIT", "HIT_MASKS", "GRAD_TUPLES", "DEFER_COLOR")
    saved = {k: getattr(ops, k) for k in knobs}
    res = []
    try:
        for plain in (False, True):
            if plain:
                for k in knobs:
                    setattr(ops, k, 0)
            p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
            vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"], sc["ang_vel"], torch.tensor(times, device=dev))
            rgb, _, radii = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"],
                                               torch.sigmoid(p["opacity_logits"]), p["sh"], vms, None, S, 1, sc["fx"],
                                               sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2, min_rgb_level=10.0,
                                               return_alpha=False)
            (rgb * wt).sum().backward()
            res.append((rgb.detach().clone(), {k: v.grad.clone() for k, v in p.items()}, ops.last_num_intersects,
                        list(ops.last_slice_intersects)))
            del p, rgb
            torch.cuda.empty_cache()
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
    (img_f, g_f, I_f, sl_f), (img_p, g_p, I_p, sl_p) = res
    assert I_f == I_p and I_p > 200_000_000 and sl_p == [I_p]        # the plain path really sorted every pair
    assert sum(sl_f) < 0.1 * I_p                                     # ... and the default path < 10 % of them
    assert torch.equal(img_f, img_p)
    for k in g_f:
        assert rel_max(g_f[k].cpu(), g_p[k].cpu()) < GRAD_RTOL, k
        touched_f = (g_f[k].reshape(n, -1) != 0).any(dim=1)
        touched_p = (g_p[k].reshape(n, -1) != 0).any(dim=1)
        assert torch.equal(touched_f, touched_p), k                  # the same Gaussians receive a gradient


_PATH_KNOBS = ("SLICE_BASE", "EXACT_TILE_CULL", "COMPACT_EMIT", "HIT_MASKS", "GRAD_TUPLES", "DEFER_COLOR",
               "RASTER_FWD_VARIANT", "RASTER_BWD_VARIANT")


def _full_size_two_paths(gs, dev, n, W, H, S, R, profile, other, min_slices=1, seed=1234, el_bar=False):
    """default path vs the path configured by `other` (knob -> value) on a full-size seeded scene: images must be
    bit-identical, the same Gaussians must receive a gradient, gradients equal up to fp32 summation order.
    el_bar: `other` is deterministic too (gradient tuples instead of atomics): the PER-ELEMENT bar of the oracle
    comparisons applies (1e-4 relative + 1e-5 of the tensor's max) instead of 3e-3 of the tensor's max."""
    from gsdeblur_amd import ops
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=seed, profile=profile), dev)
    times, _, _ = gs.subpose_schedule(S, sc["exposure_time"], R, sc["rolling_shutter_time"])
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    saved = {k: getattr(ops, k) for k in _PATH_KNOBS}
    res = []
    try:
        for knobs in ({}, other):
            for k, v in saved.items():
                setattr(ops, k, knobs.get(k, v))
            # the default path is taken in its STEADY state (round 6): the third frame through one FrameHints — what the
            # first two taught it is switched on (lazy records, nearest-first selection, the selection's size as a promise
            # to the sort), which is how bench.py and a training loop run it
            hints = ops.FrameHints() if not knobs else None
            for _rep in range(3 if not knobs else 1):
                p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
                vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"], sc["ang_vel"], torch.tensor(times, device=dev))
                rgb, _, radii = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"],
                                                   torch.sigmoid(p["opacity_logits"]), p["sh"], vms, None, S, R, sc["fx"],
                                                   sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2, min_rgb_level=10.0,
                                                   return_alpha=False, hints=hints)
                (rgb * wt).sum().backward()
            if not knobs:
                print(f"[steady state {n} {W}x{H} S={S} R={R} {profile}] third frame: selection state "
                      f"{ops.last_depth_select}, promised size {hints.select_cap}, lazy records {hints.lazy_records()}")
            res.append((rgb.detach().clone(), {k: v.grad.clone() for k, v in p.items()}, ops.last_num_intersects,
                        list(ops.last_slice_intersects)))
            del p, rgb, vms
            torch.cuda.empty_cache()
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
    (img_f, g_f, I_f, sl_f), (img_o, g_o, I_o, sl_o) = res
    # (rolling-shutter bands: the default path's projection culls / clips the pairs outside their band, the plain path
    #  counts every bounding-box pair)
    assert I_f == I_o if R == 1 else 0 < I_f < I_o
    assert sum(1 for x in sl_f if x > 0) >= min_slices, sl_f
    same_kernels = all(other.get(k, 0) == 0 for k in ("RASTER_FWD_VARIANT", "RASTER_BWD_VARIANT"))
    tag = f"{n} {W}x{H} S={S} R={R} {profile}"
    if same_kernels:
        assert torch.isfinite(img_f).all() and torch.equal(img_f, img_o)       # same arithmetic, other binning
    else:
        images_close(img_f, img_o, 5e-4, tag)          # 5e-4: the gamma chain amplifies near the floor
    for k in g_f:
        gf, go = g_f[k].cpu().numpy(), g_o[k].cpu().numpy()
        if same_kernels and el_bar:
            assert grad_el_ratio(gf, go) <= 1.0, k
        elif el_bar:
            # per element, except the rows fed by a threshold pixel: a bounded share of the rows may exceed the bar
            tol = GRAD_EL_RTOL * np.abs(go) + GRAD_EL_AFRAC * (np.abs(go).max() + 1e-300)
            bad_rows = ((np.abs(gf.astype(np.float64) - go) / tol).reshape(n, -1).max(axis=1) > 1.0)
            rows = int((np.abs(go).reshape(n, -1).max(axis=1) > 0).sum())
            print(f"[kernel-vs-kernel grad {k} {tag}] rows over the per-element bar: {int(bad_rows.sum())} of {rows} "
                  f"with a gradient, rel_max = {rel_max(gf, go):.3e}")
            assert bad_rows.sum() <= max(2, 2e-3 * rows) and rel_max(gf, go) < GRAD_RTOL, k
        else:
            assert rel_max(gf, go) < GRAD_RTOL, k
        touched_f = (g_f[k].reshape(n, -1) != 0).any(dim=1)
        touched_o = (g_o[k].reshape(n, -1) != 0).any(dim=1)
        if same_kernels:
            assert torch.equal(touched_f, touched_o), k
        else:
            # a Gaussian whose only blended pixel sits on a threshold may gain or lose its gradient
            diff = int((touched_f != touched_o).sum())
            print(f"[kernel-vs-kernel touched {k} {tag}] {diff} of {int(touched_o.sum())} rows differ")
            assert diff <= max(2, 1e-3 * int(touched_o.sum())), (k, diff)
    return sl_f, sl_o, I_o, g_f, I_f


_PLAIN = dict(SLICE_BASE=0, EXACT_TILE_CULL=0, COMPACT_EMIT=0, HIT_MASKS=0, GRAD_TUPLES=0, DEFER_COLOR=0,
              RASTER_FWD_VARIANT=2, RASTER_BWD_VARIANT=2)
# the same plain path made DETERMINISTIC: one slice holding every bounding-box pair, no
# culling, colour in the projection, the round-1 compositors — but the gradients go through per-entry tuples and the
# segmented sum instead of fp32 atomics, so two runs (and the comparison with the default path) agree element by element
_PLAIN_DET = dict(_PLAIN, GRAD_TUPLES=1)


def test_full_size_headline_equals_deterministic_plain_path_per_element(gs, dev):
    """BASELINE.json's metric configuration, default path vs the deterministic plain path: bit-identical image and the
    per-element gradient bar (not 3e-3 of the tensor's max).  238 M tuples of 48 bytes = 11.4 GB on the plain side."""
    sl_f, sl_o, I, _, _ = _full_size_two_paths(gs, dev, 1_000_000, 1920, 1080, 5, 1, "survey", _PLAIN_DET, el_bar=True)
    assert sl_o == [I] and sum(sl_f) < 0.1 * I


def test_full_size_multi_slice_equals_deterministic_plain_path_per_element(gs, dev):
    """the fitted-model-like scene (several depth slices, a third of the Gaussians with a gradient), same bar"""
    sl_f, sl_o, I, g, _ = _full_size_two_paths(gs, dev, 1_000_000, 1920, 1080, 5, 1, "trained", _PLAIN_DET, min_slices=3,
                                               el_bar=True)
    assert sl_o == [I]


def test_full_size_config3_rolling_shutter_bands_equals_plain_path(gs, dev):
    """BASELINE.json config 3: 1M Gaussians, 1080p, 10 rolling-shutter row bands (S=1, R=10) — default path vs the
    plainest one (one slice with every bounding-box pair, no culling, atomics, round-1 v_readlane compositors)."""
    sl_f, sl_o, I, _, I_banded = _full_size_two_paths(gs, dev, 1_000_000, 1920, 1080, 1, 10, "survey", _PLAIN)
    # plain path: ONE slice holding every bounding-box pair of every sub-pose's own row band (a tenth of all pairs)
    assert len(sl_o) == 1 and 0.05 * I < sl_o[0] < 0.2 * I and sum(sl_f) < 0.3 * sl_o[0]
    # round 5: the default path's band-aware projection counts exactly those pairs — the plain path's emission and the
    # projection's clipped boxes are two routes to the same integer
    assert I_banded == sl_o[0]


def test_full_size_config4_blur_and_rolling_shutter_share_equals_plain_path(gs, dev):
    """BASELINE.json config 4, one GPU's share (1 view): 2M Gaussians, 1080p, 5 samples x 2 bands."""
    sl_f, sl_o, I, _, I_banded = _full_size_two_paths(gs, dev, 2_000_000, 1920, 1080, 5, 2, "survey", _PLAIN)
    assert len(sl_o) == 1 and 0.3 * I < sl_o[0] < 0.7 * I and sum(sl_f) < 0.2 * sl_o[0] and I_banded == sl_o[0]


def test_full_size_config5_4k_10_subposes_two_slicings_agree(gs, dev):
    """BASELINE.json config 5, one GPU's share: 5M Gaussians, 3840x2160, 10 sub-poses.  The plainest path cannot
    hold its 4e9 bounding-box pairs in one 2^31-entry slice, so the default path is compared with a DIFFERENT
    slicing (4x the budget), without hit masks / deferred colour / tuples and with the round-1 compositors."""
    other = dict(SLICE_BASE=2048, HIT_MASKS=0, GRAD_TUPLES=0, DEFER_COLOR=0, RASTER_FWD_VARIANT=2, RASTER_BWD_VARIANT=2)
    sl_f, sl_o, I, _, _ = _full_size_two_paths(gs, dev, 5_000_000, 3840, 2160, 10, 1, "survey", other)
    assert I > 2 ** 31 and sum(sl_f) < 0.1 * I


def test_full_size_multi_slice_frame_equals_plain_path(gs, dev):
    """1M Gaussians, 1080p, 5 sub-poses of the fitted-model-like scene (small translucent Gaussians): the default
    path needs at least three depth slices and a large share of the Gaussians receives a gradient."""
    sl_f, sl_o, I, g, _ = _full_size_two_paths(gs, dev, 1_000_000, 1920, 1080, 5, 1, "trained", _PLAIN, min_slices=3)
    assert sl_o == [I]
    assert (g["means"] != 0).any(dim=1).float().mean().item() > 0.3


def test_more_intersections_than_the_slice_plan_covers(gs, dev):
    """Regression (found by tests/fuzz_paths.py, seed 1 trial 328): with a tiny slice budget the 16 planned
    boundaries end before the last Gaussian of a sub-pose; the last slice takes the rest and every per-slice
    buffer (hit masks!) must be sized from the sub-poses' real totals, not from the last boundary."""
    from gsdeblur_amd import ops
    n, W, H, S = 60000, 32, 32, 2                     # T = 4 tiles: budget 4 << 15 = 131k < ~240k box pairs per sub-pose
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=5, scale_mult=12.0), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    knobs = ("SLICE_BASE", "EXACT_TILE_CULL", "COMPACT_EMIT", "HIT_MASKS", "GRAD_TUPLES", "DEFER_COLOR")
    saved = {k: getattr(ops, k) for k in knobs}
    res = []
    try:
        for plain in (False, True):
            for k in knobs:
                setattr(ops, k, 0 if plain else saved[k])
            if not plain:
                ops.SLICE_BASE = 1
            p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
            vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"] * 20, sc["ang_vel"] * 10,
                                      torch.tensor(times, device=dev))
            rgb, alphas, _ = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"],
                                                torch.sigmoid(p["opacity_logits"]), p["sh"], vms, None, S, 1, sc["fx"],
                                                sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2, min_rgb_level=10.0)
            ((rgb * wt).sum() + alphas.sum()).backward()
            res.append((rgb.detach().clone(), {k: v.grad.clone() for k, v in p.items()}, ops.last_num_intersects))
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
    (img_f, g_f, I_f), (img_p, g_p, I_p) = res
    assert I_f == I_p and I_p > S * (4 << 15)           # really beyond the last planned boundary
    assert torch.equal(img_f, img_p)
    for k in g_f:
        assert rel_max(g_f[k].cpu(), g_p[k].cpu()) < GRAD_RTOL, k


@pytest.mark.parametrize("case", ["blur", "rolling_shutter", "multi_slice", "one_slice_no_plan", "pixel_velocity",
                                  "tiny_budget", "exact_rolling_shutter", "exact_rolling_shutter_multi_slice"])
def test_native_frame_orchestration_equals_python_orchestration(gs, dev, case):
    """gs_frame_forward / gs_frame_backward (csrc/frame.hip: the slice pipeline issued from C++
    out of ONE caller-owned arena) against tests/python_frame_path.py's sliced_forward / sliced_backward (the same kernels launched one by one
    from Python with torch allocations).  Same launches in the same order on the atomic-free path: images, alphas, the
    depth channel, radii and EVERY gradient must be bit-identical; the per-slice emitted counts too.  Cases: motion
    blur, rolling-shutter bands (initially closed tiles), a fitted-model-like scene that needs several depth slices,
    slicing switched off, the pixel-velocity model, and a budget so small that the 16 planned boundaries end early."""
    from gsdeblur_amd import ops
    n, W, H, S, R, prof, base, mult = 30000, 200, 136, 3, 1, "survey", None, 3.0
    if case == "rolling_shutter":
        S, R = 2, 4
    elif case == "multi_slice":
        n, prof, base, mult = 60000, "trained", 24, 6.0
    elif case == "one_slice_no_plan":
        base = 0
    elif case == "tiny_budget":
        n, W, H, S, base, mult = 60000, 32, 32, 2, 1, 12.0
    elif case == "exact_rolling_shutter_multi_slice":
        n, prof, base, mult = 60000, "trained", 24, 6.0
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=11, scale_mult=mult, profile=prof), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, R, 1 / 30 if R > 1 else 0.0)
    times_t = torch.tensor(times, device=dev)
    g = torch.Generator().manual_seed(4)
    wt, wa = torch.rand(H, W, 3, generator=g).to(dev), torch.rand(S, H, W, generator=g).to(dev)
    saved = (ops.NATIVE_FRAME, ops.SLICE_BASE, ops.SLICE_MERGE, ops.BAND_AWARE)
    res = []
    try:
        ops.SLICE_MERGE = 0.0                              # the Python orchestration issues every planned slice
        ops.BAND_AWARE = 0                                 # ... from a projection that keys every (band, Gaussian) pair
        for native in (1, 0):
            ops.NATIVE_FRAME = native
            if base is not None:
                ops.SLICE_BASE = base
            p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
            lin = (sc["lin_vel"] * 5).clone().requires_grad_(True)
            ang = (sc["ang_vel"] * 3).clone().requires_grad_(True)
            V = sc["viewmat"].clone().requires_grad_(True)
            if case.startswith("exact_rolling_shutter"):
                # continuous row time of the pixel-velocity model: box lists + the raster_rs.hip compositors
                out = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]),
                                         p["sh"], V, None, S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2,
                                         min_rgb_level=10.0, lin_vel=lin, ang_vel=ang, times=times_t, return_depth=True,
                                         rolling_shutter_time=1 / 30)
            elif case == "pixel_velocity":
                out = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]),
                                         p["sh"], V, None, S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2,
                                         min_rgb_level=10.0, lin_vel=lin, ang_vel=ang, times=times_t, return_depth=True)
            else:
                vms = gs.subpose_viewmats(V, lin, ang, times_t)
                out = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]),
                                         p["sh"], vms, None, S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2,
                                         min_rgb_level=10.0, return_depth=True)
            rgb, alphas, radii, depth = out
            assert (python_frame_path.native_ok() and native == 1) or native == 0
            ((rgb * wt).sum() + (alphas * wa).sum()).backward()
            grads = {k: v.grad.clone() for k, v in p.items()}
            grads.update(lin=lin.grad.clone(), ang=ang.grad.clone(), V=V.grad.clone())
            res.append((rgb.detach().clone(), alphas.detach().clone(), radii.clone(), depth.clone(), grads,
                        ops.last_num_intersects, [int(v) for v in ops.last_slice_intersects if int(v) > 0]))
    finally:
        ops.NATIVE_FRAME, ops.SLICE_BASE, ops.SLICE_MERGE, ops.BAND_AWARE = saved
    a, b = res
    if case.startswith("exact_rolling_shutter"):
        # round 6: the library culls the pixel-velocity lists by the swept alpha >= 1/255 ellipse
        # (gs_slice_counts_exact_swept); the Python twin keeps the swept boxes whole.  What is culled can blend nothing:
        # same pair total, fewer list entries, the same images and gradients bit for bit (asserted below)
        assert a[5] == b[5] and a[5] > 0 and 0 < sum(a[6]) < sum(b[6]), (a[5], b[5], a[6], b[6])
    else:
        assert a[5] == b[5] and a[6] == b[6] and a[5] > 0
    if case in ("multi_slice", "tiny_budget"):
        assert len(a[6]) >= 2
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    for k in a[4]:
        if k in ("lin", "ang", "V") or case.startswith("exact_rolling_shutter"):
            # camera-level tensors: two orchestrations, two block structures of the ordered sum; exact rolling shutter
            # (round 6): the library's culled lists hold fewer tuples per Gaussian than the twin's box lists, so the
            # segmented tuple sum adds the SAME non-zero tuples in another order
            assert rel_max(a[4][k].cpu(), b[4][k].cpu()) < 1e-4, k
        else:
            assert torch.equal(a[4][k], b[4][k]), k
    assert float(a[0].abs().max()) > 0 and float(a[4]["means"].abs().max()) > 0


def test_native_frame_grows_its_arena_and_reports_stage_times(gs, dev):
    """the first frame of a new shape may not fit the arena estimate: the library prices the slice plan, says what it
    needs, the host projects again and retries — the result is the one a large arena gives; the library's own HIP
    events feed ops.StageProfiler like the Python orchestration's did"""
    from gsdeblur_amd import ops
    n, W, H, S = 40000, 160, 96, 2
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=3, scale_mult=8.0), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
    vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"], sc["ang_vel"], torch.tensor(times, device=dev))

    hints = ops.FrameHints()
    ops.release_arenas()

    def render():
        return gs.render_combined(sc["means"], sc["log_scales"].exp(), sc["quats"], torch.sigmoid(sc["opacity_logits"]),
                                  sc["sh"], vms, None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2,
                                  hints=hints)[0]
    ref = render()
    assert hints.arena_bytes > (1 << 20) and hints.frames == 1
    # a far too small arena forces the retry path: pool emptied, estimate forgotten
    ops.release_arenas()
    pool_key = (str(dev), torch.cuda.current_stream(dev).cuda_stream)
    ops._arena_pool[pool_key] = [torch.empty(1 << 20, dtype=torch.uint8, device=dev)]
    hints.arena_bytes = 1 << 20
    got = render()
    assert torch.equal(ref, got) and hints.arena_bytes > (1 << 20) and hints.arena_retries >= 1
    # the grown arena is the pool's: the next frame — of ANY scene on this stream — runs in it without a retry or a malloc
    (pooled,) = ops._arena_pool[pool_key]
    before = hints.arena_retries
    got2 = render()
    assert torch.equal(ref, got2) and hints.arena_retries == before
    assert ops._arena_pool[pool_key][0].data_ptr() == pooled.data_ptr()
    ops.profiler = ops.StageProfiler()
    try:
        p = sc["means"].clone().requires_grad_(True)
        img = gs.render_combined(p, sc["log_scales"].exp(), sc["quats"], torch.sigmoid(sc["opacity_logits"]), sc["sh"], vms,
                                 None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2)[0]
        img.sum().backward()
        ms = ops.profiler.summary_ms()
    finally:
        ops.profiler = None
    for stage in ("depth_sort", "slice_count", "tile_sort", "raster_fwd", "raster_bwd", "grad_reduce", "project_fwd"):
        assert stage in ms and len(ms[stage]) >= 1 and all(t > 0 for t in ms[stage]), stage


def test_native_frame_arena_converges_over_many_ever_larger_slices(gs, dev):
    """found by tests/fuzz_paths.py (round 3): the library prices ONE slice ahead, so a first frame that needs many
    slices of doubling size (tiny budget, translucent scene, merging on) takes several arena retries — the host used
    to give up after three.  Start from a 1 MB arena: the frame must come out, equal to the Python orchestration's."""
    from gsdeblur_amd import ops
    n, W, H, S = 60000, 160, 112, 2
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=23, scale_mult=8.0, profile="trained"), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
    vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"], sc["ang_vel"], torch.tensor(times, device=dev))
    saved = (ops.NATIVE_FRAME, ops.SLICE_BASE)
    hints = ops.FrameHints()

    def render():
        return gs.render_combined(sc["means"], sc["log_scales"].exp(), sc["quats"], torch.sigmoid(sc["opacity_logits"]),
                                  sc["sh"], vms, None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2,
                                  hints=hints)[0]
    try:
        ops.SLICE_BASE = 2
        ops.NATIVE_FRAME = 0
        ref = render()
        planned = sum(1 for v in ops.last_slice_intersects if int(v) > 0)
        ops.NATIVE_FRAME = 1
        ops.release_arenas()
        ops._arena_pool[(str(dev), torch.cuda.current_stream(dev).cuda_stream)] = [torch.empty(1 << 20, dtype=torch.uint8,
                                                                                               device=dev)]
        hints.arena_bytes = 1 << 20
        got = render()
        assert python_frame_path.native_ok()
    finally:
        ops.NATIVE_FRAME, ops.SLICE_BASE = saved
        ops.release_arenas()
    assert planned >= 6, planned
    assert torch.equal(ref, got)


def test_native_frame_merges_slices_of_a_frame_that_does_not_saturate(gs, dev):
    """round 3: planned depth slices are issued together once a slice leaves most tiles open (gs_frame_desc.
    merge_open_fraction).  A fitted-model-like scene with a small budget plans many slices and closes few tiles: with
    merging fewer slices are issued (a merged slice may emit more entries: tiles that close inside it are only known
    afterwards), the image is the same bit for bit and the
    gradients agree up to the order of their sums; a threshold of 0 reproduces the planned slices one by one."""
    from gsdeblur_amd import ops
    n, W, H, S = 60000, 200, 136, 3
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=11, scale_mult=6.0, profile="trained"), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
    times_t = torch.tensor(times, device=dev)
    g = torch.Generator().manual_seed(5)
    wt = torch.rand(H, W, 3, generator=g).to(dev)
    saved = (ops.NATIVE_FRAME, ops.SLICE_BASE, ops.SLICE_MERGE)
    res = {}
    try:
        ops.NATIVE_FRAME, ops.SLICE_BASE = 1, 8
        for merge in (0.0, 0.75):
            ops.SLICE_MERGE = merge
            p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
            vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"] * 5, sc["ang_vel"] * 3, times_t)
            rgb, alphas, radii = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"],
                                                    torch.sigmoid(p["opacity_logits"]), p["sh"], vms, None, S, 1, sc["fx"],
                                                    sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2, min_rgb_level=10.0)
            assert python_frame_path.native_ok()
            (rgb * wt).sum().backward()
            res[merge] = (rgb.detach().clone(), alphas.detach().clone(), {k: v.grad.clone() for k, v in p.items()},
                          [int(v) for v in ops.last_slice_intersects if int(v) > 0])
    finally:
        ops.NATIVE_FRAME, ops.SLICE_BASE, ops.SLICE_MERGE = saved
    a, b = res[0.0], res[0.75]
    assert len(a[3]) >= 4 and len(b[3]) < len(a[3]), (a[3], b[3])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert rel_max(a[2][k].cpu(), b[2][k].cpu()) < GRAD_RTOL, k
    assert float(a[2]["means"].abs().max()) > 0


@pytest.mark.parametrize("model", ["se3", "pixel_velocity"])
def test_needle_gaussians_gradients_vs_float64_oracle(gs, oracle, dev, model):
    """VERDICT round 2 item 6a.  A scene in which 8 % of the Gaussians are NEEDLES (longest / shortest scale 30..80,
    projected conic determinant down to 1e-4 of a*c): in fp32 the chain v_conic -> cov2d -> cov3d -> scale / quaternion
    cancels along the long axis (round 2's fuzz found 2-6 % errors on such a splat on every fp32 route).  With the
    double-precision chain for needles (project_needle_hp_kernel, default) EVERY gradient element meets the bar of the
    other oracle comparisons; with it switched off (GSD_NEEDLE_HP=0 / ops.NEEDLE_HP = 0) the same scene does not —
    which is what keeps this test honest."""
    from gsdeblur_amd import ops
    O = oracle
    n, W, H, S, R = 2500, 144, 96, 3, 1
    sc = O.synthetic_scene(n, W, H, seed=77, scale_mult=5.0)
    g = torch.Generator().manual_seed(8)
    needle = torch.rand(n, generator=g) < 0.08
    ls = sc["log_scales"].clone()
    long_axis = torch.randint(0, 3, (n,), generator=g)
    ratio = 30.0 + 50.0 * torch.rand(n, generator=g)
    base = ls.mean(dim=1) - 1.0
    for a in range(3):
        is_long = long_axis == a
        ls[:, a] = torch.where(needle, base + torch.where(is_long, torch.log(ratio), torch.zeros(n)), ls[:, a])
    sc["log_scales"] = ls
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 10, sc["ang_vel"] * 5
    et, rt, gamma, mlevel = 1 / 60, 0.0, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=R, exposure_time=et,
                         rolling_shutter_time=rt, gamma=gamma, min_rgb_level=mlevel, motion_model=model)
    q = {k: sc[k].double().requires_grad_(True) for k in names}
    ref, _, ref_samples, frag, _, _ = O.render(cfg, q["means"], q["log_scales"].exp(), q["quats"],
                                               torch.sigmoid(q["opacity_logits"]), q["sh"], q["viewmat"], q["lin_vel"],
                                               q["ang_vel"], background=bg.double(), return_parts=True)
    good = ~frag
    check_fragile(frag, f"needle {model}")
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)) * good[..., None]
    (ref * wt.double()).sum().backward()
    # the needles must matter: they are visible and carry a real share of the scale gradient
    gref = q["log_scales"].grad
    assert needle.sum() > 100 and gref[needle].abs().max() > 0.05 * gref.abs().max()
    times, _, _ = gs.subpose_schedule(S, et, R, rt)
    times_t = torch.tensor(times, device=dev)
    worst = {}
    saved = ops.NEEDLE_HP
    try:
        for hp in (1, 0):
            ops.NEEDLE_HP = hp
            p = {k: sc[k].float().to(dev).requires_grad_(True) for k in names}
            if model == "pixel_velocity":
                samples, _, _ = gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"],
                                                   torch.sigmoid(p["opacity_logits"]), p["sh"], p["viewmat"], bg.to(dev),
                                                   S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, sh_degree=3,
                                                   lin_vel=p["lin_vel"], ang_vel=p["ang_vel"], times=times_t)
            else:
                vms = gs.subpose_viewmats(p["viewmat"], p["lin_vel"], p["ang_vel"], times_t)
                samples, _, _ = gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"],
                                                   torch.sigmoid(p["opacity_logits"]), p["sh"], vms, bg.to(dev), S, R,
                                                   sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, sh_degree=3)
            out = gs.combine_samples(samples, gamma, mlevel)
            (out * wt.to(dev)).sum().backward()
            if hp:
                assert (samples.detach().cpu().double() - ref_samples)[:, good].abs().max().item() < IMG_ATOL
            worst[hp] = {k: grad_el_ratio(p[k].grad.cpu().numpy(), q[k].grad.numpy()) for k in
                         ("means", "log_scales", "quats", "opacity_logits", "sh")}
    finally:
        ops.NEEDLE_HP = saved
    print(f"needles ({model}): per-element gradient error / tolerance, double chain:",
          {k: round(v, 3) for k, v in worst[1].items()}, " fp32 chain:", {k: round(v, 3) for k, v in worst[0].items()})
    for k, v in worst[1].items():
        assert v <= 1.0, (k, v)
    assert max(worst[0]["log_scales"], worst[0]["quats"]) > 1.0          # the fp32 chain fails this scene


def test_backward_transmittance_rebuild_over_a_500_entry_tile_vs_float64(gs, oracle, dev):
    """the backward rebuilds the transmittance in front of every entry with T *= rcp(1 - alpha)
    (v_rcp_f32, 1 ulp) — compounded over a list of several hundred entries.  One 16x16 tile, 700 overlapping translucent
    Gaussians (alpha ~ 0.02: about 450 are blended before T falls under 1e-4): every gradient element against the
    float64 oracle's autograd with the usual per-element bar."""
    O = oracle
    n, W, H = 700, 16, 16
    g = torch.Generator().manual_seed(12)
    sc = O.synthetic_scene(n, W, H, seed=12, scale_mult=1.0)
    # all Gaussians in front of the camera on the axis, large on screen, depth-ordered by construction
    sc["means"] = torch.stack([0.02 * torch.randn(n, generator=g), 0.02 * torch.randn(n, generator=g),
                               2.0 + 3.0 * torch.rand(n, generator=g)], -1)
    sc["log_scales"] = torch.log(torch.full((n, 3), 0.6)) + 0.35 * torch.randn(n, 3, generator=g)
    sc["opacity_logits"] = torch.full((n,), -3.9) + 0.3 * torch.randn(n, generator=g)
    sc["fx"] = sc["fy"] = 200.0                           # sigma ~ 30 px: every splat covers the whole tile
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    bg = torch.tensor([0.2, 0.3, 0.1])
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=1, rs_bands=1, exposure_time=0.0,
                         rolling_shutter_time=0.0, gamma=1.0, min_rgb_level=0.0)
    q = {k: sc[k].double().requires_grad_(True) for k in names}
    ref, _, ref_samples, frag, _, _ = O.render(cfg, q["means"], q["log_scales"].exp(), q["quats"],
                                               torch.sigmoid(q["opacity_logits"]), q["sh"], q["viewmat"], q["lin_vel"],
                                               q["ang_vel"], background=bg.double(), return_parts=True)
    good = ~frag
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)) * good[..., None]
    (ref * wt.double()).sum().backward()
    out, alpha, samples, vms, p, radii = _run_full(gs, O, dev, sc, H, W, 1, 1, 0.0, 0.0, 1.0, 0.0, 3, bg, wt)
    from gsdeblur_amd import ops
    blended = int((p["opacity_logits"].grad != 0).sum())
    assert ops.last_num_intersects >= 600 and blended >= 350, (ops.last_num_intersects, blended)
    assert float(alpha.detach().min()) > 0.99             # the whole tile saturates: the list is walked to the stop
    assert (samples.detach().cpu().double() - ref_samples)[:, good].abs().max().item() < IMG_ATOL
    worst = {k: grad_el_ratio(p[k].grad.cpu().numpy(), q[k].grad.numpy()) for k in
             ("means", "log_scales", "quats", "opacity_logits", "sh")}
    print("500-entry tile: per-element gradient error / tolerance:", {k: round(v, 3) for k, v in worst.items()},
          "Gaussians with a gradient:", blended)
    for k, v in worst.items():
        assert v <= 1.0, (k, v)


@pytest.mark.parametrize("S,W,H,n,base", [(1, 144, 128, 2500, None), (3, 128, 160, 3000, None), (2, 96, 128, 6000, 8)])
def test_exact_rolling_shutter_pixel_velocity_vs_oracle(gs, oracle, dev, S, W, H, n, base):
    """VERDICT round 2 'Missing 1' / item 5: exact per-row rolling-shutter time in the pixel-velocity model
    (csrc/raster_rs.hip; gs_project_pixvel_fwd widens the tile boxes by the sweep and hands out the pixel velocity).
    Per blur sample ONE projection: swept tile counts / radii against the float32 oracle (integers, exact), image and
    every gradient — Gaussians, mid-exposure viewmat, linear and angular velocity (the row term adds
    sum tau(y) * d/d centre to d loss / d pixel-velocity) — against the float64 oracle's continuous mode.  Last case: a
    tiny slice budget, several depth slices with carried per-pixel state."""
    from gsdeblur_amd import ops
    O = oracle
    sc = O.synthetic_scene(n, W, H, seed=900 + S, scale_mult=6.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 30, sc["ang_vel"] * 15
    et, rt, gamma, mlevel = 1 / 60, 1 / 30, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=1, exposure_time=et,
                         rolling_shutter_time=rt, gamma=gamma, min_rgb_level=mlevel, motion_model="pixel_velocity",
                         rs_exact=True)
    q = {k: sc[k].double().requires_grad_(True) for k in names}
    ref, _, ref_samples, frag, parts, _ = O.render(cfg, q["means"], q["log_scales"].exp(), q["quats"],
                                                   torch.sigmoid(q["opacity_logits"]), q["sh"], q["viewmat"],
                                                   q["lin_vel"], q["ang_vel"], background=bg.double(), return_parts=True)
    good = ~frag
    check_fragile(frag, f"exact-rs S={S} {W}x{H} n={n} base={base}")
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)) * good[..., None]
    (ref * wt.double()).sum().backward()
    p = {k: sc[k].float().to(dev).requires_grad_(True) for k in names}
    times, _, _ = gs.subpose_schedule(S, et, 1, 0.0)
    times_t = torch.tensor(times, device=dev)
    saved = ops.SLICE_BASE
    try:
        if base is not None:
            ops.SLICE_BASE = base
        samples, alphas, radii = gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"],
                                                    torch.sigmoid(p["opacity_logits"]), p["sh"], p["viewmat"], bg.to(dev),
                                                    S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, sh_degree=3,
                                                    lin_vel=p["lin_vel"], ang_vel=p["ang_vel"], times=times_t,
                                                    rolling_shutter_time=rt)
        out = gs.combine_samples(samples, gamma, mlevel)
        (out * wt.to(dev)).sum().backward()
        n_slices = sum(1 for v in ops.last_slice_intersects if int(v) > 0)
        n_isect = ops.last_num_intersects
    finally:
        ops.SLICE_BASE = saved
    if base is not None:
        assert n_slices >= 3, n_slices
    # integers: the float32 oracle sweeps the same float32 projection
    pr0 = O.project_gaussians(sc["means"], sc["log_scales"].exp(), 1.0, sc["quats"], sc["viewmat"], sc["fx"], sc["fy"],
                              sc["cx"], sc["cy"], H, W, keep_offscreen=True)
    pv = O.pixel_velocity(sc["means"], sc["viewmat"], sc["fx"], sc["fy"], sc["lin_vel"], sc["ang_vel"], 0.01, W, H)
    geom = (pr0.radii > 0).float()[:, None]
    total = 0
    for s_i, tau in enumerate(times):
        prs = O._bounds_swept(pr0, (pr0.xys + torch.tensor(tau, dtype=torch.float32) * pv) * geom, pv * geom, 0.5 * rt, H, W)
        assert np.array_equal(radii[s_i].cpu().numpy(), prs.radii.numpy()), s_i
        total += int(prs.num_tiles_hit.sum())
    assert n_isect == total                                      # swept tile counts, bit-exact
    assert (samples.detach().cpu().double() - ref_samples)[:, good].abs().max().item() < IMG_ATOL
    assert (out.detach().cpu().double() - ref.detach())[good].abs().max().item() < 5e-4
    # the bands model with one band is NOT this image: the row term is really there
    with torch.no_grad():
        flat, _, _ = gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]),
                                        p["sh"], p["viewmat"], bg.to(dev), S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"],
                                        H, W, sh_degree=3, lin_vel=p["lin_vel"], ang_vel=p["ang_vel"], times=times_t)
    assert (flat - samples.detach()).abs().max().item() > 0.05
    worst = {}
    for k in names:
        g_hip, g_ref = p[k].grad.cpu().numpy(), q[k].grad.numpy()
        if k == "viewmat":
            g_hip, g_ref = g_hip[:3], g_ref[:3]
        worst[k] = grad_el_ratio(g_hip, g_ref)
    print(f"exact rolling shutter S={S}: per-element gradient error / tolerance:",
          {k: round(v, 3) for k, v in worst.items()}, "slices", n_slices)
    for k, v in worst.items():
        assert v <= 1.0, (k, v)


@pytest.mark.parametrize("S,W,H,n,rt,base", [(3, 144, 128, 2500, 1 / 30, None), (5, 128, 160, 3000, 0.0, None),
                                             (2, 96, 128, 6000, 1 / 30, 8)])
def test_shared_list_pixel_velocity_vs_oracle(gs, oracle, dev, S, W, H, n, rt, base):
    """VERDICT round 3 'What's missing 4': ONE binning for the S blur samples of a pixel-velocity frame
    (render_subposes(shared_list=True): one projection at the centre of the sampled span, tile boxes swept over
    span + readout, one depth sort, one tile list; csrc/raster_rs.hip walks it once per sample with
    xy + (t_s - t_c + tau(y)) * velocity; the backward writes one gradient tuple per (entry, sample)).  Against the
    oracle's shared_list mode: the swept box / radii / intersection count (integers, exact), the sample images, the
    combined image and every gradient; with and without the rolling shutter; last case: several depth slices.  And
    against the per-sample lists: the same picture up to the beyond-3-sigma fringe."""
    from gsdeblur_amd import ops
    O = oracle
    sc = O.synthetic_scene(n, W, H, seed=950 + S, scale_mult=6.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 30, sc["ang_vel"] * 15
    et, gamma, mlevel = 1 / 60, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=1, exposure_time=et,
                         rolling_shutter_time=rt, gamma=gamma, min_rgb_level=mlevel, motion_model="pixel_velocity",
                         rs_exact=rt != 0.0, shared_list=True)
    q = {k: sc[k].double().requires_grad_(True) for k in names}
    ref, _, ref_samples, frag, parts, _ = O.render(cfg, q["means"], q["log_scales"].exp(), q["quats"],
                                                   torch.sigmoid(q["opacity_logits"]), q["sh"], q["viewmat"],
                                                   q["lin_vel"], q["ang_vel"], background=bg.double(), return_parts=True)
    good = ~frag
    check_fragile(frag, f"shared-list S={S} {W}x{H} n={n} rt={rt:.4f} base={base}")
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)) * good[..., None]
    (ref * wt.double()).sum().backward()
    p = {k: sc[k].float().to(dev).requires_grad_(True) for k in names}
    times, _, _ = gs.subpose_schedule(S, et, 1, 0.0)
    times_t = torch.tensor(times, device=dev)
    saved = ops.SLICE_BASE

    def run(shared):
        return gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]),
                                  p["sh"], p["viewmat"], bg.to(dev), S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
                                  sh_degree=3, lin_vel=p["lin_vel"], ang_vel=p["ang_vel"], times=times_t,
                                  rolling_shutter_time=rt, shared_list=shared)
    try:
        if base is not None:
            ops.SLICE_BASE = base
            # nothing in a multi-slice shared-list frame may depend on what its arena held before (a
            # finished sample's tile used to leave this slice's stop indices unwritten: 0x7f7f7f7f = 2.1e9 as an index)
            ops.release_arenas()
            ops._arena_pool[(str(dev), torch.cuda.current_stream(dev).cuda_stream)] = [
                torch.full((512 << 20,), 0x7f, dtype=torch.uint8, device=dev)]
        samples, alphas, radii = run(True)
        out = gs.combine_samples(samples, gamma, mlevel)
        (out * wt.to(dev)).sum().backward()
        n_slices = sum(1 for v in ops.last_slice_intersects if int(v) > 0)
        n_isect = ops.last_num_intersects
    finally:
        ops.SLICE_BASE = saved
    if base is not None:
        assert n_slices >= 3, n_slices
    # integers: ONE swept box per Gaussian — the oracle's list is the list the library walked
    pr = parts[0][0]
    assert radii.shape == (1, n)
    assert np.array_equal(radii[0].cpu().numpy(), pr.radii.numpy())
    assert n_isect == int(pr.num_tiles_hit.sum())
    assert (samples.detach().cpu().double() - ref_samples)[:, good].abs().max().item() < IMG_ATOL
    assert (out.detach().cpu().double() - ref.detach())[good].abs().max().item() < 5e-4
    worst = {}
    for k in names:
        g_hip, g_ref = p[k].grad.cpu().numpy(), q[k].grad.numpy()
        if k == "viewmat":
            g_hip, g_ref = g_hip[:3], g_ref[:3]
        worst[k] = grad_el_ratio(g_hip, g_ref)
    print(f"shared list S={S} rt={rt:.4f}: per-element gradient error / tolerance:",
          {k: round(v, 3) for k, v in worst.items()}, "slices", n_slices, "intersections", n_isect)
    for k, v in worst.items():
        assert v <= 1.0, (k, v)
    # per-sample lists: S binnings of per-sample boxes; same picture up to the fringe the swept box keeps
    with torch.no_grad():
        per_sample, _, radii_ps = run(False)
        per_isect = ops.last_num_intersects
    d = (per_sample - samples.detach()).abs()
    print(f"shared list vs per-sample lists: max {d.max().item():.4f} mean {d.mean().item():.2e}; "
          f"intersections {n_isect} vs {per_isect}")
    assert radii_ps.shape == (S, n) and n_isect < per_isect
    assert d.mean().item() < 2e-3 and d.max().item() < 0.08


def test_adaptive_slice_budget(gs, dev):
    """ops.SLICE_ADAPT (default on in the product, off in this suite's fixture): a frame that issued two or more depth
    slices doubles the first slice's budget for the next frames of that shape, up to 8x; a frame that stops after its
    first slice leaves it alone.  The image does not depend on the slicing (bit for bit), the gradients only through the
    order of fp32 sums."""
    from gsdeblur_amd import ops
    n, W, H, S = 60000, 160, 112, 2
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=23, scale_mult=8.0, profile="trained"), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
    vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"], sc["ang_vel"], torch.tensor(times, device=dev))
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    saved = (ops.SLICE_ADAPT, ops.SLICE_BASE)
    hints = ops.FrameHints()
    frames = []
    try:
        ops.SLICE_ADAPT, ops.SLICE_BASE = 1, 2
        for _ in range(4):
            p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
            rgb = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]),
                                     p["sh"], vms, None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2,
                                     hints=hints)[0]
            (rgb * wt).sum().backward()
            frames.append((rgb.detach().clone(), {k: v.grad.clone() for k, v in p.items()},
                           sum(1 for v in ops.last_slice_intersects if int(v) > 0), hints.mult))
        # a scene that needs one slice at the same budget: nothing grows.  No hints handed in: the default owner is keyed
        # by shape AND parameter storage, so the same tensors find the same hints frame after frame
        few = to_dev(gs.data.synthetic_scene(300, W, H, seed=5, scale_mult=2.0), dev)
        ops.SLICE_BASE = 512
        for _ in range(3):
            gs.render_combined(few["means"], few["log_scales"].exp(), few["quats"], torch.sigmoid(few["opacity_logits"]),
                               few["sh"], vms, None, S, 1, few["fx"], few["fy"], few["cx"], few["cy"], H, W, gamma=2.2)
        h_few = ops.hints_for((str(dev), 300, S, S, H, W, False, few["means"].untyped_storage().data_ptr()))
        assert h_few.mult == 1 and h_few.frames == 3 and h_few.settled
    finally:
        ops.SLICE_ADAPT, ops.SLICE_BASE = saved
    mults, slices = [f[3] for f in frames], [f[2] for f in frames]
    print("adaptive slice budget: multiplier after each frame", mults, "issued slices", slices)
    assert mults == [2, 4, 8, 8] and slices[0] >= 2, (mults, slices)
    for img, grads, _, _ in frames[1:]:
        assert torch.equal(img, frames[0][0])
        for k in grads:
            assert grad_el_ratio(grads[k].cpu().numpy(), frames[0][1][k].cpu().numpy()) <= 1.0, k


def test_model_exact_rolling_shutter_mode(gs, oracle, dev):
    """SplatfactoDeblurConfig(rolling_shutter_mode='exact', motion_model='pixel_velocity'): get_outputs renders with
    the continuous row time (one sub-pose per blur sample: radii is [S, N] whatever rs_bands says) and converges to
    what many bands give; the SE(3) model refuses the mode."""
    O = oracle
    n, W, H = 3000, 128, 160
    sc = O.synthetic_scene(n, W, H, seed=31, scale_mult=6.0)
    c2w = torch.eye(4)[:3].clone()
    c2w[:, 1] *= -1
    c2w[:, 2] *= -1
    cam = gs.Camera(c2w, sc["fx"], sc["fy"], sc["cx"], sc["cy"], W, H,
                    metadata=dict(cam_idx=0, camera_linear_velocity=[2.5, 0.6, 0.0],
                                  camera_angular_velocity=[0.0, 1.5, 0.6], exposure_time=1 / 60,
                                  rolling_shutter_time=1 / 30))
    imgs = {}
    for mode, bands in (("exact", 10), ("bands", 10), ("bands", 1)):
        cfg = gs.SplatfactoDeblurConfig(blur_samples=2, rs_bands=bands, gamma=2.2, min_rgb_level=10.0,
                                        motion_model="pixel_velocity", rolling_shutter_mode=mode)
        m = gs.SplatfactoDeblurModel.from_scene(cfg, sc, dev)
        out = m.get_outputs_for_camera(cam)
        imgs[(mode, bands)] = out["rgb"]
        assert m.radii.shape == ((2, n) if mode == "exact" else (2 * bands, n))
        assert out["depth"].shape == (H, W, 1) and torch.isfinite(out["depth"]).all()
    e10 = (imgs[("exact", 10)] - imgs[("bands", 10)]).abs().mean().item()
    e1 = (imgs[("exact", 10)] - imgs[("bands", 1)]).abs().mean().item()
    assert e10 < 0.35 * e1 and e1 > 1e-3, (e10, e1)
    with pytest.raises(ValueError, match="pixel_velocity"):
        bad = gs.SplatfactoDeblurModel.from_scene(gs.SplatfactoDeblurConfig(rolling_shutter_mode="exact"), sc, dev)
        bad.get_outputs_for_camera(cam)


@pytest.mark.gpu
def test_reference_velocity_fixtures_through_the_hip_subposes(gs, dev):
    """tests/golden/ref_add_velocities.json and ref_combine.json hold camera-frame velocities the REFERENCE computed
    (render_video.py::add_velocities, combine.py::process; generator: tests/golden/make_reference_fixtures.py).
    Through the model's pose / velocity handling and gs_subpose_viewmats_fwd — the kernel every frame's sub-poses come
    from — each frame's pose is carried onto its neighbours' poses; every sign / axis flip fails."""
    import test_reference_fixtures as RF

    def sub(V, lin, ang, times):
        return gs.subpose_viewmats(V.to(dev).float(), lin.to(dev).float(), ang.to(dev).float(),
                                   torch.tensor(times, device=dev, dtype=torch.float32)).cpu()
    n = 0
    for tag, c2w, lin, ang, nb in RF.neighbour_cases():
        RF.check_frame(gs, sub, c2w, lin, ang, nb, tag)
        n += 1
    for ci, case in enumerate(RF.combine_cases()):
        frames = sorted(case["combined_transforms"]["frames"], key=lambda fr: fr["file_path"])
        for i in range(1, len(frames) - 1):
            nb = [(-1, frames[i - 1]["transform_matrix"]), (1, frames[i + 1]["transform_matrix"])]
            RF.check_frame(gs, sub, frames[i]["transform_matrix"], frames[i]["camera_linear_velocity"],
                           frames[i]["camera_angular_velocity"], nb, f"combine{ci}/frame{i}")
            n += 1
    assert n >= 20


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["se3", "pixel_velocity"])
def test_raw_parameters_equal_activated_parameters(gs, oracle, dev, model):
    """Round 4: the projection kernels take splatfacto's RAW parameters (log-scales, opacity logits, features_dc +
    features_rest as two pointers; gs_project_fused_fwd param_flags / sh_rest) and return the gradients of what was
    handed in; the backward zero-fills its own outputs.  Against the round-3 route through torch (exp, sigmoid, cat in
    front of the op, their autograd behind it): same image (the kernel's expf / sigmoid against torch's: a last-bit
    difference in a scale may move a threshold pixel), gradients of the raw parameters within fp32 rounding — including
    needles (double-precision chain) and Gaussians without any gradient (exact zeros)."""
    O = oracle
    n, W, H, S = 6000, 176, 128, 3
    sc = O.synthetic_scene(n, W, H, seed=77, scale_mult=6.0)
    sc["log_scales"] = sc["log_scales"].clone()
    sc["log_scales"][::40, 0] += 2.5                                   # needles: scale ratio > 8
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
    tt = torch.tensor(times, device=dev)
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(4)).to(dev)
    names = ["means", "log_scales", "quats", "opacity_logits", "sh"]
    res = []
    for raw in (False, True):
        p = {k: sc[k].float().to(dev).clone().requires_grad_(True) for k in names}
        dc = p["sh"].detach()[:, 0, :].clone().requires_grad_(True)
        rest = p["sh"].detach()[:, 1:, :].clone().requires_grad_(True)
        V, lin, ang = (sc[k].float().to(dev) for k in ("viewmat", "lin_vel", "ang_vel"))
        kw = dict(gamma=2.2, min_rgb_level=10.0, return_alpha=False)
        if model == "pixel_velocity":
            vm, kw2 = V, dict(lin_vel=lin, ang_vel=ang, times=tt)
        else:
            vm, kw2 = gs.subpose_viewmats(V, lin, ang, tt), {}
        if raw:
            rgb, _, _ = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], dc, vm, None, S, 1,
                                           sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, sh_rest=rest, raw_params=True,
                                           **kw, **kw2)
        else:
            rgb, _, _ = gs.render_combined(p["means"], p["log_scales"].exp(), p["quats"],
                                           torch.sigmoid(p["opacity_logits"]), torch.cat([dc[:, None, :], rest], dim=1),
                                           vm, None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, **kw, **kw2)
        rgb.backward(wt)
        g = {k: p[k].grad.clone() for k in names if k != "sh"}
        g["features_dc"], g["features_rest"] = dc.grad.clone(), rest.grad.clone()
        res.append((rgb.detach().clone(), g))
    (img_a, g_a), (img_r, g_r) = res
    images_close(img_r, img_a, 5e-6, f"raw vs activated parameters, {model}", frac_max=2e-4)
    for k in g_a:
        assert g_r[k].shape == g_a[k].shape, k
        r = rel_max(g_r[k].cpu(), g_a[k].cpu())
        print(f"[kernel-vs-kernel raw-params grad {k} {model}] rel_max = {r:.3e}")
        assert r < 2e-5, (k, r)
        rows_a = (g_a[k].reshape(n, -1) != 0).any(dim=1)
        rows_r = (g_r[k].reshape(n, -1) != 0).any(dim=1)
        assert int((rows_a != rows_r).sum()) <= 2, k                   # untouched rows: exact zeros from the kernel's own fill
    assert 0.02 < float((g_a["means"] != 0).any(dim=1).float().mean()) < 0.98


# --------------------------------------------------------------------------- #
# the benchmark's own configurations at FULL size against oracle fixtures
# --------------------------------------------------------------------------- #
def _full_size_vs_oracle_fixture(gs, oracle, dev, name):
    """tests/golden/<name> (make_full_size_fixtures.py): for 64 sampled tiles of the 1M-Gaussian 1080p scene the ORACLE's
    complete tile lists (float32 projection: integers bit-exact) and its float64 composite of each tile.  Here the same
    scene goes through the HIP path at full size, twice:
      (a) projection + the unculled binning (depth pre-sort, emission of every bounding-box pair, stable tile sort, bin
          edges) + ONE compositor pass over the complete lists — the lists must equal the oracle's (length, checksum over
          the whole list, the ids a pixel can reach one by one), the composite must match colour / final T / stop index;
      (b) the default path (depth slices, exact tile culling, deferred colour): its sample images must match the same
          oracle tiles."""
    import hashlib
    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    import make_full_size_fixtures as MF
    from gsdeblur_amd import ops
    O = oracle
    d = np.load(Path(__file__).resolve().parent / "golden" / name)
    S, R, N, W, H = (int(d[k]) for k in ("S", "R", "N", "W", "H"))
    P = S * R
    sc = O.synthetic_scene(N, W, H, seed=int(d["seed"]))
    scales, opac = MF.activate(sc)
    mine = [f"{k}:{MF.tensor_hash(v)}" for k, v in (("means", sc["means"]), ("scales", scales), ("quats", sc["quats"]),
                                                    ("opacities", opac), ("sh", sc["sh"]))]
    same_inputs = mine == [str(x) for x in d["scene_hashes"]]
    print(f"[full-size {name}] scene tensors regenerated on this host are bit-identical to the fixture's: {same_inputs}")
    assert same_inputs, "the seeded scene differs from the one the fixture was made from (CPU-dependent torch kernels?)"
    means, quats, sh = (sc[k].to(dev) for k in ("means", "quats", "sh"))
    scales, opac = scales.to(dev), opac.to(dev)
    vms = torch.from_numpy(d["viewmats"]).to(dev)
    tiles = d["tiles"]
    tx_n, ty_n = (W + 15) // 16, (H + 15) // 16
    T = tx_n * ty_n
    L = gs._lib.load()
    # ---- (a) unculled lists + one compositor pass over them ----
    rec = torch.empty(P * N, ops.REC, device=dev)
    dk = torch.empty(P * N, dtype=torch.int32, device=dev)
    nt = torch.empty(P * N, dtype=torch.int32, device=dev)
    gs._lib.check(L.gs_project_fused_fwd(N, P, ops._ptr(means), ops._ptr(scales), 1.0, ops._ptr(quats), ops._ptr(opac),
                                         ops._ptr(sh), 16, 3, ops._ptr(vms), sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
                                         0.01, 1, 0, ops._ptr(rec), ops._ptr(dk), ops._ptr(nt), None, None, 0,
                                         ops._stream()), "fused")
    svals, bins, I, _ = gs.bin_and_sort_records(rec, dk, nt, P, N, H, W)
    assert I == int(d["tile_intersections_per_step"]), (I, int(d["tile_intersections_per_step"]))
    edges = ops._band_edges(H, R, dev)
    bg = torch.zeros(3, device=dev)
    out_img = torch.empty(S, H, W, 3, device=dev)
    out_T = torch.empty(S, H, W, device=dev)
    fidx = torch.empty(S, H, W, dtype=torch.int32, device=dev)
    gs._lib.check(L.gs_rasterize_fwd(ops._ptr(rec), ops._ptr(svals), ops._ptr(bins), ops._ptr(edges), ops._ptr(bg), S, R,
                                     H, W, ops._ptr(out_img), ops._ptr(out_T), ops._ptr(fidx), P * N, 0, ops._stream()),
                  "rasterize_fwd")
    # ---- (b) the default path ----
    with torch.no_grad():
        samples, _, _ = gs.render_subposes(means, scales, quats, opac, sh, vms, None, S, R, sc["fx"], sc["fy"], sc["cx"],
                                           sc["cy"], H, W, return_alpha=False)
    # ---- (b') the bench's STEADY state (round 6): the same frame three times through one FrameHints — the second frame
    # ranks only the pairs its first slice reaches (nearest-first selection) and projects records lazily, the third sorts
    # that selection under the second one's promise (one block per sub-pose finishes the sort).  Its sample images must
    # be the default path's bit for bit: what is compared with the oracle's tiles below is what bench.py times.
    h_steady = ops.FrameHints()
    steady = []
    for _ in range(3):
        with torch.no_grad():
            samples_steady, _, _ = gs.render_subposes(means, scales, quats, opac, sh, vms, None, S, R, sc["fx"], sc["fy"],
                                                      sc["cx"], sc["cy"], H, W, return_alpha=False, hints=h_steady)
        steady.append((int(ops.last_depth_select), int(h_steady.select_cap), len(ops.last_slice_intersects)))
    print(f"[full-size {name}] three frames through one FrameHints (selection state, promised size, slices): {steady}")
    assert torch.equal(samples_steady, samples)
    if R == 1:
        assert steady[1][0] == 1 and steady[2][0] == 1 and 0 < steady[1][1] <= 24576, steady
    del samples_steady
    bins_c, svals_c = bins.cpu().numpy(), None
    rows = O.band_tile_rows(H, R)
    times, samp, band = O.subpose_times(S, sc["exposure_time"], R, sc["rolling_shutter_time"])
    n_cmp = n_frag = n_px = 0
    worst = dict(a_rgb=0.0, a_T=0.0, b_rgb=0.0)
    for p in range(P):
        ty0, ty1 = rows[band[p]]
        s_img = samp[p]
        for ti, t in enumerate(tiles):
            ty, tx = divmod(int(t), tx_n)
            key = f"t{ti}_p{p}"
            if not (ty0 <= ty < ty1):
                assert key + "_n" not in d.files
                continue
            b0, b1 = (int(v) for v in bins_c[p * T + int(t)])
            n_list = int(d[key + "_n"])
            assert b1 - b0 == n_list, (key, b1 - b0, n_list)                    # list length: bit-exact
            ids = (svals[b0:b1].cpu().numpy().astype(np.int64) - p * N)
            assert MF.list_checksum(ids) == int(d[key + "_sum"]), key            # the WHOLE list, in order
            keep = d[key + "_ids"]
            assert np.array_equal(ids[:keep.size], keep.astype(np.int64)), key   # ... and its reachable head, id by id
            y0, x0 = ty * 16, tx * 16
            hh, ww = d[key + "_T"].shape
            frag = np.unpackbits(d[key + "_frag"])[:hh * ww].reshape(hh, ww).astype(bool)
            good = torch.from_numpy(~frag)
            ref_rgb = torch.from_numpy(d[key + "_rgb"])
            ref_T = torch.from_numpy(d[key + "_T"])
            a_rgb = out_img[s_img, y0:y0 + hh, x0:x0 + ww].cpu().double()
            a_T = out_T[s_img, y0:y0 + hh, x0:x0 + ww].cpu().double()
            a_stop = fidx[s_img, y0:y0 + hh, x0:x0 + ww].cpu().numpy().astype(np.int64) - b0
            b_rgb = samples[s_img, y0:y0 + hh, x0:x0 + ww].cpu().double()
            worst["a_rgb"] = max(worst["a_rgb"], float((a_rgb - ref_rgb).abs()[good].max()))
            worst["a_T"] = max(worst["a_T"], float((a_T - ref_T).abs()[good].max()))
            worst["b_rgb"] = max(worst["b_rgb"], float((b_rgb - ref_rgb).abs()[good].max()))
            # stop index (list position at which the pixel stops; list length when it never does): bit-exact off the
            # threshold pixels
            assert np.array_equal(a_stop[~frag], d[key + "_stop"].astype(np.int64)[~frag]), key
            n_cmp += 1
            n_frag += int(frag.sum())
            n_px += hh * ww
    print(f"[full-size {name}] {n_cmp} (tile, sub-pose) lists bit-exact; max |colour - oracle| unculled pass "
          f"{worst['a_rgb']:.2e}, default path {worst['b_rgb']:.2e}; max |T - oracle| {worst['a_T']:.2e}; "
          f"threshold pixels {n_frag} of {n_px}")
    assert n_cmp >= 64
    assert worst["a_rgb"] < IMG_ATOL and worst["b_rgb"] < IMG_ATOL and worst["a_T"] < 2e-5, worst
    check_fragile(np.array([n_frag / n_px]), f"full-size {name}")
    # ---- (c) GRADIENTS at full size (round 5) ----
    # d loss / d sample image = the fixture's seeded weights on the sampled tiles (zero on their fragile pixels), zero
    # everywhere else; the oracle's float64 gradients of the record fields (centre, conic, opacity, colour) of every
    # reachable entry of those tiles, summed per (sub-pose, Gaussian), against the rows the HIP backward writes —
    # (c1) the compat backward over the unculled lists, (c2) the default path's frame backward (depth slices, gradient
    # tuples + segmented reduce: the kernel bench.py's roofline is quoted on) — under both alpha-clamp conventions.
    v_img = torch.zeros(S, H, W, 3, dtype=torch.float64)
    rows_l = {"gu": [], "g": []}
    vals_l = {"gu": [], "g": []}
    for p in range(P):
        ty0, ty1 = rows[band[p]]
        for ti, t in enumerate(tiles):
            ty, tx = divmod(int(t), tx_n)
            key = f"t{ti}_p{p}"
            if not (ty0 <= ty < ty1):
                continue
            hh, ww = d[key + "_T"].shape
            frag = np.unpackbits(d[key + "_frag"])[:hh * ww].reshape(hh, ww).astype(bool)
            v_img[samp[p], ty * 16:ty * 16 + hh, tx * 16:tx * 16 + ww] = (
                MF.tile_weights(ti, p, hh, ww) * torch.from_numpy(~frag)[..., None])
            gu = d[key + "_gu"]
            g = d[key + "_g"] if key + "_g" in d.files else gu
            rws = p * N + d[key + "_ids"][:gu.shape[0]].astype(np.int64)
            for cv, arr in (("gu", gu), ("g", g)):
                rows_l[cv].append(rws)
                vals_l[cv].append(arr.astype(np.float64))
    v_img = v_img.float().to(dev)
    from gsdeblur_amd import step as step_mod
    comp = (("centre", slice(0, 2)), ("conic", slice(2, 5)), ("opacity", slice(5, 6)), ("colour", slice(6, 9)))

    def unpack(v_records, touched):
        n_rec = P * N
        vx, vc = torch.empty(n_rec, 2, device=dev), torch.empty(n_rec, 3, device=dev)
        vrgb, vo = torch.empty(n_rec, 3, device=dev), torch.empty(n_rec, 1, device=dev)
        gs._lib.check(L.gs_unpack_record_grads(n_rec, ops._ptr(v_records), ops._ptr(vx), ops._ptr(vc), ops._ptr(vrgb),
                                               ops._ptr(vo), ops._stream()), "unpack")
        allv = torch.cat([vx, vc, vo, vrgb], dim=1)
        if touched is not None:          # rows the frame backward never wrote hold whatever the allocation held
            allv = torch.where(touched.bool()[:, None], allv, torch.zeros_like(allv))
        return allv

    for cv, flags in (("gu", 7), ("g", 0)):
        rws = np.concatenate(rows_l[cv])
        uniq, inv = np.unique(rws, return_inverse=True)
        want = np.zeros((uniq.size, 9))
        np.add.at(want, inv, np.concatenate(vals_l[cv]))
        idx = torch.from_numpy(uniq).to(dev)
        with grad_convention(flags):
            # (c1) compat backward over the complete lists of pass (a)
            v_rec = torch.zeros(P * N, ops.GRAD, device=dev)
            gs._lib.check(L.gs_rasterize_bwd(ops._ptr(rec), ops._ptr(svals), ops._ptr(bins), ops._ptr(edges), ops._ptr(bg),
                                             S, R, H, W, ops._ptr(out_T), ops._ptr(fidx), ops._ptr(v_img), None,
                                             ops._ptr(v_rec), P * N, ops._bwd_variant(), ops._stream()), "rasterize_bwd")
            got_a = unpack(v_rec, None)[idx].cpu().numpy().astype(np.float64)
            # (c2) the default path: frame forward + frame backward, record gradients before the projection backward
            needs = [True] * 5 + [False] * 27
            ctx = step_mod._Ctx(needs)
            ops._RenderSubposes.forward(ctx, means, scales, quats, opac, sh, vms, None, S, R, sc["fx"], sc["fy"], sc["cx"],
                                        sc["cy"], H, W, 3, True, 1.0, 0.01, None, False, None, 0.0)
            saved = ctx.saved_tensors
            pre = ctx.prealloc
            ops.native_frame_backward(ctx.frame, saved[6], saved[10], saved[9], saved[11], v_img, None, pre["v_records"],
                                      pre["touched"], None)
            got_b = unpack(pre["v_records"], pre["touched"])[idx].cpu().numpy().astype(np.float64)
            n_slices_b = int(ctx.frame["state"].n_slices)
        msg = {}
        for tag, got in (("unculled lists", got_a), ("default path", got_b)):
            for cname, sl in comp:
                msg[f"{tag} {cname}"] = round(grad_el_ratio(got[:, sl], want[:, sl]), 3)
        nz = int((np.abs(want).sum(1) > 0).sum())
        print(f"[full-size {name}] gradients, convention {flags}: {uniq.size} (sub-pose, Gaussian) rows ({nz} non-zero), "
              f"default path in {n_slices_b} slice(s); per-element error / tolerance: {msg}")
        assert nz > 1000
        for k_, v in msg.items():
            assert v <= 1.0, (name, flags, k_, v)


@pytest.mark.gpu
def test_full_size_headline_vs_oracle_fixture(gs, oracle, dev):
    """bench.py's timed configuration (1M Gaussians, 1920x1080, 5 motion-blur sub-poses, seed 1234) against the oracle"""
    _full_size_vs_oracle_fixture(gs, oracle, dev, "full_size_headline.npz")


@pytest.mark.gpu
def test_full_size_config3_vs_oracle_fixture(gs, oracle, dev):
    """BASELINE.json config 3 (1M Gaussians, 1080p, 10 rolling-shutter row bands) against the oracle"""
    _full_size_vs_oracle_fixture(gs, oracle, dev, "full_size_config3.npz")


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["se3", "pixel_velocity"])
def test_render_step_equals_autograd_route(gs, oracle, dev, model):
    """gsdeblur_amd.render_step (forward + backward of a frame in one host call, no autograd engine; what bench.py times
    and train_step uses) issues the same C-ABI calls as ops.render_combined + Tensor.backward: same image, same
    gradients for every Gaussian parameter (bit for bit: deterministic tuple path), view matrix and velocities (their
    last reduction is a handful of fp32 atomics)."""
    O = oracle
    n, W, H, S = 5000, 160, 112, 3
    sc = O.synthetic_scene(n, W, H, seed=91, scale_mult=6.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
    tt = torch.tensor(times, device=dev)
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(4)).to(dev)
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "viewmat", "lin_vel", "ang_vel"]
    p = {k: sc[k].float().to(dev).clone().requires_grad_(True) for k in names}
    kw = dict(gamma=2.2, min_rgb_level=10.0, return_alpha=False, raw_params=True)
    if model == "pixel_velocity":
        rgb, _, _ = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"], p["viewmat"],
                                       None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, lin_vel=p["lin_vel"],
                                       ang_vel=p["ang_vel"], times=tt, **kw)
    else:
        vms = gs.subpose_viewmats(p["viewmat"], p["lin_vel"], p["ang_vel"], tt)
        rgb, _, _ = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"], vms, None, S,
                                       1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, **kw)
    rgb.backward(wt)
    q = {k: sc[k].float().to(dev).clone() for k in names}
    rgb2, g, radii = gs.render_step(q["means"], q["log_scales"], q["quats"], q["opacity_logits"], q["sh"], q["viewmat"],
                                    q["lin_vel"], q["ang_vel"], tt, None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
                                    lambda img: wt, gamma=2.2, min_rgb_level=10.0, motion_model=model)
    assert torch.equal(rgb2, rgb.detach()) and radii.shape == (S, n)
    for k, name in (("means", "means"), ("log_scales", "scales"), ("quats", "quats"), ("opacity_logits", "opacities"),
                    ("sh", "sh")):
        assert torch.equal(g[name], p[k].grad), k
    for k in ("viewmat", "lin_vel", "ang_vel"):
        assert rel_max(g[k].cpu(), p[k].grad.cpu()) < 1e-5, k


def test_render_step_callable_may_use_autograd(gs, oracle, dev):
    """render_step runs under no_grad; a grad_image callable that gets d loss / d rgb from torch.autograd
    (requires_grad_ + autograd.grad) must work — it sees a detached leaf with gradient recording switched on"""
    O = oracle
    n, W, H, S = 3000, 96, 64, 2
    sc = O.synthetic_scene(n, W, H, seed=12, scale_mult=6.0)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
    tt = torch.tensor(times, device=dev)
    q = {k: sc[k].float().to(dev) for k in ("means", "log_scales", "quats", "opacity_logits", "sh", "viewmat", "lin_vel",
                                              "ang_vel")}
    target = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(1)).to(dev)

    def by_autograd(rgb):
        rgb = rgb.requires_grad_(True)
        loss = ((rgb - target) ** 2).mean()
        (v,) = torch.autograd.grad(loss, rgb)
        return v

    args = (q["means"], q["log_scales"], q["quats"], q["opacity_logits"], q["sh"], q["viewmat"], q["lin_vel"], q["ang_vel"],
            tt, None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W)
    rgb_a, g_a, _ = gs.render_step(*args, by_autograd, gamma=2.2, min_rgb_level=10.0)
    rgb_b, g_b, _ = gs.render_step(*args, lambda rgb: 2.0 * (rgb - target) / rgb.numel(), gamma=2.2, min_rgb_level=10.0)
    assert torch.equal(rgb_a, rgb_b)
    for k in ("means", "scales", "quats", "opacities", "sh"):
        assert float(g_a[k].abs().max()) > 0 and rel_max(g_a[k].cpu(), g_b[k].cpu()) < 1e-6, k
    with pytest.raises(ValueError):
        gs.render_step(*args, lambda rgb: None, gamma=2.2, min_rgb_level=10.0)


def test_train_step_routes_agree(gs, dev):
    """training.train_step renders through model.render_and_backward -> step.render_step (ONE host
    call, the entry bench.py times); with GSD_TRAIN_AUTOGRAD it goes through model.get_outputs + loss.backward().  Same
    loss, same gradient for every Gaussian parameter, the pose / velocity adjustments and the learnable background, same
    parameters after the Adam step."""
    from gsdeblur_amd import train_step as T
    n, W, H, S = 6000, 128, 96, 3
    sc = gs.data.synthetic_scene(n, W, H, sh_degree=3, seed=31)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    c2w = torch.eye(4)[:3].clone()
    c2w[:, 1] *= -1
    c2w[:, 2] *= -1
    flip = torch.tensor([1.0, -1.0, -1.0])
    cam = gs.Camera(c2w, sc["fx"], sc["fy"], sc["cx"], sc["cy"], W, H,
                    metadata=dict(cam_idx=0, camera_linear_velocity=[float(v) for v in sc["lin_vel"] * flip],
                                  camera_angular_velocity=[float(v) for v in sc["ang_vel"] * flip],
                                  exposure_time=1 / 60, rolling_shutter_time=0.0))
    target = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)).to(dev)
    results = []
    saved = T.TRAIN_AUTOGRAD
    try:
        for autograd_route in (0, 1):
            T.TRAIN_AUTOGRAD = autograd_route
            cfg = gs.SplatfactoDeblurConfig(blur_samples=S, rolling_shutter_compensation=False, gamma=2.2, min_rgb_level=10.0,
                                            background_color="auto", use_scale_regularization=True)
            cfg.camera_optimizer.mode = "SO3xR3"
            cfg.camera_velocity_optimizer.enabled = True
            model = gs.SplatfactoDeblurModel.from_scene(cfg, sc, dev)
            model.collect_densify_stats = True
            assert T.one_call_route(model) == (not autograd_route)
            opts = T.make_optimizers(model)
            h = [T.train_step(model, opts, cam, target, 0.2) for _ in range(2)]
            grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
            params = {k: p.detach().clone() for k, p in model.named_parameters()}
            results.append((h, grads, params, model.xy_grad.clone(), model.radii.clone()))
    finally:
        T.TRAIN_AUTOGRAD = saved
    (h0, g0, p0, xy0, r0), (h1, g1, p1, xy1, r1) = results
    assert set(g0) == set(g1) and {"means", "scales", "quats", "opacities", "features_dc", "features_rest",
                                   "pose_adjustment", "velocity_adjustment", "background_param"} <= set(g0)
    for a, b in zip(h0, h1):
        assert abs(a["loss"] - b["loss"]) < 1e-6 and abs(a["psnr"] - b["psnr"]) < 1e-4
    assert torch.equal(r0, r1) and rel_max(xy0.cpu(), xy1.cpu()) < 1e-5
    for k in g0:
        assert float(g0[k].abs().max()) > 0, k
        assert rel_max(g0[k].cpu(), g1[k].cpu()) < 2e-5, k
        assert rel_max(p0[k].cpu(), p1[k].cpu()) < 2e-5, k


def test_alternating_scenes_of_one_shape_keep_their_frame_time(gs, dev):
    """two scenes of ONE shape (a headline-like one that stops after its first slice, a
    fitted-model-like one that wants a larger budget) rendered alternately, adaptive budget ON, nobody handing hints in:
    each scene finds its own FrameHints (keyed by parameter storage), both run in the SAME pooled arena, and once the
    budget and the arena have settled (four frames per scene at most) no frame pays for the other scene's frames —
    no arena retry, no new arena, no frame slower than 1.3x its scene's median."""
    import time
    from gsdeblur_amd import ops
    n, W, H, S = 200000, 640, 368, 3
    saved = (ops.SLICE_ADAPT, ops.SLICE_BASE)
    ops.release_arenas()
    try:
        ops.SLICE_ADAPT, ops.SLICE_BASE = 1, 64
        scenes = []
        for profile in ("survey", "trained"):
            sc = gs.data.synthetic_scene(n, W, H, sh_degree=3, seed=77, profile=profile)
            q = {k: sc[k].float().to(dev) for k in ("means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel")}
            scenes.append((sc, q))
        times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
        tt = torch.tensor(times, device=dev)
        wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(3)).to(dev)
        V = torch.eye(4, device=dev)
        ms = ([], [])
        slices = ([], [])
        for frame in range(24):
            i = frame % 2
            sc, q = scenes[i]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gs.render_step(q["means"], q["log_scales"], q["quats"], q["opacity_logits"], q["sh"], V, q["lin_vel"],
                           q["ang_vel"], tt, None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, wt, gamma=2.2,
                           min_rgb_level=10.0)
            torch.cuda.synchronize()
            ms[i].append((time.perf_counter() - t0) * 1e3)
            slices[i].append(len(ops.last_slice_intersects))
        hints = [ops.hints_for((str(dev), n, S, S, H, W, False, q["means"].untyped_storage().data_ptr())) for _, q in scenes]
        pool = ops._arena_pool[(str(dev), torch.cuda.current_stream(dev).cuda_stream)]
        print("alternating scenes: ms", [[round(v, 2) for v in m] for m in ms], "slices", slices,
              "budget multipliers", [h.mult for h in hints], "retries", [h.arena_retries for h in hints])
        assert hints[0] is not hints[1] and all(h.frames == 12 and h.settled for h in hints)
        assert hints[1].mult >= hints[0].mult                 # each scene learnt ITS budget
        assert len(pool) == 1                                 # one arena served all 24 frames
        # the structural facts above (own hints, one arena, no retry after settling) are deterministic; the wall-clock
        # one is not — a scheduling hiccup of the box must not fail the suite, a stall that comes from the code repeats:
        # frames that break the 1.3x bar get ONE more pass of twelve settled frames per scene
        def slow_frames(series):
            tail = sorted(series)
            med = tail[len(tail) // 2]
            return [v for v in series if v > 1.3 * med + 0.05]
        retimed = None
        if any(slow_frames(ms[i][4:]) for i in (0, 1)):
            retimed = ([], [])
            for frame in range(24):
                i = frame % 2
                sc, q = scenes[i]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gs.render_step(q["means"], q["log_scales"], q["quats"], q["opacity_logits"], q["sh"], V, q["lin_vel"],
                               q["ang_vel"], tt, None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, wt, gamma=2.2,
                               min_rgb_level=10.0)
                torch.cuda.synchronize()
                retimed[i].append((time.perf_counter() - t0) * 1e3)
            print("alternating scenes, re-timed:", [[round(v, 2) for v in m] for m in retimed])
        for i in (0, 1):
            series = ms[i][4:] if retimed is None else retimed[i]
            # (one delayed frame of twelve is the box; a frame that pays for the other scene's frames repeats every time)
            assert len(slow_frames(series)) <= (0 if retimed is None else 1), (i, ms[i], retimed)
        assert all(h.arena_retries <= 1 for h in hints) and len(pool) == 1
    finally:
        ops.SLICE_ADAPT, ops.SLICE_BASE = saved
        ops.release_arenas()


@pytest.mark.parametrize("S,rt", [(3, 1 / 30), (4, 0.0), (1, 1 / 30)])
def test_fork_style_keywords_on_the_compat_ops(gs, oracle, dev, S, rt):
    """VERDICT round 4 'missing 4' / item 8 (the fork's rasterizer takes the camera velocities as trailing
    keywords, /root/reference/README.md:196-200, train.py:46-70): project_gaussians(lin_vel=, ang_vel=, exposure_time=,
    rolling_shutter_time=, blur_samples=) -> 8th output pix_vels; rasterize_gaussians(pix_vels=, ...) renders the
    paper's model — one swept-box binning, the sample loop and the row time inside the compositor.  One frame through
    the two compat calls (+ spherical_harmonics, the way splatfacto chains them): radii / tile counts bit-exact against
    the oracle's shared_list mode and equal to render_subposes(shared_list=True)'s, samples / image / every gradient
    (Gaussians, view matrix, velocities) against the float64 oracle AND against the fused path.  Without the keywords
    the two ops are today's static ones bit for bit."""
    O = oracle
    W, H, n = 128, 96, 2500
    sc = O.synthetic_scene(n, W, H, seed=77 + S, scale_mult=6.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 30, sc["ang_vel"] * 15
    et, gamma, mlevel = 1 / 60, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=1, exposure_time=et,
                         rolling_shutter_time=rt, gamma=gamma, min_rgb_level=mlevel, motion_model="pixel_velocity",
                         rs_exact=rt != 0.0, shared_list=True)
    q = {k: sc[k].double().requires_grad_(True) for k in names}
    ref, _, ref_samples, frag, parts, _ = O.render(cfg, q["means"], q["log_scales"].exp(), q["quats"],
                                                   torch.sigmoid(q["opacity_logits"]), q["sh"], q["viewmat"],
                                                   q["lin_vel"], q["ang_vel"], background=bg.double(), return_parts=True)
    good = ~frag
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)) * good[..., None]
    (ref * wt.double()).sum().backward()

    def compat():
        p = {k: sc[k].float().to(dev).requires_grad_(True) for k in names}
        kw = dict(exposure_time=et, rolling_shutter_time=rt, blur_samples=S)
        qn = p["quats"] / p["quats"].norm(dim=-1, keepdim=True)               # splatfacto normalises before the call
        xys, depths, radii, conics, comp, ntiles, cov3d, pv = gs.project_gaussians(
            p["means"], p["log_scales"].exp(), 1.0, qn, p["viewmat"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
            lin_vel=p["lin_vel"], ang_vel=p["ang_vel"], **kw)
        Vd = p["viewmat"].detach()
        cam = -(Vd[:3, :3].T @ Vd[:3, 3])
        rgb = torch.clamp(gs.spherical_harmonics(3, p["means"].detach() - cam[None, :], p["sh"]) + 0.5, min=0.0)
        op = torch.sigmoid(p["opacity_logits"]) * comp
        samples, alphas = gs.rasterize_gaussians(xys, depths, radii, conics, ntiles, rgb, op[:, None], H, W, 16,
                                                 bg.to(dev), True, pix_vels=pv, return_samples=True, **kw)
        mean_img = gs.rasterize_gaussians(xys, depths, radii, conics, ntiles, rgb, op[:, None], H, W, 16, bg.to(dev),
                                          pix_vels=pv, **kw)
        out = gs.combine_samples(samples, gamma, mlevel)
        (out * wt.to(dev)).sum().backward()
        return p, samples.detach(), out.detach(), radii, ntiles, mean_img.detach(), cov3d.detach()

    def fused():
        p = {k: sc[k].float().to(dev).requires_grad_(True) for k in names}
        times, _, _ = gs.subpose_schedule(S, et, 1, 0.0)
        samples, alphas, radii = gs.render_subposes(p["means"], p["log_scales"].exp(), p["quats"],
                                                    torch.sigmoid(p["opacity_logits"]), p["sh"], p["viewmat"], bg.to(dev),
                                                    S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, sh_degree=3,
                                                    lin_vel=p["lin_vel"], ang_vel=p["ang_vel"],
                                                    times=torch.tensor(times, device=dev), rolling_shutter_time=rt,
                                                    shared_list=True)
        out = gs.combine_samples(samples, gamma, mlevel)
        (out * wt.to(dev)).sum().backward()
        return p, samples.detach(), out.detach(), radii

    pc, sam_c, out_c, radii_c, ntiles_c, mean_c, cov3d_c = compat()
    pf, sam_f, out_f, radii_f = fused()
    pr = parts[0][0]
    assert np.array_equal(radii_c.cpu().numpy(), pr.radii.numpy())            # integers against the oracle's swept boxes
    assert np.array_equal(ntiles_c.cpu().numpy(), pr.num_tiles_hit.numpy())
    assert torch.equal(radii_c, radii_f[0])
    assert (sam_c - sam_f).abs().max().item() < 2e-6 and (out_c - out_f).abs().max().item() < 2e-5
    assert (mean_c - sam_c.mean(dim=0)).abs().max().item() < 1e-6
    assert (sam_c.cpu().double() - ref_samples)[:, good].abs().max().item() < IMG_ATOL
    assert (out_c.cpu().double() - ref.detach())[good].abs().max().item() < 5e-4
    worst = {}
    for k in names:
        g_hip, g_ref, g_fused = pc[k].grad.cpu().numpy(), q[k].grad.numpy(), pf[k].grad.cpu().numpy()
        if k == "viewmat":
            g_hip, g_ref, g_fused = g_hip[:3], g_ref[:3], g_fused[:3]
        worst[k] = (round(grad_el_ratio(g_hip, g_ref), 3), round(rel_max(g_hip, g_fused), 6))
    print(f"fork keywords S={S} rt={rt:.4f}: (per-element error / tolerance vs oracle, rel. max vs fused path):", worst)
    for k, (a, b) in worst.items():
        assert a <= 1.0 and b < 2e-4, (k, a, b)
    # defaults: the static ops, bit for bit, 7 outputs
    with torch.no_grad():
        args = (sc["means"].to(dev), sc["log_scales"].exp().to(dev), 1.0, sc["quats"].to(dev), sc["viewmat"].to(dev),
                sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W)
        a = gs.project_gaussians(*args)
        b = gs.project_gaussians(*args, 16, 0.01, exposure_time=et, blur_samples=S)      # no velocities: still static
        assert len(a) == 7 and len(b) == 7 and all(torch.equal(x, y) for x, y in zip(a, b))
        assert torch.allclose(a[6], cov3d_c, rtol=1e-5, atol=1e-7)        # (exp on the CPU vs on the GPU: an ulp)
    with pytest.raises(ValueError):
        gs.project_gaussians(*args, lin_vel=sc["lin_vel"].to(dev))


@pytest.mark.parametrize("S,R,base", [(3, 1, 512), (3, 1, 24), (2, 5, 48), (1, 1, 0)])
def test_lazy_records_give_the_same_frame(gs, dev, S, R, base):
    """round 5: a frame projected WITHOUT records (gs_project_fused_fwd defer_color bit 4; GSD_LAZY_RECORDS) whose depth
    slices project the records of their own pairs (gs_slice_project_records) against the eager projection: image, alphas,
    radii and every Gaussian gradient bit for bit (the same slices hold the same rows), pose gradients to summation
    order.  The lazy frame runs first, on record memory poisoned with NaNs: a row nobody projected would show.
    One slice, many slices, rolling-shutter bands (band-aware), and SLICE_BASE = 0 (no plan: the frame stays eager)."""
    from gsdeblur_amd import ops
    n, W, H = 60000, 208, 176
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=23, scale_mult=3.0), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, R, 1 / 30 if R > 1 else 0.0)
    times_t = torch.tensor(times, device=dev)
    g = torch.Generator().manual_seed(8)
    wt, wa = torch.rand(H, W, 3, generator=g).to(dev), torch.rand(S, H, W, generator=g).to(dev)
    saved = (ops.LAZY_RECORDS, ops.SLICE_BASE, ops.SLICE_ADAPT)
    res = []
    try:
        ops.SLICE_BASE, ops.SLICE_ADAPT = base, 0
        for lazy in (2, 0):
            ops.LAZY_RECORDS = lazy
            ops.release_arenas()
            junk = torch.full((S * R * n * ops.REC,), float("nan"), device=dev)
            del junk
            p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
            lin = (sc["lin_vel"] * 5).clone().requires_grad_(True)
            ang = (sc["ang_vel"] * 3).clone().requires_grad_(True)
            V = sc["viewmat"].clone().requires_grad_(True)
            vms = gs.subpose_viewmats(V, lin, ang, times_t)
            rgb, alphas, radii = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"],
                                                    vms, None, S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
                                                    gamma=2.2, min_rgb_level=10.0, raw_params=True,
                                                    hints=ops.FrameHints())
            ((rgb * wt).sum() + (alphas * wa).sum()).backward()
            grads = {k: v.grad.clone() for k, v in p.items()}
            pose = dict(lin=lin.grad.clone(), ang=ang.grad.clone(), V=V.grad.clone())
            res.append((rgb.detach().clone(), alphas.detach().clone(), radii.clone(), grads, pose,
                        [int(v) for v in ops.last_slice_intersects if int(v) > 0]))
    finally:
        ops.LAZY_RECORDS, ops.SLICE_BASE, ops.SLICE_ADAPT = saved
    a, b = res
    print(f"lazy records S={S} R={R} base={base}: slices {a[5]}")
    assert a[5] == b[5] and (len(a[5]) >= 2 or base != 24)
    assert torch.isfinite(a[0]).all() and float(a[0].max()) > 0.05
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    for k in a[3]:
        assert float(a[3][k].abs().max()) > 0, k
        assert torch.equal(a[3][k], b[3][k]), (k, float((a[3][k] - b[3][k]).abs().max()))
    for k in a[4]:
        assert rel_max(a[4][k].cpu(), b[4][k].cpu()) < GRAD_RTOL, k


@pytest.mark.parametrize("model", ["se3", "pixel_velocity"])
def test_band_aware_projection_gives_the_same_frame(gs, dev, model):
    """round 5: with rolling-shutter bands the projection culls a (band, Gaussian) pair whose tile
    box misses the band's tile rows and clips a box that straddles them (gs_project_fused_fwd defer_color bit 2) — the
    depth pre-sort, the count scan and the slice plan then see a band's own pairs instead of R times as many.  Against
    the band-blind projection (GSD_BAND_AWARE=0): same image and alphas bit for bit, same radii (the un-banded
    definition), same gradients up to fp32 summation order (the slice boundaries move), several times fewer pairs."""
    from gsdeblur_amd import ops
    n, W, H, S, R = 60000, 208, 176, 2, 5
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=19, scale_mult=3.0), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, R, 1 / 30)
    times_t = torch.tensor(times, device=dev)
    g = torch.Generator().manual_seed(6)
    wt, wa = torch.rand(H, W, 3, generator=g).to(dev), torch.rand(S, H, W, generator=g).to(dev)
    saved = (ops.BAND_AWARE, ops.SLICE_BASE)
    res = []
    try:
        ops.SLICE_BASE = 48                                # several slices: the slice plan differs between the two
        for aware in (1, 0):
            ops.BAND_AWARE = aware
            p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
            lin = (sc["lin_vel"] * 5).clone().requires_grad_(True)
            ang = (sc["ang_vel"] * 3).clone().requires_grad_(True)
            V = sc["viewmat"].clone().requires_grad_(True)
            if model == "pixel_velocity":
                rgb, alphas, radii = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"],
                                                        V, None, S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
                                                        gamma=2.2, min_rgb_level=10.0, lin_vel=lin, ang_vel=ang,
                                                        times=times_t, raw_params=True)
            else:
                vms = gs.subpose_viewmats(V, lin, ang, times_t)
                rgb, alphas, radii = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"],
                                                        vms, None, S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
                                                        gamma=2.2, min_rgb_level=10.0, raw_params=True)
            ((rgb * wt).sum() + (alphas * wa).sum()).backward()
            grads = {k: v.grad.clone() for k, v in p.items()}
            grads.update(lin=lin.grad.clone(), ang=ang.grad.clone(), V=V.grad.clone())
            res.append((rgb.detach().clone(), alphas.detach().clone(), radii.clone(), grads, ops.last_num_intersects,
                        [int(v) for v in ops.last_slice_intersects if int(v) > 0]))
    finally:
        ops.BAND_AWARE, ops.SLICE_BASE = saved
    a, b = res
    print(f"band-aware projection ({model}): bounding-box pairs {a[4]} vs {b[4]} band-blind; emitted per slice {a[5]} vs {b[5]}")
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert 0 < a[4] < 0.5 * b[4] and sum(a[5]) <= sum(b[5]) and len(b[5]) >= 2
    for k in a[3]:
        assert float(a[3][k].abs().max()) > 0, k
        assert rel_max(a[3][k].cpu(), b[3][k].cpu()) < GRAD_RTOL, k
