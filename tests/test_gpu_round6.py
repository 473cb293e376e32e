"""GPU tests added in round 6 (same bars and fixtures as tests/test_gpu_parity.py: everything goes through the C ABI)."""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def to_dev(sc, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}


# --------------------------------------------------------------------------- #
# ADVICE round 5
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("n,num_bins", [(4000, 300), (70_000, 129)])
def test_bin_edges_ignore_keys_beyond_the_last_bin(gs, dev, n, num_bins):
    """gs_tile_bin_edges_u64 / _u32 take caller-supplied ids (ops.get_tile_bin_edges): a tile id >= num_bins owns no bin.
    ADVICE round 5: the self-zeroing kernel strode over the gap up to such a key and zero-filled far past the buffer.
    The bins sit in the middle of a poisoned guard region; every bin of a key < num_bins is exact, nothing outside moves."""
    from gsdeblur_amd import _lib
    from gsdeblur_amd.ops import _ptr, _stream
    L = _lib.load()
    g = torch.Generator().manual_seed(n)
    keys = torch.sort(torch.randint(0, num_bins, (n,), generator=g, dtype=torch.int64)).values
    stray = torch.tensor([num_bins, num_bins + 7, 5 * num_bins, 5 * num_bins, 2 ** 20 + 3], dtype=torch.int64)
    allk = torch.cat([keys, stray])
    guard = 4096
    for width in (64, 32):
        buf = torch.full((guard + num_bins + guard, 2), 0x7F7F7F7F, dtype=torch.int32, device=dev)
        bins = buf[guard:guard + num_bins]
        if width == 64:
            ids = ((allk << 32) | 12345).to(dev)
            _lib.check(L.gs_tile_bin_edges_u64(allk.numel(), _ptr(ids), num_bins, _ptr(bins), _stream()), "bin edges u64")
        else:
            ids = allk.to(torch.int32).to(dev)
            _lib.check(L.gs_tile_bin_edges_u32(allk.numel(), _ptr(ids), num_bins, _ptr(bins), None, _stream()), "bin edges u32")
        lo = torch.searchsorted(keys, torch.arange(num_bins), right=False)
        hi = torch.searchsorted(keys, torch.arange(num_bins), right=True)
        want = torch.stack([lo, hi], 1)
        want[hi == lo] = 0
        assert torch.equal(bins.cpu().long(), want), width
        assert bool((buf[:guard] == 0x7F7F7F7F).all()) and bool((buf[guard + num_bins:] == 0x7F7F7F7F).all()), width


def test_a_second_backward_of_one_frame_re_lends_its_arena(gs, dev):
    """a frame's arena goes back to the pool after its first backward; a second backward
    (retain_graph=True) takes it out of the pool again for its duration and gives the same gradients; once a LATER frame
    has leased the arena, the old frame's backward must refuse (lease numbers come from one global counter: an address
    handed out again by the allocator cannot make a stale frame look current)."""
    from gsdeblur_amd import ops
    n, W, H, S = 30000, 160, 96, 2
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=5, scale_mult=8.0), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
    vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"], sc["ang_vel"], torch.tensor(times, device=dev))
    ops.release_arenas()
    hints = ops.FrameHints()
    p = sc["means"].clone().requires_grad_(True)

    def render(q):
        return gs.render_combined(q, sc["log_scales"].exp(), sc["quats"], torch.sigmoid(sc["opacity_logits"]), sc["sh"],
                                  vms, None, S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2, hints=hints)[0]
    img = render(p)
    loss = img.sum()
    loss.backward(retain_graph=True)
    g1 = p.grad.clone()
    pool = ops._arena_pool[(str(dev), torch.cuda.current_stream(dev).cuda_stream)]
    assert len(pool) == 1
    p.grad = None
    loss.backward(retain_graph=True)                       # re-lends the pooled arena, returns it afterwards
    assert len(pool) == 1
    tol = 1e-5 * float(g1.abs().max())
    assert float((p.grad - g1).abs().max()) <= tol          # (same kernels; fp32 tuple sums in the same order)
    # a later frame leases the arena: the old frame may not touch it any more
    img2 = render(sc["means"].clone().requires_grad_(True))
    with pytest.raises(RuntimeError, match="later frame"):
        loss.backward()
    img2.sum().backward()
    ops.release_arenas()


# --------------------------------------------------------------------------- #
# nearest-first selection
# --------------------------------------------------------------------------- #
def _np_select_bound(keys, weights, budget):
    """reference: smallest multiple of 512 whose keys below it carry >= budget of the weights (numpy)"""
    import numpy as np
    vis = keys < 2 ** 31
    k, w = keys[vis], weights[vis].astype(np.int64)
    if w.sum() < budget:
        return 0x80000000, int(w.sum())
    order = np.argsort(k >> 9, kind="stable")
    b = (k >> 9)[order]
    cw = np.cumsum(w[order])
    first = int(np.searchsorted(cw, budget, side="left"))        # first element at which the cumulative weight reaches it
    return (int(b[first]) + 1) << 9, int(w.sum())


@pytest.mark.parametrize("P,N,budget,dist", [(1, 5000, 3000, "uniform"), (5, 100_003, 400_000, "uniform"),
                                              (3, 70_001, 10 ** 9, "uniform"), (4, 50_000, 1, "narrow"),
                                              (10, 30_000, 90_000, "narrow"), (2, 4096, 2 ** 33, "uniform")])
def test_depth_select_bound_and_selective_sort(gs, dev, P, N, budget, dist):
    """gs_depth_select + gs_segmented_sort_select_u32 at the C ABI against numpy: the bound of every segment (a multiple
    of 512; 0x80000000 when the segment carries less than the budget), the frame's total weight, the selection sorted
    stably with its packed / gathered counts, the complement [bound, culled) likewise, the source keys left intact."""
    import ctypes
    import numpy as np
    from gsdeblur_amd import _lib
    from gsdeblur_amd.ops import _ptr, _stream
    L = _lib.load()
    rng = np.random.default_rng(P * 1000 + N)
    n = P * N
    if dist == "uniform":
        z = rng.uniform(0.02, 60.0, n).astype(np.float32)
    else:
        z = (1.0 + rng.uniform(0, 1e-3, n)).astype(np.float32)      # the whole segment inside a few level-1 buckets
        z[rng.random(n) < 0.01] = np.float32(1.0)                   # ... with many equal keys
    keys = z.view(np.uint32).astype(np.int64)
    culled = rng.random(n) < 0.3
    keys[culled] = 0xFFFFFFFF
    w = np.where(rng.random(n) < 0.02, rng.integers(200, 600, n), rng.integers(1, 40, n)).astype(np.int64)
    w[culled] = 0
    dk = torch.from_numpy(keys.astype(np.uint32).view(np.int32)).to(dev)
    dw = torch.from_numpy(w.astype(np.int32)).to(dev)
    ws_b = L.gs_depth_select_workspace_bytes(P)
    ws = torch.zeros(ws_b, dtype=torch.uint8, device=dev)
    thr = torch.full((P,), -1, dtype=torch.int32, device=dev)
    grand = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.gs_depth_select(n, N, _ptr(dk), _ptr(dw), int(budget), _ptr(thr), _ptr(grand), _ptr(ws), ws_b, _stream()),
               "depth_select")
    got_thr = thr.cpu().numpy().view(np.uint32).astype(np.int64)
    tot = 0
    for p in range(P):
        want, seg_total = _np_select_bound(keys[p * N:(p + 1) * N], w[p * N:(p + 1) * N], budget)
        tot += seg_total
        assert int(got_thr[p]) == want, (p, hex(int(got_thr[p])), hex(want))
    assert (int(grand.item()) & 0xFFFFFFFF) == (tot & 0xFFFFFFFF)
    # the selective sort, both sides of the bound
    sort_ws_b = L.gs_segmented_sort_compact_workspace_bytes(n, N, 0, 31, 8)
    below_most = max(int(((keys[p * N:(p + 1) * N] < int(got_thr[p]))).sum()) for p in range(P))
    for side in ("below", "below, tail passes sized for the selection", "behind"):
        # tail_cap: a promise that no segment keeps more keys — the passes behind the compacting one are sized for it
        tail_cap = below_most + 5 if side.endswith("selection") else 0
        k0, v0, k1, v1, cnt_out = (torch.full((n,), 0x7F7F7F7F, dtype=torch.int32, device=dev) for _ in range(5))
        n_live = torch.zeros(P, dtype=torch.int32, device=dev)
        sws = torch.empty(sort_ws_b, dtype=torch.uint8, device=dev)
        res = ctypes.c_int(0)
        lo, hi = (None, thr) if side.startswith("below") else (thr, None)
        _lib.check(L.gs_segmented_sort_select_u32(n, N, _ptr(dk), _ptr(k0), _ptr(v0), _ptr(k1), _ptr(v1), 0, 31, 8,
                                                  0xFFFFFFFF, _ptr(lo), _ptr(hi), _ptr(n_live), _ptr(dw), _ptr(cnt_out),
                                                  _ptr(sws), sort_ws_b, ctypes.byref(res), tail_cap, _stream()), "sort_select")
        sk, sv = ((k1, v1) if res.value == 1 else (k0, v0))
        sk, sv, cc, nl = sk.cpu().numpy().view(np.uint32), sv.cpu().numpy(), cnt_out.cpu().numpy(), n_live.cpu().numpy()
        assert np.array_equal(dk.cpu().numpy().view(np.uint32).astype(np.int64), keys)       # source intact
        for p in range(P):
            seg = keys[p * N:(p + 1) * N]
            t = int(got_thr[p])
            keep = (seg < t) if side.startswith("below") else ((seg >= t) & (seg < 2 ** 31))
            idx = np.nonzero(keep)[0]
            order = idx[np.argsort(seg[idx], kind="stable")]
            m = order.size
            assert int(nl[p]) == m, (side, p, int(nl[p]), m)
            assert np.array_equal(sv[p * N:p * N + m].astype(np.int64), order + p * N), (side, p)
            assert np.array_equal(sk[p * N:p * N + m].astype(np.int64), seg[order]), (side, p)
            assert np.array_equal(cc[p * N:p * N + m].astype(np.int64), w[p * N:(p + 1) * N][order]), (side, p)


@pytest.mark.parametrize("spread", ["wide", "medium", "narrow", "equal"])
def test_one_block_tail_of_the_selective_sort(gs, dev, spread):
    """a promised selection of at most 24 576 keys per segment is finished by ONE block per segment
    (seg_tail_sort_kernel: registers + LDS, digits cut from key - smallest key) instead of three multi-block passes:
    against numpy's stable sort and bit-identical to the sort without the promise, over segments that keep 0, 1, 63, 1025,
    12 345 and exactly 24 576 keys; keys over the whole positive range (three 8-bit digits), over 2^26 ulps (two 9-bit
    digits: a selection's depth range), within a few thousand ulps (one digit) and all equal (none)."""
    import ctypes
    import numpy as np
    from gsdeblur_amd import _lib
    from gsdeblur_amd.ops import _ptr, _stream
    L = _lib.load()
    keeps = [0, 1, 63, 1025, 12_345, 24_576]
    P, N = len(keeps), 60_000
    n = P * N
    rng = np.random.default_rng({"wide": 1, "narrow": 2, "equal": 3, "medium": 4}[spread])
    keys = np.full(n, 0xFFFFFFFF, dtype=np.int64)
    for p, m in enumerate(keeps):
        where = rng.choice(N, m, replace=False) + p * N
        if spread == "wide":
            keys[where] = rng.integers(1, 2 ** 31 - 1, m)
        elif spread == "medium":
            keys[where] = 0x3F000000 + rng.integers(0, 2 ** 26, m)
        elif spread == "narrow":
            keys[where] = 0x3F800000 + rng.integers(0, 5000, m)
        else:
            keys[where] = 0x40490FDB
    w = rng.integers(1, 700, n).astype(np.int64)
    w[rng.random(n) < 0.01] = 9000                        # (beyond the packed field's 13 bits: these take the gather)
    dk = torch.from_numpy(keys.astype(np.uint32).view(np.int32)).to(dev)
    dw = torch.from_numpy(w.astype(np.int32)).to(dev)
    ws_b = L.gs_segmented_sort_compact_workspace_bytes(n, N, 0, 31, 8)
    got = {}
    for tail_cap in (24_576, 0):
        k0, v0, k1, v1, cnt_out = (torch.full((n,), 0x7F7F7F7F, dtype=torch.int32, device=dev) for _ in range(5))
        n_live = torch.zeros(P, dtype=torch.int32, device=dev)
        sws = torch.empty(ws_b, dtype=torch.uint8, device=dev)
        res = ctypes.c_int(-1)
        _lib.check(L.gs_segmented_sort_select_u32(n, N, _ptr(dk), _ptr(k0), _ptr(v0), _ptr(k1), _ptr(v1), 0, 31, 8,
                                                  0xFFFFFFFF, None, None, _ptr(n_live), _ptr(dw), _ptr(cnt_out),
                                                  _ptr(sws), ws_b, ctypes.byref(res), tail_cap, _stream()), "sort_select")
        sk, sv = ((k1, v1) if res.value == 1 else (k0, v0))
        got[tail_cap] = (sk.cpu().numpy().view(np.uint32).astype(np.int64), sv.cpu().numpy().astype(np.int64),
                         cnt_out.cpu().numpy().astype(np.int64), n_live.cpu().numpy())
    for tail_cap, (sk, sv, cc, nl) in got.items():
        assert list(nl) == keeps
        for p, m in enumerate(keeps):
            seg = keys[p * N:(p + 1) * N]
            idx = np.nonzero(seg != 0xFFFFFFFF)[0]
            order = idx[np.argsort(seg[idx], kind="stable")]
            assert np.array_equal(sv[p * N:p * N + m], order + p * N), (tail_cap, p)
            assert np.array_equal(sk[p * N:p * N + m], seg[order]), (tail_cap, p)
            assert np.array_equal(cc[p * N:p * N + m], w[p * N:(p + 1) * N][order]), (tail_cap, p)


def _saturating_scene(gs, dev, n, W, H, seed=23):
    """the seeded scene with its nearest 3000 Gaussians made large and opaque: every pixel stops within a few dozen
    entries, i.e. within the first depth slice of any budget — the shape of the benchmark frame at test size"""
    sc = gs.data.synthetic_scene(n, W, H, seed=seed, scale_mult=3.0)
    near = torch.argsort(sc["means"][:, 2])[:3000]
    sc["log_scales"] = sc["log_scales"].clone()
    sc["opacity_logits"] = sc["opacity_logits"].clone()
    sc["log_scales"][near] += 2.5
    sc["opacity_logits"][near] = 10.0
    return to_dev(sc, dev)


def _render_both_ways(gs, dev, sc, S, R, H, W, base, lazy, model="se3", shared=False):
    """one frame with the nearest-first selection forced on (GSD_DEPTH_SELECT=2) and one with it off: results + states"""
    from gsdeblur_amd import ops
    times, _, _ = gs.subpose_schedule(S, 1 / 60, R, 1 / 30 if R > 1 else 0.0)
    times_t = torch.tensor(times, device=dev)
    g = torch.Generator().manual_seed(8)
    wt, wa = torch.rand(H, W, 3, generator=g).to(dev), torch.rand(S, H, W, generator=g).to(dev)
    saved = (ops.DEPTH_SELECT, ops.SLICE_BASE, ops.SLICE_ADAPT, ops.LAZY_RECORDS)
    res = []
    try:
        ops.SLICE_BASE, ops.SLICE_ADAPT, ops.LAZY_RECORDS = base, 0, lazy
        for sel in (2, 0):
            ops.DEPTH_SELECT = sel
            ops.release_arenas()
            p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
            lin = (sc["lin_vel"] * 5).clone().requires_grad_(True)
            ang = (sc["ang_vel"] * 3).clone().requires_grad_(True)
            V = sc["viewmat"].clone().requires_grad_(True)
            kw = dict(gamma=2.2, min_rgb_level=10.0, raw_params=True, hints=ops.FrameHints())
            if model == "se3":
                vms = gs.subpose_viewmats(V, lin, ang, times_t)
                rgb, alphas, radii = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"],
                                                        vms, None, S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, **kw)
            else:
                rgb, alphas, radii = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"],
                                                        V, None, S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W,
                                                        lin_vel=lin, ang_vel=ang, times=times_t, shared_list=shared, **kw)
            ((rgb * wt).sum() + (alphas * wa).sum()).backward()
            res.append(dict(rgb=rgb.detach().clone(), alphas=alphas.detach().clone(), radii=radii.clone(),
                            grads={k: v.grad.clone() for k, v in p.items()},
                            pose=dict(lin=lin.grad.clone(), ang=ang.grad.clone(), V=V.grad.clone()),
                            slices=[int(v) for v in ops.last_slice_intersects if int(v) > 0],
                            state=ops.last_depth_select, n_total=ops.last_num_intersects))
    finally:
        ops.DEPTH_SELECT, ops.SLICE_BASE, ops.SLICE_ADAPT, ops.LAZY_RECORDS = saved
        ops.release_arenas()
    return res


@pytest.mark.parametrize("tag,S,R,base,lazy,profile,model,shared,want_state", [
    ("one slice", 3, 1, 512, 0, "saturating", "se3", False, 1),
    ("one slice, lazy records", 3, 1, 512, 2, "saturating", "se3", False, 1),
    ("the first slice leaves tiles open", 3, 1, 24, 0, "survey", "se3", False, 2),
    ("the first slice leaves tiles open, lazy records", 2, 1, 16, 2, "trained", "se3", False, 2),
    ("rolling-shutter bands", 2, 5, 48, 2, "survey", "se3", False, 2),
    ("pixel velocity, per-sample lists", 3, 1, 32, 0, "survey", "pixel_velocity", False, 2),
    ("pixel velocity, shared list", 3, 1, 32, 0, "survey", "pixel_velocity", True, 2),
    ("a frame that fits its budget whole", 2, 1, 100_000, 0, "survey", "se3", False, 1)])
def test_nearest_first_selection_gives_the_same_frame(gs, dev, tag, S, R, base, lazy, profile, model, shared, want_state):
    """round 6: the depth pre-sort restricted to the pairs the first slice's budget reaches (gs_frame_desc.depth_select)
    against the full pre-sort — image, alphas, radii bit for bit, the frame's pair total, every gradient (bit for bit
    when the frame is one slice on both sides: the same Gaussians' tuples are summed in the same order; to fp32 summation
    order when slice boundaries move).  Cases: one slice; a first slice that leaves tiles open, so that the pairs behind
    the selection are sorted and planned afterwards (state 2); rolling-shutter bands; the pixel-velocity model with
    per-sample and shared lists; a frame whose pairs all fit the budget (bound = everything)."""
    n, W, H = 60000, 208, 176
    if profile == "saturating":
        sc = _saturating_scene(gs, dev, n, W, H)
    else:
        sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=23, scale_mult=3.0, profile=profile), dev)
    a, b = _render_both_ways(gs, dev, sc, S, R, H, W, base, lazy, model, shared)
    print(f"nearest-first selection [{tag}]: slices {a['slices']} (selection) vs {b['slices']} (full sort), "
          f"state {a['state']}, pairs {a['n_total']}")
    assert a["state"] == want_state and b["state"] == 0, (a["state"], b["state"])
    assert a["n_total"] == b["n_total"]
    assert torch.isfinite(a["rgb"]).all() and float(a["rgb"].max()) > 0.05
    assert torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["alphas"], b["alphas"]) and torch.equal(a["radii"], b["radii"])
    one_slice = len(a["slices"]) == 1 and len(b["slices"]) == 1
    for k in a["grads"]:
        ga, gb = a["grads"][k], b["grads"][k]
        assert float(ga.abs().max()) > 0, k
        if one_slice and model == "se3":
            assert torch.equal(ga, gb), (k, float((ga - gb).abs().max()))
        else:
            assert float((ga - gb).abs().max()) <= 3e-5 * float(gb.abs().max()), (k, float((ga - gb).abs().max()))
    for k in a["pose"]:
        d = float((a["pose"][k] - b["pose"][k]).abs().max())
        assert d <= 3e-3 * float(b["pose"][k].abs().max()) + 1e-12, (k, d)


def test_frame_hints_switch_the_selection_on_and_off(gs, dev):
    """FrameHints.depth_select(): the first frame of a scene sorts everything; a scene whose last frame stopped in ONE
    slice holding under half of its pairs selects from the next frame on; a frame whose selection turns out short
    (state 2) switches it off again (and, with SLICE_ADAPT, grows the budget) — all through ONE hints object, the way
    SplatfactoDeblurModel owns one."""
    from gsdeblur_amd import ops
    n, W, H, S = 60000, 208, 176, 2
    sc = _saturating_scene(gs, dev, n, W, H)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, 1, 0.0)
    vms = gs.subpose_viewmats(sc["viewmat"], sc["lin_vel"], sc["ang_vel"], torch.tensor(times, device=dev))
    saved = (ops.DEPTH_SELECT, ops.SLICE_BASE, ops.SLICE_ADAPT)

    def frame(h):
        with torch.no_grad():
            img = gs.render_combined(sc["means"], sc["log_scales"], sc["quats"], sc["opacity_logits"], sc["sh"], vms, None, S, 1,
                                     sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2, raw_params=True, hints=h)[0]
        return img, ops.last_depth_select, len([v for v in ops.last_slice_intersects if int(v) > 0])
    try:
        ops.DEPTH_SELECT, ops.SLICE_ADAPT = 1, 1
        ops.SLICE_BASE = 512
        h = ops.FrameHints()
        img0, st0, k0 = frame(h)
        assert st0 == 0 and k0 == 1 and h.depth_select()            # first frame: full sort; it stopped in one small slice
        img1, st1, k1 = frame(h)
        assert st1 == 1 and k1 == 1 and torch.equal(img0, img1)      # from now on: the selection
        assert h.select_cap > 0                                        # the selection's size is remembered ...
        img2, st2, _ = frame(h)
        assert st2 == 1 and torch.equal(img0, img2) and h.select_misses == 0 and h.settled
        pairs2, slice2 = ops.last_num_intersects, [int(v) for v in ops.last_slice_intersects]
        # ... and sizes the next frame's tail passes; a promise that turns out too small is noticed and sorted again —
        # with the SAME bound (the selection is not run a second time: same slice, same frame total)
        h.select_cap = 64
        img2b, st2b, _ = frame(h)
        assert st2b == 1 and torch.equal(img0, img2b) and h.select_overflows == 1 and h.select_cap > 64
        assert ops.last_num_intersects == pairs2 and [int(v) for v in ops.last_slice_intersects] == slice2
        img2c, _, _ = frame(h)
        assert torch.equal(img0, img2c) and h.select_overflows == 1
        # a budget this scene does not stop within: the selection of the next frame falls short once, then it is off
        ops.SLICE_BASE = 2
        h2 = ops.FrameHints()
        h2.last_slices, h2.box_share = 1, 0.1                        # (as if the last frame had stopped early)
        img3, st3, k3 = frame(h2)
        assert st3 == 2 and k3 >= 2 and h2.select_misses == 1 and not h2.depth_select()
        img4, st4, _ = frame(h2)
        assert st4 == 0 and torch.equal(img3, img4) and torch.equal(img3, img0)
    finally:
        ops.DEPTH_SELECT, ops.SLICE_BASE, ops.SLICE_ADAPT = saved
        ops.release_arenas()


# --------------------------------------------------------------------------- #
# camera-level gradients are deterministic
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("model,S,R", [("se3", 3, 1), ("se3", 2, 4), ("pixel_velocity", 3, 1), ("pixel_velocity_shared", 3, 1),
                                       ("compat", 1, 1)])
def test_camera_level_gradients_are_bit_identical_from_run_to_run(gs, dev, model, S, R):
    """the view-matrix and velocity gradients — what the pose and velocity optimizers consume
    (/root/reference/train.py:40,66) — are sums over every Gaussian and sub-pose.  Rounds 1-5 finished them with fp32
    atomics (run-to-run differences in the last bits, a 3x wider test bar to absorb them); since round 6 the blocks' sums
    go through a scratch row each and are added in block order, the sub-poses in sub-pose order.  Three runs of one frame:
    torch.equal on viewmat / lin_vel / ang_vel gradients (and on every Gaussian gradient, which always was)."""
    n, W, H = 40000, 176, 144
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=29, scale_mult=4.0), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, R, 1 / 30 if R > 1 else 0.0)
    tt = torch.tensor(times, device=dev)
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    runs = []
    for _ in range(3):
        p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
        lin = (sc["lin_vel"] * 5).clone().requires_grad_(True)
        ang = (sc["ang_vel"] * 3).clone().requires_grad_(True)
        V = sc["viewmat"].clone().requires_grad_(True)
        if model == "compat":
            xys, depths, radii, conics, comp, ntiles, _ = gs.project_gaussians(p["means"], p["log_scales"].exp(), 1.0, p["quats"],
                                                                             V, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W)
            g = torch.Generator().manual_seed(5)
            ((xys * torch.rand(n, 2, generator=g).to(dev)).sum() + (conics * torch.rand(n, 3, generator=g).to(dev)).sum()
             + (depths * torch.rand(n, generator=g).to(dev)).sum()).backward()
            runs.append(dict(V=V.grad.clone(), means=p["means"].grad.clone()))
            continue
        kw = dict(gamma=2.2, min_rgb_level=10.0, raw_params=True, return_alpha=False)
        if model == "se3":
            vms = gs.subpose_viewmats(V, lin, ang, tt)
            rgb = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"], vms, None, S, R,
                                     sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, **kw)[0]
        else:
            rgb = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"], V, None, S, R,
                                     sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, lin_vel=lin, ang_vel=ang, times=tt,
                                     shared_list=model.endswith("shared"), **kw)[0]
        (rgb * wt).sum().backward()
        runs.append(dict(V=V.grad.clone(), lin=lin.grad.clone(), ang=ang.grad.clone(),
                         **{k: v.grad.clone() for k, v in p.items()}))
    for k in runs[0]:
        assert float(runs[0][k].abs().max()) > 0, k
        for r in runs[1:]:
            assert torch.equal(runs[0][k], r[k]), (model, k, float((runs[0][k] - r[k]).abs().max()))


# --------------------------------------------------------------------------- #
# a training-shaped sequence: many cameras through ONE FrameHints
# --------------------------------------------------------------------------- #
def test_view_sweep_keeps_every_view_near_its_fixed_view_time(gs, dev):
    """bench.view_sweep at test size: 12 distinct cameras cycled through ONE FrameHints (adaptive budget, lazy records and
    the nearest-first selection all live, as in SplatfactoDeblurModel) — after the first cycle no view may take more than
    1.2x the time the same view takes when it is rendered back to back through hints of its own (median frame of the three
    later cycles over the median of the view's own fixed-view frames: sub-millisecond frames timed by wall clock; the
    slowest single frame is printed beside it).  A sweep that breaks the bar is repeated once (a scheduling hiccup does not
    repeat, a stall that comes from the code does)."""
    import bench
    from gsdeblur_amd import ops
    saved = (ops.SLICE_ADAPT, ops.SLICE_BASE, ops.DEPTH_SELECT, ops.LAZY_RECORDS)
    try:
        ops.SLICE_ADAPT, ops.SLICE_BASE, ops.DEPTH_SELECT, ops.LAZY_RECORDS = 1, 512, 1, 1
        wl = bench.Workload(gs, dev, 0, 1, 300_000, 960, 544, 3, 1, "survey", "sparse")
        wl.warm_until_settled(3)
        # every camera through its own memory inside the one FrameHints (hints.view(i): what the Model does) ...
        res = bench.view_sweep(wl, ops, n_views=12, cycles=4)
        fixed = res.pop("_fixed")
        print("view sweep:", res)
        # (wall clock with a synchronize per frame: ONE frame that a busy host delays must not fail the suite — a sweep
        #  that misses either bar is measured once more, and a stall that comes from the code repeats)
        if res["worst_view_median_over_its_fixed_time"] > 1.2 or res["frames_over_1p5x_their_fixed_time"] > 2:
            res = bench.view_sweep(wl, ops, n_views=12, cycles=4)
            fixed = res.pop("_fixed")
            print("view sweep, again:", res)
        assert res["worst_view_median_over_its_fixed_time"] <= 1.2, res
        # single frames: a stall that comes from the code hits every visit of a camera (round 5's budget blow-up: all 36);
        # one or two frames that a busy host delayed are not that
        assert res["frames_over_1p5x_their_fixed_time"] <= 2, res
        assert res["arena_retries_later_cycles"] == 0, res
        assert res["frames_timed"] == 36
        # a camera that has been seen decides for itself: no selection of the later cycles falls short
        assert res["selection_misses_after_first_cycle"] == 0, res
        # ... and through ONE last-frame memory for all of them (a caller without camera keys): the cheap views sort
        # everything whenever they follow an expensive one — slower, within the same bar
        one = bench.view_sweep(wl, ops, n_views=12, cycles=4, per_camera=False, fixed=fixed)
        one.pop("_fixed")
        print("view sweep, one memory for all cameras:", one)
        assert one["worst_view_median_over_its_fixed_time"] <= 1.3, one
        assert one["arena_retries_later_cycles"] == 0, one
    finally:
        ops.SLICE_ADAPT, ops.SLICE_BASE, ops.DEPTH_SELECT, ops.LAZY_RECORDS = saved
        ops.release_arenas()


@pytest.mark.parametrize("base", [16, 64])
def test_third_frame_of_a_camera_sequence_vs_float64_oracle(gs, oracle, dev, base):
    """the oracle comparisons ran with ops.SLICE_ADAPT = 0 (tests/conftest.py) and fresh hints —
    never a frame rendered through hints that had learned something.  Here three DIFFERENT cameras go through one
    FrameHints with the product's default state machine on (SLICE_ADAPT = 1, lazy records and nearest-first selection on
    auto); the third frame — budget multiplier grown (base 16: every frame needs several slices) or selection / lazy
    records switched on by the first two (base 64, a scene that stops within its first slice) — is held against the float64 oracle: image, per-sample composites,
    every gradient per element."""
    import dataclasses
    import numpy as np
    from gsdeblur_amd import ops
    from test_gpu_parity import grad_el_ratio, check_fragile, IMG_ATOL
    O = oracle
    n, W, H, S = 5000, 192, 128, 3
    sc = O.synthetic_scene(n, W, H, seed=77, scale_mult=5.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    if base == 64:
        # a frame that stops within its first slice: the nearest Gaussians large and opaque
        near = torch.argsort(sc["means"][:, 2])[:400]
        sc["log_scales"] = sc["log_scales"].clone(); sc["opacity_logits"] = sc["opacity_logits"].clone()
        sc["log_scales"][near] += 1.5
        sc["opacity_logits"][near] = 8.0
    et, gamma, mlevel = 1 / 60, 2.2, 10.0
    bg = torch.tensor([0.05, 0.1, 0.15])
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel", "viewmat"]
    times, _, _ = gs.subpose_schedule(S, et, 1, 0.0)
    tt = torch.tensor(times, device=dev)
    # three cameras: the scene's own and two moved / rotated ones
    cams = []
    for k, (shift, rot) in enumerate([((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)), ((0.08, -0.03, 0.02), (0.02, 0.05, -0.01)),
                                      ((-0.06, 0.04, 0.05), (-0.03, -0.04, 0.02))]):
        V = O.subpose_viewmats(sc["viewmat"].double(), torch.tensor(shift, dtype=torch.float64),
                               torch.tensor(rot, dtype=torch.float64), [1.0])[0].float()
        cams.append((V, sc["lin_vel"] * (1.0 + 0.3 * k), sc["ang_vel"] * (1.0 - 0.2 * k)))
    saved = (ops.SLICE_ADAPT, ops.SLICE_BASE, ops.DEPTH_SELECT, ops.LAZY_RECORDS)
    hints = ops.FrameHints()
    states = []
    try:
        ops.SLICE_ADAPT, ops.SLICE_BASE, ops.DEPTH_SELECT, ops.LAZY_RECORDS = 1, base, 1, 1
        for k, (V, lin, ang) in enumerate(cams):
            p = {kk: sc[kk].float().to(dev).requires_grad_(True) for kk in names[:5]}
            p["viewmat"], p["lin_vel"], p["ang_vel"] = (x.float().to(dev).requires_grad_(True) for x in (V, lin, ang))
            decided = (hints.mult, bool(hints.lazy_records()), bool(hints.depth_select()))
            vms = gs.subpose_viewmats(p["viewmat"], p["lin_vel"], p["ang_vel"], tt)
            samples, _, _ = gs.render_subposes(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"], vms,
                                               bg.to(dev), S, 1, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, raw_params=True,
                                               hints=hints)
            states.append(decided + (ops.last_depth_select, len([v for v in ops.last_slice_intersects if int(v) > 0])))
    finally:
        ops.SLICE_ADAPT, ops.SLICE_BASE, ops.DEPTH_SELECT, ops.LAZY_RECORDS = saved
    print(f"camera sequence, base {base}: (budget multiplier, lazy, selection) decided before each frame + (selection state, "
          f"slices) after it: {states}")
    if base == 16:
        assert states[2][0] >= 4 and states[2][4] >= 1            # two multi-slice frames doubled the budget twice
    else:
        assert states[2][2] and states[2][3] == 1                 # the third frame ran on the nearest-first selection
    # the third frame against the oracle
    V, lin, ang = cams[2]
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], blur_samples=S, rs_bands=1, exposure_time=et,
                         rolling_shutter_time=0.0, gamma=gamma, min_rgb_level=mlevel)
    q = {kk: sc[kk].double().requires_grad_(True) for kk in names[:5]}
    q["viewmat"], q["lin_vel"], q["ang_vel"] = (x.double().requires_grad_(True) for x in (V, lin, ang))
    ref, _, ref_samples, frag, _, _ = O.render(cfg, q["means"], q["log_scales"].exp(), q["quats"], torch.sigmoid(q["opacity_logits"]),
                                               q["sh"], q["viewmat"], q["lin_vel"], q["ang_vel"], background=bg.double(),
                                               return_parts=True)
    good = ~frag
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5)) * good[..., None]
    (ref * wt.double()).sum().backward()
    out = gs.combine_samples(samples, gamma, mlevel)
    (out * wt.to(dev)).sum().backward()
    assert float(frag.float().mean()) < 0.10
    assert (samples.detach().cpu().double() - ref_samples)[:, good].abs().max().item() < IMG_ATOL
    assert (out.detach().cpu().double() - ref.detach())[good].abs().max().item() < 5e-4
    worst = {}
    for kk in names:
        g_hip, g_ref = p[kk].grad.cpu().numpy(), q[kk].grad.numpy()
        if kk == "viewmat":
            g_hip, g_ref = g_hip[:3], g_ref[:3]
        worst[kk] = grad_el_ratio(g_hip, g_ref)
    print(f"third frame vs oracle, base {base}: per-element gradient error / tolerance:", {k_: round(v, 3) for k_, v in worst.items()})
    for k_, v in worst.items():
        assert v <= 1.0, (k_, v)


# --------------------------------------------------------------------------- #
# multi-GPU readiness on the one GPU there is
# --------------------------------------------------------------------------- #
def test_bench_runs_every_exchange_mode_over_rccl_at_world_1(dev):
    """`bench.py --force-exchange --allreduce <mode>` for the three exchange modes: the timed step then contains the
    whole pack -> RCCL collective -> scatter chain on the nccl (= RCCL) backend at world size 1, and the line reports its
    device time as exchange_ms.  Asserted: the run succeeds, says which mode it ran, exchange_ms is positive and inside the
    step, per-rank diagnostics are present.  No scaling claim follows from a single rank (RCCL runs a single-rank
    collective as a copy); the three numbers are printed for profiles/."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    out = {}
    for i, mode in enumerate(("sparse", "allreduce", "rs_ag")):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + i + os.getpid() % 300),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, str(root / "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline",
                            "--no-secondary", "--no-view-sweep", "--force-exchange", "--allreduce", mode, "--gaussians",
                            "300000", "--width", "960", "--height", "544", "--subposes", "3"],
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, (mode, r.stderr[-2000:])
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        c = line["config"]
        assert c["gradient_exchange"] == mode and c["gradient_exchange_forced_at_world_1"] is True
        assert line["exchange_ms"] is not None and 0.0 < line["exchange_ms"] < line["ms_per_step"], (mode, line["exchange_ms"])
        assert c["per_rank"] and c["per_rank"][0]["readback"]["local_rank"] == 0 and c["rccl_version"]
        out[mode] = (line["exchange_ms"], line["ms_per_step"], c["per_rank"][0]["gaussians_with_gradient"])
    print("exchange modes over RCCL at world 1 (exchange_ms, ms_per_step, rows with a gradient):", out)


# --------------------------------------------------------------------------- #
# the splat-parallel backward (built and measured — a negative, kept selectable)
# --------------------------------------------------------------------------- #
@pytest.mark.parametrize("base,R", [(512, 1), (24, 1), (48, 3)])
def test_splat_parallel_backward_gives_the_same_gradients(gs, dev, base, R):
    """csrc/raster_bwd.hip raster_bwd_splat_kernel (lane = list entry of a 64-entry chunk, the tile's pixels streamed
    through the wave, a lane accumulates its entry's sums in registers) against the tile-per-wave kernel: same image
    (the forward is untouched), every gradient equal up to fp32 summation order — one slice, many slices (the
    reverse-traversal state goes through HBM between them), rolling-shutter bands."""
    from gsdeblur_amd import ops
    n, W, H, S = 40000, 208, 176, 2
    sc = to_dev(gs.data.synthetic_scene(n, W, H, seed=41, scale_mult=3.0), dev)
    times, _, _ = gs.subpose_schedule(S, 1 / 60, R, 1 / 30 if R > 1 else 0.0)
    tt = torch.tensor(times, device=dev)
    g = torch.Generator().manual_seed(8)
    wt, wa = torch.rand(H, W, 3, generator=g).to(dev), torch.rand(S, H, W, generator=g).to(dev)
    saved = (ops.BWD_SPLAT, ops.SLICE_BASE)
    res = []
    try:
        ops.SLICE_BASE = base
        for splat in (1, 0):
            ops.BWD_SPLAT = splat
            p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "log_scales", "quats", "opacity_logits", "sh")}
            V = sc["viewmat"].clone().requires_grad_(True)
            vms = gs.subpose_viewmats(V, sc["lin_vel"] * 5, sc["ang_vel"] * 3, tt)
            rgb, alphas, _ = gs.render_combined(p["means"], p["log_scales"], p["quats"], p["opacity_logits"], p["sh"], vms, None,
                                                S, R, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, gamma=2.2, min_rgb_level=10.0,
                                                raw_params=True, hints=ops.FrameHints())
            ((rgb * wt).sum() + (alphas * wa).sum()).backward()
            res.append((rgb.detach().clone(), {k: v.grad.clone() for k, v in p.items()}, V.grad.clone(),
                        len([v for v in ops.last_slice_intersects if int(v) > 0])))
    finally:
        ops.BWD_SPLAT, ops.SLICE_BASE = saved
    a, b = res
    print(f"splat-parallel backward, base {base} R={R}: slices {a[3]}")
    assert torch.equal(a[0], b[0]) and a[3] == b[3] and (a[3] >= 2 or base == 512)
    for k in a[1]:
        ga, gb = a[1][k], b[1][k]
        assert float(gb.abs().max()) > 0
        assert float((ga - gb).abs().max()) <= 2e-5 * float(gb.abs().max()), (k, float((ga - gb).abs().max()), float(gb.abs().max()))
    assert float((a[2] - b[2]).abs().max()) <= 1e-4 * float(b[2].abs().max())
