"""Reference-independent known-answer tests that pin the CPU oracle (parity is otherwise
unpinned: the reference's implementation of this path is not vendored — SURVEY.md §8c)."""
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).resolve().parent / "golden"


def _cfg(O, sc, H, W, **kw):
    kw.setdefault("upstream_grads", 0)      # the known-answer tests differentiate: finite differences see TRUE derivatives
    return O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], **kw)


def _render64(O, sc, cfg, **over):
    g = lambda k: over.get(k, sc[k]).double()
    return O.render(cfg, g("means"), g("log_scales").exp(), g("quats"), torch.sigmoid(g("opacity_logits")),
                    g("sh"), g("viewmat"), g("lin_vel"), g("ang_vel"))


def test_single_gaussian_analytic(oracle):
    """One isotropic Gaussian on the optical axis: image = o * exp(-r^2 / (2 s^2)) * colour."""
    O = oracle
    H = W = 96
    fx = fy = 40.0
    z, s3d, o = 4.0, 0.3, 0.8
    means = torch.tensor([[0.0, 0.0, z]], dtype=torch.float64)
    scales = torch.full((1, 3), s3d, dtype=torch.float64)
    quats = torch.tensor([[1.0, 0, 0, 0]], dtype=torch.float64)
    pr = O.project_gaussians(means, scales, 1.0, quats, torch.eye(4, dtype=torch.float64), fx, fy, W / 2, H / 2, H, W)
    var = (fx * s3d / z) ** 2 + O.DILATION
    assert abs(pr.conics[0, 0].item() - 1 / var) < 1e-12 and abs(pr.conics[0, 1].item()) < 1e-12
    assert pr.radii[0].item() == int(np.ceil(3 * np.sqrt(var)))
    assert abs(pr.compensation[0].item() - ((fx * s3d / z) ** 2) / var) < 1e-12  # sqrt(det0/det) = s^2/(s^2+.3)
    col = torch.tensor([[0.2, 0.5, 0.9]], dtype=torch.float64)
    img, r = O.rasterize_gaussians(pr.xys, pr.depths, pr.radii, pr.conics, pr.num_tiles_hit, col,
                                   torch.tensor([o], dtype=torch.float64), H, W, proj=pr)
    yy, xx = np.meshgrid(np.arange(H) + 0.5, np.arange(W) + 0.5, indexing="ij")
    r2 = (xx - W / 2) ** 2 + (yy - H / 2) ** 2
    a = o * np.exp(-0.5 * r2 / var)
    a[a < 1 / 255] = 0
    # tiles the bbox does not reach stay empty
    ref = a[..., None] * col.numpy()[0]
    got = img.numpy()
    covered = np.zeros((H, W), bool)
    tmn, tmx = pr.tile_min[0].numpy(), pr.tile_max[0].numpy()
    covered[tmn[1] * 16:tmx[1] * 16, tmn[0] * 16:tmx[0] * 16] = True
    assert np.abs(got[covered] - ref[covered]).max() < 1e-12
    assert np.abs(got[~covered]).max() == 0


def test_opaque_front_gaussian_stops_the_rest(oracle):
    """Front-to-back order + early stop: behind a stack that drives T below 1e-4 nothing contributes."""
    O = oracle
    H = W = 16
    n_front = 8
    means = torch.tensor([[0.0, 0.0, 2.0 + 0.01 * i] for i in range(n_front)] + [[0.0, 0.0, 5.0]], dtype=torch.float64)
    scales = torch.full((n_front + 1, 3), 5.0, dtype=torch.float64)
    quats = torch.tensor([[1.0, 0, 0, 0]] * (n_front + 1), dtype=torch.float64)
    pr = O.project_gaussians(means, scales, 1.0, quats, torch.eye(4, dtype=torch.float64), 20, 20, 8, 8, H, W)
    cols = torch.tensor([[1.0, 0, 0]] * n_front + [[0, 1.0, 0]], dtype=torch.float64)
    op = torch.full((n_front + 1,), 0.99, dtype=torch.float64)
    img, r = O.rasterize_gaussians(pr.xys, pr.depths, pr.radii, pr.conics, pr.num_tiles_hit, cols, op, H, W, proj=pr)
    assert img[..., 1].abs().max().item() == 0.0          # the green one behind is never blended
    assert (r.final_T > oracle.T_MIN).all()                 # stop happens BEFORE T would fall to <= 1e-4
    assert (r.final_idx < n_front + 1).all()


def test_static_equals_zero_velocity_and_S_average(oracle):
    O = oracle
    W, H = 64, 48
    sc = O.synthetic_scene(300, W, H, seed=3, scale_mult=8.0)
    zero = torch.zeros(3)
    base, _ = _render64(O, sc, _cfg(O, sc, H, W))
    blur0, _ = _render64(O, sc, _cfg(O, sc, H, W, blur_samples=4, rs_bands=3, exposure_time=0.02,
                                     rolling_shutter_time=0.03), lin_vel=zero, ang_vel=zero)
    assert torch.allclose(base, blur0, atol=1e-12)
    # S-sample average == mean of S independent single-pose renders at the sub-pose viewmats
    cfgS = _cfg(O, sc, H, W, blur_samples=3, exposure_time=0.05)
    outS, _ = _render64(O, sc, cfgS)
    times, _, _ = O.subpose_times(3, 0.05, 1, 0.0)
    vms = O.subpose_viewmats(sc["viewmat"].double(), sc["lin_vel"].double(), sc["ang_vel"].double(), times)
    acc = 0
    for V in vms:
        o, _ = _render64(O, sc, _cfg(O, sc, H, W), viewmat=V, lin_vel=zero, ang_vel=zero)
        acc = acc + o
    assert torch.allclose(outS, acc / 3, atol=1e-12)


def test_gamma_mean_identities(oracle):
    O = oracle
    s = torch.rand(4, 5, 6, 3, dtype=torch.float64)
    assert torch.allclose(O.combine_samples(s, 1.0, 0.0), s.mean(0))
    same = s[:1].expand(4, -1, -1, -1)
    assert torch.allclose(O.combine_samples(same, 2.2, 0.0), s[0], atol=1e-12)   # mean of equal samples is identity
    m = 10.0
    out = O.combine_samples(s * 0.01, 2.2, m)                                      # everything below the floor
    assert torch.allclose(out, torch.full_like(out, m / 255.0), atol=1e-12)


def test_permutation_invariance(oracle):
    O = oracle
    W, H = 48, 32
    sc = O.synthetic_scene(200, W, H, seed=5, scale_mult=8.0)
    cfg = _cfg(O, sc, H, W)
    a, _ = _render64(O, sc, cfg)
    perm = torch.randperm(200, generator=torch.Generator().manual_seed(0))
    sc2 = {k: (v[perm] if isinstance(v, torch.Tensor) and v.shape[:1] == (200,) else v) for k, v in sc.items()}
    b, _ = _render64(O, sc2, cfg)
    assert torch.allclose(a, b, atol=1e-12)


@pytest.mark.parametrize("sh_degree", [0, 3])
def test_finite_difference_gradients(oracle, sh_degree):
    """autograd through the float64 oracle vs central differences on a few coordinates.
    View directions are detached (splatfacto semantics: no gradient through SH dirs), so with
    sh_degree=3 only the parameters that do not move the view direction are differenced."""
    O = oracle
    W, H = 32, 32
    sc = O.synthetic_scene(40, W, H, seed=9, scale_mult=10.0, sh_degree=sh_degree)
    cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], sh_degree=sh_degree, blur_samples=2,
                         rs_bands=2, exposure_time=0.02, rolling_shutter_time=0.02, gamma=2.2, min_rgb_level=5.0,
                         upstream_grads=0)           # finite differences see the TRUE derivatives
    names = ["means", "log_scales", "quats", "opacity_logits", "sh", "lin_vel", "ang_vel"]
    fd_names = names if sh_degree == 0 else ["log_scales", "quats", "opacity_logits", "sh"]
    ps = {k: sc[k].double().clone().requires_grad_(True) for k in names}
    wt = torch.rand(H, W, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(1))

    def loss(p):
        out, _ = O.render(cfg, p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["opacity_logits"]),
                          p["sh"], sc["viewmat"].double(), p["lin_vel"], p["ang_vel"])
        return (out * wt).sum()

    loss(ps).backward()
    rng = np.random.default_rng(0)
    eps = 1e-6
    worst = {}
    for k in fd_names:
        flat = ps[k].detach().reshape(-1)
        for idx in rng.choice(flat.numel(), size=min(5, flat.numel()), replace=False):
            def at(delta):
                q = {kk: vv.detach().clone() for kk, vv in ps.items()}
                q[k].reshape(-1)[idx] += delta
                return loss(q).item()
            fd = (at(eps) - at(-eps)) / (2 * eps)
            an = ps[k].grad.reshape(-1)[idx].item()
            worst[k] = max(worst.get(k, 0.0), abs(fd - an) / (abs(an) + abs(fd) + 1e-5))
    # thresholds (alpha>=1/255, T<=1e-4, bbox tiles) make the image piecewise smooth; with eps=1e-6 a
    # flip is improbable on 40 Gaussians and would show up as an O(1) relative error
    assert max(worst.values()) < 5e-4, worst


def test_sh_orthonormal_and_independent_forms(oracle):
    """All 25 real SH basis functions are orthonormal on the sphere (Gauss-Legendre x uniform-phi quadrature)."""
    O = oracle
    x, w = np.polynomial.legendre.leggauss(64)
    ph = np.arange(128) * 2 * np.pi / 128
    th = np.arccos(x)
    T, Pm = np.meshgrid(th, ph, indexing="ij")
    dirs = np.stack([np.sin(T) * np.cos(Pm), np.sin(T) * np.sin(Pm), np.cos(T)], -1).reshape(-1, 3)
    ww = (w[:, None] * np.ones(128)[None, :] * 2 * np.pi / 128).reshape(-1)
    B = O.sh_basis(4, torch.from_numpy(dirs)).numpy()
    G = (B * ww[:, None]).T @ B
    assert np.abs(G - np.eye(25)).max() < 1e-12


def test_se3_screw_properties(oracle):
    O = oracle
    V0 = O.subpose_viewmats(torch.eye(4, dtype=torch.float64), torch.tensor([0.3, -0.2, 0.5], dtype=torch.float64),
                            torch.tensor([0.2, 0.4, -0.1], dtype=torch.float64), [1.0])[0]
    lin = torch.tensor([0.1, 0.2, -0.3], dtype=torch.float64)
    ang = torch.tensor([-0.5, 0.25, 0.4], dtype=torch.float64)
    Vs = O.subpose_viewmats(V0, lin, ang, [-0.2, 0.0, 0.2, 0.4])
    assert torch.allclose(Vs[1], V0, atol=1e-14)                                  # t=0 is the identity
    R = Vs[:, :3, :3]
    assert torch.allclose(R @ R.transpose(1, 2), torch.eye(3, dtype=torch.float64).expand(4, 3, 3), atol=1e-13)
    # one-parameter subgroup: V(t1+t2) = Exp(-t2 xi) V(t1)
    V02 = O.subpose_viewmats(Vs[2], lin, ang, [0.2])[0]
    assert torch.allclose(V02, Vs[3], atol=1e-13)
    # pure translation: camera centre moves by +t*R_c2w*v
    Vt = O.subpose_viewmats(V0, lin, torch.zeros(3, dtype=torch.float64), [0.5])[0]
    c0 = -(V0[:3, :3].T @ V0[:3, 3])
    c1 = -(Vt[:3, :3].T @ Vt[:3, 3])
    assert torch.allclose(c1 - c0, 0.5 * (V0[:3, :3].T @ lin), atol=1e-13)


def test_binning_consistency(oracle):
    O = oracle
    W, H = 80, 64
    sc = O.synthetic_scene(400, W, H, seed=21, scale_mult=10.0)
    pr = O.project_gaussians(sc["means"], sc["log_scales"].exp(), 1.0, sc["quats"], sc["viewmat"], sc["fx"], sc["fy"],
                             sc["cx"], sc["cy"], H, W)
    keys, gids = O.map_gaussian_to_intersects(pr, W)
    assert len(keys) == int(pr.num_tiles_hit.sum())
    sk, sg = O.sort_intersects(keys, gids)
    assert (np.diff(sk) >= 0).all()
    T = 5 * 4
    bins = O.get_tile_bin_edges(sk, T)
    assert bins[:, 1].max() == len(sk) and ((bins[:, 1] - bins[:, 0]).sum() == len(sk))
    for t in range(T):
        s, e = bins[t]
        assert ((sk[s:e] >> 32) == t).all()
        d = pr.depths.numpy()[sg[s:e]]
        assert (np.diff(d) >= 0).all()


@pytest.mark.parametrize("name", ["static_small", "blur_rs_small", "pixvel_exact_rs_posed_small"])
def test_oracle_matches_committed_golden(oracle, name):
    """The oracle reproduces the committed fixtures (guards against silent drift of the checker)."""
    O = oracle
    d = np.load(GOLD / f"{name}.npz")
    H, W, S, R, deg = (int(v) for v in d["cfg"])
    et, rt, gamma, mlevel = (float(v) for v in d["cfg_f"])
    pixvel, exact = (int(v) for v in d["cfg_model"]) if "cfg_model" in d.files else (0, 0)
    cfg = O.RenderConfig(H, W, float(d["fx"]), float(d["fy"]), float(d["cx"]), float(d["cy"]), sh_degree=deg,
                         blur_samples=S, rs_bands=R, exposure_time=et, rolling_shutter_time=rt, gamma=gamma,
                         min_rgb_level=mlevel, motion_model="pixel_velocity" if pixvel else "se3", rs_exact=bool(exact))
    t = lambda k: torch.from_numpy(d[k]).double()
    out, alpha = O.render(cfg, t("means"), t("log_scales").exp(), t("quats"), torch.sigmoid(t("opacity_logits")),
                          t("sh"), t("viewmat"), t("lin_vel"), t("ang_vel"), background=t("background"))
    assert np.abs(out.numpy() - d["out"]).max() < 1e-12
    assert np.abs(alpha.numpy() - d["alpha"]).max() < 1e-12


# --------------------------------------------------------------------------- #
# pixel-velocity model (the paper's first-order blur / rolling-shutter model)
# --------------------------------------------------------------------------- #
def _pv_scene(oracle, n=300, W=96, H=64, seed=3):
    sc = oracle.synthetic_scene(n, W, H, seed=seed, scale_mult=6.0)
    return {k: (v.double() if isinstance(v, torch.Tensor) else v) for k, v in sc.items()}, W, H


def test_pixel_velocity_agrees_with_se3_reprojection_to_second_order(oracle):
    """centre(t) under the screw-interpolated pose = centre(0) + t * pixel_velocity + O(t^2): halving t must quarter
    the gap (this also pins the sign / frame conventions of v' = -J (w x p_c + v) to the SE(3) model's)"""
    O = oracle
    sc, W, H = _pv_scene(O)
    lin, ang = sc["lin_vel"] * 20, sc["ang_vel"] * 10
    args = (sc["means"], sc["log_scales"].exp(), 1.0, sc["quats"])
    pr0 = O.project_gaussians(*args, sc["viewmat"], sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, keep_offscreen=True)
    pv = O.pixel_velocity(sc["means"], sc["viewmat"], sc["fx"], sc["fy"], lin, ang)
    gaps = []
    for t in (0.02, 0.01, 0.005):
        Vt = O.subpose_viewmats(sc["viewmat"], lin, ang, [t])[0]
        pr = O.project_gaussians(*args, Vt, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W, keep_offscreen=True)
        ok = (pr.radii > 0) & (pr0.radii > 0)
        gaps.append((pr.xys - (pr0.xys + t * pv))[ok].abs().max().item())
    assert gaps[0] > 1e-3                                        # the motion is visible at all
    assert 3.5 < gaps[0] / gaps[1] < 4.5 and 3.5 < gaps[1] / gaps[2] < 4.5


def _posed(sc, O):
    """the scene seen from a rotated and translated camera: world = R^T (camera - t), viewmat = [R | t] (every other
    oracle test renders from the identity pose, where pre- and post-multiplication of the pose cannot be told apart)"""
    import math

    def rot(ax, a):
        c, s_ = math.cos(a), math.sin(a)
        R = torch.eye(3, dtype=torch.float64)
        i, j = [(1, 2), (0, 2), (0, 1)][ax]
        R[i, i] = c; R[j, j] = c; R[i, j] = -s_; R[j, i] = s_
        return R
    R = rot(1, 0.7) @ rot(0, -0.4) @ rot(2, 1.1)
    t = torch.tensor([0.3, -0.2, 0.5], dtype=torch.float64)
    V = torch.eye(4, dtype=torch.float64)
    V[:3, :3], V[:3, 3] = R, t
    out = dict(sc)
    out["means"] = (sc["means"] - t) @ R
    out["viewmat"] = V
    return out


def test_pixel_velocity_follows_se3_from_a_real_pose_and_ignores_grazing_gaussians(oracle):
    """round 3, found end to end (tools/rs_forward_check.py): from a real camera pose, inside a cloud of Gaussians, the
    pixel-velocity frame was 11 dB away from the SE(3) frame although every far Gaussian's centre followed to 0.2 px.
    Cause: Gaussians just in front of the camera plane at grazing angles (z = 0.02, |x| = 2: x/z = 100) — far off screen
    in every true sub-pose, but their first-order pixel velocity is 1e5 px/s and dragged them across the image.  The
    A centre outside the projection's own fov guard band (1.3 tan(fov/2)) now moves with the Jacobian of the band's
    edge — where the covariance projection takes its Jacobian too (round 3 culled such Gaussians outright, and a floor
    or wall whose centre lies beside the image vanished from this model only).
    Known answers: (1) identity and non-identity pose: the pixel-velocity render is much closer to the SE(3) render
    than the static render is; (2) adding grazing Gaussians changes neither render; (3) adding large Gaussians a metre
    away whose centres are out of band but whose footprints reach well into the image changes BOTH renders, and the
    first-order frame still follows the SE(3) frame better than the static frame does (the band-edge Jacobian
    under-estimates the depth-motion term of such a centre by up to x/z : 0.8125 — a first-order model's price)."""
    O = oracle
    H, W, n = 64, 96, 500
    sc = O.synthetic_scene(n, W, H, seed=3, dtype=torch.float64, scale_mult=8.0)
    lin, ang = sc["lin_vel"] * 40, sc["ang_vel"] * 40
    # two row bands rendered a fifth of a second apart: the frame differs from the static one at FIRST order in the
    # twist (a symmetric blur alone only differs at second order, like the model's own error)
    kw = dict(blur_samples=2, exposure_time=1 / 30, rs_bands=2, rolling_shutter_time=1 / 5, gamma=2.2, min_rgb_level=0.0)

    def renders(s_):
        base = (s_["means"], s_["log_scales"].exp(), s_["quats"], torch.sigmoid(s_["opacity_logits"]), s_["sh"], s_["viewmat"])
        z = torch.zeros(3, dtype=torch.float64)
        mk = lambda mm: O.RenderConfig(H, W, s_["fx"], s_["fy"], s_["cx"], s_["cy"], motion_model=mm, **kw)   # noqa: E731
        return (O.render(mk("se3"), *base, lin, ang)[0], O.render(mk("pixel_velocity"), *base, lin, ang)[0],
                O.render(mk("se3"), *base, z, z)[0])
    mse = lambda a, b: float(((a - b) ** 2).mean())     # noqa: E731
    for posed in (False, True):
        s0 = _posed(sc, O) if posed else sc
        se3, pv, static = renders(s0)
        # (the model moves centres only — shapes, opacities and depth order stay those of the mid pose —, so it closes
        #  about two thirds of the gap, whatever the twist's magnitude)
        assert mse(pv, se3) < 0.5 * mse(static, se3), (posed, mse(pv, se3), mse(static, se3))
        # grazing Gaussians: camera-space z = 0.02 .. 0.05, |x| or |y| around 2 (x/z up to 100), large and opaque
        g = torch.Generator().manual_seed(5)
        k = 40
        pc = torch.stack([(torch.rand(k, generator=g, dtype=torch.float64) * 2 + 1) * torch.sign(torch.rand(k, generator=g, dtype=torch.float64) - 0.5),
                          (torch.rand(k, generator=g, dtype=torch.float64) * 2 - 1) * 2,
                          0.02 + 0.03 * torch.rand(k, generator=g, dtype=torch.float64)], dim=1)
        Vm = s0["viewmat"]
        world = (pc - Vm[:3, 3]) @ Vm[:3, :3]
        s1 = dict(s0)
        s1["means"] = torch.cat([s0["means"], world])
        s1["log_scales"] = torch.cat([s0["log_scales"], torch.full((k, 3), -2.5, dtype=torch.float64)])
        s1["quats"] = torch.cat([s0["quats"], s0["quats"][:k]])
        s1["opacity_logits"] = torch.cat([s0["opacity_logits"], torch.full((k,), 4.0, dtype=torch.float64)])
        s1["sh"] = torch.cat([s0["sh"], s0["sh"][:k] + 0.5])
        se3_g, pv_g, _ = renders(s1)
        assert mse(se3_g, se3) < 1e-8, posed                 # they are never on screen in a true sub-pose ...
        assert mse(pv_g, pv) < 1e-8, posed                   # ... and no longer in the first-order model's either
        # walls: x/z = +-1.0 or y/z = +-0.75 (band: 0.8125), one metre away, 0.37 m wide
        wall = torch.tensor([[1.0, 0.1, 1.0], [-1.0, -0.2, 1.0], [0.15, 0.75, 1.0], [-0.1, -0.75, 1.0]], dtype=torch.float64)
        k2 = wall.shape[0]
        s2 = dict(s0)
        s2["means"] = torch.cat([s0["means"], (wall - Vm[:3, 3]) @ Vm[:3, :3]])
        s2["log_scales"] = torch.cat([s0["log_scales"], torch.full((k2, 3), -1.0, dtype=torch.float64)])
        s2["quats"] = torch.cat([s0["quats"], s0["quats"][:k2]])
        s2["opacity_logits"] = torch.cat([s0["opacity_logits"], torch.full((k2,), 0.5, dtype=torch.float64)])
        s2["sh"] = torch.cat([s0["sh"], s0["sh"][:k2] + 0.5])
        se3_w, pv_w, static_w = renders(s2)
        assert mse(se3_w, se3) > 1e-4 and mse(pv_w, pv) > 1e-4, (posed, mse(se3_w, se3), mse(pv_w, pv))
        print(f"walls posed={posed}: pixel velocity vs SE(3) {mse(pv_w, se3_w):.2e}, static vs SE(3) {mse(static_w, se3_w):.2e}")
        assert mse(pv_w, se3_w) < 0.8 * mse(static_w, se3_w), (posed, mse(pv_w, se3_w), mse(static_w, se3_w))


def test_pixel_velocity_render_static_limit_and_autograd(oracle):
    """zero twist: every sub-pose renders the static frame; non-zero twist: autograd of the render w.r.t. the
    twist matches central finite differences"""
    O = oracle
    sc, W, H = _pv_scene(O, n=120, W=48, H=32)
    kw = dict(blur_samples=3, rs_bands=2, exposure_time=1 / 60, rolling_shutter_time=1 / 30, gamma=2.2, min_rgb_level=10.0)
    cfg_pv = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], motion_model="pixel_velocity", upstream_grads=0,
                            **kw)
    cfg_static = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], gamma=2.2, min_rgb_level=10.0)
    base = (sc["means"], sc["log_scales"].exp(), sc["quats"], torch.sigmoid(sc["opacity_logits"]), sc["sh"], sc["viewmat"])
    z = torch.zeros(3, dtype=torch.float64)
    a, _ = O.render(cfg_pv, *base, z, z)
    b, _ = O.render(cfg_static, *base, z, z)
    assert (a - b).abs().max().item() < 1e-12
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    lin = (sc["lin_vel"] * 20).requires_grad_(True)
    ang = (sc["ang_vel"] * 10).requires_grad_(True)
    out, _ = O.render(cfg_pv, *base, lin, ang)
    (out * wt).sum().backward()
    for name, x in (("lin", lin), ("ang", ang)):
        for j in range(3):
            d = torch.zeros(3, dtype=torch.float64)
            d[j] = 1e-6
            lp = [(lin + d) if name == "lin" else lin, (ang + d) if name == "ang" else ang]
            lm = [(lin - d) if name == "lin" else lin, (ang - d) if name == "ang" else ang]
            with torch.no_grad():
                fp = (O.render(cfg_pv, *base, *lp)[0] * wt).sum()
                fm = (O.render(cfg_pv, *base, *lm)[0] * wt).sum()
            fd = ((fp - fm) / 2e-6).item()
            assert abs(fd - x.grad[j].item()) < 1e-4 * (abs(fd) + 1e-3), (name, j, fd, x.grad[j].item())


def _pixel_loop():
    import importlib.util
    spec = importlib.util.spec_from_file_location("pixel_loop_oracle",
                                                  Path(__file__).resolve().parents[1] / "oracle" / "pixel_loop_oracle.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed,n,H,W,hot", [(3, 100, 40, 56, False), (11, 90, 33, 47, True)])
def test_vectorised_oracle_equals_independent_pixel_loop(oracle, seed, n, H, W, hot):
    """VERDICT round 2 'Missing 6': `rasterize_sorted` (vectorised per tile, autograd backward) against a literal
    per-pixel / per-Gaussian loop with a real `break` and a hand-derived reverse loop, written from SURVEY App. A
    alone (oracle/pixel_loop_oracle.py shares no code with gs_oracle.py).  Binning: sorted ids and bin edges against
    a brute-force "members of every tile, ordered by (depth bits, id)"; compositing: image, final T and final index;
    gradients of a random loss (image + alpha) with respect to xy, conic, colour and opacity.  Ragged image sizes,
    opacities above the 0.999 clamp in the second case (both clamp-gradient conventions)."""
    O, PL = oracle, _pixel_loop()
    sc = O.synthetic_scene(n, W, H, seed=seed, dtype=torch.float64, scale_mult=20.0)
    sc["opacity_logits"] = sc["opacity_logits"] + 1.5
    if hot:
        sc["opacity_logits"] = sc["opacity_logits"] + 6.0              # many opacities > 0.999: the clamp is active
    pr = O.project_gaussians(sc["means"], sc["log_scales"].exp(), 1.0, sc["quats"], sc["viewmat"], sc["fx"], sc["fy"],
                             sc["cx"], sc["cy"], H, W)
    g = torch.Generator().manual_seed(seed)
    colors = torch.rand(n, 3, generator=g, dtype=torch.float64)
    opac = torch.sigmoid(sc["opacity_logits"])
    bg = torch.tensor([0.3, 0.1, 0.6], dtype=torch.float64)
    # --- binning
    keys, gids = O.sort_intersects(*O.map_gaussian_to_intersects(pr, W))
    T = ((W + 15) // 16) * ((H + 15) // 16)
    bins = O.get_tile_bin_edges(keys, T)
    ids_bf, bins_bf = PL.tile_lists_by_brute_force(pr.tile_min.numpy(), pr.tile_max.numpy(), pr.depths.numpy(),
                                                   (pr.num_tiles_hit > 0).numpy(), H, W)
    assert np.array_equal(ids_bf, gids) and np.array_equal(bins_bf, bins)
    assert len(gids) > 4 * T                                            # lists are long enough to mean something
    # --- forward
    leaves = [t.clone().requires_grad_(True) for t in (pr.xys.detach(), pr.conics.detach(), colors, opac)]
    r = O.rasterize_sorted(leaves[0], leaves[1], leaves[2], leaves[3], gids, bins, H, W, bg, upstream=0)
    xys, conics = pr.xys.detach().numpy(), pr.conics.detach().numpy()
    f = PL.composite_pixel_loop(xys, conics, colors.numpy(), opac.numpy(), gids, bins, H, W, bg.tolist())
    assert np.array_equal(f["final_idx"], r.final_idx.numpy())
    assert np.abs(f["img"] - r.img.detach().numpy()).max() < 1e-13
    assert np.abs(f["final_T"] - r.final_T.detach().numpy()).max() < 1e-13
    assert f["stops"] > (0.02 if hot else 0.0) * H * W                  # the early stop is exercised
    # --- backward: random loss on image and alpha
    v_img = torch.rand(H, W, 3, generator=g, dtype=torch.float64) - 0.3
    v_alpha = torch.rand(H, W, generator=g, dtype=torch.float64) - 0.5
    ((r.img * v_img).sum() + (r.alpha * v_alpha).sum()).backward()
    b = PL.composite_backward_pixel_loop(xys, conics, colors.numpy(), opac.numpy(), gids, bins, H, W, f, v_img.numpy(),
                                         v_alpha.numpy(), bg.tolist(), clamp_blocks_gradient=True)
    for name, leaf in zip(("v_xy", "v_conic", "v_colors", "v_opacity"), leaves):
        want = leaf.grad.numpy().reshape(b[name].shape)
        assert np.abs(b[name] - want).max() <= 1e-11 * max(1.0, np.abs(want).max()), name
    if hot:
        # upstream's convention lets the gradient through the clamp: it must differ exactly on the clamped Gaussians
        bu = PL.composite_backward_pixel_loop(xys, conics, colors.numpy(), opac.numpy(), gids, bins, H, W, f,
                                              v_img.numpy(), v_alpha.numpy(), bg.tolist(), clamp_blocks_gradient=False)
        changed = np.abs(bu["v_opacity"] - b["v_opacity"]) > 0
        assert changed.any() and not changed[opac.numpy() <= 0.999].any()
        assert np.array_equal(bu["v_colors"], b["v_colors"])
        # ... and the vectorised oracle's UPSTREAM mode (round 5: straight-through min(0.999, .), the default of the
        # oracle and of the product) is that convention: every gradient against the loop's hand-derived reverse pass
        lu = [t.clone().requires_grad_(True) for t in (pr.xys.detach(), pr.conics.detach(), colors, opac)]
        ru = O.rasterize_sorted(lu[0], lu[1], lu[2], lu[3], gids, bins, H, W, bg, upstream=O.UP_ALPHA_CLAMP)
        assert torch.equal(ru.img.detach(), r.img.detach()) and torch.equal(ru.final_T.detach(), r.final_T.detach())
        ((ru.img * v_img).sum() + (ru.alpha * v_alpha).sum()).backward()
        for name, leaf in zip(("v_xy", "v_conic", "v_colors", "v_opacity"), lu):
            want = leaf.grad.numpy().reshape(bu[name].shape)
            assert np.abs(bu[name] - want).max() <= 1e-11 * max(1.0, np.abs(want).max()), name
        assert (lu[3].grad - leaves[3].grad).abs().max() > 0


def test_exact_rolling_shutter_compositing_equals_independent_pixel_loop(oracle):
    """round 3: the per-row centre shift of the exact rolling-shutter mode (`rasterize_sorted(row_shift=...)`, what
    raster_rs.hip is held against) against the independent per-pixel loop with the same shift written out literally —
    image, final T, final index, and the gradients to xy, conic, colour, opacity AND pixel velocity (the loop's
    hand-derived v_pix_vel = sum over pixels of tau(row) * v_xy)."""
    O, PL = oracle, _pixel_loop()
    H, W, n, seed = 40, 56, 100, 8
    sc = O.synthetic_scene(n, W, H, seed=seed, dtype=torch.float64, scale_mult=20.0)
    sc["opacity_logits"] = sc["opacity_logits"] + 1.5
    pr = O.project_gaussians(sc["means"], sc["log_scales"].exp(), 1.0, sc["quats"], sc["viewmat"], sc["fx"], sc["fy"],
                             sc["cx"], sc["cy"], H, W, keep_offscreen=True)
    g = torch.Generator().manual_seed(seed)
    colors = torch.rand(n, 3, generator=g, dtype=torch.float64)
    opac = torch.sigmoid(sc["opacity_logits"])
    bg = torch.tensor([0.3, 0.1, 0.6], dtype=torch.float64)
    pv = (torch.rand(n, 2, generator=g, dtype=torch.float64) - 0.5) * 600.0             # px / s
    T_ro = 1 / 30
    tau = ((torch.arange(H, dtype=torch.float64) + 0.5) / H - 0.5) * T_ro
    prs = O._bounds_swept(pr, pr.xys.detach(), pv, 0.5 * T_ro, H, W)                     # swept tile boxes
    keys, gids = O.sort_intersects(*O.map_gaussian_to_intersects(prs, W))
    Tn = ((W + 15) // 16) * ((H + 15) // 16)
    bins = O.get_tile_bin_edges(keys, Tn)
    leaves = [t.clone().requires_grad_(True) for t in (pr.xys.detach(), pr.conics.detach(), colors, opac, pv)]
    r = O.rasterize_sorted(leaves[0], leaves[1], leaves[2], leaves[3], gids, bins, H, W, bg, row_shift=(leaves[4], tau))
    xys, conics = pr.xys.detach().numpy(), pr.conics.detach().numpy()
    f = PL.composite_pixel_loop(xys, conics, colors.numpy(), opac.numpy(), gids, bins, H, W, bg.tolist(),
                                pix_vel=pv.numpy(), row_time=tau.numpy())
    still = PL.composite_pixel_loop(xys, conics, colors.numpy(), opac.numpy(), gids, bins, H, W, bg.tolist())
    assert np.abs(f["img"] - still["img"]).max() > 0.05                  # the shift really moves things
    assert np.array_equal(f["final_idx"], r.final_idx.numpy())
    assert np.abs(f["img"] - r.img.detach().numpy()).max() < 1e-13
    assert np.abs(f["final_T"] - r.final_T.detach().numpy()).max() < 1e-13
    v_img = torch.rand(H, W, 3, generator=g, dtype=torch.float64) - 0.3
    v_alpha = torch.rand(H, W, generator=g, dtype=torch.float64) - 0.5
    ((r.img * v_img).sum() + (r.alpha * v_alpha).sum()).backward()
    b = PL.composite_backward_pixel_loop(xys, conics, colors.numpy(), opac.numpy(), gids, bins, H, W, f, v_img.numpy(),
                                         v_alpha.numpy(), bg.tolist(), clamp_blocks_gradient=True, pix_vel=pv.numpy(),
                                         row_time=tau.numpy())
    for name, leaf in zip(("v_xy", "v_conic", "v_colors", "v_opacity", "v_pix_vel"), leaves):
        want = leaf.grad.numpy().reshape(b[name].shape)
        assert np.abs(want).max() > 0, name
        assert np.abs(b[name] - want).max() <= 1e-11 * max(1.0, np.abs(want).max()), name


def test_shared_list_mode_known_answers(oracle):
    """round 4: RenderConfig.shared_list — ONE swept-box binning for all blur samples of the pixel-velocity model
    (what render_subposes(shared_list=True) is held against).  Reference-independent known answers:
      * zero twist: the swept boxes are the plain boxes, every sample is the static frame, and the mode equals the
        per-sample lists bit for bit;
      * the swept box of the frame contains every sample's own box, so the shared list is a superset of each sample's
        list and the walk reaches the same stops: where no splat has a fringe beyond its 3-sigma box (opacity below
        1/(255 e^-4.5) = 0.353, alpha at the box edge under 1/255) the two modes render the same image;
      * with opaque splats the modes differ, only by the fringe: per pixel at most a few times alpha = op e^-4.5;
      * autograd through the mode agrees with central finite differences of the twist and of a sample-time-weighted
        centre (the (t_s - t_c) term reaches d loss / d pixel-velocity)."""
    O = oracle
    H, W, n = 64, 96, 300
    sc = O.synthetic_scene(n, W, H, seed=77, dtype=torch.float64, scale_mult=8.0)
    sc["sh"][:, 1:] = 0.0
    base = lambda s_: (s_["means"], s_["log_scales"].exp(), s_["quats"], torch.sigmoid(s_["opacity_logits"]), s_["sh"],  # noqa: E731
                       s_["viewmat"])
    kw = dict(blur_samples=4, exposure_time=1 / 50, gamma=2.2, min_rgb_level=10.0, motion_model="pixel_velocity")

    def cfg(shared, rt=0.0):
        return O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], rolling_shutter_time=rt, rs_exact=rt != 0.0,
                              shared_list=shared, upstream_grads=0, **kw)
    z = torch.zeros(3, dtype=torch.float64)
    a, _ = O.render(cfg(True), *base(sc), z, z)
    b, _ = O.render(cfg(False), *base(sc), z, z)
    assert torch.equal(a, b)
    lin, ang = sc["lin_vel"] * 40, sc["ang_vel"] * 25
    # translucent splats: no fringe, same picture (with and without the row term)
    faint = dict(sc)
    faint["opacity_logits"] = torch.clamp(sc["opacity_logits"], max=-0.7)          # sigmoid(-0.7) = 0.33 < 0.353
    for rt in (0.0, 1 / 30):
        a, _, sa, _, parts_a, _ = O.render(cfg(True, rt), *base(faint), lin, ang, return_parts=True)
        b, _, sb, _, parts_b, _ = O.render(cfg(False, rt), *base(faint), lin, ang, return_parts=True)
        assert (sa - sb).abs().max().item() < 1e-7, rt      # (the mode rounds the sample times to float32, like the library)
        shared_pairs = int(parts_a[0][0].num_tiles_hit.sum())
        assert shared_pairs < sum(int(p[0].num_tiles_hit.sum()) for p in parts_b)     # one list, fewer pairs in all
        for p in parts_b:                                                           # ... but a superset of each sample's
            assert bool((parts_a[0][0].tile_min <= p[0].tile_min)[p[0].radii > 0].all())
            assert bool((parts_a[0][0].tile_max >= p[0].tile_max)[p[0].radii > 0].all())
    # opaque splats: only the fringe differs
    a, _, sa, _, _, _ = O.render(cfg(True), *base(sc), lin, ang, return_parts=True)
    b, _, sb, _, _, _ = O.render(cfg(False), *base(sc), lin, ang, return_parts=True)
    d = (sa - sb).abs()
    assert 0 < d.max().item() < 0.05 and d.mean().item() < 1e-4, (d.max().item(), d.mean().item())
    # autograd vs central differences (twist)
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(1), dtype=torch.float64)
    lin_r, ang_r = lin.clone().requires_grad_(True), ang.clone().requires_grad_(True)
    (O.render(cfg(True, 1 / 30), *base(sc), lin_r, ang_r)[0] * wt).sum().backward()
    f = lambda l_, a_: float((O.render(cfg(True, 1 / 30), *base(sc), l_, a_)[0] * wt).sum())   # noqa: E731
    eps = 1e-6                    # (small: the alpha >= 1/255 gate makes the render piecewise smooth)
    for vec, grad, which in ((lin, lin_r.grad, 0), (ang, ang_r.grad, 1)):
        for i in range(3):
            e = torch.zeros(3, dtype=torch.float64); e[i] = eps
            fd = ((f(lin + e, ang) - f(lin - e, ang)) if which == 0 else (f(lin, ang + e) - f(lin, ang - e))) / (2 * eps)
            assert abs(fd - float(grad[i])) <= 2e-4 * max(1.0, abs(fd)), (which, i, fd, float(grad[i]))


@pytest.mark.parametrize("real_pose", [False, True])
def test_exact_rolling_shutter_is_the_limit_of_row_bands_and_differentiable(oracle, real_pose):
    """VERDICT round 2 'Missing 1': the continuous per-row rolling shutter of the pixel-velocity model
    (RenderConfig.rs_exact; SURVEY App. A 'row time (y/H - 1/2) T_ro').  Known answers:
      * zero readout time -> exactly the render without rolling shutter;
      * R tile-row bands converge to it: the gap shrinks as R grows (bands evaluate tau at their centre row, the exact
        form at every row: first-order error ~ band height);
      * autograd through the row term agrees with central finite differences (twist and a Gaussian centre)."""
    O = oracle
    H, W, n = 128, 96, 400
    sc = O.synthetic_scene(n, W, H, seed=21, dtype=torch.float64, scale_mult=8.0)
    sc["lin_vel"], sc["ang_vel"] = sc["lin_vel"] * 40, sc["ang_vel"] * 25          # ~10 px of readout motion
    sc["sh"][:, 1:] = 0.0            # view-independent colour: the oracle (like splatfacto) detaches the view direction,
    #                                  a finite difference of a mean would see that dependence
    if real_pose:                    # (round 3) the same known answers from a rotated and translated camera
        sc = _posed(sc, O)
    kw = dict(blur_samples=2, exposure_time=1 / 60, gamma=2.2, min_rgb_level=10.0, motion_model="pixel_velocity")
    base = _cfg(O, sc, H, W, rs_bands=1, rolling_shutter_time=0.0, **kw)
    exact0 = _cfg(O, sc, H, W, rs_bands=1, rolling_shutter_time=0.0, rs_exact=True, **kw)
    img0, _ = _render64(O, sc, base)
    img0e, _ = _render64(O, sc, exact0)
    assert torch.equal(img0, img0e)
    T_ro = 1 / 30
    exact = _cfg(O, sc, H, W, rs_bands=1, rolling_shutter_time=T_ro, rs_exact=True, **kw)
    ref, _ = _render64(O, sc, exact)
    assert (ref - img0).abs().max() > 0.05                     # the readout really moves things
    gaps = []
    for R in (1, 2, 4, 8):
        img, _ = _render64(O, sc, _cfg(O, sc, H, W, rs_bands=R, rolling_shutter_time=T_ro, **kw))
        gaps.append(float((img - ref).abs().mean()))
    assert gaps[0] > gaps[1] > gaps[2] > gaps[3] and gaps[3] < 0.3 * gaps[0], gaps
    # gradients through the row term: d/d lin_vel and d/d one mean, against central differences
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    lin = sc["lin_vel"].clone().requires_grad_(True)
    means = sc["means"].clone().requires_grad_(True)
    out, _ = O.render(exact, means, sc["log_scales"].exp(), sc["quats"], torch.sigmoid(sc["opacity_logits"]), sc["sh"],
                      sc["viewmat"], lin, sc["ang_vel"])
    (out * wt).sum().backward()

    def loss(lv, mm):
        o, _ = O.render(exact, mm, sc["log_scales"].exp(), sc["quats"], torch.sigmoid(sc["opacity_logits"]), sc["sh"],
                        sc["viewmat"], lv, sc["ang_vel"])
        return float((o * wt).sum())
    eps = 1e-6
    for j in range(3):
        d = torch.zeros(3, dtype=torch.float64); d[j] = eps
        fd = (loss(sc["lin_vel"] + d, sc["means"]) - loss(sc["lin_vel"] - d, sc["means"])) / (2 * eps)
        assert abs(fd - float(lin.grad[j])) <= 2e-4 * max(1.0, abs(fd)), (j, fd, float(lin.grad[j]))
    g_idx = int(means.grad.abs().sum(dim=1).argmax())
    d = torch.zeros_like(sc["means"]); d[g_idx, 0] = eps
    fd = (loss(sc["lin_vel"], sc["means"] + d) - loss(sc["lin_vel"], sc["means"] - d)) / (2 * eps)
    assert abs(fd - float(means.grad[g_idx, 0])) <= 2e-4 * max(1.0, abs(fd))


def test_upstream_gradient_conventions_are_straight_through_rules(oracle):
    """round 5: the oracle's UPSTREAM mode — the reference's gradient conventions as recollected
    (SURVEY App. A "Backward"), the default of the oracle and of the product — against known answers:
      * no value changes under any flag (forward identical bit for bit);
      * UP_ALPHA_CLAMP: only Gaussians whose alpha reaches the 0.999 clamp see another gradient, and for ONE opaque
        splat over a background the closed form: d C / d o = e^{-sigma} (rgb - bg) where alpha is clamped (true: 0);
      * UP_QUAT_RAW: raw gradient g and true gradient are related by the normalisation's Jacobian,
        v_q = (g - qn (g . qn)) / |q|, and g is NOT tangent to the sphere for a non-unit q;
      * UP_FOV_CLAMP: only Gaussians outside the 1.3 tan(fov/2) guard band change, and there d tx / d px = 1;
      * render() (the fused path) ignores UP_QUAT_RAW: it stands for splatfacto's q/|q| + the kernel."""
    O = oracle
    W, H, n = 64, 48, 300
    sc = O.synthetic_scene(n, W, H, seed=21, dtype=torch.float64, scale_mult=8.0)
    means = sc["means"].clone()
    means[:30, 0] *= 3.0                                                 # beyond the guard band
    quats = sc["quats"] * (0.5 + torch.rand(n, 1, generator=torch.Generator().manual_seed(1), dtype=torch.float64))
    logits = sc["opacity_logits"].clone()
    logits[30:90] = 12.0                                                 # opacity ~ 1: the alpha clamp is reached
    wt = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(2), dtype=torch.float64)

    def grads(up):
        p = {"means": means.clone().requires_grad_(True), "log_scales": sc["log_scales"].clone().requires_grad_(True),
             "quats": quats.clone().requires_grad_(True), "logits": logits.clone().requires_grad_(True)}
        cfg = O.RenderConfig(H, W, sc["fx"], sc["fy"], sc["cx"], sc["cy"], antialiased=False, upstream_grads=up)
        out, _ = O.render(cfg, p["means"], p["log_scales"].exp(), p["quats"], torch.sigmoid(p["logits"]), sc["sh"],
                          sc["viewmat"], sc["lin_vel"], sc["ang_vel"])
        (out * wt).sum().backward()
        return out.detach(), {k: v.grad.clone() for k, v in p.items()}
    img0, g0 = grads(0)
    img7, g7 = grads(O.UPSTREAM)
    img2, g2 = grads(O.UP_QUAT_RAW)
    assert torch.equal(img0, img7) and torch.equal(img0, img2)
    for k in g0:
        assert torch.equal(g2[k], g0[k]), k                              # the fused path ignores the quaternion flag
    _, g4 = grads(O.UP_ALPHA_CLAMP)
    d_op = (g4["logits"] - g0["logits"]).abs()
    assert d_op[30:90].max() > 0 and d_op[90:].max() <= 1e-12 * d_op.max()
    _, g1 = grads(O.UP_FOV_CLAMP)
    lim_x, lim_y = 1.3 * 0.5 * W / sc["fx"], 1.3 * 0.5 * H / sc["fy"]
    inside = ((means[:, 0] / means[:, 2]).abs() <= lim_x) & ((means[:, 1] / means[:, 2]).abs() <= lim_y)
    d_m = (g1["means"] - g0["means"]).abs().sum(1)
    assert d_m[~inside].max() > 0 and d_m[inside].max() <= 1e-12 * d_m.max()
    # all three at once = the sum of what each does where it acts (they touch disjoint parts of the chain)
    assert torch.allclose(g7["logits"], g4["logits"] + (g1["logits"] - g0["logits"]), rtol=0, atol=1e-12)
    # compat op: raw quaternion gradient vs the normalisation's Jacobian
    q_raw = quats.clone().requires_grad_(True)
    q_true = quats.clone().requires_grad_(True)
    wc = torch.rand(n, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    for q_, up in ((q_raw, O.UP_QUAT_RAW), (q_true, 0)):
        pr = O.project_gaussians(means, sc["log_scales"].exp(), 1.0, q_, sc["viewmat"], sc["fx"], sc["fy"], sc["cx"],
                                 sc["cy"], H, W, upstream=up)
        (pr.conics * wc).sum().backward()
    g, nq = q_raw.grad, quats.norm(dim=1, keepdim=True)
    qn = quats / nq
    assert torch.allclose((g - qn * (qn * g).sum(1, keepdim=True)) / nq, q_true.grad, rtol=1e-9, atol=1e-12)
    assert (qn * g).sum(1).abs().max() > 1e-3 * g.abs().max()           # raw: a radial component survives
    assert (qn * q_true.grad).sum(1).abs().max() < 1e-9 * q_true.grad.abs().max()
    # closed form, one opaque splat centred on a pixel over a background: the centre pixel's alpha sits on the clamp
    H1 = W1 = 16
    xy = torch.tensor([[8.5, 8.5]], dtype=torch.float64)
    conic = torch.tensor([[0.05, 0.0, 0.05]], dtype=torch.float64)
    rgb = torch.tensor([[0.9, 0.5, 0.2]], dtype=torch.float64)
    bg = torch.tensor([0.1, 0.3, 0.6], dtype=torch.float64)
    bins = np.array([[0, 1]], dtype=np.int32)
    gid = np.array([0], dtype=np.int32)
    for up, expect_centre in ((O.UP_ALPHA_CLAMP, 1.0), (0, 0.0)):
        o = torch.tensor([0.9999], dtype=torch.float64, requires_grad=True)
        r = O.rasterize_sorted(xy, conic, rgb, o, gid, bins, H1, W1, bg, upstream=up)
        r.img[8, 8, 0].backward()
        # C = alpha rgb + (1 - alpha) bg at the centre (sigma = 0): d C_r / d o = (rgb_r - bg_r) when the clamp lets it pass
        assert abs(float(o.grad) - expect_centre * (0.9 - 0.1)) < 1e-12


def test_regenerate_from_reference_plumbing_with_a_stand_in_module(oracle, tmp_path):
    """tests/golden/make_golden.regenerate_from_reference is the function tests/golden/make_reference_fixtures.py calls
    the day a gsplat with `_torch_impl` imports (none does here: /root/reference's submodules are empty).  It cannot be
    run against the reference today, so its PLUMBING is: a stand-in module with the recollected gsplat 0.1.11 names and
    argument shapes (built on the oracle, so the differences it reports are zero) goes through it end to end — the
    name-bound calls, the 9-tuple unpacking, the fixture file — and a signature with an unknown required parameter is
    reported by name instead of guessed."""
    import sys
    import types
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))
    import make_golden as MG
    O = oracle

    def project_gaussians_forward(means3d, scales, glob_scale, quats, viewmat, intrins, img_size, block_width,
                                  clip_thresh=0.01):
        fx, fy, cx, cy = intrins
        pr = O.project_gaussians(means3d, scales, glob_scale, quats, viewmat, fx, fy, cx, cy, img_size[1], img_size[0],
                                 block_width, clip_thresh)
        return (pr.cov3d, None, pr.xys, pr.depths, pr.radii, pr.conics, pr.compensation, pr.num_tiles_hit, pr.radii > 0)

    def rasterize_forward(xys, depths, radii, conics, num_tiles_hit, colors, opacities, img_height, img_width, block_width,
                          background):
        img, r = O.rasterize_gaussians(xys, depths, radii, conics, num_tiles_hit, colors, opacities.reshape(-1), img_height,
                                       img_width, block_width, background)
        return img, r.final_T, r.final_idx

    stand_in = types.SimpleNamespace(project_gaussians_forward=project_gaussians_forward, rasterize_forward=rasterize_forward)
    diffs = MG.regenerate_from_reference(stand_in, out_dir=tmp_path)
    assert diffs["radii_mismatches"] == 0 and diffs["num_tiles_hit_mismatches"] == 0
    assert diffs["xys_max_abs"] == 0.0 and diffs["image_max_abs_outside_fragile"] < 1e-6
    d = np.load(tmp_path / "ref_static_small.npz")
    assert d["ref_image"].shape == (48, 64, 3) and d["ref_radii"].shape == (600,) and "diff_values" in d.files

    def odd_signature(means3d, scales, quats, viewmat, something_new):
        return None
    with pytest.raises(RuntimeError, match="something_new"):
        MG.regenerate_from_reference(types.SimpleNamespace(project_gaussians_forward=odd_signature,
                                                           rasterize_forward=rasterize_forward), out_dir=tmp_path)
