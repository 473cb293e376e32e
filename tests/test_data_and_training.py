"""transforms.json loader (CPU) and loss / optimizer step (GPU)."""
import json
import math

import numpy as np
import pytest
import torch


def _write_scene(tmp_path, n_frames=17, prefix=False):
    frames = []
    for i in range(n_frames):
        c2w = np.eye(4)
        c2w[0, 3] = 0.1 * i
        name = f"{i:03d}.png"
        if prefix:
            name = ("eval_" if i % 4 == 0 else "train_") + name
        fr = {"file_path": f"./images/{name}", "transform_matrix": c2w.tolist(),
              "camera_linear_velocity": [0.0, 0.0, 0.0] if i % 8 == 0 else [0.1 * i, 0.0, -0.2],
              "camera_angular_velocity": [0.0, 0.0, 0.0] if i % 8 == 0 else [0.0, 0.01 * i, 0.0]}
        if i % 3 == 0:
            fr["motion_blur_score"] = float(i)
        frames.append(fr)
    rng = np.random.default_rng(0)
    rng.shuffle(frames)                                   # the loader must sort by file_path
    meta = {"aabb_scale": 16, "w": 400, "h": 300, "cx": 200.0, "cy": 150.0, "orientation_override": "none",
            "exposure_time": 1 / 60, "rolling_shutter_time": 1 / 50, "fl_x": 320.0, "fl_y": 318.0, "k1": 0, "k2": 0,
            "p1": 0, "p2": 0, "frames": frames, "ply_file_path": "./sparse_pc.ply"}
    (tmp_path / "transforms.json").write_text(json.dumps(meta))
    ply = ["ply", "format ascii 1.0", "element vertex 3", "property float x", "property float y", "property float z",
           "property uint8 red", "property uint8 green", "property uint8 blue", "end_header",
           "0.0 1.0 2.0 255 0 0", "1.5 -1.0 0.5 0 128 0", "-2.0 0.25 3.0 10 20 30"]
    (tmp_path / "sparse_pc.ply").write_text("\n".join(ply) + "\n")
    return meta


def test_load_transforms_contract(gs, tmp_path):
    _write_scene(tmp_path)
    sc = gs.load_transforms(str(tmp_path))
    assert len(sc.cameras) == 17 and sc.exposure_time == pytest.approx(1 / 60) and sc.rolling_shutter_time == pytest.approx(0.02)
    assert [c.metadata["cam_idx"] for c in sc.cameras] == list(range(17))
    assert [p.split("/")[-1] for p in sc.image_paths] == [f"{i:03d}.png" for i in range(17)]      # sorted by path
    assert sc.eval_indices == [0, 8, 16] and len(sc.train_indices) == 14                            # i % 8 == 0
    for i in sc.eval_indices:          # eval frames of the synthetic sets are static (process_synthetic_inputs.py:287-293)
        assert sum(map(abs, sc.cameras[i].metadata["camera_linear_velocity"])) == 0
        assert sc.cameras[i].metadata["is_eval"]
    c5 = sc.cameras[5]
    assert c5.metadata["camera_linear_velocity"] == pytest.approx([0.5, 0.0, -0.2])
    assert c5.camera_to_world.shape == (3, 4) and c5.camera_to_world[0, 3].item() == pytest.approx(0.5)
    assert (c5.fx, c5.fy, c5.cx, c5.cy, c5.width, c5.height) == (320.0, 318.0, 200.0, 150.0, 400, 300)
    assert sc.cameras[3].metadata["motion_blur_score"] == 3.0 and "motion_blur_score" not in sc.cameras[4].metadata
    xyz, rgb = gs.load_seed_points_ply(sc.ply_file_path)
    assert xyz.shape == (3, 3) and torch.allclose(rgb[0], torch.tensor([1.0, 0.0, 0.0]))
    half = gs.load_transforms(str(tmp_path / "transforms.json"), downscale=2)
    assert (half.cameras[0].width, half.cameras[0].fx) == (200, 160.0)


def test_eval_split_modes(gs, tmp_path):
    _write_scene(tmp_path, prefix=True)
    sc = gs.load_transforms(str(tmp_path), eval_mode="filename")
    names = [p.split("/")[-1] for p in sc.image_paths]
    assert all(names[i].startswith("eval_") for i in sc.eval_indices) and len(sc.eval_indices) == 5
    assert all(names[i].startswith("train_") for i in sc.train_indices) and len(sc.train_indices) == 12
    al = gs.load_transforms(str(tmp_path), eval_mode="all")
    assert al.train_indices == list(range(17)) == al.eval_indices
    with pytest.raises(ValueError):
        gs.data.split_indices(["a.png", "b.png"], "filename")
    with pytest.raises(KeyError):
        (tmp_path / "bad.json").write_text(json.dumps({"w": 1, "frames": []}))
        gs.load_transforms(str(tmp_path / "bad.json"))


def test_ssim_and_loss_identities(gs):
    T = gs.training
    g = torch.Generator().manual_seed(0)
    a = torch.rand(40, 48, 3, generator=g)
    assert T.ssim(a, a).item() == pytest.approx(1.0, abs=1e-6)
    assert T.image_loss(a, a).item() == pytest.approx(0.0, abs=1e-6)
    b = (a + 0.2 * torch.randn(40, 48, 3, generator=g)).clamp(0, 1)
    assert 0 < T.ssim(a, b).item() < 0.95 and T.psnr(a, a) == float("inf") and 5 < T.psnr(a, b) < 30
    ls = torch.log(torch.tensor([[1.0, 1.0, 1.0], [1.0, 1.0, 30.0]]))
    assert T.scale_regularization(ls).item() == pytest.approx(0.1 * (20.0 / 2), rel=1e-5)      # only the needle is penalised


@pytest.mark.gpu
def test_training_recovers_blurred_target(gs, oracle, dev):
    """End-to-end: targets are motion-blurred renders of ground-truth Gaussians; a perturbed copy trained with
    the reference's loss through the HIP forward/backward must raise PSNR by a wide margin."""
    O = oracle
    W, H, n = 128, 96, 1500
    sc = O.synthetic_scene(n, W, H, seed=5, scale_mult=7.0)
    cfg = gs.SplatfactoDeblurConfig(blur_samples=3, rs_bands=2, gamma=2.2, min_rgb_level=10.0, background_color="black",
                                    use_scale_regularization=True)
    c2w = torch.eye(4)[:3].clone()
    c2w[:, 1] *= -1
    c2w[:, 2] *= -1
    cams = [gs.Camera(c2w.clone(), sc["fx"], sc["fy"], sc["cx"], sc["cy"], W, H,
                      metadata=dict(cam_idx=i, camera_linear_velocity=[0.6 * (i - 1), 0.2, 0.0],
                                    camera_angular_velocity=[0.0, 0.2 * (i - 1), 0.1], exposure_time=1 / 60,
                                    rolling_shutter_time=1 / 30)) for i in range(3)]
    gt_model = gs.SplatfactoDeblurModel.from_scene(cfg, sc, dev, num_cameras=3)
    with torch.no_grad():
        gt_model.train()
        targets = [gt_model.get_outputs(c)["rgb"].clone() for c in cams]
    g = torch.Generator().manual_seed(1)
    pert = dict(sc)
    pert["sh"] = sc["sh"] + 0.3 * torch.randn(sc["sh"].shape, generator=g)
    pert["opacity_logits"] = sc["opacity_logits"] + 0.5 * torch.randn(n, generator=g)
    pert["means"] = sc["means"] + 0.01 * torch.randn(n, 3, generator=g)
    model = gs.SplatfactoDeblurModel.from_scene(cfg, pert, dev, num_cameras=3)
    opts = gs.training.make_optimizers(model, lr_scale=4.0)
    with torch.no_grad():
        model.train()
        psnr0 = np.mean([gs.training.psnr(model.get_outputs(c)["rgb"], t) for c, t in zip(cams, targets)])
    hist = []
    for it in range(150):
        i = it % 3
        hist.append(gs.training.train_step(model, opts, cams[i], targets[i]))
    with torch.no_grad():
        psnr1 = np.mean([gs.training.psnr(model.get_outputs(c)["rgb"], t) for c, t in zip(cams, targets)])
    assert all(math.isfinite(h["loss"]) for h in hist)
    assert np.mean([h["loss"] for h in hist[-9:]]) < 0.6 * np.mean([h["loss"] for h in hist[:9]])
    assert psnr1 > psnr0 + 4.0, (psnr0, psnr1)


def test_dataset_writer_round_trips_through_the_loader(gs, tmp_path):
    """write_transforms / write_seed_points_ply / save_image emit what load_transforms / load_seed_points_ply /
    load_image read: the wire format of /root/reference/process_synthetic_inputs.py:113-129,171-176,203-219"""
    frames = []
    for i in range(9):
        c2w = torch.eye(4)
        c2w[0, 3] = 0.1 * i
        frames.append(dict(file_path=f"./images/{i:03d}.png", transform_matrix=c2w.tolist(),
                           camera_linear_velocity=[0.0] * 3 if i % 8 == 0 else [0.1, 0.2, -0.3],
                           camera_angular_velocity=[0.0] * 3 if i % 8 == 0 else [0.01, 0.0, 0.02]))
    gs.data.write_transforms(str(tmp_path), 64, 48, 50.0, 51.0, 32.0, 24.0, 1 / 30, 1 / 60, frames, "./sparse_pc.ply")
    meta = json.loads((tmp_path / "transforms.json").read_text())
    assert set(meta) >= {"aabb_scale", "cx", "cy", "exposure_time", "fl_x", "fl_y", "frames", "h", "k1", "k2",
                         "orientation_override", "p1", "p2", "rolling_shutter_time", "w"}       # :113-129
    assert set(meta["frames"][0]) == {"camera_angular_velocity", "camera_linear_velocity", "file_path",
                                      "transform_matrix"}                                      # :171-176
    g = torch.Generator().manual_seed(0)
    xyz, rgb = torch.randn(5, 3, generator=g), torch.rand(5, 3, generator=g)
    gs.data.write_seed_points_ply(str(tmp_path / "sparse_pc.ply"), xyz, rgb)
    img = torch.rand(48, 64, 3, generator=g)
    gs.data.save_image(str(tmp_path / "images" / "000.png"), img)
    sc = gs.load_transforms(str(tmp_path))
    assert sc.eval_indices == [0, 8] and sc.exposure_time == pytest.approx(1 / 30)
    assert sc.cameras[3].metadata["camera_linear_velocity"] == pytest.approx([0.1, 0.2, -0.3])
    x2, c2 = gs.load_seed_points_ply(sc.ply_file_path)
    assert torch.allclose(x2, xyz, atol=1e-5) and (c2 - rgb).abs().max() <= 0.5 / 255 + 1e-6
    back = gs.data.load_image(sc.image_paths[0])
    assert back.shape == (48, 64, 3) and (back - img).abs().max() <= 0.5 / 255 + 1e-6


@pytest.mark.gpu
def test_end_to_end_deblurring_on_a_self_generated_dataset(gs, dev, tmp_path):
    """The reference's own experiment in miniature (/root/reference/train.py:29-76 variants, :78-109 scoring): a
    dataset in the reference's wire format whose TRAINING frames are motion-blurred (64 dense sub-poses of a
    ground-truth scene) and whose EVALUATION frames (i % 8 == 0) are sharp; the same initial model is trained with
    blur_samples = 0 (no compensation) and 5 (the default), with both motion models, and scored on the sharp
    frames.  Modelling the blur must pay: >= 1 dB PSNR and a higher SSIM."""
    import synthetic_dataset as SD          # tools/synthetic_dataset.py (conftest puts tools/ on sys.path)
    root = str(tmp_path / "ds")
    info = SD.generate(root, dev, width=160, height=120, n_frames=17, n_gaussians=4000, speed=1.5, dense_samples=64)
    scene = gs.load_transforms(root)
    assert scene.eval_indices == [0, 8, 16]
    images = [gs.data.load_image(p, dev) for p in scene.image_paths]
    gt = info["scene"]
    g = torch.Generator().manual_seed(1)
    start = dict(gt)
    start["sh"] = gt["sh"] + 0.15 * torch.randn(gt["sh"].shape, generator=g) * (torch.arange(16) == 0)[None, :, None]
    start["means"] = gt["means"] + 0.004 * torch.randn(gt["means"].shape, generator=g)
    start["log_scales"] = gt["log_scales"] + 0.1
    res = {}
    for name, bs, mm in (("static", 0, "se3"), ("se3", 5, "se3"), ("pixel_velocity", 5, "pixel_velocity")):
        cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=bs, gamma=2.2 if bs else 1.0, min_rgb_level=0.0,
                                        rolling_shutter_compensation=False, motion_model=mm)
        model = gs.SplatfactoDeblurModel.from_scene(cfg, start, dev, num_cameras=len(scene.cameras))
        res[name] = gs.training.train_scene(model, scene, images, iterations=700, lr_scale=1.0)["results"]
    print("sharp-frame scores:", {k: {m: round(v, 3) for m, v in r.items()} for k, r in res.items()})
    for name in ("se3", "pixel_velocity"):
        assert res[name]["psnr"] > res["static"]["psnr"] + 1.0, res
        assert res[name]["ssim"] > res["static"]["ssim"], res


@pytest.mark.gpu
@pytest.mark.parametrize("motion_model", ["se3", "pixel_velocity"])
def test_velocity_optimizer_recovers_known_velocities_from_rolling_shutter_frames(gs, dev, motion_model):
    """--camera-velocity-optimizer.enabled / .zero-initial-velocities (/root/reference/train.py:66,70), end to end:
    ground-truth Gaussians (constants), frames rendered with rolling shutter and motion blur from KNOWN velocities (a
    per-pixel-row ground truth that uses none of the renderer's rolling-shutter machinery), a model that starts from
    zero velocities.  The rasterizer's twist gradients alone must find the motion: the loss falls in every frame (by 3x to
    40x in most) and the learned angular velocity points along the true one (cosine > 0.9; blur alone would leave its
    SIGN open, the rolling shutter does not) — frames of reference (OpenGL data, OpenCV kernels), signs and the time
    conventions of both rolling-shutter modes are all in that number."""
    import synthetic_dataset as SD          # tools/synthetic_dataset.py (conftest puts tools/ on sys.path)
    from gsdeblur_amd.model import Camera
    H, W, exposure, t_ro = 120, 160, 1 / 15, 1 / 15
    gt = SD.make_gt_scene(4000, 0)
    traj = SD.trajectory(17, 1.5, 0)
    frames = [i for i in range(len(traj)) if i % 8 != 0][:6]
    ref_cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=1, gamma=2.2, min_rgb_level=0.0, background_color="black",
                                        rolling_shutter_compensation=False)
    ref_model = gs.SplatfactoDeblurModel.from_scene(ref_cfg, gt, dev).eval()
    cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=5, gamma=2.2, min_rgb_level=0.0, background_color="black",
                                    rolling_shutter_compensation=True, rs_bands=8, motion_model=motion_model,
                                    rolling_shutter_mode="exact" if motion_model == "pixel_velocity" else "bands")
    cfg.camera_velocity_optimizer.enabled = True
    cfg.camera_velocity_optimizer.zero_initial_velocities = True
    model = gs.SplatfactoDeblurModel.from_scene(cfg, gt, dev, num_cameras=len(traj))
    opts = gs.training.make_optimizers(model)
    cams, imgs = {}, {}
    with torch.no_grad():
        for i in frames:
            fr = traj[i]
            md = dict(cam_idx=i, camera_linear_velocity=fr["lin"].tolist(), camera_angular_velocity=fr["ang"].tolist(),
                      exposure_time=exposure, rolling_shutter_time=t_ro)
            cams[i] = Camera(fr["c2w"][:3], 0.75 * W, 0.75 * W, W / 2.0, H / 2.0, W, H, metadata=md)
            imgs[i] = SD.render_rolling_shutter_frame(ref_model, cams[i], exposure, t_ro, 2.2)
    first = {}
    for it in range(200):
        for i in frames:
            loss = gs.training.eval_camera_step(model, opts, cams[i], imgs[i])
            first.setdefault(i, loss)
    flip = torch.tensor([1.0, -1.0, -1.0])
    good = 0
    for i in frames:
        adj = model.velocity_adjustment[i].detach().cpu()
        ang_t = traj[i]["ang"] * flip
        last = gs.training.eval_camera_step(model, opts, cams[i], imgs[i])
        c_a = float((adj[3:] * ang_t).sum() / (adj[3:].norm() * ang_t.norm() + 1e-12))
        print(f"{motion_model} frame {i}: loss {first[i]:.4f} -> {last:.4f}, angular velocity cosine {c_a:+.3f}")
        assert last < 0.85 * first[i], (i, first[i], last)
        good += int(c_a > 0.9)
    assert good >= len(frames) - 1, good
    for k, v in model.gauss_params().items():
        assert v.grad is None, k                         # the Gaussians stayed constants


@pytest.mark.gpu
@pytest.mark.parametrize("convention", [0, 6, 7])
def test_pose_optimizer_pulls_perturbed_cameras_back_to_the_true_pose(gs, dev, convention):
    """--camera-optimizer.mode=SO3xR3 (/root/reference/train.py:40), end to end: ground-truth Gaussians (constants),
    sharp frames rendered from the TRUE poses, cameras handed to the model with a known perturbation (composed in the
    camera frame, c2w @ exp(delta), as nerfstudio's camera optimizer does).  Through the rasterizer's viewmat gradients
    alone the pose adjustment must undo it: the composed pose c2w_perturbed @ exp(adj) ends closer to the true pose in
    translation AND rotation — which pins the viewmat gradient's frame, sign and the OpenGL -> OpenCV flip.
    `convention` = ops.UPSTREAM_GRADS: with the TRUE derivatives (0) and with the product's DEFAULT (6, round 6) all four
    frames must recover (0.03-0.8 cm) — the default configuration is not given a relaxed bar.  With all three of the
    reference's conventions as recollected (7, opt-in) three do and frame 3 — whose view
    holds large splats centred beyond the 1.3 tan(fov/2) guard band — is pushed AWAY (1.3 -> 3.8 cm): the straight-through
    gradient of the fov clamp (bit 1) tells the optimizer that moving the camera changes those splats' footprints when it
    does not (bisected on the GPU: masks 7 and 3 fail, 6 and 0 recover; the other reading of "as if inactive", the VJP of
    the unclamped EWA projection, was built and loses 11.8 cm).  Recorded here as what that convention costs, DESIGN.md
    §1.2; that measurement is why the default is 6."""
    import math
    from gsdeblur_amd import ops
    saved_convention = ops.UPSTREAM_GRADS
    ops.UPSTREAM_GRADS = convention
    try:
        _pose_optimizer_case(gs, dev, convention)
    finally:
        ops.UPSTREAM_GRADS = saved_convention


def _pose_optimizer_case(gs, dev, convention):
    import math
    import synthetic_dataset as SD          # tools/synthetic_dataset.py (conftest puts tools/ on sys.path)
    from gsdeblur_amd.model import Camera, _so3_exp
    H, W = 120, 160
    gt = SD.make_gt_scene(4000, 0)
    traj = SD.trajectory(9, 1.0, 0)
    frames = [1, 3, 5, 7]
    cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=0, gamma=1.0, min_rgb_level=0.0, background_color="black",
                                    rolling_shutter_compensation=False)
    ref_model = gs.SplatfactoDeblurModel.from_scene(cfg, gt, dev).eval()
    cfg2 = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=0, gamma=1.0, min_rgb_level=0.0, background_color="black",
                                     rolling_shutter_compensation=False)
    cfg2.camera_optimizer.mode = "SO3xR3"
    model = gs.SplatfactoDeblurModel.from_scene(cfg2, gt, dev, num_cameras=len(traj))
    opts = gs.training.make_optimizers(model, lr_scale=10.0)
    g = torch.Generator().manual_seed(2)
    cams, imgs, true_c2w, pert_c2w = {}, {}, {}, {}
    z3 = [0.0, 0.0, 0.0]
    with torch.no_grad():
        for i in frames:
            c2w = traj[i]["c2w"][:3].clone()
            md = dict(cam_idx=i, camera_linear_velocity=z3, camera_angular_velocity=z3, exposure_time=0.0,
                      rolling_shutter_time=0.0)
            imgs[i] = ref_model.get_outputs(Camera(c2w, 0.75 * W, 0.75 * W, W / 2.0, H / 2.0, W, H, metadata=md))["rgb"]
            d_t = 0.03 * (torch.rand(3, generator=g) - 0.5)
            d_r = 0.03 * (torch.rand(3, generator=g) - 0.5)
            p = c2w.clone()
            p[:, 3] = c2w[:, 3] + c2w[:, :3] @ d_t
            p[:, :3] = c2w[:, :3] @ _so3_exp(d_r)
            true_c2w[i], pert_c2w[i] = c2w, p
            cams[i] = Camera(p, 0.75 * W, 0.75 * W, W / 2.0, H / 2.0, W, H, metadata=md)
    first = {}
    for it in range(300):
        for i in frames:
            loss = gs.training.eval_camera_step(model, opts, cams[i], imgs[i])
            first.setdefault(i, loss)

    def pose_error(a, b):
        dr = a[:, :3].T @ b[:, :3]
        ang = math.acos(max(-1.0, min(1.0, (float(dr.trace()) - 1.0) / 2.0)))
        return float((a[:, 3] - b[:, 3]).norm()), ang
    recovered = {}
    for i in frames:
        adj = model.pose_adjustment[i].detach().cpu()
        cur = pert_c2w[i].clone()
        cur[:, 3] = pert_c2w[i][:, 3] + pert_c2w[i][:, :3] @ adj[:3]
        cur[:, :3] = pert_c2w[i][:, :3] @ _so3_exp(adj[3:])
        t0, r0 = pose_error(pert_c2w[i], true_c2w[i])
        t1, r1 = pose_error(cur, true_c2w[i])
        last = gs.training.eval_camera_step(model, opts, cams[i], imgs[i])
        print(f"frame {i}: loss {first[i]:.4f} -> {last:.4f}; translation error {t0 * 100:.2f} -> {t1 * 100:.2f} cm, "
              f"rotation error {math.degrees(r0):.2f} -> {math.degrees(r1):.2f} deg")
        recovered[i] = last < 0.5 * first[i] and t1 < 0.6 * t0 and r1 < 0.6 * r0
    print(f"pose optimizer, gradient convention {convention}: frames recovered {recovered}")
    if convention in (0, 6):
        assert all(recovered.values()), recovered
    else:
        assert sum(recovered.values()) >= 3 and recovered[1] and recovered[5], recovered


@pytest.mark.gpu
def test_optimize_eval_cameras_moves_only_the_eval_cameras(gs, dev, tmp_path):
    """--optimize-eval-cameras (/root/reference/train.py:180-183): a step on an evaluation frame updates that
    frame's pose / velocity adjustment and nothing else — no gradient reaches the Gaussians"""
    import synthetic_dataset as SD          # tools/synthetic_dataset.py (conftest puts tools/ on sys.path)
    root = str(tmp_path / "ds")
    info = SD.generate(root, dev, width=96, height=64, n_frames=9, n_gaussians=1500, speed=1.0, dense_samples=16)
    scene = gs.load_transforms(root)
    images = [gs.data.load_image(p, dev) for p in scene.image_paths]
    cfg = gs.SplatfactoDeblurConfig(sh_degree=3, blur_samples=3, rolling_shutter_compensation=False)
    cfg.camera_optimizer.mode = "SO3xR3"
    cfg.camera_velocity_optimizer.enabled = True
    model = gs.SplatfactoDeblurModel.from_scene(cfg, info["scene"], dev, num_cameras=len(scene.cameras))
    opts = gs.training.make_optimizers(model)
    before = {k: v.detach().clone() for k, v in model.gauss_params().items()}
    e = scene.eval_indices[1]
    with torch.no_grad():                      # a wrong evaluation pose to correct
        scene.cameras[e].camera_to_world[:, 3] += torch.tensor([0.03, -0.02, 0.01])
    l0 = gs.training.eval_camera_step(model, opts, scene.cameras[e], images[e])
    for _ in range(40):
        l1 = gs.training.eval_camera_step(model, opts, scene.cameras[e], images[e])
    assert l1 < l0
    for k, v in model.gauss_params().items():
        assert torch.equal(v.detach(), before[k]) and v.grad is None, k
    adj = model.pose_adjustment.detach()
    assert adj[e].abs().sum().item() > 0 and adj[[i for i in range(len(scene.cameras)) if i != e]].abs().sum().item() == 0


def test_resolution_schedule_and_downscaled_ground_truth(gs):
    """`num_downscales` (/root/reference/train.py:14; splatfacto 1.1.0 `_get_downscale_factor`): while training the
    model renders at 1 / 2^max(n - step // schedule, 0) of the camera's resolution and the ground truth is averaged
    over d x d blocks; evaluation is always full size"""
    cfg = gs.SplatfactoDeblurConfig(num_downscales=2, resolution_schedule=100)
    sc = gs.data.synthetic_scene(50, 64, 48, seed=1)
    m = gs.SplatfactoDeblurModel.from_scene(cfg, sc, "cpu")
    m.train()
    got = []
    for step in (0, 99, 100, 199, 200, 5000):
        m.step = step
        got.append(m.downscale_factor())
    assert got == [4, 4, 2, 2, 1, 1]
    m.eval()
    m.step = 0
    assert m.downscale_factor() == 1
    cam = gs.Camera(torch.eye(4)[:3], 50.0, 52.0, 32.0, 24.0, 65, 49, {"cam_idx": 0})
    c4 = cam.rescaled(4)
    assert (c4.width, c4.height) == (16, 12) and c4.fx == 12.5 and c4.cy == 6.0 and c4.metadata is cam.metadata
    img = torch.arange(49 * 65 * 3, dtype=torch.float32).reshape(49, 65, 3)
    small = gs.training.downscale_image(img, 4)
    assert small.shape == (12, 16, 3)
    assert torch.allclose(small[2, 3], img[8:12, 12:16].reshape(-1, 3).mean(0))
    assert gs.training.downscale_image(img, 1) is img


def test_undistort_image_inverts_the_opencv_lens_model(gs):
    """k1, k2, p1, p2 of transforms.json (/root/reference/process_synthetic_inputs.py:113-129; combine.py:109-131 may
    carry COLMAP's estimates) are no longer parsed and ignored: `undistort_image` resamples a frame to the pinhole
    camera the rasterizer models.  Check: render an analytic pattern THROUGH the lens model, undistort, compare with
    the pattern seen by the ideal pinhole."""
    H, W, fx, fy, cx, cy = 96, 128, 90.0, 92.0, 64.0, 48.0
    dist = {"k1": -0.12, "k2": 0.03, "p1": 0.004, "p2": -0.003}

    def pattern(x, y):                                   # smooth in normalised pinhole coordinates
        return torch.stack([0.5 + 0.4 * torch.sin(5 * x) * torch.cos(4 * y), 0.5 + 0.4 * torch.cos(3 * x + 2 * y),
                            0.5 + 0.3 * torch.sin(6 * y)], -1)

    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    xd, yd = (u + 0.5 - cx) / fx, (v + 0.5 - cy) / fy     # what a distorted sensor pixel measures ...
    x, y = xd.clone(), yd.clone()                         # ... is the scene point whose DISTORTED position it is:
    for _ in range(30):                                   # invert the model by fixed-point iteration
        r2 = x * x + y * y
        rad = 1 + r2 * (dist["k1"] + r2 * dist["k2"])
        dx = 2 * dist["p1"] * x * y + dist["p2"] * (r2 + 2 * x * x)
        dy = dist["p1"] * (r2 + 2 * y * y) + 2 * dist["p2"] * x * y
        x, y = (xd - dx) / rad, (yd - dy) / rad
    distorted = pattern(x, y).float()
    ideal = pattern(xd, yd).float()                       # the pinhole image on the same pixel grid
    got = gs.data.undistort_image(distorted, fx, fy, cx, cy, dist)
    inner = (slice(8, H - 8), slice(10, W - 10))          # away from the border (source pixels outside the frame)
    assert (got[inner] - ideal[inner]).abs().max() < 2e-3
    assert (distorted[inner] - ideal[inner]).abs().max() > 2e-2      # the lens really moved things
    assert gs.data.undistort_image(distorted, fx, fy, cx, cy, {"k1": 0, "k2": 0, "p1": 0, "p2": 0}) is distorted


def test_undistorted_frames_are_cropped_to_valid_pixels_and_the_camera_follows(gs, tmp_path):
    """with pincushion (k1 > 0) or tangential coefficients the border of the undistorted frame looks
    outside the sensor; left black and unmasked it pulls the model towards black at the edges.  load_scene_images crops
    every undistorted frame to the rectangle in which all pixels are valid and replaces the scene's cameras by those of
    the cropped frames (same focal lengths, principal point shifted by the crop origin) — nerfstudio's datamanager does
    the equivalent with cv2.getOptimalNewCameraMatrix(alpha=0) + ROI.  A bright frame must come out without a single
    dark pixel, and a 3D point must project to the same picture content before and after."""
    import json
    H, W, fx, fy, cx, cy = 90, 120, 100.0, 101.0, 61.0, 44.0
    dist = {"k1": 0.15, "k2": 0.02, "p1": 0.01, "p2": -0.008}
    x0, y0, x1, y1 = gs.data.undistort_roi(H, W, fx, fy, cx, cy, dist)
    assert 0 < x0 < x1 < W and 0 < y0 < y1 < H and (x1 - x0) * (y1 - y0) > 0.5 * H * W
    white = torch.ones(H, W, 3)
    full = gs.data.undistort_image(white, fx, fy, cx, cy, dist)
    assert full.min() < 0.5                                             # the uncropped frame HAS black border pixels
    crop, roi = gs.data.undistort_image(white, fx, fy, cx, cy, dist, crop=True)
    assert roi == (x0, y0, x1, y1) and crop.shape == (y1 - y0, x1 - x0, 3)
    assert crop.min() > 1 - 1e-5                                        # ... the cropped one has none
    # growing the rectangle by one pixel on any side brings an invalid pixel in
    for dx0, dy0, dx1, dy1 in ((-1, 0, 0, 0), (0, -1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)):
        a, b, c, d = x0 + dx0, y0 + dy0, x1 + dx1, y1 + dy1
        if a >= 0 and b >= 0 and c <= W and d <= H:
            assert full[b:d, a:c].min() < 1 - 1e-5, (dx0, dy0, dx1, dy1)
    # a scene on disk: the loader crops and the cameras follow
    root = tmp_path / "scene"
    (root / "images").mkdir(parents=True)
    grad = torch.linspace(0.2, 1.0, W)[None, :, None].expand(H, W, 3).contiguous()
    for i in range(2):
        gs.data.save_image(str(root / "images" / f"f{i}.png"), grad)
    meta = dict(w=W, h=H, fl_x=fx, fl_y=fy, cx=cx, cy=cy, **dist,
                frames=[dict(file_path=f"./images/f{i}.png", transform_matrix=torch.eye(4).tolist()) for i in range(2)])
    with open(root / "transforms.json", "wt") as f:
        json.dump(meta, f)
    scene = gs.data.load_transforms(str(root), eval_mode="all")
    imgs = gs.data.load_scene_images(scene)
    for cam, img in zip(scene.cameras, imgs):
        assert (cam.width, cam.height) == (x1 - x0, y1 - y0) == (img.shape[1], img.shape[0])
        assert cam.fx == fx and cam.fy == fy and cam.cx == cx - x0 and cam.cy == cy - y0
        assert img.min() > 0.19
    again = gs.data.load_scene_images(scene)                            # idempotent: the cameras are not cropped twice
    assert torch.equal(again[0], imgs[0]) and scene.cameras[0].width == x1 - x0
    # the raw frames of a scene whose cameras were already cropped come back in the cameras' size
    raw = gs.data.load_scene_images(scene, undistort=False)
    assert raw[0].shape == (y1 - y0, x1 - x0, 3) and scene.cameras[0].width == x1 - x0
    assert torch.allclose(raw[0], grad[y0:y1, x0:x1], atol=1 / 255)   # ... untouched pixels of the stored frame
    fresh = gs.data.load_transforms(str(root), eval_mode="all")
    assert gs.data.load_scene_images(fresh, undistort=False)[0].shape == (H, W, 3) and fresh.cameras[0].width == W
