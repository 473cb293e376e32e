"""The hot path's input contract against fixtures the REFERENCE computed (tests/golden/make_reference_fixtures.py
imports /root/reference/render_video.py::add_velocities and /root/reference/combine.py::process in the build
container; only their outputs travel).  CPU: through the model's pose / velocity handling and the oracle's screw
interpolation.  The `-m gpu` twin (tests/test_gpu_parity.py::test_reference_velocity_fixtures_through_the_hip_subposes)
runs the same check through gs_subpose_viewmats_fwd."""
import json
import os
from pathlib import Path

import pytest
import torch

GOLDEN = Path(__file__).resolve().parent / "golden"
FLIPS = (("-lin", lambda l, a: (-l, a)), ("-ang", lambda l, a: (l, -a)), ("-both", lambda l, a: (-l, -a)),
         ("lin axes", lambda l, a: (l * torch.tensor([1.0, -1.0, -1.0]), a)),
         ("ang axes", lambda l, a: (l, a * torch.tensor([1.0, -1.0, -1.0]))))


def gl_c2w_to_cv_viewmat(c2w):
    c2w = torch.as_tensor(c2w, dtype=torch.float64)
    Rcv = c2w[:3, :3] * torch.tensor([1.0, -1.0, -1.0], dtype=torch.float64)[None, :]
    V = torch.eye(4, dtype=torch.float64)
    V[:3, :3] = Rcv.T
    V[:3, 3] = -(Rcv.T @ c2w[:3, 3])
    return V


def _model(gs):
    cfg = gs.SplatfactoDeblurConfig(background_color="black")
    return gs.SplatfactoDeblurModel(cfg, torch.zeros(4, 3), torch.zeros(4, 3), torch.ones(4, 4), torch.zeros(4),
                                    torch.zeros(4, 3), torch.zeros(4, 15, 3))


def neighbour_cases():
    """(case id, c2w of frame i, its reference velocities, [(dt, c2w of the neighbour at i + dt)])"""
    d = json.load(open(GOLDEN / "ref_add_velocities.json"))
    out = []
    for ci, case in enumerate(d["cases"]):
        fr = case["frames"]
        n = len(fr)
        for i in range(n):
            ip, inx = (i - 1, i + 1)
            if case["loop"]:
                if ip < 0 or inx >= n:
                    continue             # the reference's delta_t = i_next - i_prev is not a time step across the wrap
            else:
                ip, inx = max(0, ip), min(n - 1, inx)
            nb = [(j - i, fr[j]["camera_to_world"]) for j in (ip, inx) if j != i]
            out.append((f"path{ci}/frame{i}", fr[i]["camera_to_world"], fr[i]["camera_linear_velocity"],
                        fr[i]["camera_angular_velocity"], nb))
    return out


def check_frame(gs, subposes, c2w, lin_cam, ang_cam, neighbours, tag):
    """the sub-poses at the neighbours' times, from the frame's pose and the REFERENCE's velocities, land on the
    neighbours' poses (up to the second-order gap between a screw and the reference's translate-and-rotate
    definition); every sign / axis flip misses by several times that"""
    model = _model(gs)
    cam = gs.Camera(torch.tensor(c2w, dtype=torch.float32)[:3], 100, 100, 32, 32, 64, 64,
                    metadata=dict(camera_linear_velocity=lin_cam, camera_angular_velocity=ang_cam, exposure_time=1.0,
                                  rolling_shutter_time=0.0))
    V, lin, ang = model._viewmat_and_velocity(cam)
    assert torch.allclose(V.double(), gl_c2w_to_cv_viewmat(c2w), atol=1e-6), tag
    times = [float(dt) for dt, _ in neighbours]
    want = torch.stack([gl_c2w_to_cv_viewmat(m) for _, m in neighbours])
    span = float((want - gl_c2w_to_cv_viewmat(c2w)[None]).abs().max())

    def gap(l, a):
        return float((subposes(V, l, a, times).double() - want).abs().max())
    good = gap(lin, ang)
    assert span > 0.05 and good < 0.08 * span, (tag, good, span)
    for name, f in FLIPS:
        l2, a2 = f(lin.cpu(), ang.cpu())
        assert gap(l2.to(lin.device), a2.to(ang.device)) > 4 * good, (tag, name, good)
    return good, span


def test_add_velocities_fixture_reproduces_neighbour_poses(gs, oracle):
    cases = neighbour_cases()
    assert len(cases) >= 14
    sub = lambda V, l, a, t: oracle.subpose_viewmats(V.double(), l.double(), a.double(), t)
    for tag, c2w, lin, ang, nb in cases:
        check_frame(gs, sub, c2w, lin, ang, nb, tag)


def combine_cases():
    return json.load(open(GOLDEN / "ref_combine.json"))["cases"]


def test_combine_fixture_loads_and_its_rescaled_velocities_move_the_colmap_poses(gs, oracle, tmp_path):
    """transforms.json as /root/reference/combine.py wrote it: the loader takes its fields as they are; the linear
    velocity the reference rescaled into COLMAP's scene scale (and the untouched angular velocity) carry each COLMAP
    pose onto its neighbours — in a world that differs from the VIO world by scale, rotation and translation."""
    sub = lambda V, l, a, t: oracle.subpose_viewmats(V.double(), l.double(), a.double(), t)
    for ci, case in enumerate(combine_cases()):
        comb, vio = case["combined_transforms"], case["vio_transforms"]
        root = tmp_path / f"case{ci}"
        os.makedirs(root)
        with open(root / "transforms.json", "wt") as f:
            json.dump(comb, f)
        scene = gs.data.load_transforms(str(root), eval_mode="all")
        assert scene.exposure_time == vio["exposure_time"] and scene.rolling_shutter_time == vio["rolling_shutter_time"]
        assert scene.distortion["k1"] == comb["k1"] and scene.applied_transform is not None
        frames = sorted(comb["frames"], key=lambda fr: fr["file_path"])
        vio_frames = sorted(vio["frames"], key=lambda fr: fr["file_path"])
        assert len(scene.cameras) == len(frames) == len(vio_frames)
        s = case["scale"]
        for cam, fr, vf in zip(scene.cameras, frames, vio_frames):
            assert cam.metadata["camera_linear_velocity"] == fr["camera_linear_velocity"]
            assert cam.metadata["camera_angular_velocity"] == fr["camera_angular_velocity"] == vf["camera_angular_velocity"]
            assert torch.allclose(torch.tensor(fr["camera_linear_velocity"], dtype=torch.float64),
                                  s * torch.tensor(vf["camera_linear_velocity"], dtype=torch.float64), rtol=1e-9, atol=0)
            assert cam.metadata["motion_blur_score"] == vf["motion_blur_score"]
        n = len(frames)
        for i in range(1, n - 1):
            nb = [(-1, frames[i - 1]["transform_matrix"]), (1, frames[i + 1]["transform_matrix"])]
            check_frame(gs, sub, frames[i]["transform_matrix"], frames[i]["camera_linear_velocity"],
                        frames[i]["camera_angular_velocity"], nb, f"combine{ci}/frame{i}")
        # the UNSCALED VIO velocity does not fit the COLMAP poses (scale 2.5 / 0.4): the rescale is what is pinned
        i = n // 2
        nb = [(-1, frames[i - 1]["transform_matrix"]), (1, frames[i + 1]["transform_matrix"])]
        with pytest.raises(AssertionError):
            check_frame(gs, sub, frames[i]["transform_matrix"], vio_frames[i]["camera_linear_velocity"],
                        frames[i]["camera_angular_velocity"], nb, "unscaled")
