"""Reference-made fixtures for the hot path's INPUT CONTRACT.

Runs in the BUILD CONTAINER only: it imports the reference's own Python from /root/reference, feeds it pose
sequences, and commits what the reference computed as data (poses in, camera-frame velocities out):

  ref_add_velocities.json   /root/reference/render_video.py::add_velocities (:85-115), finite-difference
                            camera-frame velocities of camera paths (open and looped)
  ref_combine.json          /root/reference/combine.py::process (:6-153), COLMAP poses merged with VIO velocities,
                            linear velocity rescaled by the scene-scale ratio (:89-101), exposure / readout times
                            copied (:135-137)

Nothing of the reference travels: only these JSON files do.  tests/test_reference_fixtures.py (CPU, through the
oracle) and tests/test_gpu_parity.py (-m gpu, through gs_subpose_viewmats_fwd) must reproduce the previous / next
pose of every frame from the velocities the REFERENCE computed, and fail for every sign or axis flip.

The compositing path itself (the SpectacularAI forks of gsplat / nerfstudio) is not vendored.  If a
gsplat with `_torch_impl` is ever importable here, the guarded branch at the bottom regenerates tests/golden/*.npz
from it; today it prints "hot path unpinned".
"""
import argparse
import importlib.util
import json
import math
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
from scipy.spatial.transform import Rotation

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent


def _load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, REF / (name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pose_sequence(n, seed, dt=1.0):
    """n OpenGL camera-to-world poses of a camera moving with constant WORLD-frame linear and angular velocity
    (per frame), from a general start pose"""
    rng = np.random.default_rng(seed)
    R0 = Rotation.from_rotvec(rng.uniform(-1.2, 1.2, 3)).as_matrix()
    p0 = rng.uniform(-2, 2, 3)
    v = rng.uniform(-0.25, 0.25, 3)
    w = rng.uniform(-0.12, 0.12, 3)
    out = []
    for i in range(n):
        m = np.eye(4)
        m[:3, :3] = Rotation.from_rotvec(w * i * dt).as_matrix() @ R0
        m[:3, 3] = p0 + v * i * dt
        out.append(m)
    return out


def make_add_velocities():
    rv = _load("render_video")
    cases = []
    for seed, n, loop in ((11, 7, False), (12, 5, False), (13, 6, True)):
        poses = pose_sequence(n, seed)
        cp = {"camera_path": [{"camera_to_world": m.tolist()} for m in poses]}
        rv.add_velocities(cp, loop=loop)
        cases.append({"loop": loop, "frames": cp["camera_path"]})
    return {"source": "/root/reference/render_video.py::add_velocities (:85-115), called unmodified",
            "units": "per frame index (delta_t counts frames)", "cases": cases}


def make_combine():
    cb = _load("combine")
    rv = _load("render_video")
    out = {"source": "/root/reference/combine.py::process (:6-153), called unmodified on a temporary folder",
           "cases": []}
    for seed, scale in ((21, 2.5), (22, 0.4)):
        rng = np.random.default_rng(seed)
        n = 6
        poses = pose_sequence(n, seed)
        # VIO ("sai") side: poses + camera-frame velocities (the reference's own finite differences), times
        cp = {"camera_path": [{"camera_to_world": m.tolist()} for m in poses]}
        rv.add_velocities(cp, loop=False)
        exposure, readout = 0.012, 0.025
        sai = {"exposure_time": exposure, "rolling_shutter_time": readout, "fl_x": 500.0, "fl_y": 500.0, "cx": 320.0,
               "cy": 240.0, "w": 640, "h": 480, "frames": []}
        for i, f in enumerate(cp["camera_path"]):
            sai["frames"].append({"file_path": "images/frame_%05d.png" % i, "transform_matrix": f["camera_to_world"],
                                  "camera_linear_velocity": f["camera_linear_velocity"],
                                  "camera_angular_velocity": f["camera_angular_velocity"],
                                  "motion_blur_score": float(rng.uniform(0, 1))})
        # COLMAP side: the same cameras in a similarity-transformed world (scale, rotation, translation)
        Rg = Rotation.from_rotvec(rng.uniform(-1, 1, 3)).as_matrix()
        tg = rng.uniform(-3, 3, 3)
        src = {"fl_x": 510.0, "fl_y": 511.0, "cx": 321.0, "cy": 239.0, "w": 640, "h": 480, "k1": 0.01, "k2": -0.002,
               "p1": 0.0, "p2": 0.0, "applied_transform": np.eye(4)[:3].tolist(), "frames": []}
        for i, m in enumerate(poses):
            c = np.eye(4)
            c[:3, :3] = Rg @ m[:3, :3]
            c[:3, 3] = scale * (Rg @ m[:3, 3]) + tg
            src["frames"].append({"file_path": "images/frame_%05d.png" % i, "transform_matrix": c.tolist(),
                                  "colmap_im_id": i + 1})
        with tempfile.TemporaryDirectory() as td:
            inp, sai_dir, outp = (os.path.join(td, d) for d in ("colmap/scene", "sai/scene", "out"))
            for d in (inp, sai_dir):
                os.makedirs(os.path.join(d, "images"))
                with open(os.path.join(d, "sparse_pc.ply"), "wt") as f:
                    f.write("ply\n")
            with open(os.path.join(inp, "transforms.json"), "wt") as f:
                json.dump(src, f)
            with open(os.path.join(sai_dir, "transforms.json"), "wt") as f:
                json.dump(sai, f)
            args = argparse.Namespace(override_calibration=None, sai_input_folder=sai_dir, pose_opt_pass_dir=None,
                                      tolerate_missing=False, keep_intrinsics=False, dataset="fixture",
                                      set_rolling_shutter_to=None, output_folder=outp, dry_run=False,
                                      model_name="splatfacto")
            cb.process(inp, args)
            with open(os.path.join(outp, "transforms.json")) as f:
                combined = json.load(f)
        out["cases"].append({"scale": scale, "similarity_rotation": Rg.tolist(), "similarity_translation": tg.tolist(),
                             "vio_transforms": sai, "combined_transforms": combined})
    return out


def main():
    if not REF.exists():
        sys.exit("needs /root/reference (build container only)")
    for name, fn in (("ref_add_velocities.json", make_add_velocities), ("ref_combine.json", make_combine)):
        with open(HERE / name, "wt") as f:
            json.dump(fn(), f, indent=1)
        print("wrote", HERE / name)
    # ---- the compositing path: regenerate the self-golden fixtures from the reference if it is ever importable ----
    try:
        import gsplat                                    # noqa: F401  (the fork, or upstream 0.1.11 with _torch_impl)
        from gsplat import _torch_impl                   # noqa: F401
    except Exception as e:                               # ModuleNotFoundError in this container
        print(f"hot path unpinned: gsplat / _torch_impl not importable here ({type(e).__name__}: {e}); "
              "tests/golden/*.npz stay self-golden (make_golden.py)")
        return
    import make_golden                                   # type: ignore
    make_golden.regenerate_from_reference(_torch_impl)   # -> tests/golden/ref_static_small.npz (REFERENCE-golden)


if __name__ == "__main__":
    main()
