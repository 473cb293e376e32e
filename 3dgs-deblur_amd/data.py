"""`transforms.json` loader: the wire format between the reference's data preparation and the model
(App. B).  Field contract, all from the reference tree:
  top level   w h cx cy fl_x fl_y k1 k2 p1 p2 exposure_time rolling_shutter_time frames[]
              (/root/reference/process_synthetic_inputs.py:113-129), optional ply_file_path (:298),
              applied_transform, k3 (/root/reference/combine.py:118,127)
  per frame   file_path transform_matrix[4][4] camera_linear_velocity[3] camera_angular_velocity[3]
              (/root/reference/process_synthetic_inputs.py:171-176), optional motion_blur_score
              (/root/reference/combine.py:79-81)
  conventions camera-to-world in OpenGL axes; velocities in that camera frame
              (/root/reference/process_synthetic_inputs.py:157-165)
  eval split  "interval": sorted index i % 8 == 0 (/root/reference/train.py:174-177,
              /root/reference/process_synthetic_inputs.py:287-293); "filename": eval_/train_ prefixes
              (/root/reference/train_eval_split_by_blur_score.py:34-37, /root/reference/train.py:169-172);
              "all" (/root/reference/train.py:164-167)
Host code (json + torch tensors); `load_image` decodes the frames (PNG / JPEG through PIL, or .npy float arrays)
and `write_transforms` / `write_seed_points_ply` emit the same wire format (used by the self-generated datasets of
``tools/synthetic_dataset.py``, the offline stand-in for the Zenodo downloads of /root/reference/download_data.py:21-33).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from .model import Camera


@dataclass
class TransformsScene:
    cameras: List[Camera]
    image_paths: List[str]
    train_indices: List[int]
    eval_indices: List[int]
    exposure_time: float
    rolling_shutter_time: float
    distortion: Dict[str, float] = field(default_factory=dict)
    ply_file_path: Optional[str] = None
    applied_transform: Optional[torch.Tensor] = None


def split_indices(file_paths: List[str], eval_mode: str = "interval", eval_interval: int = 8) -> Tuple[List[int], List[int]]:
    """(train, eval) indices into the path-sorted frame list."""
    n = len(file_paths)
    if eval_mode == "all":
        return list(range(n)), list(range(n))
    if eval_mode == "interval":
        ev = [i for i in range(n) if i % eval_interval == 0]
        return [i for i in range(n) if i % eval_interval != 0], ev
    if eval_mode == "filename":
        ev = [i for i, p in enumerate(file_paths) if os.path.basename(p).startswith("eval_")]
        tr = [i for i, p in enumerate(file_paths) if os.path.basename(p).startswith("train_")]
        if not ev and not tr:
            raise ValueError("eval_mode='filename' needs eval_/train_ prefixed file names")
        return tr, ev
    raise ValueError(f"unknown eval_mode {eval_mode!r}")


def load_transforms(path: str, eval_mode: str = "interval", eval_interval: int = 8,
                    downscale: int = 1) -> TransformsScene:
    """Parse a nerfstudio-style transforms.json written by the reference's converters."""
    json_path = os.path.join(path, "transforms.json") if os.path.isdir(path) else path
    root = os.path.dirname(json_path)
    with open(json_path, "rt") as f:
        meta = json.load(f)
    for k in ("w", "h", "fl_x", "fl_y", "cx", "cy", "frames"):
        if k not in meta:
            raise KeyError(f"transforms.json misses '{k}'")
    frames = sorted(meta["frames"], key=lambda fr: fr["file_path"])
    exposure = float(meta.get("exposure_time", 0.0))
    readout = float(meta.get("rolling_shutter_time", 0.0))
    s = 1.0 / float(downscale)
    W, H = int(round(meta["w"] * s)), int(round(meta["h"] * s))
    cams, paths = [], []
    is_eval_set = set()
    tr, ev = split_indices([fr["file_path"] for fr in frames], eval_mode, eval_interval)
    is_eval_set.update(ev)
    for i, fr in enumerate(frames):
        c2w = torch.tensor(fr["transform_matrix"], dtype=torch.float32)
        if c2w.shape != (4, 4):
            raise ValueError(f"frame {fr['file_path']}: transform_matrix must be 4x4")
        md = {
            "cam_idx": i,
            "camera_linear_velocity": [float(v) for v in fr.get("camera_linear_velocity", (0.0, 0.0, 0.0))],
            "camera_angular_velocity": [float(v) for v in fr.get("camera_angular_velocity", (0.0, 0.0, 0.0))],
            "exposure_time": exposure,
            "rolling_shutter_time": readout,
            "is_eval": i in is_eval_set and eval_mode != "all",
        }
        if "motion_blur_score" in fr:
            md["motion_blur_score"] = float(fr["motion_blur_score"])
        cams.append(Camera(c2w[:3], float(fr.get("fl_x", meta["fl_x"])) * s, float(fr.get("fl_y", meta["fl_y"])) * s,
                           float(fr.get("cx", meta["cx"])) * s, float(fr.get("cy", meta["cy"])) * s, W, H, md))
        paths.append(os.path.normpath(os.path.join(root, fr["file_path"])))
    at = meta.get("applied_transform")
    return TransformsScene(
        cameras=cams, image_paths=paths, train_indices=tr, eval_indices=ev, exposure_time=exposure,
        rolling_shutter_time=readout,
        distortion={k: float(meta[k]) for k in ("k1", "k2", "k3", "p1", "p2") if k in meta},
        ply_file_path=(os.path.normpath(os.path.join(root, meta["ply_file_path"])) if "ply_file_path" in meta else None),
        applied_transform=(torch.tensor(at, dtype=torch.float32) if at is not None else None),
    )


def _undistort_grid(H: int, W: int, fx: float, fy: float, cx: float, cy: float, distortion: Dict[str, float], device):
    """source pixel INDEX coordinates (us, vs) [H,W] float64 that the undistorted pixel grid samples"""
    k1, k2, k3 = (float(distortion.get(k, 0.0)) for k in ("k1", "k2", "k3"))
    p1, p2 = float(distortion.get("p1", 0.0)), float(distortion.get("p2", 0.0))
    dt = torch.float64
    v, u = torch.meshgrid(torch.arange(H, device=device, dtype=dt), torch.arange(W, device=device, dtype=dt), indexing="ij")
    # pixel CENTRES at +0.5 (the convention of the rasterizer and of nerfstudio's cameras)
    x, y = (u + 0.5 - cx) / fx, (v + 0.5 - cy) / fy
    r2 = x * x + y * y
    radial = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * radial + 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
    yd = y * radial + p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
    return xd * fx + cx - 0.5, yd * fy + cy - 0.5


def _has_distortion(distortion: Dict[str, float]) -> bool:
    return any(float(distortion.get(k, 0.0)) != 0.0 for k in ("k1", "k2", "k3", "p1", "p2"))


def undistort_roi(H: int, W: int, fx: float, fy: float, cx: float, cy: float,
                  distortion: Dict[str, float]) -> Tuple[int, int, int, int]:
    """(x0, y0, x1, y1): the largest axis-aligned rectangle of the undistorted frame, grown out from the principal
    point, in which EVERY pixel samples inside the source image.  With pincushion (k1 > 0) or tangential terms the
    border of the undistorted frame looks outside the sensor; nerfstudio's datamanager crops to the valid region
    (cv2.getOptimalNewCameraMatrix(alpha=0) + its ROI) and adopts the new intrinsics, so no black, unmasked pixels reach
    the loss.  Here the focal lengths stay and the frame is cropped: cx, cy shift by (x0, y0)."""
    if not _has_distortion(distortion):
        return 0, 0, W, H
    us, vs = _undistort_grid(H, W, fx, fy, cx, cy, distortion, "cpu")
    ok = (us >= 0.0) & (us <= W - 1.0) & (vs >= 0.0) & (vs <= H - 1.0)      # the bilinear footprint lies inside
    x0, y0, x1, y1 = 0, 0, W, H
    # shrink the side whose border line has the largest share of invalid pixels until all four lines are clean
    while x1 - x0 > 2 and y1 - y0 > 2:
        sides = {"l": (~ok[y0:y1, x0]).float().mean(), "r": (~ok[y0:y1, x1 - 1]).float().mean(),
                 "t": (~ok[y0, x0:x1]).float().mean(), "b": (~ok[y1 - 1, x0:x1]).float().mean()}
        side, bad = max(sides.items(), key=lambda kv: float(kv[1]))
        if float(bad) == 0.0:
            break
        if side == "l":
            x0 += 1
        elif side == "r":
            x1 -= 1
        elif side == "t":
            y0 += 1
        else:
            y1 -= 1
    if not bool(ok[y0:y1, x0:x1].all()):
        raise ValueError("lens distortion leaves no valid rectangle around the principal point")
    return x0, y0, x1, y1


def undistort_image(img: torch.Tensor, fx: float, fy: float, cx: float, cy: float,
                    distortion: Dict[str, float], crop: bool = False):
    """[H,W,3] image taken through OpenCV's radial-tangential lens model (k1, k2, k3, p1, p2 — the fields
    /root/reference/process_synthetic_inputs.py:113-129 and combine.py:109-131 write) -> the image an ideal pinhole
    camera with the SAME focal lengths would see (bilinear).  nerfstudio's datamanager undistorts the training images
    once up front — the rasterizer only knows pinhole cameras.  All-zero coefficients return the input unchanged.
    crop=False: same size and principal point; pixels that look outside the source stay black (no mask!).
    crop=True: -> (image cropped to undistort_roi, (x0, y0, x1, y1)): every pixel valid; the caller shifts cx, cy by
    (x0, y0) and adopts the new size (load_scene_images does)."""
    H, W = img.shape[0], img.shape[1]
    if not _has_distortion(distortion):
        return (img, (0, 0, W, H)) if crop else img
    us, vs = _undistort_grid(H, W, fx, fy, cx, cy, distortion, img.device)
    grid = torch.stack([(us + 0.5) / W * 2.0 - 1.0, (vs + 0.5) / H * 2.0 - 1.0], dim=-1)[None].to(img.dtype)
    out = torch.nn.functional.grid_sample(img.permute(2, 0, 1)[None], grid, mode="bilinear", padding_mode="zeros",
                                          align_corners=False)
    out = out[0].permute(1, 2, 0).contiguous()
    if not crop:
        return out
    x0, y0, x1, y1 = undistort_roi(H, W, fx, fy, cx, cy, distortion)
    return out[y0:y1, x0:x1].contiguous(), (x0, y0, x1, y1)


def load_scene_images(scene: "TransformsScene", device="cpu", undistort: bool = True) -> List[torch.Tensor]:
    """every frame's image, undistorted with the scene's lens coefficients when any is non-zero (what nerfstudio's
    datamanager does before the first iteration).  An undistorted frame is CROPPED to the rectangle in which every pixel
    is valid and `scene.cameras[i]` is replaced by the camera of the cropped frame (cx, cy shifted, new width / height;
    same focal lengths): no black border reaches the loss.  undistort=False returns the frames as stored — cut to the
    cropped cameras' rectangle if an earlier load of this scene object already replaced its cameras."""
    out = []
    lens = undistort and _has_distortion(scene.distortion)
    for i, (cam, path) in enumerate(zip(scene.cameras, scene.image_paths)):
        img = load_image(path, device)
        if lens and not cam.metadata.get("undistorted", False):
            img, (x0, y0, x1, y1) = undistort_image(img, cam.fx, cam.fy, cam.cx, cam.cy, scene.distortion, crop=True)
            md = dict(cam.metadata, undistorted=True, undistort_roi=(x0, y0, x1, y1))
            scene.cameras[i] = Camera(cam.camera_to_world, cam.fx, cam.fy, cam.cx - x0, cam.cy - y0, x1 - x0, y1 - y0, md)
        elif lens:
            # the camera already describes the cropped frame (a second load of the same scene): same crop again
            x0, y0, x1, y1 = cam.metadata["undistort_roi"]
            full = undistort_image(img, cam.fx, cam.fy, cam.cx + x0, cam.cy + y0, scene.distortion)
            img = full[y0:y1, x0:x1].contiguous()
        elif cam.metadata.get("undistorted", False):
            # undistort=False on a scene whose cameras an earlier load already replaced by those of the cropped frames
            # (ADVICE round 4): the raw frame is cut to the same rectangle, so that image size and intrinsics agree —
            # the pixels stay distorted, which is what the caller asked for
            x0, y0, x1, y1 = cam.metadata["undistort_roi"]
            img = img[y0:y1, x0:x1].contiguous()
        out.append(img)
    return out


def load_image(path: str, device="cpu") -> torch.Tensor:
    """[H,W,3] float32 in 0..1.  `.npy` (float, already 0..1) or anything PIL decodes (8-bit PNG / JPEG)."""
    if path.endswith(".npy"):
        import numpy as np
        return torch.from_numpy(np.load(path).astype("float32"))[..., :3].to(device)
    import numpy as np
    from PIL import Image
    with Image.open(path) as im:
        arr = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return (torch.from_numpy(arr.copy()).to(device).float() / 255.0)


def save_image(path: str, img: torch.Tensor) -> None:
    """[H,W,3] float 0..1 -> 8-bit PNG (or .npy float32 when the name says so)"""
    import numpy as np
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    a = img.detach().clamp(0, 1).cpu()
    if path.endswith(".npy"):
        np.save(path, a.numpy().astype("float32"))
        return
    from PIL import Image
    Image.fromarray((a * 255.0 + 0.5).to(torch.uint8).numpy()).save(path)


def write_transforms(root: str, width: int, height: int, fx: float, fy: float, cx: float, cy: float,
                     exposure_time: float, rolling_shutter_time: float, frames: List[Dict],
                     ply_file_path: Optional[str] = None) -> str:
    """Write transforms.json with exactly the fields of /root/reference/process_synthetic_inputs.py:113-129 (top
    level) and :171-176 (per frame: camera_linear_velocity, camera_angular_velocity, file_path, transform_matrix)."""
    meta = {"aabb_scale": 16, "cx": float(cx), "cy": float(cy), "exposure_time": float(exposure_time),
            "fl_x": float(fx), "fl_y": float(fy), "frames": [], "h": int(height), "k1": 0, "k2": 0,
            "orientation_override": "none", "p1": 0, "p2": 0, "rolling_shutter_time": float(rolling_shutter_time),
            "w": int(width)}
    for fr in frames:
        meta["frames"].append({
            "camera_angular_velocity": [float(v) for v in fr["camera_angular_velocity"]],
            "camera_linear_velocity": [float(v) for v in fr["camera_linear_velocity"]],
            "file_path": fr["file_path"],
            "transform_matrix": [[float(v) for v in row] for row in fr["transform_matrix"]]})
    if ply_file_path is not None:
        meta["ply_file_path"] = ply_file_path
    os.makedirs(root, exist_ok=True)
    out = os.path.join(root, "transforms.json")
    with open(out, "wt") as f:
        json.dump(meta, f, indent=4)
    return out


def write_seed_points_ply(path: str, xyz: torch.Tensor, rgb: torch.Tensor) -> None:
    """ASCII PLY `x y z red green blue` like /root/reference/process_synthetic_inputs.py:203-219"""
    n = xyz.shape[0]
    lines = ["ply", "format ascii 1.0", f"element vertex {n}", "property float x", "property float y",
             "property float z", "property uint8 red", "property uint8 green", "property uint8 blue", "end_header"]
    c = (rgb.clamp(0, 1) * 255.0 + 0.5).to(torch.int64)
    for i in range(n):
        lines.append(f"{xyz[i, 0].item():.6f} {xyz[i, 1].item():.6f} {xyz[i, 2].item():.6f} "
                     f"{int(c[i, 0])} {int(c[i, 1])} {int(c[i, 2])}")
    with open(path, "wt") as f:
        f.write("\n".join(lines) + "\n")


def load_seed_points_ply(path: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """ASCII PLY seed cloud `x y z red green blue` (/root/reference/process_synthetic_inputs.py:203-219)
    -> (xyz [N,3] float32, rgb [N,3] float32 in 0..1)."""
    with open(path, "rt") as f:
        lines = f.read().splitlines()
    if not lines or lines[0].strip() != "ply":
        raise ValueError("not a PLY file")
    n, props, i = 0, [], 0
    for i, ln in enumerate(lines):
        t = ln.split()
        if t[:2] == ["element", "vertex"]:
            n = int(t[2])
        elif t[:1] == ["property"]:
            props.append(t[-1])
        elif t[:1] == ["format"] and t[1] != "ascii":
            raise ValueError("only ASCII PLY seed clouds are supported")
        elif t[:1] == ["end_header"]:
            break
    rows = [[float(v) for v in ln.split()] for ln in lines[i + 1:i + 1 + n]]
    data = torch.tensor(rows, dtype=torch.float32).reshape(n, len(props))
    col = {p: j for j, p in enumerate(props)}
    xyz = data[:, [col["x"], col["y"], col["z"]]]
    rgb = data[:, [col["red"], col["green"], col["blue"]]] / 255.0 if "red" in col else torch.full((n, 3), 0.5)
    return xyz, rgb


def synthetic_scene(n: int, width: int, height: int, sh_degree: int = 3, seed: int = 1234,
                    scale_mult: float = 1.0, profile: str = "survey") -> Dict:
    """The seeded benchmark scene of SURVEY.md §8d (the inputs bench.py measures on): camera at the origin,
    OpenCV axes, fx = fy = 0.8*W; Gaussians uniform in the frustum slab z in [1,10] reaching 1.2x the field
    of view, log-scales ~ N(log(0.004 * 5.5 * scale_mult), 0.6^2), random unit quaternions, opacity logits
    ~ N(0, 2^2), SH dc ~ N(0, 0.5^2), higher bands ~ N(0, 0.05^2); camera velocity 0.1*U[-1,1]^3 m/s and
    0.2*U[-1,1]^3 rad/s, exposure 1/60 s, readout 1/30 s.  Raw (pre-activation) float32 parameters on the CPU.
    The test oracle draws the same scene from the same generator sequence (tests/test_host_logic.py pins it).
    profile="trained" keeps every random draw but shapes the scene like a fitted model instead of SURVEY §8d's
    saturated one: world-space scale proportional to depth (0.006 * z: every Gaussian is ~7 px wide on screen,
    none fills the view) and mostly translucent opacities (logits ~ N(-3.3, 1.5^2)) — pixels need hundreds of
    list entries before they saturate, a large share of the Gaussians receives a gradient and the depth-sliced
    path needs several slices (bench.py reports it as config.secondary)."""
    import math
    g = torch.Generator().manual_seed(seed)
    fx = fy = 0.8 * width
    z = 1.0 + 9.0 * torch.rand(n, generator=g)
    u = (torch.rand(n, generator=g) * 2 - 1) * 1.2
    v = (torch.rand(n, generator=g) * 2 - 1) * 1.2
    means = torch.stack([u * z * (0.5 * width / fx), v * z * (0.5 * height / fy), z], -1)
    log_scales = math.log(0.004 * 5.5 * scale_mult) + 0.6 * torch.randn(n, 3, generator=g)
    quats = torch.randn(n, 4, generator=g)
    quats = quats / quats.norm(dim=-1, keepdim=True)
    opacity_logits = 2.0 * torch.randn(n, generator=g)
    if profile == "trained":
        log_scales = log_scales - math.log(0.004 * 5.5) + torch.log(0.006 * z)[:, None]
        opacity_logits = opacity_logits * 0.75 - 3.3
    elif profile != "survey":
        raise ValueError(f"unknown scene profile {profile!r}")
    K = (sh_degree + 1) ** 2
    sh = torch.cat([0.5 * torch.randn(n, 1, 3, generator=g), 0.05 * torch.randn(n, K - 1, 3, generator=g)], 1)
    lin_vel = 0.1 * (torch.rand(3, generator=g) * 2 - 1)
    ang_vel = 0.2 * (torch.rand(3, generator=g) * 2 - 1)
    return dict(means=means, log_scales=log_scales, quats=quats, opacity_logits=opacity_logits, sh=sh,
                viewmat=torch.eye(4), lin_vel=lin_vel, ang_vel=ang_vel, fx=fx, fy=fy, cx=width / 2.0,
                cy=height / 2.0, exposure_time=1.0 / 60.0, rolling_shutter_time=1.0 / 30.0)
