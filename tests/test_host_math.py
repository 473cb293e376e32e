"""The product's device math header (3dgs-deblur_amd/csrc/gs_math.h) compiled for the host with
g++ and compared with the oracle: integer outputs bit-exact against the float32 restatement,
hand-derived backward passes against float64 autograd.  CPU-only; the HIP kernels inline the very
same functions."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "host_math" / "host_math.cpp"
LIB = ROOT / "tests" / "host_math" / "libhost_math.so"
f = ctypes.c_float


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


@pytest.fixture(scope="module")
def hm():
    hdr = ROOT / "3dgs-deblur_amd" / "csrc" / "gs_math.h"
    if not LIB.exists() or LIB.stat().st_mtime < max(SRC.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC",
                               f"-I{hdr.parent}", str(SRC), "-o", str(LIB)])
    return ctypes.CDLL(str(LIB))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.fixture(scope="module")
def scene(oracle):
    O = oracle
    W, H = 256, 192
    sc = O.synthetic_scene(20000, W, H, seed=7, scale_mult=4.0)
    means = sc["means"].clone()
    means[:50, 2] = -1.0          # behind the camera
    means[50:100, 0] *= 5.0       # far outside the 1.3x fov clamp
    scales = sc["log_scales"].exp()
    quats = sc["quats"] * 1.7     # un-normalised on purpose
    V = O.subpose_viewmats(torch.eye(4, dtype=torch.float64), torch.tensor([0.1, 0.05, -0.2], dtype=torch.float64),
                           torch.tensor([0.05, -0.08, 0.03], dtype=torch.float64), [1.0])[0].float()
    return dict(sc=sc, W=W, H=H, means=means, scales=scales, quats=quats, V=V)


def test_projection_integers_bit_exact(hm, oracle, scene):
    O, sc, W, H = oracle, scene["sc"], scene["W"], scene["H"]
    pr = O.project_gaussians(scene["means"], scene["scales"], 1.0, scene["quats"], scene["V"], sc["fx"], sc["fy"],
                             sc["cx"], sc["cy"], H, W)
    n = scene["means"].shape[0]
    m, s, q, Vn = (np.ascontiguousarray(scene[k].numpy()) for k in ("means", "scales", "quats", "V"))
    xys = np.zeros((n, 2), np.float32); dep = np.zeros(n, np.float32); rad = np.zeros(n, np.int32)
    con = np.zeros((n, 3), np.float32); comp = np.zeros(n, np.float32); nt = np.zeros(n, np.int32)
    c3 = np.zeros((n, 6), np.float32); tb = np.zeros((n, 4), np.int32)
    hm.hm_project(n, P(m), P(s), f(1.0), P(q), P(Vn), f(sc["fx"]), f(sc["fy"]), f(sc["cx"]), f(sc["cy"]), W, H,
                  f(0.01), P(xys), P(dep), P(rad), P(con), P(comp), P(nt), P(c3), P(tb))
    assert (rad > 0).sum() > 10000 and (rad == 0).sum() > 100
    assert (rad == pr.radii.numpy()).all()
    assert (nt == pr.num_tiles_hit.numpy()).all()
    assert (tb[:, :2] == pr.tile_min.numpy()).all() and (tb[:, 2:] == pr.tile_max.numpy()).all()
    assert (dep.view(np.int32) == pr.depths.numpy().view(np.int32)).all()      # sort-key bits
    ok = rad > 0
    assert (xys[ok].view(np.int32) == pr.xys.numpy()[ok].view(np.int32)).all()
    assert (con[ok].view(np.int32) == pr.conics.numpy()[ok].view(np.int32)).all()
    assert np.abs(comp[ok] - pr.compensation.numpy()[ok]).max() < 1e-6


def test_projection_backward_vs_autograd(hm, oracle, scene):
    O, sc, W, H = oracle, scene["sc"], scene["W"], scene["H"]
    n = scene["means"].shape[0]
    md = scene["means"].double().requires_grad_(True)
    sd = scene["scales"].double().requires_grad_(True)
    qd = scene["quats"].double().requires_grad_(True)
    Vd = scene["V"].double().requires_grad_(True)
    prd = O.project_gaussians(md, sd, 1.0, qd, Vd, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W)
    g = torch.Generator().manual_seed(1)
    vx, vd, vc, vcomp = (torch.randn(*shp, generator=g) for shp in ((n, 2), (n,), (n, 3), (n,)))
    loss = (prd.xys * vx.double()).sum() + (prd.depths * vd.double() * (prd.radii > 0)).sum() + \
        (prd.conics * vc.double()).sum() + (prd.compensation * vcomp.double()).sum()
    loss.backward()
    m, s, q, Vn = (np.ascontiguousarray(scene[k].numpy()) for k in ("means", "scales", "quats", "V"))
    vm = np.zeros((n, 3), np.float32); vs = np.zeros((n, 3), np.float32); vq = np.zeros((n, 4), np.float32)
    vV = np.zeros(12, np.float32)
    hm.hm_project_bwd(n, P(m), P(s), f(1.0), P(q), P(Vn), f(sc["fx"]), f(sc["fy"]), f(sc["cx"]), f(sc["cy"]), W, H,
                      f(0.01), P(np.ascontiguousarray(vx.numpy())), P(np.ascontiguousarray(vd.numpy())),
                      P(np.ascontiguousarray(vc.numpy())), P(np.ascontiguousarray(vcomp.numpy())), P(vm), P(vs),
                      P(vq), P(vV))
    # float32 evaluation of the analytic backward vs float64 autograd: tolerance 2e-5 of the max magnitude
    assert rel(vm, md.grad.numpy()) < 2e-5
    assert rel(vs, sd.grad.numpy()) < 2e-5
    assert rel(vq, qd.grad.numpy()) < 2e-5
    assert rel(vV, Vd.grad.numpy()[:3].reshape(-1)) < 2e-5


def test_sh_basis_two_formulations(hm, oracle):
    d = torch.randn(1000, 3, generator=torch.Generator().manual_seed(0))
    d = d / d.norm(dim=-1, keepdim=True)
    for deg in range(5):
        B = np.zeros((1000, (deg + 1) ** 2), np.float32)
        hm.hm_sh_basis(1000, deg, P(np.ascontiguousarray(d.numpy())), P(B))
        assert np.abs(B - oracle.sh_basis(deg, d.double()).numpy()).max() < 1e-6


@pytest.mark.parametrize("zero_ang", [False, True])
def test_se3_closed_form_vs_matrix_exp(hm, oracle, scene, zero_ang):
    O = oracle
    V = scene["V"]
    lin = torch.tensor([0.1, 0.05, -0.2])
    ang = torch.zeros(3) if zero_ang else torch.tensor([0.05, -0.08, 0.03])
    times = np.array([-0.01, 0.0, 0.02, 0.5], np.float32)
    V0 = np.ascontiguousarray(V.numpy()); out = np.zeros((4, 16), np.float32)
    linn, angn = np.ascontiguousarray(lin.numpy()), np.ascontiguousarray(ang.numpy())
    hm.hm_subpose_viewmats(4, P(V0), P(linn), P(angn), P(times), P(out))
    Vd, ld, ad = (t.double().requires_grad_(True) for t in (V, lin, ang))
    ref = O.subpose_viewmats(Vd, ld, ad, times.tolist())
    assert np.abs(out.reshape(4, 4, 4) - ref.detach().numpy()).max() < 1e-6
    go = torch.randn(4, 4, 4, generator=torch.Generator().manual_seed(3)); go[:, 3, :] = 0
    (ref * go.double()).sum().backward()
    vV0 = np.zeros(16, np.float32); vl = np.zeros(3, np.float32); va = np.zeros(3, np.float32)
    hm.hm_subpose_viewmats_bwd(4, P(V0), P(linn), P(angn), P(times), P(np.ascontiguousarray(go.numpy().reshape(4, 16))),
                               P(vV0), P(vl), P(va))
    assert rel(vV0[:12], Vd.grad.numpy()[:3].reshape(-1)) < 1e-5
    assert rel(vl, ld.grad.numpy()) < 1e-5
    assert rel(va, ad.grad.numpy()) < 1e-5
