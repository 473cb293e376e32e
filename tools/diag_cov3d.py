"""Diagnostic: where does cov3d differ between the GPU kernel, the host-compiled gs_math.h and the torch oracle?"""
import ctypes, subprocess, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import gs_oracle as O
import gsdeblur_amd as gs
hdr = ROOT / "3dgs-deblur_amd" / "csrc"
lib = "/tmp/libhost_math.so"
subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC", f"-I{hdr}",
                       str(ROOT / "tests/host_math/host_math.cpp"), "-o", lib])
hm = ctypes.CDLL(lib)
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
f = ctypes.c_float
W, H, n = 256, 256, 5000
sc = O.synthetic_scene(n, W, H, seed=7, scale_mult=4.0)
scales, quats = sc["log_scales"].exp(), sc["quats"] * 1.7
V = torch.eye(4)
pr = O.project_gaussians(sc["means"], scales, 1.0, quats, V, sc["fx"], sc["fy"], sc["cx"], sc["cy"], H, W)
m, s, q, Vn = (np.ascontiguousarray(t.numpy()) for t in (sc["means"], scales, quats, V))
xys = np.zeros((n, 2), np.float32); dep = np.zeros(n, np.float32); rad = np.zeros(n, np.int32)
con = np.zeros((n, 3), np.float32); comp = np.zeros(n, np.float32); nt = np.zeros(n, np.int32)
c3 = np.zeros((n, 6), np.float32); tb = np.zeros((n, 4), np.int32)
hm.hm_project(n, P(m), P(s), f(1.0), P(q), P(Vn), f(sc["fx"]), f(sc["fy"]), f(sc["cx"]), f(sc["cy"]), W, H, f(0.01),
              P(xys), P(dep), P(rad), P(con), P(comp), P(nt), P(c3), P(tb))
dev = torch.device("cuda:0")
out = gs.project_gaussians(sc["means"].to(dev), scales.to(dev), 1.0, quats.to(dev), V.to(dev), sc["fx"], sc["fy"],
                           sc["cx"], sc["cy"], H, W, 16)
g_c3 = out[6].cpu().numpy()
o_c3 = pr.cov3d.numpy()
def nbad(a, b): return int((a.view(np.int32) != b.view(np.int32)).sum())
print("cov3d mismatching words: host-vs-oracle", nbad(c3, o_c3), " gpu-vs-oracle", nbad(g_c3, o_c3), " gpu-vs-host", nbad(g_c3, c3), "of", c3.size)
bad = np.argwhere(g_c3.view(np.int32) != c3.view(np.int32))
if len(bad):
    i = bad[0][0]
    print("first bad gaussian", i, "quat", q[i], "scale", s[i])
    print(" gpu ", g_c3[i]); print(" host", c3[i]); print(" orcl", o_c3[i])
    # recompute by hand in numpy float32
    qq = q[i].astype(np.float32)
    n2 = np.float32(np.float32(np.float32(qq[0]*qq[0] + qq[1]*qq[1]) + qq[2]*qq[2]) + qq[3]*qq[3])
    inv = np.float32(1.0) / np.sqrt(n2)
    print(" n2", n2, "inv", inv, inv.view(np.int32))
