#!/bin/bash
set -u
OUT=gpurun_out/r3_run4
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_training.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest_training.log 2>&1
tail -25 $OUT/pytest_training.log
timeout 300 python - > $OUT/train_timing.log 2>&1 <<'PY'
import torch, time, sys
sys.path.insert(0, '.')
import gsdeblur_amd as gs
dev = torch.device('cuda:0')
H, W = 1080, 1920
pred = torch.rand(H, W, 3, device=dev).requires_grad_(True); gt = torch.rand(H, W, 3, device=dev)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def fused():
    pred.grad = None; gs.fused.image_loss(pred, gt, 0.2).backward()
def torch_loss():
    pred.grad = None; gs.training.image_loss_torch(pred, gt, 0.2).backward()
print('image loss fwd+bwd 1080p: HIP %.3f ms, torch %.3f ms' % (t(fused), t(torch_loss)))
N = 1_000_000
shapes = [(N, 3), (N, 3), (N, 4), (N, 1), (N, 3), (N, 15, 3)]
pa = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
pb = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
for p in pa + pb: p.grad = torch.randn_like(p)
oa = [gs.fused.HipAdam([p], lr=1e-3, eps=1e-15) for p in pa]
ob = [torch.optim.Adam([p], lr=1e-3, eps=1e-15) for p in pb]
print('Adam step, 1M Gaussians x 59 floats: HIP %.3f ms, torch %.3f ms' % (t(lambda: gs.fused.adam_step_all(oa)), t(lambda: [o.step() for o in ob])))
PY
cat $OUT/train_timing.log
