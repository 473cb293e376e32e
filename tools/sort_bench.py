"""Micro-benchmark of the radix sort passes (one configuration per process, so that a `rocprofv3 --kernel-trace` of it
shows that configuration's launches last).  usage: python tools/sort_bench.py <config> [reps]
  tile      5.2M (tile key, emission index) pairs, 16-bit keys, host-side count
  tile_dev  the same through the device-side count (capacity 20.9M)
  depth     5 x 1M 32-bit depth keys, 27% visible, full segmented sort + count gather + scan
  depthc    the same through the compacting sort
"""
import sys
import time

import torch

sys.path.insert(0, ".")
import gsdeblur_amd as gs                      # noqa: E402
from gsdeblur_amd import ops                   # noqa: E402
sys.path.insert(0, "tests")
import python_frame_path                       # noqa: E402  (the Python depth pre-sort wrapper lives with the test twin)
python_frame_path.install()

cfg = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(1)


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


if cfg.startswith("tile"):
    n, T = 5_240_000, 5 * 8160
    # emission order: Gaussians in depth order, each writing a small box of tiles
    keys = torch.randint(0, T, (n,), generator=g, dtype=torch.int32).to(dev)
    vals = torch.randint(0, 5_000_000, (n + 8,), generator=g, dtype=torch.int32).to(dev)
    if cfg == "tile_dev":
        cap = 20_900_000
        kbuf = torch.zeros(cap, dtype=torch.int32, device=dev)
        kbuf[:n] = keys
        n_dev = torch.tensor([n], dtype=torch.int32, device=dev)
        fn = lambda: ops.radix_sort_pairs(kbuf.clone(), None, 0, 16, gather_src=vals, n_dev=n_dev)   # noqa: E731
    else:
        fn = lambda: ops.radix_sort_pairs(keys.clone(), None, 0, 16, gather_src=vals)                # noqa: E731
else:
    P, N = 5, 1_000_000
    depth = (torch.rand(P * N, generator=g) * 19 + 1).float()
    keys = depth.view(torch.int32).clone()
    culled = torch.rand(P * N, generator=g) > 0.27
    keys[culled] = -1                      # 0xFFFFFFFF
    nt = torch.randint(1, 40, (P * N,), generator=g, dtype=torch.int32)
    nt[culled] = 0
    keys, nt = keys.to(dev), nt.to(dev)
    rec = torch.zeros(1, device=dev)
    ops.DEPTH_SORT_COMPACT = 1 if cfg == "depthc" else 0
    fn = lambda: python_frame_path._depth_rank(rec, keys.clone(), nt, P, N)                                        # noqa: E731
print(cfg, f"{timed(fn):.4f} ms per call (includes one clone of the keys)")
