"""Device-side cost of the row-sparse gradient exchange on ONE GPU (everything but the collective).
First line: the sync-free form dp.allreduce_gradients(mode="sparse") uses (masked pack into a fixed-capacity payload with
a header, zero-fill, `world` count-from-header scatter-adds).  Second line: the index-list form (row mask, nonzero —
a host sync —, pack, zero-fill, scatter-adds) kept for comparison.
usage: python tools/dp_bench.py [N] [rows] [world]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import gsdeblur_amd as gs  # noqa: E402
from gsdeblur_amd.dp import _RowOps  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 5400
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda", 0)
shapes = [(N, 3), (N, 3), (N, 4), (N,), (N, 16, 3)]
g = torch.Generator().manual_seed(0)
touched = torch.zeros(N, dtype=torch.bool)
touched[torch.randperm(N, generator=g)[:rows]] = True
grads = [(torch.randn(s, generator=g) * touched.view(-1, *([1] * (len(s) - 1)))).to(dev) for s in shapes]
ops = _RowOps(grads)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6, out


cap = int(rows * 1.25) + 64
t_pm, pay_m = timed(lambda: ops.pack_masked(cap))
t_zero0, _ = timed(lambda: [x.zero_() for x in grads])
t_sc0, _ = timed(lambda: [ops.scatter_add_payload(pay_m, cap, 1.0 / world) for _ in range(world)])
print(f"dp exchange, device side, sync-free, N={N} rows={rows} cap={cap} world={world}: mask + masked pack {t_pm:.0f} us, "
      f"zero-fill {t_zero0:.0f} us, {world} payload scatter-adds {t_sc0:.0f} us, total {t_pm + t_zero0 + t_sc0:.0f} us; "
      f"payload per rank {pay_m.numel() * 4 / 1e6:.2f} MB")
grads = [(torch.randn(s, generator=g) * touched.view(-1, *([1] * (len(s) - 1)))).to(dev) for s in shapes]
ops = _RowOps(grads)
t_mask, mask = timed(ops.row_mask)
t_nz, idx = timed(lambda: mask.nonzero(as_tuple=False).reshape(-1))
t_pack, pay = timed(lambda: ops.pack(idx, idx.numel()))
t_zero, _ = timed(lambda: [x.zero_() for x in grads])
t_scatter, _ = timed(lambda: [ops.scatter_add(pay, idx.numel(), 1.0 / world) for _ in range(world)])
total = t_mask + t_nz + t_pack + t_zero + t_scatter
print(f"dp exchange, device side, index-list form, N={N} rows={idx.numel()} world={world}: row_mask {t_mask:.0f} us, nonzero {t_nz:.0f} us, "
      f"pack {t_pack:.0f} us, zero-fill {t_zero:.0f} us, {world} scatter-adds {t_scatter:.0f} us, total {total:.0f} us; "
      f"payload per rank {pay.numel() * 4 / 1e6:.2f} MB (dense bucket {sum(x.numel() for x in grads) * 4 / 1e6:.0f} MB)")
