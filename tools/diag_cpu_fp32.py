"""Diagnostic: are torch / numpy float32 elementwise ops on THIS cpu correctly rounded (== fp64 op rounded to fp32)?"""
import numpy as np, torch
print(torch.__version__, torch.backends.cpu.get_cpu_capability(), "threads", torch.get_num_threads())
g = torch.Generator().manual_seed(0)
for n in (7, 64, 100003):
    a = (torch.rand(n, generator=g) * 4 + 0.01)
    b = (torch.randn(n, generator=g))
    c = (torch.randn(n, generator=g))
    a64, b64, c64 = a.double(), b.double(), c.double()
    checks = {
        "sqrt": (torch.sqrt(a), torch.sqrt(a64).float(), np.sqrt(a.numpy())),
        "recip": (1.0 / a, (1.0 / a64).float(), np.float32(1.0) / a.numpy()),
        "div": (b / a, (b64 / a64).float(), b.numpy() / a.numpy()),
        "mul": (b * c, (b64 * c64).float(), b.numpy() * c.numpy()),
        "add": (b + c, (b64 + c64).float(), b.numpy() + c.numpy()),
        "rsqrt_expr": (1.0 / torch.sqrt(a), (1.0 / torch.sqrt(a64).float().double()).float(), np.float32(1.0) / np.sqrt(a.numpy())),
        "exp": (torch.exp(b), torch.exp(b64).float(), np.exp(b.numpy())),
    }
    for k, (t32, ref, np32) in checks.items():
        bt = int((t32.view(torch.int32) != ref.view(torch.int32)).sum())
        bn = int((torch.from_numpy(np32).view(torch.int32) != ref.view(torch.int32)).sum())
        print(f"n={n:7d} {k:10s} torch-mismatch {bt:6d}  numpy-mismatch {bn:6d}")
