#!/usr/bin/env python
"""rocBLAS/hipBLASLt fp32 (torch.mm / F.linear) on the bench's GEMM shapes, as an external reference point for
wsi::gemm_f32_kernel (tools/gemm_bench.py).  Ungrouped: one call per shape with all 80000 rows."""
import torch
dev = torch.device("cuda:0")


def t(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


n = 80000
for name, K, N in [("adapt", 1024, 512), ("kqv", 512, 1536), ("a_lin", 512, 512)]:
    x = torch.randn(n, K, device=dev); w = torch.randn(N, K, device=dev) * 0.03; gy = torch.randn(n, N, device=dev)
    y = torch.empty(n, N, device=dev); gx = torch.empty(n, K, device=dev); gw = torch.empty(N, K, device=dev)
    fl = 2.0 * n * K * N
    for nm, fn in (("NT fwd", lambda: torch.mm(x, w.t(), out=y)), ("NN dX", lambda: torch.mm(gy, w, out=gx)), ("TN dW", lambda: torch.mm(gy.t(), x, out=gw))):
        ms = t(fn)
        print(f"rocBLAS {name:6s} {nm}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TFLOP/s")
