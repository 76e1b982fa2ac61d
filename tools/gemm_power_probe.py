#!/usr/bin/env python
"""Is the bf16x6 GEMM schedule-bound or power-bound?  Same kernels, same shapes, operands random vs all-zero: identical
instruction streams, only the data toggling (= matrix-core power) differs.  GPU."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import ops, _native as N

dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


M, K, Nn = 80000, 512, 1536
out = {}
for fill in ("random", "zeros"):
    A = torch.randn(M, K, device=dev) if fill == "random" else torch.zeros(M, K, device=dev)
    B = torch.randn(Nn, K, device=dev) if fill == "random" else torch.zeros(Nn, K, device=dev)
    C = torch.empty(M, Nn, device=dev)
    fl = 2.0 * M * K * Nn
    Ap, Bp = ops.split_planes(A), ops.split_planes(B)
    g = dict(Ap=N.ptr(Ap), ldap=Ap.stride(0), Bp=N.ptr(Bp), ldbp=Bp.stride(0), C=N.ptr(C), ldc=Nn, M=M, N=Nn, K=K)
    t = timeit(lambda: ops.gemm_p3(N.WSI_GEMM_NT, 0, [g], dev))
    res = {"p3_nt_TF": round(fl / t / 1e9, 1)}
    for mode in ("bf16x6", "fp32"):
        ops.set_gemm_precision(mode)
        gg = dict(A=N.ptr(A), lda=K, B=N.ptr(B), ldb=K, C=N.ptr(C), ldc=Nn, M=M, N=Nn, K=K)
        tm = timeit(lambda: ops._gemm(N.WSI_GEMM_NT, 0, [gg], dev))
        res[mode + "_nt_TF"] = round(fl / tm / 1e9, 1)
    out[fill] = res
    print(fill, res, flush=True)
print(json.dumps(out))
