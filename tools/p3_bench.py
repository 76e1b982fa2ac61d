#!/usr/bin/env python
"""Kernel-level rates of the pre-split GEMM (csrc/gemm_p3.hip) on the bench shapes, beside the in-kernel-split bf16x6 kernel
and the exact fp32 kernel: python tools/p3_bench.py  (GPU)."""
import json
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import ops, _native as N

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


out = {}
Mrows = 80000
for name, K, Nn in (("adapt", 1024, 512), ("kqv", 512, 1536), ("a_lin", 512, 512)):
    A = torch.randn(Mrows, K, device=dev)
    B = torch.randn(Nn, K, device=dev)
    C = torch.empty(Mrows, Nn, device=dev)
    Ap, Bp = ops.split_planes(A), ops.split_planes(B)
    Cp = ops.empty_planes(Mrows, Nn, dev)
    fl = 2.0 * Mrows * K * Nn
    g = dict(Ap=N.ptr(Ap), ldap=Ap.stride(0), Bp=N.ptr(Bp), ldbp=Bp.stride(0), C=N.ptr(C), ldc=Nn, M=Mrows, N=Nn, K=K)
    t = timeit(lambda: ops.gemm_p3(N.WSI_GEMM_NT, 0, [g], dev))
    g2 = dict(g, Cp=N.ptr(Cp), ldcp=Cp.stride(0))
    t2 = timeit(lambda: ops.gemm_p3(N.WSI_GEMM_NT, 0, [g2], dev))
    ts = timeit(lambda: ops.split_planes(A, out=Ap))
    res = {"p3_nt_ms": t, "p3_nt_TF": fl / t / 1e9, "p3_nt_with_planes_out_ms": t2, "split_A_ms": ts,
           "split_GBps": (A.numel() * 10) / ts / 1e6}
    for mode in ("fp32", "bf16x6"):
        ops.set_gemm_precision(mode)
        gg = dict(A=N.ptr(A), lda=K, B=N.ptr(B), ldb=K, C=N.ptr(C), ldc=Nn, M=Mrows, N=Nn, K=K)
        tm = timeit(lambda: ops._gemm(N.WSI_GEMM_NT, 0, [gg], dev))
        res[mode + "_nt_ms"] = tm
        res[mode + "_nt_TF"] = fl / tm / 1e9
    # TN: dW [Nn, K] = dY[Mrows, Nn]^T X[Mrows, K]
    dY = torch.randn(Mrows, Nn, device=dev)
    dYp = ops.split_planes(dY)
    W = torch.empty(Nn, K, device=dev)
    cs = torch.empty(Nn, device=dev)
    gt = dict(Ap=N.ptr(dYp), ldap=dYp.stride(0), Bp=N.ptr(Ap), ldbp=Ap.stride(0), C=N.ptr(W), ldc=K, colsum_out=N.ptr(cs), M=Nn, N=K, K=Mrows)
    tt = timeit(lambda: ops.gemm_p3(N.WSI_GEMM_TN, 0, [gt], dev))
    res["p3_tn_ms"] = tt
    res["p3_tn_TF"] = fl / tt / 1e9
    for mode in ("fp32", "bf16x6"):
        ops.set_gemm_precision(mode)
        gg = dict(A=N.ptr(dY), lda=Nn, B=N.ptr(A), ldb=K, C=N.ptr(W), ldc=K, colsum_out=N.ptr(cs), M=Nn, N=K, K=Mrows)
        tm = timeit(lambda: ops._gemm(N.WSI_GEMM_TN, 0, [gg], dev))
        res[mode + "_tn_ms"] = tm
        res[mode + "_tn_TF"] = fl / tm / 1e9
    ops.set_gemm_precision("fp32")
    out[name] = {k: round(v, 3) for k, v in res.items()}
    print(name, out[name], flush=True)
print(json.dumps(out))
