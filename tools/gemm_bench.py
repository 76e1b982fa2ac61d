#!/usr/bin/env python
"""Micro-benchmark of the grouped fp32 MFMA GEMM on the bench shapes (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from wsi_hgnn_amd import ops, _native as N

dev = torch.device("cuda:0")
rows = [(0, 40000), (40000, 64000), (64000, 80000)]
n = 80000


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def case(name, K, Nout, nproj):
    x = torch.randn(n, K, device=dev)
    ws = [torch.randn(Nout, K, device=dev) * 0.03 for _ in range(3 * nproj)]
    bs = [torch.randn(Nout, device=dev) for _ in range(3 * nproj)]
    srows, cols = [], []
    for r in rows:
        for j in range(nproj):
            srows.append(r); cols.append(j * Nout)
    spec = ops.LinearSpec(srows, cols, nproj * Nout, n)
    y = torch.empty(n, nproj * Nout, device=dev)
    gy = torch.randn(n, nproj * Nout, device=dev)
    gx = torch.empty(n, K, device=dev)
    flops = 2.0 * n * K * Nout * nproj

    def fwd():
        groups = []
        for i, w in enumerate(ws):
            r0, r1 = srows[i]
            groups.append(dict(A=N.ptr(x, r0 * K * 4), lda=K, B=N.ptr(w), ldb=K, C=N.ptr(y, (r0 * spec.out_cols + cols[i]) * 4),
                               ldc=spec.out_cols, bias=N.ptr(bs[i]), M=r1 - r0, N=Nout, K=K))
        ops._gemm(N.WSI_GEMM_NT, N.WSI_EPI_BIAS, groups, dev)

    def dx():
        if nproj == 1:
            groups = []
            for t, (r0, r1) in enumerate(rows):
                groups.append(dict(A=N.ptr(gy, r0 * spec.out_cols * 4), lda=spec.out_cols, B=N.ptr(ws[t]), ldb=K,
                                   C=N.ptr(gx, r0 * K * 4), ldc=K, M=r1 - r0, N=K, K=Nout))
            ops._gemm(N.WSI_GEMM_NN, 0, groups, dev)
            return
        # what the fused layer launches: ONE NN GEMM whose reduction runs over the three weight matrices (chunked B) with the
        # (1-s)*g_out residual added in the epilogue
        groups = []
        for t, (r0, r1) in enumerate(rows):
            w3 = ws[t * nproj:(t + 1) * nproj]
            groups.append(dict(A=N.ptr(gy, r0 * spec.out_cols * 4), lda=spec.out_cols, B=N.ptr(w3[0]), B1=N.ptr(w3[1]), B2=N.ptr(w3[2]),
                               b_chunk=Nout, ldb=K, C=N.ptr(gx, r0 * K * 4), ldc=K, R=N.ptr(x, r0 * K * 4), ldr=K,
                               M=r1 - r0, N=K, K=nproj * Nout))
        ops._gemm(N.WSI_GEMM_NN, N.WSI_EPI_ADD_R, groups, dev)

    gws = [torch.empty_like(w) for w in ws]

    def dw():
        groups = []
        for i, w in enumerate(ws):
            r0, r1 = srows[i]
            groups.append(dict(A=N.ptr(gy, (r0 * spec.out_cols + cols[i]) * 4), lda=spec.out_cols, B=N.ptr(x, r0 * K * 4), ldb=K,
                               C=N.ptr(gws[i]), ldc=K, M=Nout, N=K, K=r1 - r0))
        ops._gemm(N.WSI_GEMM_TN, 0, groups, dev)

    for nm, fn in (("NT fwd", fwd), ("NN dX", dx), ("TN dW", dw)):
        ms = timeit(fn)
        print(f"{name:8s} {nm:7s} K={K:5d} N={Nout * nproj:5d}: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)


M = 4096
a = torch.randn(M, M, device=dev); b = torch.randn(M, M, device=dev); c = torch.empty(M, M, device=dev)
def sq():
    ops._gemm(N.WSI_GEMM_NT, 0, [dict(A=N.ptr(a), lda=M, B=N.ptr(b), ldb=M, C=N.ptr(c), ldc=M, M=M, N=M, K=M)], dev)

for mode in ("fp32", "bf16x6", "fp32", "bf16x6"):
    ops.set_gemm_precision(mode)
    print(f"--- precision {mode}")
    case("adapt", 1024, 512, 1)
    case("kqv", 512, 512, 3)
    case("a_lin", 512, 512, 1)
    ms = timeit(sq)
    print(f"square 4096^3 NT: {ms:.3f} ms {2.0 * M ** 3 / ms / 1e9:.1f} TFLOP/s")
ops.set_gemm_precision("fp32")
ms = timeit(lambda: torch.mm(a, b.t(), out=c))
print(f"torch.mm (rocBLAS/hipBLASLt) 4096^3: {ms:.3f} ms {2.0 * M ** 3 / ms / 1e9:.1f} TFLOP/s")
