#!/usr/bin/env python
"""Error against float64 and rate of the three GEMM arithmetics (fp32 MFMA / bf16x6 / fp16x3) on the bench's projection
shapes and on operands chosen to stress the fp16 scaling (rows spread over 60 binades, outliers inside a row, gradient-like
tiny values).  GPU."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import ops, _native as N

dev = torch.device("cuda:0")
MODES = ("fp32", "bf16x6", "fp16x3")


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


def run(op, A, B, M, Nn, K):
    C = torch.empty(M, Nn, device=dev)
    g = dict(A=N.ptr(A), lda=A.stride(0), B=N.ptr(B), ldb=B.stride(0), C=N.ptr(C), ldc=Nn, M=M, N=Nn, K=K)
    ops._gemm(op, 0, [g], dev)
    return C


def operands(op, M, Nn, K, kind):
    torch.manual_seed(5)
    a = torch.randn(M, K, device=dev)
    b = torch.randn(Nn, K, device=dev)
    if kind == "rows60":          # every row of A / B on its own binade over 2^-30..2^30
        a = a * torch.exp2(torch.randint(-30, 31, (M, 1), device=dev).float())
        b = b * torch.exp2(torch.randint(-30, 31, (Nn, 1), device=dev).float())
    elif kind == "elem12":        # element-wise spread of 2^-6..2^6 (the bf16x6 test's operands)
        a = a * torch.exp2(torch.randint(-6, 7, (M, K), device=dev).float())
        b = b * torch.exp2(torch.randint(-6, 7, (Nn, K), device=dev).float())
    elif kind == "outlier":       # one element per row 2^20 above the rest
        a[torch.arange(M), torch.randint(0, K, (M,))] *= 2.0 ** 20
        b[torch.arange(Nn), torch.randint(0, K, (Nn,))] *= 2.0 ** 20
    elif kind == "tiny":          # gradient-like magnitudes
        a = a * 1e-9
        b = b * 3e-2
    # logical A [M,K], B [N,K]; lay them out for the op
    if op == N.WSI_GEMM_NT:
        return a.contiguous(), b.contiguous(), a, b
    if op == N.WSI_GEMM_NN:
        return a.contiguous(), b.t().contiguous(), a, b            # B stored [K,N]
    return a.t().contiguous(), b.t().contiguous(), a, b              # TN: A stored [K,M], B stored [K,N]


out = {"error": {}, "rate": {}}
for opname, op in (() if os.environ.get("RATE_ONLY") else (("NT", N.WSI_GEMM_NT), ("NN", N.WSI_GEMM_NN), ("TN", N.WSI_GEMM_TN))):
    for kind in ("normal", "elem12", "rows60", "outlier", "tiny"):
        M, Nn, K = 515, 389, 4100 if opname != "NT" else 4096
        As, Bs, a, b = operands(op, M, Nn, K, kind)
        ref = a.double() @ b.double().t()
        scale = a.abs().double() @ b.abs().double().t()
        row = {}
        for mode in MODES:
            ops.set_gemm_precision(mode)
            C = run(op, As, Bs, M, Nn, K)
            C2 = run(op, As, Bs, M, Nn, K)
            assert torch.equal(C, C2), (opname, kind, mode)
            err = (C.double() - ref)
            row[mode] = {"fro": float(err.norm() / ref.norm()), "max_over_sumabs": float((err.abs() / scale).max()),
                         "mean_signed_over_sumabs": float((err / scale).mean())}
        out["error"][f"{opname}/{kind}"] = row
        print(opname, kind, {m: (f"{row[m]['fro']:.2e}", f"{row[m]['max_over_sumabs']:.2e}") for m in MODES}, flush=True)

# bench shapes (80k nodes, hidden 512): K/Q/V projection NT, its dgrad NN, its wgrad TN
M = 80000
for opname, op, (m, n, k) in (("NT kqv", N.WSI_GEMM_NT, (M, 1536, 512)), ("NT out", N.WSI_GEMM_NT, (M, 512, 512)),
                              ("NN dgrad", N.WSI_GEMM_NN, (M, 512, 1536)), ("TN wgrad", N.WSI_GEMM_TN, (1536, 512, M)),
                              ("NT in", N.WSI_GEMM_NT, (M, 512, 1024)),
                              # HEATNet2 / configs[1] sizes (40k nodes, hidden 256): where does the scaled-fp16 kernel stop paying?
                              ("NT kqv h256", N.WSI_GEMM_NT, (40000, 768, 256)), ("NT out h256", N.WSI_GEMM_NT, (40000, 256, 256)),
                              ("NN dgrad h256", N.WSI_GEMM_NN, (40000, 256, 768)), ("NT in h256", N.WSI_GEMM_NT, (40000, 256, 1024)),
                              ("NT K=384", N.WSI_GEMM_NT, (80000, 384, 384)), ("NT K=128", N.WSI_GEMM_NT, (80000, 128, 128))):
    As, Bs, a, b = operands(op, m, n, k, "normal")
    fl = 2.0 * m * n * k
    row = {}
    for mode in MODES:
        ops.set_gemm_precision(mode)
        C = torch.empty(m, n, device=dev)
        g = dict(A=N.ptr(As), lda=As.stride(0), B=N.ptr(Bs), ldb=Bs.stride(0), C=N.ptr(C), ldc=n, M=m, N=n, K=k)
        t = timeit(lambda: ops._gemm(op, 0, [g], dev))
        row[mode] = {"ms": round(t, 4), "TF_fp32_equiv": round(fl / t / 1e9, 1)}
    out["rate"][opname] = row
    print(opname, (m, n, k), row, flush=True)
ops.set_gemm_precision("fp32")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/emu_probe.json", "w"), indent=1)
