#!/usr/bin/env python
"""Where the column-scaled fp16 weight-gradient kernel (gemm_tn16_kernel, its column-statistics pass inside the launch: the worst case) overtakes bf16x6:
one TN launch per shape, both arithmetics on the same box - for the weight-gradient threshold of WSI_GEMM_AUTO.  GPU.  usage: python tools/tn_threshold_probe.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import ops, _native as N
dev = torch.device("cuda:0")


def timeit(fn, iters=20):
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


rows = []
for M, Nn, ng in ((256, 256, 1), (256, 256, 6), (256, 1024, 3), (512, 512, 1), (512, 512, 3), (128, 128, 6)):
    for K in (2500, 5000, 10000, 20000, 40000, 80000):
        dY = torch.randn(K, M * ng, device=dev) * 1e-3
        X = torch.randn(K, Nn, device=dev)
        outs = [torch.empty(M, Nn, device=dev) for _ in range(ng)]
        groups = [dict(A=N.ptr(dY, j * M * 4), lda=M * ng, B=N.ptr(X), ldb=Nn, C=N.ptr(outs[j]), ldc=Nn, M=M, N=Nn, K=K) for j in range(ng)]
        t = {}
        for mode in ("bf16x6", "fp16x3"):
            ops.set_gemm_precision(mode)
            t[mode] = timeit(lambda: ops._gemm(N.WSI_GEMM_TN, 0, groups, dev))
        gf = 2.0 * M * Nn * K * ng / 1e9
        rows.append(dict(M=M, N=Nn, groups=ng, K=K, gflop=round(gf, 2), bf16x6_us=round(t["bf16x6"] * 1e3, 1), fp16x3_us=round(t["fp16x3"] * 1e3, 1),
                         fp16x3_over_bf16x6=round(t["fp16x3"] / t["bf16x6"], 3)))
        print(rows[-1], flush=True)
ops.set_gemm_precision("fp32")
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
