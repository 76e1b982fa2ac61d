#!/usr/bin/env python
"""Where the scaled-fp16 kernel (with its absmax + pack pre-pass) overtakes bf16x6: one launch per shape, both arithmetics on the
same box, for the threshold of WSI_GEMM_AUTO (csrc/gemm_f32.hip::kernel_precision).  GPU.  usage: python tools/auto_threshold_probe.py [out.json]
(pre-pass included on both sides: the launch brings neither row scales nor packed weights - the worst case for the scaled kernel)"""
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


rows = []
for K, Nn in ((128, 128), (256, 256), (256, 768), (384, 1152), (512, 1536), (512, 512), (1024, 512)):
    for M in (2500, 5000, 10000, 20000, 40000, 80000):
        a = torch.randn(M, K, device=dev)
        b = torch.randn(Nn, K, device=dev)
        c = torch.empty(M, Nn, device=dev)
        g = dict(A=N.ptr(a), lda=K, B=N.ptr(b), ldb=K, C=N.ptr(c), ldc=Nn, M=M, N=Nn, K=K)
        t = {}
        for mode in ("bf16x6", "fp16x3"):
            ops.set_gemm_precision(mode)
            t[mode] = timeit(lambda: ops._gemm(N.WSI_GEMM_NT, 0, [g], dev))
        gf = 2.0 * M * Nn * K / 1e9
        rows.append(dict(M=M, N=Nn, K=K, gflop=round(gf, 2), bf16x6_us=round(t["bf16x6"] * 1e3, 1), fp16x3_us=round(t["fp16x3"] * 1e3, 1),
                         fp16x3_over_bf16x6=round(t["fp16x3"] / t["bf16x6"], 3)))
        print(rows[-1], flush=True)
ops.set_gemm_precision("fp32")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/auto_threshold.json", "w"), indent=1)
