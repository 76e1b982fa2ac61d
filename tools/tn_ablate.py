#!/usr/bin/env python
"""Where the weight-gradient kernel's time goes (gemm_bf16x6_kernel<TN>): its measurement variants in the -DWSI_ABLATE build
(WSI_BF16_ABL: 1 no LDS writes, 2 no split arithmetic, 3 no fragment reads after the first stage; garbage results) on the bench's four dW shapes,
interleaved in one process.  GPU.  usage: python tools/tn_ablate.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import _native as N
N.use_measurement_library()
from wsi_hgnn_amd import build
build.build_native(ablate=True)
from wsi_hgnn_amd import ops
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16x6")
rows = [26667, 26667, 26666]
shapes = {"in_proj dW (512 x 1024)": (512, 1024, 1), "K|Q|V dW (3 x 512 x 512)": (512, 512, 3), "K|Q dW": (512, 512, 2), "a_linear dW": (512, 512, 1)}
res = {}
for name, (M, Nn, nproj) in shapes.items():
    n = sum(rows)
    dY = torch.randn(n, M * nproj, device=dev)
    X = torch.randn(n, Nn, device=dev)
    outs = [torch.empty(M, Nn, device=dev) for _ in range(3 * nproj)]
    groups, r0 = [], 0
    for i, r in enumerate(rows):
        for j in range(nproj):
            groups.append(dict(A=N.ptr(dY, (r0 * M * nproj + j * M) * 4), lda=M * nproj, B=N.ptr(X, r0 * Nn * 4), ldb=Nn, C=N.ptr(outs[i * nproj + j]), ldc=Nn, M=M, N=Nn, K=r))
        r0 += r
    flops = sum(2.0 * g["M"] * g["N"] * g["K"] for g in groups)
    rec = {}
    for rep in range(3):
        for abl in ("0", "1", "2", "3"):
            os.environ["WSI_BF16_ABL"] = abl
            for _ in range(2):
                ops._gemm(N.WSI_GEMM_TN, 0, groups, dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops._gemm(N.WSI_GEMM_TN, 0, groups, dev)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            rec.setdefault(abl, []).append(ms)
    res[name] = {k: {"ms": round(min(v), 4), "tf_eq": round(flops / min(v) / 1e9, 1)} for k, v in rec.items()}
    print(name, res[name], flush=True)
os.environ["WSI_BF16_ABL"] = "0"
print(json.dumps(res))
