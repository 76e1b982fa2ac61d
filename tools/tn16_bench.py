#!/usr/bin/env python
"""The weight-gradient launches of one bench step (8 x 10k nodes: 40000 / 24000 / 16000 rows per node type) under the six-product bf16 kernel and
under the column-scaled three-product fp16 kernel of csrc/gemm_tn16.hip in both tile forms (WSI_TN16_CFG, measurement build), interleaved in one
process; whole-call times (column-maxima pre-pass and split-K second stage included) and the error against float64.  GPU.
usage: python tools/tn16_bench.py [--json out.json]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import _native as N
N.use_measurement_library()
from wsi_hgnn_amd import build
build.build_native(ablate=True)
from wsi_hgnn_amd import ops
dev = torch.device("cuda:0")
rows = [40000, 24000, 16000]
shapes = {"in_proj dW (512 x 1024)": (512, 1024, 1), "K|Q|V dW (3 x 512 x 512)": (512, 512, 3), "K|Q dW (2 x 512 x 512)": (512, 512, 2), "a_linear dW (512 x 512)": (512, 512, 1)}
variants = [("bf16x6", None), ("fp16x3", "256"), ("fp16x3", "128"), ("fp32", None)]
res = {}
torch.manual_seed(0)
for name, (M, Nn, nproj) in shapes.items():
    n = sum(rows)
    dY = torch.randn(n, M * nproj, device=dev) * 1e-3
    X = torch.randn(n, Nn, device=dev)
    outs = [torch.empty(M, Nn, device=dev) for _ in range(3 * nproj)]
    bias = [torch.empty(M, device=dev) for _ in range(3 * nproj)]
    groups, r0 = [], 0
    for i, r in enumerate(rows):
        for j in range(nproj):
            groups.append(dict(A=N.ptr(dY, (r0 * M * nproj + j * M) * 4), lda=M * nproj, B=N.ptr(X, r0 * Nn * 4), ldb=Nn, C=N.ptr(outs[i * nproj + j]), ldc=Nn,
                               colsum_out=N.ptr(bias[i * nproj + j]), M=M, N=Nn, K=r))
        r0 += r
    flops = sum(2.0 * g["M"] * g["N"] * g["K"] for g in groups)
    ref = (dY[:rows[0], :M].double().t() @ X[:rows[0]].double())
    sc = (dY[:rows[0], :M].double().abs().t() @ X[:rows[0]].double().abs())
    refb = dY[:rows[0], :M].double().sum(0)
    rec, err = {}, {}
    for rep in range(3):
        for mode, cfg in variants:
            ops.set_gemm_precision(mode)
            os.environ["WSI_TN16_CFG"] = cfg or "128"
            key = mode + (("/" + cfg) if cfg else "")
            for _ in range(2):
                ops._gemm(N.WSI_GEMM_TN, 0, groups, dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops._gemm(N.WSI_GEMM_TN, 0, groups, dev)
            e1.record(); torch.cuda.synchronize()
            rec.setdefault(key, []).append(e0.elapsed_time(e1) / 10)
            if rep == 0:
                err[key] = (((outs[0].double() - ref).abs() / sc).max().item(), ((bias[0].double() - refb).abs().max() / refb.abs().max()).item())
    res[name] = {k: {"ms": round(min(v), 4), "tf_eq": round(flops / min(v) / 1e9, 1), "err_vs_sum_abs": float(f"{err[k][0]:.3e}"), "bias_err": float(f"{err[k][1]:.2e}")} for k, v in rec.items()}
    print(name, json.dumps(res[name]), flush=True)
ops.set_gemm_precision("fp32")
if "--json" in sys.argv:
    json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
print("total ms per step:", {k: round(sum(res[s][k]["ms"] for s in res), 3) for k in next(iter(res.values()))})
