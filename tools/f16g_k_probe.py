#!/usr/bin/env python
"""Where gemm_fp16x3g_kernel's time goes as a function of the contraction length: the K|Q|V-shaped launch (80000 rows, 1536 columns, three node-type groups)
at K = 256 ... 8192 - a tile's prologue and epilogue are a fixed cost, the main loop scales with K - under the 16-byte (LDS-transposed) epilogue and under the
element-wise one (WSI_F16G_EPI=g, measurement build).  TFLOP/s fp32-equivalent, pre-passes excluded (row scales and packed weights supplied).  GPU.
usage: python tools/f16g_k_probe.py [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import _native as N
N.use_measurement_library()
import __graft_entry__
__graft_entry__.build()
from wsi_hgnn_amd import ops
dev = torch.device("cuda:0")
ops.set_gemm_precision("fp16x3")
rows = [(0, 40000), (40000, 64000), (64000, 80000)]
n, Nout = 80000, 1536


def timeit(fn, iters=10):
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


res = {}
for K in (256, 512, 1024, 2048, 4096, 8192):
    x = torch.randn(n, K, device=dev)
    ws = [torch.randn(Nout, K, device=dev) * 0.03 for _ in rows]
    y = torch.empty(n, Nout, device=dev)
    bits = ops.row_absmax(x)
    def run():
        ops._gemm(N.WSI_GEMM_NT, 0, [dict(A=N.ptr(x, r0 * K * 4), lda=K, B=N.ptr(ws[t]), ldb=K, C=N.ptr(y, r0 * Nout * 4), ldc=Nout, M=r1 - r0, N=Nout, K=K,
                                          a_absmax=N.ptr(bits, r0 * 4), a_absmax_parts=1) for t, (r0, r1) in enumerate(rows)], dev)
    fl = 2.0 * n * K * Nout
    rec = {}
    for rnd in range(2):
        for epi in ("vec", "guarded"):
            if epi == "guarded":
                os.environ["WSI_F16G_EPI"] = "g"
            else:
                os.environ.pop("WSI_F16G_EPI", None)
            ms = timeit(run)
            rec.setdefault(epi, []).append(round(fl / ms / 1e9, 1))
    os.environ.pop("WSI_F16G_EPI", None)
    res[f"K={K}"] = rec
    print(K, rec, flush=True)
    del x, ws, y
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
