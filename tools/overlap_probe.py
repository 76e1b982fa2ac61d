#!/usr/bin/env python
"""Can the matrix-bound weight-gradient GEMMs run UNDER the fabric-bound attention backward?  Times, on the bench shapes,
(a) dW GEMM (TN, kqv shape) alone, (b) attention backward alone, (c) both back to back on one stream, (d) both on two streams.  GPU."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import _native as N
N.use_measurement_library()        # WSI_GEMM_LDS_PAD (residency cap of the GEMM workgroups) exists only in the -DWSI_ABLATE build
import wsi_hgnn_amd as W
from wsi_hgnn_amd import ops, synthetic

dev = torch.device("cuda:0")
ops.set_gemm_precision(os.environ.get("MODE", "bf16x6"))
G = W.batch([synthetic.hetero_graph(10000, 8, seed=611 + i) for i in range(8)]).to(dev)
plan, sim = G.plan(), G.cat_edata_csr("sim")
n, D, H = plan.num_nodes, 512, 4
kqv = torch.randn(n, 3 * D, device=dev, requires_grad=True)
ew = torch.tensor([[0.7]], device=dev, requires_grad=True)
eb = torch.tensor([0.3], device=dev, requires_grad=True)
t = ops.heat_attention(kqv, ew, eb, plan, sim, D, H)
gt = torch.randn_like(t)
# dW GEMM operands (kqv shape: dY [n, 1536], X [n, 512])
dY = torch.randn(n, 3 * D, device=dev)
X = torch.randn(n, D, device=dev)
gw = torch.empty(3 * D, D, device=dev)
grp = [dict(A=N.ptr(dY), lda=3 * D, B=N.ptr(X), ldb=D, C=N.ptr(gw), ldc=D, M=3 * D, N=D, K=n)]


def gemm():
    ops._gemm(N.WSI_GEMM_TN, 0, grp, dev)


def attn():
    torch.autograd.grad(t, (kqv, ew, eb), gt, retain_graph=True)


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


side = torch.cuda.Stream(priority=int(os.environ.get("SIDE_PRIORITY", "0")))      # (-1 = high, 0 = default)


def both_seq():
    gemm()
    attn()


def both_par():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    if os.environ.get("ATTN_ON_SIDE") == "1":
        with torch.cuda.stream(side):
            attn()
        gemm()
    else:
        with torch.cuda.stream(side):
            gemm()
        attn()
    cur.wait_stream(side)


res = {"gemm_tn_ms": timeit(gemm), "attn_bwd_ms": timeit(attn), "sequential_ms": timeit(both_seq), "two_streams_ms": timeit(both_par)}
res["settings"] = {k: os.environ.get(k) for k in ("MODE", "SIDE_PRIORITY", "WSI_GEMM_LDS_PAD", "ATTN_ON_SIDE")}
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))
