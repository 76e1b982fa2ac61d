#!/usr/bin/env python
"""What the fused epilogues of the scaled-fp16 NT projection cost: the a_linear forward shape (80000 x 512 x 512, three node types)
with no epilogue / bias / gated skip (bias + gate + residual) / gated skip + row-scale output, TFLOP/s fp32-equivalent.  GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from wsi_hgnn_amd import ops, _native as N
dev = torch.device("cuda:0")
ops.set_gemm_precision("fp16x3")
rows = [(0, 40000), (40000, 64000), (64000, 80000)]
n, K, Nout = 80000, 512, 512
x = torch.rand(n, K, device=dev); h = torch.randn(n, Nout, device=dev)
ws = [torch.randn(Nout, K, device=dev) * 0.03 for _ in range(3)]
bs = [torch.randn(Nout, device=dev) for _ in range(3)]
gate = torch.ones(3, device=dev)
y = torch.empty(n, Nout, device=dev)
bits = ops.row_absmax(x)
cm = torch.zeros(n, N.gemm_absmax_parts(Nout), dtype=torch.int32, device=dev)
def run(epi, cmax):
    groups = []
    for t, (r0, r1) in enumerate(rows):
        g = dict(A=N.ptr(x, r0 * K * 4), lda=K, B=N.ptr(ws[t]), ldb=K, C=N.ptr(y, r0 * Nout * 4), ldc=Nout, M=r1 - r0, N=Nout, K=K,
                 bias=N.ptr(bs[t]), R=N.ptr(h, r0 * Nout * 4), ldr=Nout, gate=N.ptr(gate, 4 * t), a_absmax=N.ptr(bits, r0 * 4), a_absmax_parts=1)
        if cmax:
            g.update(c_absmax=N.ptr(cm, r0 * cm.shape[1] * 4), c_absmax_parts=cm.shape[1], c_absmax_first=0)
        groups.append(g)
    ops._gemm(N.WSI_GEMM_NT, epi, groups, dev)
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
fl = 2.0 * n * K * Nout
for rnd in range(2):
    for name, epi, cmax in (("none", 0, False), ("bias", N.WSI_EPI_BIAS, False), ("bias+cmax", N.WSI_EPI_BIAS, True), ("add_r", N.WSI_EPI_ADD_R, False),
                            ("gated_skip", N.WSI_EPI_GATED_SKIP, False), ("gated_skip+cmax", N.WSI_EPI_GATED_SKIP, True)):
        print(rnd, name, round(fl / timeit(lambda: run(epi, cmax)) / 1e9, 1), flush=True)
ops.set_gemm_precision("fp32")
