#!/usr/bin/env python
"""K / M sweep of the NT GEMM to separate per-K-tile cost from per-output-tile fixed cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from wsi_hgnn_amd import ops, _native as N
dev = torch.device("cuda:0")

def timeit(fn, reps=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def run(M, Nn, K):
    a = torch.randn(M, K, device=dev); b = torch.randn(Nn, K, device=dev); c = torch.empty(M, Nn, device=dev)
    g = [dict(A=N.ptr(a), lda=K, B=N.ptr(b), ldb=K, C=N.ptr(c), ldc=Nn, M=M, N=Nn, K=K)]
    ms = timeit(lambda: ops._gemm(N.WSI_GEMM_NT, 0, g, dev))
    tiles = ((M + 127) // 128) * ((Nn + 127) // 128)
    print(f"M={M:6d} N={Nn:5d} K={K:5d} tiles={tiles:6d} ({tiles/768:6.2f} rounds)  {ms:8.3f} ms  {2.0*M*Nn*K/ms/1e9:7.1f} TF  us/tile/ktile={ms*1e3/tiles/(K/32)*768:7.3f}", flush=True)

for K in (128, 256, 512, 1024, 2048, 4096):
    run(98304, 512, K)      # 3072 tiles = exactly 4 rounds of 768
for M in (24576, 49152, 98304, 80000):
    run(M, 1536, 512)
run(98304, 512, 512)
run(98304 + 128 * 100, 512, 512)
