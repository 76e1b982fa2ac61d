#!/usr/bin/env python
"""Where the time of gemm_fp16x3g_kernel goes: the shipped kernel (0) beside its measurement variants (WSI_F16G_ABL: 1 no split
arithmetic, 2 no C stores, 3 no DMA after the first two stages, 4 no B fragment reads, 5 all of them) and the register-fragment
kernel ('w'), interleaved in one process on the bench's projection shapes.  TFLOP/s fp32-equivalent incl. the pre-pass.  GPU."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import _native as N
N.use_measurement_library()        # the WSI_* kernel switches below exist only in the -DWSI_ABLATE build (csrc/common.h::knob)
import __graft_entry__
__graft_entry__.build()
from wsi_hgnn_amd import ops

dev = torch.device("cuda:0")
ops.set_gemm_precision("fp16x3")
rows = [(0, 40000), (40000, 64000), (64000, 80000)]
n = 80000


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


def select(v):
    """'0'..'7': ablation code; 'p': plain C stores; 'w': the register-fragment kernel."""
    os.environ.pop("WSI_GEMM_F16_KERNEL", None)
    os.environ.pop("WSI_F16G_ABL", None)
    os.environ.pop("WSI_F16G_NT", None)
    if v == "p":                               # the shipped kernel with default-policy (not non-temporal) C stores
        os.environ["WSI_F16G_NT"] = "0"
        return
    if v == "w":
        os.environ["WSI_GEMM_F16_KERNEL"] = "w"
    elif v != "0":
        os.environ["WSI_F16G_ABL"] = v


VARIANTS = sys.argv[2].split(",") if len(sys.argv) > 2 else ("0", "p", "w", "1", "2", "3", "4", "5", "6", "7")
out = {}
for name, K, Nout, nproj in (("kqv", 512, 512, 3), ("a_lin", 512, 512, 1), ("adapt", 1024, 512, 1)):
    x = torch.rand(n, K, device=dev)
    ws = [torch.randn(Nout, K, device=dev) * 0.03 for _ in range(3 * nproj)]
    y = torch.empty(n, nproj * Nout, device=dev)
    bits = ops.row_absmax(x)

    def fwd():
        groups = []
        for t, (r0, r1) in enumerate(rows):
            for j in range(nproj):
                groups.append(dict(A=N.ptr(x, r0 * K * 4), lda=K, B=N.ptr(ws[t * nproj + j]), ldb=K,
                                   C=N.ptr(y, (r0 * nproj * Nout + j * Nout) * 4), ldc=nproj * Nout, M=r1 - r0, N=Nout, K=K,
                                   a_absmax=N.ptr(bits, r0 * 4), a_absmax_parts=1))
        ops._gemm(N.WSI_GEMM_NT, 0, groups, dev)

    fl = 2.0 * n * K * Nout * nproj
    res = {}
    for rnd in range(3):
        for v in VARIANTS:
            select(v)
            res.setdefault(v, []).append(round(fl / timeit(fwd) / 1e9, 1))
    select("0")
    out[name] = res
    print(name, res, flush=True)
ops.set_gemm_precision("fp32")
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
