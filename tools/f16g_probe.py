#!/usr/bin/env python
"""The LDS-DMA scaled-fp16 GEMM (gemm_fp16x3g_kernel) against another form of it - `w`: the register-fragment kernel it replaced
(gemm_fp16x3w_kernel), `q` (round 5): the same kernel with the waves as a 2 x 2 grid of 64 x 64 tiles (gemm_fp16x3q_kernel) - selected with
WSI_GEMM_F16_KERNEL in the measurement build: bit equality over shapes / epilogues / edge tiles / scale exchange / column statistics, then rates on
the bench's projection shapes, interleaved in one process.  GPU.  `python tools/f16g_probe.py [out.json] [w|q]`"""
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
OTHER = sys.argv[2] if len(sys.argv) > 2 else "w"


def kernel(which):
    if which == "g":
        os.environ.pop("WSI_GEMM_F16_KERNEL", None)
    else:
        os.environ["WSI_GEMM_F16_KERNEL"] = OTHER


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


def one(op, M, Nn, K, epi=0, parts=0, seed=0, chunks=1, want_cmax=False):
    """Run one single-group launch under both kernels; returns (C_g, C_w, cmax_g, cmax_w, fp64 reference)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = (torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-8, 9, (M, 1), generator=g).float())).to(dev)
    w = (torch.randn(Nn, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(Nn, generator=g).to(dev)
    R = torch.randn(M, Nn, generator=g).to(dev)
    Mm = (torch.rand(M, Nn, generator=g) > 0.2).float().to(dev) * 1.25
    gate = torch.tensor([0.3], device=dev)
    C0 = torch.randn(M, Nn, generator=g).to(dev)
    outs = []
    for which in ("g", OTHER):
        kernel(which)
        C = C0.clone()
        grp = dict(A=N.ptr(a), lda=K, C=N.ptr(C), ldc=Nn, M=M, N=Nn, K=K, bias=N.ptr(bias), R=N.ptr(R), ldr=Nn,
                   gate=N.ptr(gate), Mm=N.ptr(Mm), ldm=Nn)
        if op == N.WSI_GEMM_NT:
            grp.update(B=N.ptr(w), ldb=K)
        else:
            wt = w.t().contiguous()                    # [K, N]
            if chunks == 1:
                grp.update(B=N.ptr(wt), ldb=Nn)
                keep = wt
            else:
                kc = K // chunks
                keep = [wt[i * kc:(i + 1) * kc].clone() for i in range(chunks)]
                grp.update(B=N.ptr(keep[0]), B1=N.ptr(keep[1]), B2=N.ptr(keep[2]) if chunks > 2 else None, b_chunk=kc, ldb=Nn)
        if parts:
            bits = ops.row_absmax(a)
            tab = torch.zeros(M, parts, dtype=torch.int32, device=dev)
            tab[:, seed % parts] = bits[:, 0]
            grp.update(a_absmax=N.ptr(tab), a_absmax_parts=parts)
        cm = None
        if want_cmax:
            cm = torch.zeros(M, N.gemm_absmax_parts(Nn), dtype=torch.int32, device=dev)
            grp.update(c_absmax=N.ptr(cm), c_absmax_parts=cm.shape[1], c_absmax_first=0)
        cs = None
        if want_cmax and OTHER in ("q", "p"):                 # (the register-fragment kernel leaves no column statistics)
            parts_m = (M + 127) // 128
            ldc_ = (Nn + 3) & ~3
            cs = (torch.zeros(parts_m, ldc_, dtype=torch.int32, device=dev), torch.zeros(parts_m, ldc_, device=dev))
            grp.update(c_colmax=N.ptr(cs[0]), c_colsum=N.ptr(cs[1]), c_col_ld=ldc_)
        ops._gemm(op, epi, [grp], dev)
        torch.cuda.synchronize()
        outs.append((C, cm if cs is None else torch.cat([cm.reshape(-1).float(), cs[0].reshape(-1).float(), cs[1].reshape(-1)])))
    kernel("g")
    ref = a.double() @ w.double().t()
    return outs[0][0], outs[1][0], outs[0][1], outs[1][1], ref


report = {"equal": [], "rates": {}}
bad = 0
E = N
cases = []
for op in (N.WSI_GEMM_NT, N.WSI_GEMM_NN):
    for (M, Nn, K) in ((128, 128, 32), (256, 128, 64), (1000, 512, 512), (333, 200, 96), (4096, 1536, 512), (77, 50, 1024), (129, 257, 160)):
        cases.append((op, M, Nn, K, 0, 0, 1, False))
    cases.append((op, 640, 512, 512, N.WSI_EPI_BIAS, 0, 1, True))
    cases.append((op, 650, 500, 512, N.WSI_EPI_BIAS | N.WSI_EPI_GELU, 3, 1, True))
    cases.append((op, 512, 512, 256, N.WSI_EPI_GATED_SKIP, 2, 1, True))
    cases.append((op, 512, 384, 256, N.WSI_EPI_GATED_SKIP | N.WSI_EPI_MUL_M, 0, 1, False))
    cases.append((op, 300, 128, 128, N.WSI_EPI_ACCUMULATE, 0, 1, False))
    cases.append((op, 384, 256, 128, N.WSI_EPI_SCALE_GATE, 1, 1, True))
    cases.append((op, 384, 256, 128, N.WSI_EPI_ADD_R | N.WSI_EPI_R_1MG, 1, 1, True))
cases.append((N.WSI_GEMM_NN, 1024, 512, 1536, N.WSI_EPI_ADD_R, 2, 3, True))       # chunked B (the dX of K|Q|V)
cases.append((N.WSI_GEMM_NN, 500, 512, 1024, 0, 0, 2, False))
for i, (op, M, Nn, K, epi, parts, chunks, cmax) in enumerate(cases):
    cg, cw, mg, mw, ref = one(op, M, Nn, K, epi, parts, seed=i, chunks=chunks, want_cmax=cmax)
    eq = bool(torch.equal(cg, cw)) and (mg is None or bool(torch.equal(mg, mw)))
    rec = {"op": "NT" if op == N.WSI_GEMM_NT else "NN", "M": M, "N": Nn, "K": K, "epi": epi, "parts": parts, "chunks": chunks, "equal": eq,
           "finite": bool(torch.isfinite(cg).all())}
    if epi == 0:
        rec["rel_err_vs_fp64"] = float(((cg.double() - ref).norm() / ref.norm()).item())
    if not eq:
        bad += 1
        d = (cg - cw).abs()
        rec["max_abs_diff"] = float(d.max().item())
        rec["n_diff"] = int((d > 0).sum().item())
        idx = torch.nonzero(d > 0)[:5].tolist()
        rec["first_diff"] = idx
    report["equal"].append(rec)
    print(rec, flush=True)
report["all_equal"] = bad == 0

# ---- rates on the bench's projection shapes (three node-type groups of 40000 / 24000 / 16000 rows), interleaved
rows = [(0, 40000), (40000, 64000), (64000, 80000)]
n = 80000


def shape(name, K, Nout, nproj, nn):
    x = torch.rand(n, K, device=dev)
    ws = [torch.randn(Nout, K, device=dev) * 0.03 for _ in range(3 * nproj)]
    y = torch.empty(n, nproj * Nout, device=dev)
    gy = torch.randn(n, nproj * Nout, device=dev)
    gx = torch.empty(n, K, device=dev)
    bits = ops.row_absmax(x)
    gbits = ops.row_absmax(gy)

    def fwd():
        groups = []
        for t, (r0, r1) in enumerate(rows):
            for j in range(nproj):
                w = ws[t * nproj + j]
                groups.append(dict(A=N.ptr(x, r0 * K * 4), lda=K, B=N.ptr(w), ldb=K, C=N.ptr(y, (r0 * nproj * Nout + j * Nout) * 4),
                                   ldc=nproj * Nout, M=r1 - r0, N=Nout, K=K, a_absmax=N.ptr(bits, r0 * 4), a_absmax_parts=1))
        ops._gemm(N.WSI_GEMM_NT, 0, groups, dev)

    def dx():
        groups = []
        for t, (r0, r1) in enumerate(rows):
            w3 = ws[t * nproj:(t + 1) * nproj]
            g = dict(A=N.ptr(gy, r0 * nproj * Nout * 4), lda=nproj * Nout, B=N.ptr(w3[0]), ldb=K, C=N.ptr(gx, r0 * K * 4), ldc=K,
                     M=r1 - r0, N=K, K=nproj * Nout, a_absmax=N.ptr(gbits, r0 * 4), a_absmax_parts=1)
            if nproj == 3:
                g.update(B1=N.ptr(w3[1]), B2=N.ptr(w3[2]), b_chunk=Nout)
            groups.append(g)
        ops._gemm(N.WSI_GEMM_NN, 0, groups, dev)

    fl = 2.0 * n * K * Nout * nproj
    res = {}
    for rnd in range(3):
        for which in ("g", OTHER):
            kernel(which)
            for nm, fn in (("NT", fwd),) + ((("NN", dx),) if nn else ()):
                ms = timeit(fn)
                res.setdefault(f"{nm}_{which}", []).append(round(fl / ms / 1e9, 1))
    kernel("g")
    report["rates"][name] = res
    print(name, res, flush=True)


shape("adapt K=1024 N=512", 1024, 512, 1, False)
shape("kqv K=512 N=1536", 512, 512, 3, True)
shape("a_lin K=512 N=512", 512, 512, 1, True)
ops.set_gemm_precision("fp32")
if len(sys.argv) > 1:
    json.dump(report, open(sys.argv[1], "w"), indent=1)
print("ALL EQUAL" if bad == 0 else f"{bad} MISMATCHING CASES")
