"""Entry points of the S-row algebra of the layer under a readout (csrc/pooled.hip, wsi_pool_factors, wsi_gate_grad, wsi_gemm_small_pair):
each against the tensor-operation formulation it replaced, in float64 where a sum is involved.  The model-level parity of the path these
serve is tests/test_headline_path_gpu.py and test_readout_shortcuts_equal_the_full_depth_path."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _ptrs(ts):
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


def _type_major_ranges(T, Bg, gen, empty_at=None):
    sizes = [int(x) for x in torch.randint(40, 700, (T * Bg,), generator=gen)]
    if empty_at is not None:
        sizes[empty_at] = 0
    ranges, pos = [], 0
    for n in sizes:
        ranges.append((pos, pos + n))
        pos += n
    return ranges, pos


@pytest.mark.parametrize("T,H,D,Bg", [(3, 4, 512, 8), (6, 4, 256, 3), (2, 8, 128, 1), (3, 20, 320, 2)])
def test_pool_factors_matches_the_permuted_weighted_sums(T, H, D, Bg):
    from wsi_hgnn_amd import ops, _native as N
    gen = torch.Generator().manual_seed(7 + T + H)
    ranges, n = _type_major_ranges(T, Bg, gen, empty_at=1 if Bg > 1 else None)
    rp = ops.ReducePlan.from_ranges(ranges, _dev())
    h = torch.randn(n, D, generator=gen).to(_dev())
    ctab = torch.rand(n, T * H, generator=gen).to(_dev())
    hp, csum = ops._pooled_factors(h, ctab, rp, T, H)
    S = T * Bg
    # reference: [source seg = tau * Bg + g][j = b * H + hh][D] in float64, then the permutation the consumers want
    hw = torch.zeros(S, T * H, D, dtype=torch.float64)
    cs = torch.zeros(S, T * H, dtype=torch.float64)
    hc, cc = h.double().cpu(), ctab.double().cpu()
    for s, (a, b) in enumerate(ranges):
        hw[s] = cc[a:b].t() @ hc[a:b]
        cs[s] = cc[a:b].sum(dim=0)
    ref_hp = hw.view(T, Bg, T, H, D).permute(2, 1, 3, 0, 4).reshape(S, H, T, D)
    ref_cs = cs.view(T, Bg, T, H).permute(0, 2, 1, 3).reshape(T, S, H)
    assert hp.shape == (S, H, T, D) and csum.shape == (T, S, H)
    assert (hp.double().cpu() - ref_hp).abs().max() <= 1e-5 * ref_hp.abs().max()
    assert (csum.double().cpu() - ref_cs).abs().max() <= 1e-5 * ref_cs.abs().max()
    # the old entry point gives the same numbers (same stage 1, same summation order): bit for bit
    old = ops.segment_weighted_sums(h, ctab, rp).view(T, Bg, T, H, D).permute(2, 1, 3, 0, 4).reshape(S, H, T, D)
    assert torch.equal(old, hp)
    # the same pass with one more weight column of ones: the segment means of h beside unchanged hp / csum (an empty segment reads 0)
    hp2, csum2, hm = ops._pooled_factors(h, ctab, rp, T, H, with_mean=True)
    assert torch.equal(hp2, hp) and torch.equal(csum2, csum)
    ref_hm = torch.stack([hc[a:b].mean(dim=0) if b > a else torch.zeros(D, dtype=torch.float64) for a, b in ranges])
    assert (hm.double().cpu() - ref_hm).abs().max() <= 1e-6 * max(float(ref_hm.abs().max()), 1.0)
    assert (hm - ops._segment_reduce_raw(h, rp, N.WSI_RED_MEAN)[0]).abs().max().item() <= 1e-6


@pytest.mark.parametrize("T,H,D,S,with_scale", [(3, 4, 512, 24, True), (6, 8, 256, 18, True), (20, 2, 64, 5, False)])
def test_pool_tmean(T, H, D, S, with_scale):
    from wsi_hgnn_amd import _native as N
    lib = N.load()
    gen = torch.Generator().manual_seed(11)
    tpart = torch.randn(T, S, D, generator=gen).to(_dev())
    csum = torch.rand(T, S, H, generator=gen).to(_dev())
    bv = [torch.randn(D, generator=gen).to(_dev()) for _ in range(T)]
    if T > 2:
        bv[1] = None                                   # a Linear without bias
    scale = torch.rand(S, generator=gen).to(_dev()) if with_scale else None
    out = torch.empty(S, D, device=_dev())
    N.check(lib.wsi_pool_tmean(N.ptr(tpart), T, S, D, H, N.ptr(csum), _ptrs(bv), N.ptr(scale), N.ptr(out), N.stream()), "wsi_pool_tmean")
    dk = D // H
    ref = tpart.double().sum(dim=0)
    for tau in range(T):
        if bv[tau] is not None:
            ref = ref + (csum[tau].double().unsqueeze(-1) * bv[tau].double().view(1, H, dk)).reshape(S, D)
    if scale is not None:
        ref = ref * scale.double().view(S, 1)
    assert (out.double() - ref).abs().max() <= 2e-6 * ref.abs().max()


@pytest.mark.parametrize("op", ["sum", "mean"])
@pytest.mark.parametrize("S,D,T,n_gates", [(24, 512, 3, 3), (12, 200, 6, 4), (3, 64, 3, 5)])
def test_pool_bwd_prep(op, S, D, T, n_gates):
    from wsi_hgnn_amd import _native as N
    lib = N.load()
    gen = torch.Generator().manual_seed(5 + S)
    g_pool = torch.randn(S, D, generator=gen).to(_dev())
    counts = torch.randint(1, 900, (S,), generator=gen).float()
    counts[S // 2] = 0.0                               # an empty segment
    counts = counts.to(_dev())
    z, hm = torch.randn(S, D, generator=gen).to(_dev()), torch.randn(S, D, generator=gen).to(_dev())
    Bg = S // T
    type_gate = [(i * 2) % n_gates if i != 1 else -1 for i in range(T)]      # type 1: passed through (no gate)
    seg_gate = [type_gate[s // Bg] for s in range(S)]
    skip = torch.randn(n_gates, generator=gen).to(_dev())
    tg = torch.tensor(type_gate, dtype=torch.int32, device=_dev())
    sg = torch.tensor(seg_gate, dtype=torch.int32, device=_dev())
    g_row, g_sum = torch.empty_like(g_pool), torch.empty_like(g_pool)
    g_skip, omg = torch.empty(n_gates, device=_dev()), torch.empty(T, device=_dev())
    N.check(lib.wsi_pool_bwd_prep(N.ptr(g_pool), S, D, N.WSI_RED_MEAN if op == "mean" else N.WSI_RED_SUM, N.ptr(counts), N.ptr(z), N.ptr(hm), N.ptr(sg),
                                  N.ptr(skip), n_gates, N.ptr(tg), T, N.ptr(g_row), N.ptr(g_sum), N.ptr(g_skip), N.ptr(omg), N.stream()), "wsi_pool_bwd_prep")
    c = counts.double().view(S, 1)
    gp = g_pool.double()
    if op == "mean":
        r_row = gp * torch.where(c > 0, 1.0 / c.clamp(min=1), torch.zeros_like(c))
        r_sum = gp * (c > 0).double()
    else:
        r_row, r_sum = gp, gp * c
    dots = (r_sum * (z.double() - hm.double())).sum(dim=1)
    sig = torch.sigmoid(skip.double())
    r_skip = torch.stack([sum((dots[s] for s in range(S) if seg_gate[s] == g), torch.zeros((), dtype=torch.float64, device=_dev())) for g in range(n_gates)]) * (1 - sig)
    r_omg = torch.stack([(1 - sig[type_gate[i]]) if type_gate[i] >= 0 else torch.ones((), dtype=torch.float64, device=_dev()) for i in range(T)])
    assert (g_row.double() - r_row).abs().max() <= 1e-6 * r_row.abs().max()
    assert (g_sum.double() - r_sum).abs().max() <= 1e-6 * r_sum.abs().max()
    assert (g_skip.double() - r_skip).abs().max() <= 1e-5 * max(float(r_skip.abs().max()), 1.0)
    assert (omg.double() - r_omg).abs().max() <= 1e-6


@pytest.mark.parametrize("T,H,D,S", [(3, 4, 512, 24), (6, 8, 256, 12), (18, 2, 64, 4)])
def test_pool_bwd_bias(T, H, D, S):
    from wsi_hgnn_amd import _native as N
    lib = N.load()
    gen = torch.Generator().manual_seed(3)
    gt = torch.randn(S, D, generator=gen).to(_dev())
    bv = [torch.randn(D, generator=gen).to(_dev()) for _ in range(T)]
    csum = torch.rand(T, S, H, generator=gen).to(_dev())
    beta, gbv = torch.empty(T, S, H, device=_dev()), torch.empty(T, D, device=_dev())
    N.check(lib.wsi_pool_bwd_bias(N.ptr(gt), T, S, D, H, _ptrs(bv), N.ptr(csum), N.ptr(beta), N.ptr(gbv), N.stream()), "wsi_pool_bwd_bias")
    dk = D // H
    r_beta = (gt.double().view(1, S, H, dk) * torch.stack(bv).double().view(T, 1, H, dk)).sum(dim=-1)
    r_gbv = (csum.double().unsqueeze(-1) * gt.double().view(1, S, H, dk)).sum(dim=1).reshape(T, D)
    assert (beta.double() - r_beta).abs().max() <= 1e-5 * r_beta.abs().max()
    assert (gbv.double() - r_gbv).abs().max() <= 1e-5 * r_gbv.abs().max()
    # one output at a time
    gbv2 = torch.empty_like(gbv)
    N.check(lib.wsi_pool_bwd_bias(N.ptr(gt), T, S, D, H, None, N.ptr(csum), None, N.ptr(gbv2), N.stream()), "wsi_pool_bwd_bias")
    assert torch.equal(gbv, gbv2)


def test_gate_grad_matches_segment_dots():
    from wsi_hgnn_amd import ops, _native as N
    lib = N.load()
    gen = torch.Generator().manual_seed(2)
    ranges = [(0, 3000), (3000, 3000), (3000, 7500), (7500, 9000)]       # one empty node type
    n, D, n_gates = 9000, 512, 3
    rp = ops.ReducePlan.from_ranges(ranges, _dev())
    g, a, b = (torch.randn(n, D, generator=gen).to(_dev()) for _ in range(3))
    seg_gate = [2, 0, -1, 2]
    skip = torch.randn(n_gates, generator=gen).to(_dev())
    sg = torch.tensor(seg_gate, dtype=torch.int32, device=_dev())
    out = torch.empty(n_gates, device=_dev())
    partial = torch.empty(rp.num_chunks * ((D + 255) // 256), device=_dev())
    N.check(lib.wsi_gate_grad(N.ptr(g), D, N.ptr(a), D, N.ptr(b), D, D, N.ptr(rp.chunk_row), rp.num_chunks, N.ptr(rp.seg_chunk), rp.num_segs,
                              N.ptr(sg), N.ptr(skip), n_gates, N.ptr(partial), N.ptr(out), N.stream()), "wsi_gate_grad")
    dots = ops.segment_dot_diff(g, a, b, rp).double()
    ref = torch.zeros(n_gates, dtype=torch.float64, device=_dev())
    for s, gt in enumerate(seg_gate):
        if gt >= 0:
            ref[gt] += dots[s]
    ref = ref * (1 - torch.sigmoid(skip.double()))
    assert (out.double() - ref).abs().max() <= 1e-6 * max(float(ref.abs().max()), 1.0)
    assert float(out[1]) == 0.0                                      # a gate no segment maps to: written, zero


def test_gemm_small_pair_equals_the_two_launches():
    """dX and dW (+ db) of the classifier head's levels in one launch: the same fma chains as wsi_gemm_grouped runs for such shapes, bit for bit."""
    from wsi_hgnn_amd import ops, _native as N
    gen = torch.Generator().manual_seed(9)
    rows, cin, cout = 24, 512, 256
    x = torch.randn(rows, cin, generator=gen).to(_dev())
    ws = [torch.randn(cout, cin, generator=gen).to(_dev()) for _ in range(3)]
    gy = torch.randn(rows, 3 * cout, generator=gen).to(_dev())

    def run(pair):
        gx = torch.empty(rows, cin, device=_dev())
        gws = [torch.empty_like(w) for w in ws]
        gbs = [torch.empty(cout, device=_dev()) for _ in ws]
        dx = [dict(A=N.ptr(gy, (i * 8 * 3 * cout + i * cout) * 4), lda=3 * cout, B=N.ptr(ws[i]), ldb=cin, C=N.ptr(gx, i * 8 * cin * 4), ldc=cin, M=8, N=cin, K=cout)
              for i in range(3)]
        dw = [dict(A=N.ptr(gy, (i * 8 * 3 * cout + i * cout) * 4), lda=3 * cout, B=N.ptr(x, i * 8 * cin * 4), ldb=cin, C=N.ptr(gws[i]), ldc=cin,
                   colsum_out=N.ptr(gbs[i]), M=cout, N=cin, K=8) for i in range(3)]
        if pair:
            assert ops._gemm_small_pair(dx, 0, dw, 0, _dev())
        else:
            ops._gemm(N.WSI_GEMM_NN, 0, dx, _dev())
            ops._gemm(N.WSI_GEMM_TN, 0, dw, _dev())
        return [gx] + gws + gbs

    a, b = run(True), run(False)
    assert all(torch.equal(p, q) for p, q in zip(a, b))
    # against float64
    for i in range(3):
        gyi = gy[i * 8:(i + 1) * 8, i * cout:(i + 1) * cout].double()
        assert (a[0][i * 8:(i + 1) * 8].double() - gyi @ ws[i].double()).abs().max() <= 1e-4
        assert (a[1 + i].double() - gyi.t() @ x[i * 8:(i + 1) * 8].double()).abs().max() <= 1e-4
        assert (a[4 + i].double() - gyi.sum(dim=0)).abs().max() <= 1e-5
    # a group that is not small is refused (the caller then makes its two calls)
    big = [dict(A=N.ptr(gy), lda=3 * cout, B=N.ptr(ws[0]), ldb=cin, C=N.ptr(a[0]), ldc=cin, M=64, N=cin, K=cout)]
    assert not ops._gemm_small_pair(big, 0, [dict(A=N.ptr(gy), lda=3 * cout, B=N.ptr(x), ldb=cin, C=N.ptr(a[1]), ldc=cin, M=cout, N=cin, K=8)], 0, _dev())


@pytest.mark.parametrize("dst_mode", ["uniform", "hub"])
def test_plan_assembly_kernel_equals_the_tensor_formulation(dst_mode):
    """wsi_plan_assemble (graph.assemble_plan on a GPU: one launch over segment descriptors) against graph.assemble_plan_torch (the same
    concatenations / offset additions / table lookups as ~90 tensor operations): every table of the batch's plan bit for bit - slides of
    different sizes, hub destinations (the exact hub list in front of order_dst)."""
    from wsi_hgnn_amd import synthetic, graph as G_
    from wsi_hgnn_amd.data import StoredGraph
    gs = [synthetic.hetero_graph(n, 16, seed=40 + i, dst_mode=dst_mode) for i, n in enumerate([3000, 500, 7000, 1200, 2500])]
    its = [StoredGraph(g, i % 2, _dev(), True) for i, g in enumerate(gs)]
    ntypes, rels = its[0].ntypes, its[0].rels
    T = len(ntypes)
    counts = [[it.num_nodes[t] for it in its] for t in range(T)]
    hd = G_.PlanHeader(ntypes, rels, [sum(c) for c in counts])
    pa, sa = G_.assemble_plan(hd, [it.pieces for it in its], _dev(), counts)
    pb, sb = G_.assemble_plan_torch(hd, [it.pieces for it in its], _dev(), counts)
    assert torch.equal(sa, sb)
    for name in ("node_seg", "inv_rd", "rowptr", "colptr", "src", "csc_eid", "csc_dst", "order_dst", "order_src", "readout_ptr"):
        a, b = getattr(pa, name), getattr(pb, name)
        assert a.dtype == b.dtype and a.shape == b.shape and torch.equal(a, b), name
    for name in ("num_nodes", "num_edges", "num_segs", "num_heavy", "num_src_rows", "batch_size", "locality", "heavy_degree", "type_off", "rel_slots"):
        assert getattr(pa, name) == getattr(pb, name), name
    if dst_mode == "hub":
        assert pa.num_heavy > 0
