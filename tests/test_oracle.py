"""CPU tests of the oracle itself (no GPU): two formulations agree, gradcheck, known answers, and the
plan-level kernel reference reproduces the module-level oracle (so the CSR/CSC plan is right)."""
import math

import pytest
import torch

from oracle import models as OM, dense, dgl_semantics as S, kernel_ref
import wsi_hgnn_amd as W
from wsi_hgnn_amd import synthetic

ND = {"0": 0, "1": 1, "2": 2}


def _double_graph(g):
    for t in g.ntypes:
        g.nodes[t].data["feat"] = g.nodes[t].data["feat"].double()
    return g


@pytest.mark.parametrize("dst_mode", ["uniform", "hub"])
def test_scatter_vs_dense_layer(dst_mode):
    torch.manual_seed(0)
    m = OM.HEATNet4(16, 32, 2, 2, 4, ND, 0.0).double()
    g = _double_graph(synthetic.hetero_graph(120, 16, seed=5, dst_mode=dst_mode))
    h = {nt: m.adapt_ws[ND[nt]](g.nodes[nt].data["feat"]) for nt in g.ntypes}
    sim = g.edata["sim"]
    a = m.gcs[0](g, h, sim)
    b = dense.heat_layer_dense(m.gcs[0], g, h, sim)
    for k in a:
        assert (a[k] - b[k]).abs().max().item() < 1e-12


def test_plan_reference_matches_module_oracle():
    torch.manual_seed(1)
    D, H = 32, 4
    m = OM.HEATNet4(16, D, 2, 1, H, ND, 0.0).double()
    g = _double_graph(W.batch([synthetic.hetero_graph(90, 16, seed=s, dst_mode="hub") for s in (1, 2)]))
    layer = m.gcs[0]
    h = {nt: m.adapt_ws[ND[nt]](g.nodes[nt].data["feat"]) for nt in g.ntypes}
    want = layer(g, h, g.edata["sim"])
    plan = g.plan()
    hcat = torch.cat([h[t] for t in g.ntypes])
    rows = list(zip(plan.type_off[:-1], plan.type_off[1:]))
    kqv = torch.zeros(plan.num_nodes, 3 * D, dtype=torch.float64)
    for (a, b), t in zip(rows, g.ntypes):
        i = ND[t]
        kqv[a:b, 0:D] = layer.k_linears[i](hcat[a:b])
        kqv[a:b, D:2 * D] = layer.q_linears[i](hcat[a:b])
        kqv[a:b, 2 * D:] = layer.v_linears[i](hcat[a:b])
    tt = kernel_ref.heat_attention_ref(kqv, layer.e_linear.weight, layer.e_linear.bias, plan, g.cat_edata_csr("sim").double(), D, H)
    for (a, b), t in zip(rows, g.ntypes):
        i = ND[t]
        al = torch.sigmoid(layer.skip[i])
        got = al * layer.a_linears[i](tt[a:b]) + (1 - al) * hcat[a:b]
        assert (got - want[t]).abs().max().item() < 1e-12


def test_csc_is_a_permutation_of_csr():
    g = synthetic.hetero_graph(200, 4, seed=9, dst_mode="hub")
    p = g.plan()
    assert sorted(p.csc_eid.tolist()) == list(range(p.num_edges))
    # destination of every CSR edge, independently of the plan: the COO destinations (relation-major) through plan.perm
    off = dict(zip(g.ntypes, p.type_off))
    dst = torch.cat([g.edges(r)[1] + off[r[2]] for r in g.canonical_etypes])[p.perm]
    pdst, seg = kernel_ref.plan_edge_tables(p)
    assert torch.equal(pdst, dst)
    assert torch.equal(dst[p.csc_eid.long()].int(), p.csc_dst)
    src_sorted = p.src[p.csc_eid.long()]
    assert torch.all(src_sorted[1:] >= src_sorted[:-1])
    assert p.rowptr[-1].item() == p.num_edges and p.colptr[-1].item() == p.num_edges
    # every edge sits in the segment of its dst node and relation slot
    assert torch.all((seg >= p.node_seg[dst].long()) & (seg < p.node_seg[dst + 1].long()))


def test_gradcheck_attention_reference():
    g = synthetic.hetero_graph(24, 4, seed=2, dst_mode="hub")
    plan = g.plan()
    D, H = 8, 2
    torch.manual_seed(0)
    kqv = torch.randn(plan.num_nodes, 3 * D, dtype=torch.float64, requires_grad=True)
    ew = torch.tensor([[0.6]], dtype=torch.float64, requires_grad=True)
    eb = torch.tensor([0.2], dtype=torch.float64, requires_grad=True)
    sim = g.cat_edata_csr("sim").double()
    assert torch.autograd.gradcheck(lambda a, b, c: kernel_ref.heat_attention_ref(a, b, c, plan, sim, D, H), (kqv, ew, eb), atol=1e-7)


# ----------------------------------------------------------------------------- analytic known answers
def _tiny(rel_edges, n=(3, 2, 1), D=8):
    from collections import OrderedDict
    nn_ = OrderedDict(zip(["0", "1", "2"], n))
    edges = OrderedDict()
    sim = {}
    for r, (u, v, s) in rel_edges.items():
        edges[r] = (torch.tensor(u, dtype=torch.int64), torch.tensor(v, dtype=torch.int64))
        sim[r] = torch.tensor(s, dtype=torch.float64)
    g = W.HeteroGraph.from_coo(nn_, edges, sim=sim)
    return g


def test_known_answers():
    torch.manual_seed(4)
    D, H = 8, 2
    layer = OM.HEATLayer(D, D, ND, H, 0.0).double()
    h = {"0": torch.randn(3, D, dtype=torch.float64), "1": torch.randn(2, D, dtype=torch.float64),
         "2": torch.randn(1, D, dtype=torch.float64)}
    # (1) sim == 0 and b_e == 0  ->  all logits 0 -> uniform attention: t = mean of v[src]
    with torch.no_grad():
        layer.e_linear.bias.zero_()
    g = _tiny({("1", "pos", "0"): ([0, 1, 1], [0, 0, 2], [0.0, 0.0, 0.0])})
    out = layer(g, h, g.edata["sim"] if isinstance(g.edata["sim"], dict) else {g.canonical_etypes[0]: g.edata["sim"]})
    v = layer.v_linears[1](h["1"])
    al = torch.sigmoid(layer.skip[0])
    t0 = (v[0] + v[1]) / 2           # node 0: two in-edges, uniform
    t2 = v[1]                        # node 2: one in-edge -> a = 1
    t1 = torch.zeros(D, dtype=torch.float64)   # node 1: isolated -> 0
    want = torch.stack([t0, t1, t2])
    want = al * layer.a_linears[0](want) + (1 - al) * h["0"]
    assert (out["0"] - want).abs().max().item() < 1e-12
    # types 1 and 2 have no incoming relation -> passthrough
    assert torch.equal(out["1"], h["1"]) and torch.equal(out["2"], h["2"])
    # (2) R_d = 2 with one EMPTY relation: t = m / 2
    g2 = _tiny({("1", "pos", "0"): ([0, 1, 1], [0, 0, 2], [0.0, 0.0, 0.0]), ("2", "neg", "0"): ([], [], [])})
    out2 = layer(g2, h, g2.edata["sim"])
    want2 = al * layer.a_linears[0](torch.stack([t0, t1, t2]) / 2) + (1 - al) * h["0"]
    assert (out2["0"] - want2).abs().max().item() < 1e-12


def test_readout_semantics():
    x = torch.arange(12, dtype=torch.float64).reshape(6, 2)
    bnn = torch.tensor([2, 0, 4])
    assert torch.equal(S.segment_readout(x, bnn, "sum"), torch.tensor([[2., 4.], [0., 0.], [28., 32.]], dtype=torch.float64))
    assert torch.equal(S.segment_readout(x, bnn, "mean"), torch.tensor([[1., 2.], [0., 0.], [7., 8.]], dtype=torch.float64))
    assert torch.equal(S.segment_readout(x, bnn, "max"), torch.tensor([[2., 3.], [0., 0.], [10., 11.]], dtype=torch.float64))


def test_linear_attention_block_is_identity():
    torch.manual_seed(0)
    blk = OM.LinearAttentionBlock(16)
    l = torch.randn(5, 16, requires_grad=True)
    g = torch.randn(5, 16)
    out = blk(l, g)
    assert torch.equal(out, l)
    out.sum().backward()
    assert blk.op.weight.grad.abs().max().item() == 0.0


def test_construct_oracle_known_answers():
    """oracle/construct.py (restating graph_constructor.py:263-296) on cases with answers known by hand."""
    import numpy as np
    from oracle import construct as OC
    # points on a line at 0, 1, 3, 7: 2 nearest others, ties impossible
    pts = np.array([[0.0, 0], [1, 0], [3, 0], [7, 0]], dtype=np.float32)
    nbr, d2 = OC.knn_bruteforce(pts, radius=3)
    assert nbr.tolist() == [[1, 2], [0, 2], [1, 0], [2, 1]]
    assert d2.tolist() == [[1, 9], [1, 4], [4, 9], [16, 36]]
    # equidistant neighbours: the smaller index wins
    sq = np.array([[0.0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]], dtype=np.float32)
    assert OC.knn_bruteforce(sq, radius=3)[0][0].tolist() == [1, 2]
    # Pearson typing: perfectly correlated -> 'pos' (1), anti-correlated -> 'neg' (0); `corr > 0` is strict
    f = np.array([[1, 2, 3, 4], [2, 4, 6, 8.5], [4, 3, 2, 1], [8, 6, 4, 2.5]], dtype=np.float32)
    a, b, et, es = OC.edge_lists(f, radius=2)
    assert a.tolist() == [0, 1, 2, 3]
    for i, j, t, s in zip(a, b, et, es):
        assert t == (1 if s > 0 else 0)
    assert abs(es[0] - np.corrcoef(f[0], f[b[0]])[0, 1]) < 1e-6
    # to_heterogeneous: ids renumbered per type in increasing homogeneous id; relations lexicographic in type ids
    ids, rels = OC.to_heterogeneous(4, np.array([0, 1, 2, 3]), np.array([1, 0, 3, 2]), [1, 0, 1, 0], [1, 1, 0, 0], ["0", "1"], ["neg", "pos"])
    assert ids["0"].tolist() == [1, 3] and ids["1"].tolist() == [0, 2]
    assert list(rels.keys()) == [("0", "neg", "1"), ("0", "pos", "1"), ("1", "neg", "0"), ("1", "pos", "0")]
    u, v, m = rels[("1", "pos", "0")]
    assert (u.tolist(), v.tolist(), m.tolist()) == ([0], [0], [0])


def test_locality_order_positions_and_processing_order():
    """graph.apply_locality_order: '_pos' is a slide-wide permutation (every position 0..N-1 exactly once across the node
    types), the graph is isomorphic to the input (same multiset of (src feature row, dst feature row, sim)), and a plan built
    from such graphs walks destination nodes graph by graph in ascending position (after the hub prefix)."""
    g = synthetic.hetero_graph(120, 4, seed=3, dst_mode="hub")
    h = W.apply_locality_order(g)
    pos = torch.cat([h.nodes[t].data["_pos"] for t in h.ntypes])
    assert sorted(pos.tolist()) == list(range(g.num_nodes()))
    for r in g.canonical_etypes:
        s, _, d = r
        def sig(G):
            u, v = G.edges(r)
            rows = torch.cat([G.nodes[s].data["feat"][u], G.nodes[d].data["feat"][v], G.edata["sim"][r][:, None]], dim=1)
            return sorted(map(tuple, rows.tolist()))
        assert sig(g) == sig(h)
    b = W.batch([h, W.apply_locality_order(synthetic.hetero_graph(80, 4, seed=4))])
    p = b.plan()
    assert p.locality and not g.plan().locality
    order = p.order_dst.long()[p.num_heavy:]
    off = b.type_offsets()
    gid = torch.cat([torch.repeat_interleave(torch.arange(2), b.batch_num_nodes(t)) for t in b.ntypes])
    bpos = torch.cat([b.nodes[t].data["_pos"] for t in b.ntypes])
    key = gid[order] * 10_000 + bpos[order]
    assert torch.all(key[1:] >= key[:-1])
    assert sorted(p.order_dst.tolist()) == list(range(b.num_nodes())) and sorted(p.order_src.tolist()) == list(range(b.num_nodes()))


def test_real_schema_generator_matches_the_graph_constructor_schema():
    """synthetic.real_schema_graph: 6 node types '0'..'5', every patch sends out_edges edges, a relation exists only if it has
    edges (dgl.to_heterogeneous, graph_constructor.py:285-297), 'pos' edges carry sim > 0 and 'neg' edges sim < 0."""
    g = synthetic.real_schema_graph(600, 4, seed=1)
    assert g.ntypes == [str(i) for i in range(6)] and g.num_edges() == 600 * 8
    assert all(g.num_edges(r) > 0 for r in g.canonical_etypes) and len(g.canonical_etypes) <= 72
    assert g.canonical_etypes == sorted(g.canonical_etypes)
    for (s, e, d) in g.canonical_etypes:
        sim = g.edata["sim"][(s, e, d)]
        assert bool((sim > 0).all()) if e == "pos" else bool((sim < 0).all())
    outdeg = torch.zeros(600, dtype=torch.int64)
    off = dict(zip(g.ntypes, g.type_offsets()))
    for (s, e, d) in g.canonical_etypes:
        outdeg.index_add_(0, g.edges((s, e, d))[0] + off[s], torch.ones(g.num_edges((s, e, d)), dtype=torch.int64))
    assert bool((outdeg == 8).all())


def test_remove_nodes_dgl_semantics():
    """graph.remove_nodes: the node and its incident edges go, remaining ids shift down, fields follow, empty relations stay."""
    from collections import OrderedDict
    g = W.HeteroGraph.from_coo(OrderedDict([("0", 4), ("1", 2)]),
                               OrderedDict([(("0", "pos", "0"), (torch.tensor([0, 1, 2, 3]), torch.tensor([1, 2, 3, 0]))),
                                            (("1", "neg", "0"), (torch.tensor([0, 1]), torch.tensor([2, 2])))]),
                               feat={"0": torch.arange(8.).view(4, 2), "1": torch.arange(4.).view(2, 2)},
                               sim={("0", "pos", "0"): torch.tensor([.1, .2, .3, .4]), ("1", "neg", "0"): torch.tensor([-.5, -.6])})
    h = W.remove_nodes(g, torch.tensor([2]), "0")
    assert h.num_nodes("0") == 3 and h.num_nodes("1") == 2 and h.canonical_etypes == g.canonical_etypes
    u, v = h.edges(("0", "pos", "0"))
    assert u.tolist() == [0, 2] and v.tolist() == [1, 0]                     # edges 0->1 and 3->0 survive, node 3 is now 2
    assert h.edata["sim"][("0", "pos", "0")].tolist() == pytest.approx([.1, .4])
    assert h.num_edges(("1", "neg", "0")) == 0                               # both edges pointed at the removed node: relation kept, empty
    assert torch.equal(h.nodes["0"].data["feat"], g.nodes["0"].data["feat"][[0, 1, 3]])
    assert torch.equal(h.nodes["1"].data["feat"], g.nodes["1"].data["feat"])


def test_row_scales_travel_on_the_tensor_object():
    """Host logic of the fp16x3 scale exchange (ops.attach_row_scales / row_scales_of): the scales answer for the very tensor object they were
    attached to, at the version they were taken at - not for a view or another tensor, not after an in-place write through any alias, not outside
    the scaled modes; nothing is shared between tensors (no process-wide registry: a thousand other tensors cannot evict an entry)."""
    import torch
    from wsi_hgnn_amd import ops
    try:
        ops.set_gemm_precision("fp16x3")
        x = torch.zeros(6, 4)
        bits = torch.zeros(6, 2, dtype=torch.int32)
        ops.attach_row_scales(x, bits)
        assert ops.row_scales_of(x) is bits
        assert ops.row_scales_of(x.contiguous()) is bits           # (contiguous() of a contiguous tensor is the tensor itself)
        assert ops.row_scales_of(x.view(6, 4)) is None and ops.row_scales_of(x.t()) is None and ops.row_scales_of(x[1:]) is None
        assert ops.row_scales_of(x.detach()) is None and ops.row_scales_of(torch.zeros(6, 4)) is None
        x.view(6, 4).add_(1)                                        # in-place write through an alias: the version counter is shared
        assert ops.row_scales_of(x) is None
        ops.attach_row_scales(x, bits)
        others = [torch.zeros(1) for _ in range(1000)]
        for o in others:
            ops.attach_row_scales(o, torch.zeros(1, 1, dtype=torch.int32))
        assert ops.row_scales_of(x) is bits
        ops.set_gemm_precision("bf16x6")
        assert ops.row_scales_of(x) is None and ops._new_row_scale(4, 2, "cpu") is None
        ops.set_gemm_precision("auto")
        assert ops.row_scales_of(x) is bits                         # a fact about the tensor, not about the mode it was taken under
        assert ops._new_row_scale(40, 2, "cpu").shape == (40, 2)
        assert ops._new_row_scale(32, 2, "cpu") is None            # short tensors live on the skinny kernels: no scales
        assert ops._new_row_scale(80000, 8, "cpu", 512).shape == (80000, 8)       # a consumer with K = 512 over 80k rows: fp16x3
        assert ops._new_row_scale(80000, 4, "cpu", 200) is None and ops._new_row_scale(500, 8, "cpu", 512) is None   # bf16x6 anyway
    finally:
        ops.set_gemm_precision("fp32")


def test_tensor_annotations_survive_autograd_hand_over_and_die_with_accumulation():
    """What the explicit hand-over relies on: a tensor returned by one autograd node's backward reaches the next node's backward as the SAME Python
    object (annotation intact); when the engine accumulates two contributions the fact is gone (in-place accumulation moves the version,
    out-of-place creates a new tensor)."""
    import torch
    from wsi_hgnn_amd import ops
    seen = []

    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x + 1.0

        @staticmethod
        def backward(ctx, g):
            gx = g * 2.0
            ops._annotate(gx, "_wsi_test_fact", "from-producer")
            return gx

    class Consumer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            y = x * 3.0
            ops._annotate(y, "_wsi_test_fact", "fwd")
            return y

        @staticmethod
        def backward(ctx, g):
            seen.append(ops._annotation(g, "_wsi_test_fact"))
            return g * 3.0

    w = torch.randn(5, requires_grad=True)
    y = Consumer.apply(w * 1.0)
    assert ops._annotation(y, "_wsi_test_fact") == "fwd"
    Producer.apply(y).sum().backward()
    assert seen == ["from-producer"]
    seen.clear()
    y = Consumer.apply(w * 1.0)
    (Producer.apply(y) + Producer.apply(y)).sum().backward()        # two contributions to y's gradient: accumulated by the engine
    assert seen == [None]


def test_edge_softmax_against_dgl_published_example():
    """Known-answer test from DGL's own documentation of ``dgl.nn.functional.edge_softmax`` (the example in its docstring, DGL >= 0.5):
        g = dgl.graph((th.tensor([0, 0, 0, 1, 1, 2]), th.tensor([0, 1, 2, 1, 2, 2]))); edata = th.ones(6, 1)
        edge_softmax(g, edata)                 -> [1, .5, .3333, .5, .3333, .3333]      (norm_by='dst', the default the models use)
        edge_softmax(g, edata, norm_by='src')  -> [.3333, .3333, .3333, .5, .5, 1]
    The oracle's restatement (oracle/dgl_semantics.py::edge_softmax_dst, reached from models/HEATNet4.py:113) must reproduce both: the
    second one by exchanging the roles of the end points.  This pins the one DGL primitive whose semantics the whole path hangs on
    to a vector DGL itself publishes (DGL cannot be imported here)."""
    from oracle import dgl_semantics as D
    src = torch.tensor([0, 0, 0, 1, 1, 2])
    dst = torch.tensor([0, 1, 2, 1, 2, 2])
    e = torch.ones(6, 1)
    by_dst = D.edge_softmax_dst(e, dst, 3)
    assert torch.allclose(by_dst.flatten(), torch.tensor([1.0, 0.5, 1 / 3, 0.5, 1 / 3, 1 / 3]), atol=1e-7)
    by_src = D.edge_softmax_dst(e, src, 3)
    assert torch.allclose(by_src.flatten(), torch.tensor([1 / 3, 1 / 3, 1 / 3, 0.5, 0.5, 1.0]), atol=1e-7)
    # per trailing dimension (heads) independently, max-subtracted: large logits do not overflow
    big = torch.tensor([[1000.0, 0.0], [1000.0, 1.0], [999.0, 2.0], [5.0, 5.0], [5.0, 5.0], [7.0, 7.0]])
    a = D.edge_softmax_dst(big, dst, 3)
    assert torch.isfinite(a).all()
    for d in range(3):
        assert torch.allclose(a[dst == d].sum(0), torch.ones(2), atol=1e-6)


def test_reduce_plan_helpers_and_broadcast_annotation():
    """Host-side pieces of the S-row path, no GPU: the per-segment tables of a ReducePlan (counts, reciprocals with 0 for an
    empty segment, row -> segment map, mapping of node-type row ranges onto runs of segments) and the annotation by which a readout's
    backward tells the layer below that its gradient is a broadcast - valid for that very tensor at that version, never a different or a modified one."""
    from wsi_hgnn_amd import ops
    ptr = [0, 3, 3, 7, 12, 12, 20]                      # six segments, two of them empty
    rp = ops.ReducePlan.from_ptr(ptr, "cpu")
    assert rp.num_segs == 6 and rp.num_rows == 20 and rp.has_empty()
    assert rp.counts().view(-1).tolist() == [3.0, 0.0, 4.0, 5.0, 0.0, 8.0]
    assert torch.allclose(rp.inv_counts().view(-1), torch.tensor([1 / 3, 0.0, 0.25, 0.2, 0.0, 0.125]), rtol=1e-6, atol=0)
    assert rp.nonempty().view(-1).tolist() == [1.0, 0.0, 1.0, 1.0, 0.0, 1.0]
    assert rp.row_segment().tolist() == [0] * 3 + [2] * 4 + [3] * 5 + [5] * 8
    assert rp.segments_of([(0, 7), (7, 12), (12, 20)]) == [(0, 3), (3, 5), (5, 6)]       # an empty segment on a boundary goes with the range on its left: no overlap
    assert rp.segments_of([(0, 5), (5, 20)]) is None                                      # 5 is not a segment boundary
    assert rp.segments_of([(0, 3), (3, 3), (3, 20)]) == [(0, 2), (0, 0), (2, 6)]            # an empty row range maps to no segment
    rp2 = ops.ReducePlan.from_ptr(ptr, "cpu")          # a plan whose builder recorded which segments are whose: that record wins
    rp2.type_rows, rp2.type_segments = [(0, 3), (3, 12), (12, 20)], [(0, 2), (2, 4), (4, 6)]
    assert rp2.segments_of([(0, 3), (3, 12), (12, 20)]) == [(0, 2), (2, 4), (4, 6)]
    g = torch.zeros(20, 4)
    info = object()
    ops._annotate(g, "_wsi_broadcast", info)
    assert ops._annotation(g, "_wsi_broadcast") is info and ops._annotation(g, "_wsi_row_scales") is None
    assert ops._annotation(torch.zeros(20, 4), "_wsi_broadcast") is None and ops._annotation(g[1:], "_wsi_broadcast") is None
    g.add_(1.0)                                         # accumulated into: no longer the broadcast the readout wrote
    assert ops._annotation(g, "_wsi_broadcast") is None


def test_take_rows_equals_advanced_indexing_for_a_permutation_prefix():
    """pooling/ASAP.py::_TakeRows (x[perm] for a perm without repeats, scatter backward) against autograd's own indexing: values and gradients, 1-D and 2-D."""
    from wsi_hgnn_amd.pooling.ASAP import _TakeRows
    g = torch.Generator().manual_seed(3)
    perm = torch.randperm(37, generator=g)[:19]
    for shape in ((37, 5), (37,)):
        x = torch.randn(*shape, generator=g, dtype=torch.float64, requires_grad=True)
        y = torch.randn(*((19,) + shape[1:]), generator=g, dtype=torch.float64)
        (_TakeRows.apply(x, perm) * y).sum().backward()
        got, x.grad = x.grad.clone(), None
        (x[perm] * y).sum().backward()
        assert torch.equal(_TakeRows.apply(x, perm), x[perm]) and torch.equal(got, x.grad)
