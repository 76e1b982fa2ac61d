"""Parity at BASELINE.json's full sizes (configs[1..4]).

Where the CPU oracle finishes in seconds (one 5k/10k/20k-node graph) the HIP path is compared with it directly;
at the bench size (batch of 8 x 10k nodes) size-independent properties are checked instead: invariance of the logits
under node relabelling and edge reordering, batch == per-graph results, run-to-run bit reproducibility.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
ND = {"0": 0, "1": 1, "2": 2}


def _dev():
    return torch.device("cuda:0")


def _oracle_copy(m, cls, *args):
    o = cls(*args)
    o.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    return o


def test_config2_heatnet2_5k_nodes_vs_oracle(gemm_mode):
    """configs[1]: HEATNet2 on a BRCA-shaped synthetic hetero graph (3 node types, 6 rels, 5k nodes, 1024-d; hidden 256)."""
    from wsi_hgnn_amd import models, synthetic
    from oracle import models as OM
    torch.manual_seed(611)
    args = (1024, 256, 2, 2, 4, ND, 0.0, "mean")
    m = models.HEATNet2(*args).to(_dev())
    o = _oracle_copy(m, OM.HEATNet2, *args)
    g = synthetic.hetero_graph(5000, 1024, seed=611, dst_mode="hub")
    y = torch.tensor([1])
    out = m(g.to(_dev()))
    loss = torch.nn.functional.cross_entropy(out, y.to(_dev()))
    loss.backward()
    ref = o(g)
    rloss = torch.nn.functional.cross_entropy(ref, y)
    rloss.backward()
    assert (out.cpu() - ref).abs().max().item() < 1e-4 and abs(loss.item() - rloss.item()) < 1e-4
    og = dict(o.named_parameters())
    for k, p in m.named_parameters():
        if og[k].grad is not None:
            assert (p.grad.cpu() - og[k].grad).abs().max().item() <= 1e-7 + 1e-4 * og[k].grad.abs().max().item(), k


@pytest.mark.parametrize("dst_mode", ["uniform", "hub"])
def test_config3_heatnet4_10k_nodes_vs_oracle_single_graph(dst_mode, gemm_mode):
    """configs[2] shape, one graph: HEATNet4(1024,512,2 layers,4 heads) on a 10k-node / 80k-edge graph vs the oracle; uniform
    destinations and kNN-like hub destinations (in-degrees up to several hundred: the cooperative hub kernels run)."""
    from wsi_hgnn_amd import models, synthetic
    from oracle import models as OM
    torch.manual_seed(611)
    args = (1024, 512, 2, 2, 4, ND, 0.0, "mean")
    m = models.HEATNet4(*args).to(_dev())
    o = _oracle_copy(m, OM.HEATNet4, *args)
    g = synthetic.hetero_graph(10000, 1024, seed=612, dst_mode=dst_mode)
    y = torch.tensor([0])
    out = m(g.to(_dev()))
    loss = torch.nn.functional.cross_entropy(out, y.to(_dev()))
    loss.backward()
    with torch.no_grad():
        pass
    ref = o(g)
    rloss = torch.nn.functional.cross_entropy(ref, y)
    rloss.backward()
    assert (out.cpu() - ref).abs().max().item() < 1e-4 and abs(loss.item() - rloss.item()) < 1e-4
    og = dict(o.named_parameters())
    worst = 0.0
    for k, p in m.named_parameters():
        if og[k].grad is not None:
            worst = max(worst, (p.grad.cpu() - og[k].grad).abs().max().item() / (og[k].grad.abs().max().item() + 1e-12))
    assert worst < 1e-4, worst


def test_config3_bench_batch_properties():
    """configs[2]/[3] at the bench size (8 x 10k nodes per GPU): properties that need no oracle."""
    import wsi_hgnn_amd as W
    from wsi_hgnn_amd import models, synthetic
    from collections import OrderedDict
    torch.manual_seed(611)
    m = models.HEATNet4(1024, 512, 2, 2, 4, ND, 0.0, "mean").to(_dev())
    graphs = [synthetic.hetero_graph(10000, 1024, seed=611 + i, dst_mode="hub") for i in range(8)]
    G = W.batch(graphs).to(_dev())
    with torch.no_grad():
        out = m(G)
        # (1) batched == per graph (block-diagonal batching does not mix graphs)
        solo = torch.cat([m(g.to(_dev())) for g in graphs[:3]])
        assert (out[:3] - solo).abs().max().item() < 2e-5
        # (2) bit-reproducible (no atomics anywhere on the path)
        assert torch.equal(out, m(G))
        # (3) node relabelling within each type + edge reordering leave the logits unchanged (up to fp32 summation order)
        g0 = graphs[0]
        gen = torch.Generator().manual_seed(3)
        perm = {t: torch.randperm(g0.num_nodes(t), generator=gen) for t in g0.ntypes}
        inv = {t: torch.argsort(perm[t]) for t in g0.ntypes}          # old id -> new id
        edges, sim = OrderedDict(), {}
        for r in g0.canonical_etypes:
            u, v = g0.edges(r)
            eo = torch.randperm(u.numel(), generator=gen)
            edges[r] = (inv[r[0]][u][eo], inv[r[2]][v][eo])
            sim[r] = g0.edata["sim"][r][eo]
        feat = {t: g0.nodes[t].data["feat"][perm[t]] for t in g0.ntypes}
        gp = W.HeteroGraph.from_coo(OrderedDict((t, g0.num_nodes(t)) for t in g0.ntypes), edges, feat=feat, sim=sim)
        assert (m(gp.to(_dev())) - out[:1]).abs().max().item() < 2e-5


def test_config5_hgt_20k_nodes_vs_oracle(gemm_mode):
    """configs[4] shape (single GPU, one graph): HGT hidden 200, 4 heads, 20k nodes (10k/6k/4k), 6 out-edges per node."""
    from wsi_hgnn_amd import models, synthetic
    from oracle import models as OM
    rels = [(str(s), r, str(t)) for r in ("pos", "neg") for s in range(3) for t in range(3)]
    ed = {et: i for i, et in enumerate(rels)}
    torch.manual_seed(611)
    m = models.HGT(ND, ed, 1024, 200, 2, 2, 4).to(_dev()).eval()
    o = OM.HGT(ND, ed, 1024, 200, 2, 2, 4).eval()
    o.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    g = synthetic.hetero_graph(20000, 1024, seed=5, dst_mode="hub", edges_per_dst=3)   # sum E = 6 N (ESCA radius 7)
    y = torch.tensor([1])
    out = m(g.to(_dev()))
    loss = torch.nn.functional.cross_entropy(out, y.to(_dev()))
    loss.backward()
    ref = o(g)
    rloss = torch.nn.functional.cross_entropy(ref, y)
    rloss.backward()
    scale = max(1.0, ref.abs().max().item())
    assert (out.cpu() - ref).abs().max().item() < 1e-4 * scale and abs(loss.item() - rloss.item()) < 1e-4 * scale
    # every parameter gradient at full size against the oracle evaluated in float64 (the rule of tests/test_headline_path_gpu.py::_compare: within
    # 1e-4 of the tensor's largest entry; round 5 compared logits and loss only here and left the gradients to the 400-node case)
    o64 = OM.HGT(ND, ed, 1024, 200, 2, 2, 4).eval().double()
    o64.load_state_dict({k: v.detach().cpu().double() for k, v in m.state_dict().items()})
    r64 = o64(g, {t: g.nodes[t].data["feat"].double() for t in g.ntypes})
    torch.nn.functional.cross_entropy(r64, y).backward()
    assert (out.detach().double().cpu() - r64.detach()).abs().max().item() < 1e-4 * scale
    report = []
    got = dict(m.named_parameters())
    largest = max(p.grad.abs().max().item() for p in o64.parameters() if p.grad is not None)
    for k, p in o64.named_parameters():
        if p.grad is None:
            assert got[k].grad is None or float(got[k].grad.abs().max()) == 0.0, k
            continue
        assert got[k].grad is not None, k
        # (the key biases' exact gradient is ZERO - a key bias shifts every logit of a destination by the same q . b_k W_att, which the softmax
        # ignores: float64 leaves 1e-20 of rounding there, fp32 1e-10 - so a tensor's scale has a floor of 1e-6 of the largest gradient of the model)
        rel = (got[k].grad.double().cpu() - p.grad).abs().max().item() / max(p.grad.abs().max().item(), 1e-6 * largest)
        if rel >= 1e-4:
            report.append((k, rel))
    assert not report, report


def test_attention_kernel_properties_at_bench_size():
    """wsi_heat_attn_fwd/bwd at the bench size (80k nodes, 640k edges, D=512, H=4), properties that need no oracle:
    (1) V = ones  ->  t[w] = (#non-empty relation segments of w) / (#relation slots of w): every softmax sums to one;
    (2) linearity in V:  t(v1 + a v2) = t(v1) + a t(v2);
    (3) the backward is the adjoint of (2):  <g, t(v)> = <dL/dv, v>  for L = <g, t>."""
    import wsi_hgnn_amd as W
    from wsi_hgnn_amd import ops, synthetic
    torch.manual_seed(5)
    G = W.batch([synthetic.hetero_graph(10000, 8, seed=611 + i, dst_mode="hub") for i in range(8)]).to(_dev())
    plan = G.plan()
    sim = G.cat_edata_csr("sim")
    n, D, H = plan.num_nodes, 512, 4
    ew = torch.tensor([[0.8]], device=_dev())
    eb = torch.tensor([0.25], device=_dev())
    kq = torch.randn(n, 2 * D, device=_dev()) * 0.2

    def attend(v):
        kqv = torch.cat([kq[:, :D], kq[:, D:], v], dim=1).contiguous()
        return ops.heat_attention(kqv, ew, eb, plan, sim, D, H)

    with torch.no_grad():
        t1 = attend(torch.ones(n, D, device=_dev()))
        ns, rp = plan.node_seg.long(), plan.rowptr.long()
        nonempty = (rp[1:] - rp[:-1] > 0).to(torch.float32)
        seg_cnt = torch.zeros(n + 1, device=_dev()).index_add_(0, torch.bucketize(torch.arange(plan.num_segs, device=_dev()), ns[1:], right=True), nonempty)[:n]
        slots = (ns[1:] - ns[:-1]).to(torch.float32)
        expect = torch.where(slots > 0, seg_cnt / slots.clamp(min=1), torch.zeros_like(slots))
        assert (t1 - expect[:, None]).abs().max().item() < 2e-6
        v1, v2 = torch.randn(n, D, device=_dev()), torch.randn(n, D, device=_dev())
        lin = attend(v1 + 0.37 * v2) - (attend(v1) + 0.37 * attend(v2))
        assert lin.abs().max().item() < 2e-5
    v = torch.randn(n, D, device=_dev(), requires_grad=True)
    g = torch.randn(n, D, device=_dev())
    t = attend(v)
    (t * g).sum().backward()
    lhs = (t.detach().double() * g.double()).sum().item()
    rhs = (v.grad.double() * v.detach().double()).sum().item()
    assert abs(lhs - rhs) < 1e-6 * max(1.0, abs(lhs)) + 1e-3


def test_graph_construction_properties_at_full_size():
    """construct.knn_pearson on a full-size slide (10k patches x 1024-d, radius 9), properties that need no oracle: no self
    edges, no duplicate neighbours, distances ascending and equal to a direct recomputation, r in [-1, 1] and symmetric for
    mutual neighbours, 50 random rows against the fp64 brute force."""
    from wsi_hgnn_amd import construct
    g = torch.Generator().manual_seed(42)
    n, F, radius = 10000, 1024, 9
    centres = torch.rand(32, F, generator=g)
    x = (centres[torch.randint(0, 32, (n,), generator=g)] + 0.15 * torch.randn(n, F, generator=g)).clamp_(min=0).float()
    xd = x.to(_dev())
    nbr, corr, d2 = construct.knn_pearson(xd, radius)
    assert nbr.shape == (n, radius - 1)
    assert (nbr != torch.arange(n, device=_dev())[:, None]).all()
    srt = torch.sort(nbr, dim=1).values
    assert (srt[:, 1:] != srt[:, :-1]).all()
    assert (d2[:, 1:] >= d2[:, :-1]).all()
    direct = ((xd[:, None, :] - xd[nbr]) ** 2).sum(-1)
    assert ((direct - d2).abs() <= 2e-5 * d2 + 1e-6).all()
    assert (corr.abs() <= 1.0).all()
    # mutual neighbours: the same pair seen from both ends has the same distance and correlation
    back = nbr[nbr]                                                 # [n, k, k]: neighbours of my neighbours
    me = torch.arange(n, device=_dev())[:, None, None]
    mutual = (back == me)
    i_idx, a_idx, b_idx = mutual.nonzero(as_tuple=True)
    assert i_idx.numel() > 1000
    j = nbr[i_idx, a_idx]
    assert (corr[i_idx, a_idx] - corr[j, b_idx]).abs().max().item() < 2e-6
    assert ((d2[i_idx, a_idx] - d2[j, b_idx]).abs() <= 2e-6 * d2[i_idx, a_idx] + 1e-7).all()
    rows = torch.randint(0, n, (50,), generator=g)
    x64 = x.double()
    for r in rows.tolist():
        dd = ((x64 - x64[r]) ** 2).sum(1)
        dd[r] = float("inf")
        ref = torch.topk(dd, radius - 1, largest=False).values
        assert ((d2[r].double().cpu() - ref).abs() <= 2e-5 * ref + 1e-9).all(), r


def test_config1_gcn_2k_nodes_vs_oracle(gemm_mode):
    """configs[0] on the HIP path: GCN(1024,256,2 classes,2 layers,relu,0 dropout,mean) on one homogeneous 2k-node patch graph
    (8 out-edges per node + self loops, 1024-d) — models/GCN.py:64-79, configs/COAD/GCN_Kimia_v2.yml:41-50."""
    import torch.nn.functional as F
    from wsi_hgnn_amd import models, synthetic
    from oracle import models as OM
    torch.manual_seed(611)
    m = models.GCN(1024, 256, 2, 2, F.relu, 0.0, "mean").to(_dev())
    o = OM.GCN(1024, 256, 2, 2, F.relu, 0.0, "mean")
    o.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    g = synthetic.homogeneous_graph(2000, 1024, seed=611)
    y = torch.tensor([1])
    out = m(g.to(_dev()))
    loss = F.cross_entropy(out, y.to(_dev()))
    loss.backward()
    ref = o(g)
    rloss = F.cross_entropy(ref, y)
    rloss.backward()
    assert (out.cpu() - ref).abs().max().item() < 1e-4 and abs(loss.item() - rloss.item()) < 1e-4
    og = dict(o.named_parameters())
    for k, p in m.named_parameters():
        if og[k].grad is not None:
            assert (p.grad.cpu() - og[k].grad).abs().max().item() <= 1e-7 + 1e-4 * og[k].grad.abs().max().item(), k


HGT_RELS = [(str(s), r, str(t)) for r in ("pos", "neg") for s in range(3) for t in range(3)]
HGT_ED = {et: i for i, et in enumerate(HGT_RELS)}


@pytest.mark.parametrize("hidden,B", [(32, 2), (200, 1)])
def test_hgt_asap_matches_oracle(hidden, B, gemm_mode):
    """The HGT + ASAPPooling composition of configs[4] (models/HGT_ASAP.py; the reference has no such model) against the
    oracle composed from the restated reference parts (oracle HGT layers + the DENSE restatement of pooling/ASAP.py:142-199),
    at sizes the O(N^2 F) dense oracle holds: logits, loss, every parameter gradient."""
    import wsi_hgnn_amd as W
    from wsi_hgnn_amd import models, synthetic
    from oracle import models as OM
    torch.manual_seed(611)
    args = (ND, HGT_ED, 48, hidden, 2, 2, 4)
    m = models.HGTASAP(*args).to(_dev()).eval()
    o = OM.HGTASAP(*args).eval()
    o.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    gs = [synthetic.hetero_graph(150, 48, seed=70 + i, dst_mode="hub", edges_per_dst=3) for i in range(B)]
    g = W.batch(gs) if B > 1 else gs[0]
    y = torch.arange(B) % 2
    out = m(g.to(_dev()))
    loss = torch.nn.functional.cross_entropy(out, y.to(_dev()))
    loss.backward()
    ref = o(g)
    rloss = torch.nn.functional.cross_entropy(ref, y)
    rloss.backward()
    scale = max(1.0, ref.abs().max().item())
    assert (out.detach().cpu() - ref.detach()).abs().max().item() < 1e-4 * scale
    assert abs(loss.item() - rloss.item()) < 1e-4 * scale
    og = dict(o.named_parameters())
    for k, p in m.named_parameters():
        if og[k].grad is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        assert (p.grad.cpu() - og[k].grad).abs().max().item() <= 1e-6 + 2e-4 * og[k].grad.abs().max().item(), k
    dead = set(m.dead_parameter_names())
    assert dead == {k for k, p in m.named_parameters() if p.grad is None}


def test_config5_hgt_asap_20k_nodes():
    """configs[4] at its stated size on one GPU: HGT (hidden 200, 4 heads, 18-relation edge_dict) + ASAPPooling on a batch of
    4 x 20k-node ESCA-shaped graphs (10k/6k/4k nodes per type, 6 out-edges per node).  The dense ASAP oracle cannot hold 20k
    nodes, so (1) the HGT half is pinned to the oracle at this size by test_config5_hgt_20k_nodes_vs_oracle, (2) the ASAP half
    is checked against an independent eager-torch scatter formulation of pooling/ASAP.py:150-185 in fp64 on the SAME 80k-node
    input, and (3) properties: per-graph pooled counts ceil(0.8 n), fitness-descending order inside a graph, a coalesced
    pooled edge list with unit self loops and no cross-graph edges, batched == per-graph logits, finite gradients."""
    import math
    import wsi_hgnn_amd as W
    from wsi_hgnn_amd import models, synthetic
    from wsi_hgnn_amd.pooling import ASAP as PA
    torch.manual_seed(611)
    m = models.HGTASAP(ND, HGT_ED, 1024, 200, 2, 2, 4).to(_dev()).eval()
    graphs = [synthetic.hetero_graph(20000, 1024, seed=500 + i, dst_mode="hub", edges_per_dst=3) for i in range(4)]
    G = W.batch(graphs).to(_dev())
    y = torch.tensor([0, 1, 1, 0], device=_dev())
    out = m(G)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    assert torch.isfinite(out).all()
    for k, p in m.named_parameters():
        if k not in set(m.dead_parameter_names()):
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    with torch.no_grad():
        solo = m(graphs[1].to(_dev()))
        assert (solo - out[1:2]).abs().max().item() < 1e-4 * max(1.0, out.abs().max().item())
        # ---- the ASAP half alone on a realistic 80k-node input
        x = torch.randn(G.num_nodes(), 200, device=_dev()) * 0.5
        ei, batch = m.homogeneous_view(G, _dev())
        xp, ei2, ew2, b2, perm = m.asap(x, ei, None, batch)
        n_per = torch.bincount(batch, minlength=4).tolist()
        k_per = [int(math.ceil(0.8 * n)) for n in n_per]
        assert torch.bincount(b2, minlength=4).tolist() == k_per == m.pooled_counts(G)
        assert (b2[1:] >= b2[:-1]).all() and torch.equal(b2, batch[perm])
        assert perm.unique().numel() == perm.numel()
        # eager fp64 restatement of :150-185 (scatter form) for the pooled features
        N = x.shape[0]
        a = m.asap
        ei_l, _ = PA.add_remaining_self_loops(ei, None, 1.0, N)
        i, j = ei_l[0], ei_l[1]
        xd = x.double()
        deg = torch.zeros(N, dtype=torch.float64, device=_dev()).index_add_(0, j, torch.ones_like(j, dtype=torch.float64))
        dis = deg.pow(-0.5)
        h = xd @ a.gnn_intra_cluster.lin.weight.double().t()
        x_pool = torch.zeros_like(h).index_add_(0, j, (dis[i] * dis[j]).view(-1, 1) * h[i]) + a.gnn_intra_cluster.bias.double()
        X_q = torch.full((N, 200), float("-inf"), dtype=torch.float64, device=_dev()).scatter_reduce(
            0, i.view(-1, 1).expand(-1, 200), x_pool[j], reduce="amax", include_self=True)
        M_q = X_q @ a.lin_q.weight.double().t() + a.lin_q.bias.double()
        sc = torch.cat([M_q[i], x_pool[j]], dim=-1) @ a.gat_att.weight.double().t() + a.gat_att.bias.double()
        sc = PA.segment_softmax(torch.nn.functional.leaky_relu(sc, a.negative_slope), i, N)
        outn = torch.zeros_like(xd).index_add_(0, i, xd[j] * sc)
        nl = i != j
        g_ = a.gnn_score
        hh = outn @ g_.weight.double()
        degl = torch.zeros(N, dtype=torch.float64, device=_dev()).index_add_(0, i[nl], torch.ones(int(nl.sum()), dtype=torch.float64, device=_dev()))
        aggr = torch.zeros(N, 1, dtype=torch.float64, device=_dev()).index_add_(0, i[nl], hh[j[nl]])
        fit = torch.sigmoid(degl.view(-1, 1) * (outn @ g_.lin1.weight.double().t() + g_.lin1.bias.double()) + aggr
                            + (outn @ g_.lin2.weight.double().t() + g_.lin2.bias.double())).view(-1)
        got_fit = fit[perm]
        same_graph = b2[1:] == b2[:-1]
        assert (got_fit[1:][same_graph] <= got_fit[:-1][same_graph] + 1e-6).all()           # descending inside each graph
        # nothing outside the selection beats the weakest selected node of its graph (up to fp32 rounding of the fitness)
        sel = torch.zeros(N, dtype=torch.bool, device=_dev())
        sel[perm] = True
        for b in range(4):
            inb = batch == b
            if (~sel & inb).any():
                assert fit[~sel & inb].max().item() <= fit[sel & inb].min().item() + 1e-5
        ref_x = (outn[perm] * fit[perm].view(-1, 1))
        assert (xp.double() - ref_x).abs().max().item() < 1e-4 * max(1.0, ref_x.abs().max().item())
        # ---- pooled connectivity
        kN = perm.numel()
        assert int(ei2.min()) >= 0 and int(ei2.max()) < kN
        assert (b2[ei2[0]] == b2[ei2[1]]).all()                                               # E = S^T A S is block-diagonal
        key = ei2[0] * kN + ei2[1]
        assert key.unique().numel() == key.numel()                                            # coalesced: no duplicate pair
        loops = ei2[0] == ei2[1]
        assert int(loops.sum()) == kN and (ew2[loops] == 1.0).all()                           # one unit self loop per pooled node
        assert torch.isfinite(ew2).all() and (ew2 >= 0).all()


def test_loader_batches_feed_every_model_family():
    """A GraphBatchLoader batch carries its assembled HEAT plan; everything else a model may derive from the edges (HGT's
    per-relation-source plan, the homogeneous view of NTPoolGCN / HGTASAP, batch(), save_graph) must see the SAME edges as
    graph.batch() of the stored graphs — round 1 handed those consumers an edge-less graph."""
    import wsi_hgnn_amd as W
    from wsi_hgnn_amd import models, synthetic
    from wsi_hgnn_amd.data import GraphBatchLoader
    from wsi_hgnn_amd.graph import to_homogeneous
    import torch.nn.functional as F
    gs = [synthetic.hetero_graph(300, 64, seed=40 + i, dst_mode="hub", edges_per_dst=3) for i in range(4)]
    loader = GraphBatchLoader(gs, [0, 1, 0, 1], 4, "cuda", shuffle=False, resident=True)
    (Gl, yl), = list(loader)
    Gd = W.batch(gs).to(_dev())
    assert Gl.to("cuda") is Gl and Gl.to(_dev()) is Gl                    # 'cuda' == 'cuda:0': no rebuild, caches kept
    assert Gl.plan().num_edges == Gd.plan().num_edges == Gl.plan(per_relation_src=True).num_edges
    for r in Gd.canonical_etypes:
        assert torch.equal(Gl.edges(r)[0], Gd.edges(r)[0]) and torch.equal(Gl.edges(r)[1], Gd.edges(r)[1])
        assert torch.equal(Gl.edata["sim"][r], Gd.edata["sim"][r])
    assert to_homogeneous(Gl).num_edges() == to_homogeneous(Gd).num_edges() == Gd.num_edges()
    for G_ in (Gl, Gd):          # '_ID' of the homogeneous node table (graph_constructor.py:285-303), identity here
        off = G_.type_offsets()
        G_.ndata["_ID"] = {t: off[i] + torch.arange(G_.num_nodes(t), device=_dev()) for i, t in enumerate(G_.ntypes)}
    torch.manual_seed(3)
    nets = [models.HGT(ND, HGT_ED, 64, 64, 2, 2, 4).to(_dev()).eval(),
            models.HGTASAP(ND, HGT_ED, 64, 64, 2, 2, 4).to(_dev()).eval(),
            models.NTPoolGCN(64, 32, 2, ND, 2, F.relu, 0.0, "mean").to(_dev()).eval(),
            models.HEATNet4(64, 128, 2, 2, 4, ND, 0.0, "mean").to(_dev()).eval()]
    for net in nets:
        a = net(Gl)
        b = net(Gd)
        assert a.abs().max().item() > 0
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()), type(net).__name__
    # a later edit of the per-relation field reaches the kernels (no stale CSR copy)
    r0 = Gd.canonical_etypes[0]
    heat = nets[3]
    base = heat(Gd)
    Gd.edata["sim"][r0].mul_(-1.0)
    changed = heat(Gd)
    assert (changed - base).abs().max().item() > 0
    Gl.edata["sim"][r0].mul_(-1.0)
    assert (heat(Gl) - changed).abs().max().item() <= 2e-5 * max(1.0, changed.abs().max().item())


def test_dead_parameter_names_match_autograd():
    """dist.GradBucket.from_model leaves out exactly the parameters no forward reaches: for every model family the declared
    dead set equals the set of parameters without a gradient after a full-schema backward."""
    import torch.nn.functional as F
    import wsi_hgnn_amd as W
    from wsi_hgnn_amd import models, synthetic
    from wsi_hgnn_amd.dist import GradBucket
    g = W.batch([synthetic.hetero_graph(200, 32, seed=9 + i, dst_mode="hub") for i in range(2)]).to(_dev())
    hg = synthetic.homogeneous_graph(200, 32, seed=5).to(_dev())
    etypes = {r: str(i) for i, r in enumerate(g.canonical_etypes)}
    ed = {r: i for i, r in enumerate(g.canonical_etypes)}
    cases = [(models.HEATNet4(32, 64, 2, 2, 4, ND, 0.0, "mean"), g), (models.HEATNet2(32, 64, 2, 2, 4, ND, 0.0, "att"), g),
             (models.HGT(ND, ed, 32, 64, 2, 2, 4), g), (models.HGTASAP(ND, ed, 32, 64, 2, 2, 4), g),
             (models.HeteroRGCN(32, 64, 2, 2, etypes, ND), g), (models.GCN(32, 64, 2, 2, F.relu, 0.0, "mean"), hg)]
    for net, graph in cases:
        net = net.to(_dev()).eval()
        net(graph).sum().backward()
        none = {k for k, p in net.named_parameters() if p.grad is None}
        assert none == set(net.dead_parameter_names()), (type(net).__name__, none ^ set(net.dead_parameter_names()))
        b = GradBucket.from_model(net)
        assert len(b.params) == sum(1 for _ in net.parameters()) - len(none)


def test_locality_order_changes_the_schedule_not_the_result():
    """apply_locality_order (RCM node ids + '_pos') switches the attention kernels to position-ordered, XCD-contiguous
    processing: logits and gradients are those of the unordered graphs (permutation equivariance; fp32 summation order only),
    directly batched and through the loader."""
    import wsi_hgnn_amd as W
    from wsi_hgnn_amd import models, synthetic
    from wsi_hgnn_amd.data import GraphBatchLoader
    torch.manual_seed(611)
    m = models.HEATNet4(64, 128, 2, 2, 4, ND, 0.0, "mean").to(_dev())
    raw = [synthetic.hetero_graph(700, 64, seed=80 + i, dst_mode="hub") for i in range(3)]
    ordered = [W.apply_locality_order(g) for g in raw]
    Ga, Gb = W.batch(raw).to(_dev()), W.batch(ordered).to(_dev())
    assert Gb.plan().locality and not Ga.plan().locality
    y = torch.tensor([0, 1, 1], device=_dev())
    outs, grads = [], []
    for G in (Ga, Gb):
        m.zero_grad(set_to_none=True)
        o = m(G)
        torch.nn.functional.cross_entropy(o, y).backward()
        outs.append(o.detach())
        grads.append(torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]))
    assert (outs[0] - outs[1]).abs().max().item() < 2e-5
    assert (grads[0] - grads[1]).abs().max().item() <= 1e-6 + 1e-4 * grads[0].abs().max().item()
    loader = GraphBatchLoader(ordered, [0, 1, 1], 3, _dev(), shuffle=False, resident=True)
    (Gl, yl), = list(loader)
    assert Gl.plan().locality
    with torch.no_grad():
        assert (m(Gl) - outs[1]).abs().max().item() < 2e-5


def test_heatnet4_on_the_real_schema_batch():
    """The bench's --schema real workload at a small size: 6 node types, the union of the slides' (up to 72) relations with the
    missing ones present-but-empty (dgl.batch semantics), HEATNet4 vs the oracle."""
    import wsi_hgnn_amd as W
    from collections import OrderedDict
    from wsi_hgnn_amd import models, synthetic
    from oracle import models as OM
    nd6 = {str(i): i for i in range(6)}
    gs = [synthetic.real_schema_graph(300, 32, seed=5 + i, dst_mode="hub") for i in range(2)]
    rels = sorted({r for g in gs for r in g.canonical_etypes})
    empty = torch.empty(0, dtype=torch.int64)
    gs = [W.HeteroGraph.from_coo(OrderedDict((t, g.num_nodes(t)) for t in g.ntypes),
                                 OrderedDict((r, g.edges(r) if r in g.canonical_etypes else (empty, empty)) for r in rels),
                                 feat={t: g.nodes[t].data["feat"] for t in g.ntypes},
                                 sim={r: (g.edata["sim"][r] if r in g.canonical_etypes else torch.empty(0)) for r in rels}) for g in gs]
    G = W.batch(gs)
    torch.manual_seed(611)
    args = (32, 64, 2, 2, 4, nd6, 0.0, "mean")
    m = models.HEATNet4(*args).to(_dev())
    o = _oracle_copy(m, OM.HEATNet4, *args)
    y = torch.tensor([1, 0])
    out = m(G.to(_dev()))
    loss = torch.nn.functional.cross_entropy(out, y.to(_dev()))
    loss.backward()
    ref = o(G)
    rloss = torch.nn.functional.cross_entropy(ref, y)
    rloss.backward()
    assert (out.detach().cpu() - ref.detach()).abs().max().item() < 1e-4 and abs(loss.item() - rloss.item()) < 1e-4
    og = dict(o.named_parameters())
    for k, p in m.named_parameters():
        if og[k].grad is not None:
            assert (p.grad.cpu() - og[k].grad).abs().max().item() <= 1e-7 + 1e-4 * og[k].grad.abs().max().item(), k


def test_loader_hands_over_the_feature_row_scales():
    """fp16x3 / auto: a resident GraphBatchLoader concatenates the stored graphs' own feature row scales (each scanned once) into
    the batch's table; they are exactly what a scan of the assembled feature table finds, the input projection picks them up, and the
    model's output on the loader batch is bit-identical to the output on graph.batch() of the same slides."""
    import wsi_hgnn_amd as W
    from wsi_hgnn_amd import models, synthetic, ops
    from wsi_hgnn_amd.data import GraphBatchLoader
    gs = [synthetic.hetero_graph(3000, 512, seed=70 + i, dst_mode="hub") for i in range(4)]
    torch.manual_seed(5)
    net = models.HEATNet4(512, 512, 2, 2, 4, ND, 0.0, "mean").to(_dev()).eval()
    try:
        ops.set_gemm_precision("fp16x3")
        loader = GraphBatchLoader(gs, [0, 1, 0, 1], 4, "cuda", shuffle=False, resident=True)
        (Gl, _), = list(loader)
        x = Gl.cat_ndata("feat")
        cached = ops.row_scales_of(x)                  # attached to the batch's feature table by the loader
        assert cached is not None and torch.equal(cached, ops.row_absmax(x))
        with torch.no_grad():
            ops.EXCHANGE_STATS["row_scale_hits"] = 0
            a = net(Gl)
            assert ops.EXCHANGE_STATS["row_scale_hits"] > 0
            b = net(W.batch(gs).to(_dev()))
        assert torch.equal(a, b)
    finally:
        ops.set_gemm_precision("fp32")
