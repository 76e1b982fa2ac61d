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


def test_config2_heatnet2_5k_nodes_vs_oracle():
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


def test_config3_heatnet4_10k_nodes_vs_oracle_single_graph():
    """configs[2] shape, one graph: HEATNet4(1024,512,2 layers,4 heads) on a 10k-node / 80k-edge graph vs the oracle."""
    from wsi_hgnn_amd import models, synthetic
    from oracle import models as OM
    torch.manual_seed(611)
    args = (1024, 512, 2, 2, 4, ND, 0.0, "mean")
    m = models.HEATNet4(*args).to(_dev())
    o = _oracle_copy(m, OM.HEATNet4, *args)
    g = synthetic.hetero_graph(10000, 1024, seed=612, dst_mode="uniform")
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


def test_config5_hgt_20k_nodes_vs_oracle():
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
