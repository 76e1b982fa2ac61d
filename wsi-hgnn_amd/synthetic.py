"""Synthetic WSI patch graphs with the shapes SURVEY.md §8d / BASELINE.md §3 prescribe.

Real graphs come from construct_graph/graph_constructor.py:256-303 (kNN in feature space with
``radius-1`` = 8 out-edges per node, Pearson-signed ``sim``, HoVer-Net node types); that
pipeline is out of scope, so benchmarks and tests use this generator: 3 node types split
50/30/20 %, 6 canonical relations (each type is the destination of exactly two), ``4*N_dst``
edges per relation (sum = 8*N), ``feat ~ U(0,1)``, ``sim ~ +U(0,1)`` for 'pos' and ``-U(0,1)``
for 'neg' relations.  Seeds follow ``611 + 1000*rank + graph_idx`` (611 = main.py:15).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List, Optional, Sequence, Tuple

import torch

from .graph import HeteroGraph, batch

HEAT_RELATIONS: List[Tuple[str, str, str]] = [
    ("0", "pos", "0"), ("1", "pos", "0"), ("0", "pos", "1"),
    ("2", "neg", "1"), ("1", "neg", "2"), ("2", "pos", "2"),
]


def hetero_graph(num_nodes: int = 10000, in_dim: int = 1024, seed: int = 611, dst_mode: str = "uniform",
                 fractions: Sequence[float] = (0.5, 0.3, 0.2), edges_per_dst: int = 4,
                 relations: Optional[List[Tuple[str, str, str]]] = None,
                 dtype: torch.dtype = torch.float32) -> HeteroGraph:
    """One synthetic heterogeneous patch graph (CPU tensors). ``dst_mode``: 'uniform' | 'hub'."""
    gen = torch.Generator().manual_seed(int(seed))
    relations = HEAT_RELATIONS if relations is None else relations
    ntypes = [str(i) for i in range(len(fractions))]
    counts = [int(round(num_nodes * f)) for f in fractions]
    counts[0] += num_nodes - sum(counts)
    nn_ = OrderedDict(zip(ntypes, counts))
    edges = OrderedDict()
    sim = {}
    for (s, e, d) in relations:
        ns, nd = nn_[s], nn_[d]
        ne = edges_per_dst * nd if (ns > 0 and nd > 0) else 0
        src = torch.randint(0, max(ns, 1), (ne,), generator=gen, dtype=torch.int64)
        if dst_mode == "uniform":
            dst = torch.randint(0, max(nd, 1), (ne,), generator=gen, dtype=torch.int64)
        elif dst_mode == "hub":
            u = torch.rand(ne, generator=gen, dtype=torch.float64)
            dst = torch.clamp((nd * u * u).floor().to(torch.int64), max=max(nd - 1, 0))
        else:
            raise ValueError(dst_mode)
        edges[(s, e, d)] = (src, dst)
        mag = torch.rand(ne, generator=gen, dtype=torch.float32)
        sim[(s, e, d)] = mag if e == "pos" else -mag
    feat = {t: torch.rand(nn_[t], in_dim, generator=gen, dtype=torch.float32).to(dtype) for t in ntypes}
    return HeteroGraph.from_coo(nn_, edges, feat=feat, sim=sim)


def hetero_batch(batch_size: int = 8, num_nodes: int = 10000, in_dim: int = 1024, rank: int = 0,
                 dst_mode: str = "uniform", **kw) -> Tuple[HeteroGraph, torch.Tensor]:
    """Block-diagonal batch of ``batch_size`` graphs + labels ~ U{0,1} (seed 611 + 1000*rank + idx)."""
    graphs = [hetero_graph(num_nodes, in_dim, seed=611 + 1000 * rank + i, dst_mode=dst_mode, **kw)
              for i in range(batch_size)]
    gen = torch.Generator().manual_seed(611 + 1000 * rank + 999)
    labels = torch.randint(0, 2, (batch_size,), generator=gen, dtype=torch.int64)
    return batch(graphs), labels


def homogeneous_graph(num_nodes: int = 2000, in_dim: int = 1024, out_edges: int = 8, seed: int = 611,
                      self_loops: bool = True) -> HeteroGraph:
    """BASELINE config 1: homogeneous patch graph, 8 out-edges per node (+ self loops, data.py:120-121)."""
    gen = torch.Generator().manual_seed(int(seed))
    src = torch.arange(num_nodes, dtype=torch.int64).repeat_interleave(out_edges)
    dst = torch.randint(0, num_nodes, (num_nodes * out_edges,), generator=gen, dtype=torch.int64)
    if self_loops:
        keep = src != dst
        loop = torch.arange(num_nodes, dtype=torch.int64)
        src = torch.cat([src[keep], loop])
        dst = torch.cat([dst[keep], loop])
    feat = torch.rand(num_nodes, in_dim, generator=gen, dtype=torch.float32)
    return HeteroGraph.homogeneous(num_nodes, src, dst, feat=feat)


REAL_TYPE_FRACTIONS = (0.34, 0.24, 0.18, 0.12, 0.08, 0.04)


def real_schema_graph(num_nodes: int = 10000, in_dim: int = 1024, seed: int = 611, n_types: int = 6, out_edges: int = 8,
                      p_pos: float = 0.7, dst_mode: str = "uniform",
                      fractions: Sequence[float] = REAL_TYPE_FRACTIONS) -> HeteroGraph:
    """A graph with the SCHEMA the reference's graph constructor emits (construct_graph/graph_constructor.py:276-303,
    configs/COAD/HEAT4_kimia_classification_v2.yml:44 ``n_node_types: 6``): node types '0'..'5' (HoVer-Net nucleus classes,
    skewed frequencies), every patch sends ``out_edges`` = radius-1 edges to other patches whatever their type, an edge is
    'pos' or 'neg' by the sign of its Pearson ``sim``, and ``dgl.to_heterogeneous`` keeps one canonical relation per
    (src type, sign, dst type) combination THAT OCCURS — up to 2*6*6 = 72 relations, 12 relation slots per destination node,
    many of them small.  Targets are random (``uniform`` or kNN-like ``hub`` skew), not feature-space neighbours."""
    gen = torch.Generator().manual_seed(int(seed))
    fr = torch.tensor(list(fractions[:n_types]), dtype=torch.float64)
    fr = fr / fr.sum()
    counts = [int(round(num_nodes * float(f))) for f in fr]
    counts[0] += num_nodes - sum(counts)
    ntypes = [str(i) for i in range(n_types)]
    off = [0]
    for c in counts:
        off.append(off[-1] + c)
    n = off[-1]
    type_of = torch.repeat_interleave(torch.arange(n_types), torch.tensor(counts))
    src = torch.arange(n, dtype=torch.int64).repeat_interleave(out_edges)
    if dst_mode == "uniform":
        dst = torch.randint(0, n, (n * out_edges,), generator=gen, dtype=torch.int64)
    elif dst_mode == "hub":
        u = torch.rand(n * out_edges, generator=gen, dtype=torch.float64)
        dst = torch.clamp((n * u * u).floor().to(torch.int64), max=n - 1)
        dst = torch.randperm(n, generator=gen)[dst]           # hubs spread over all types, not only the first ids
    else:
        raise ValueError(dst_mode)
    pos = torch.rand(n * out_edges, generator=gen) < p_pos
    mag = torch.rand(n * out_edges, generator=gen, dtype=torch.float32)
    ts, td = type_of[src], type_of[dst]
    offt = torch.tensor(off[:-1], dtype=torch.int64)
    edges, sim = OrderedDict(), {}
    for e, is_pos in (("neg", False), ("pos", True)):
        for s in range(n_types):
            for d in range(n_types):
                m = (ts == s) & (td == d) & (pos == is_pos)
                if not bool(m.any()):
                    continue                                  # to_heterogeneous only creates relations that occur
                r = (str(s), e, str(d))
                edges[r] = (src[m] - offt[s], dst[m] - offt[d])
                sim[r] = mag[m] if is_pos else -mag[m]
    feat = {t: torch.rand(counts[i], in_dim, generator=gen, dtype=torch.float32) for i, t in enumerate(ntypes)}
    return HeteroGraph.from_coo(OrderedDict(zip(ntypes, counts)), edges, feat=feat, sim=sim)


def real_schema_batch(batch_size: int = 8, num_nodes: int = 10000, in_dim: int = 1024, rank: int = 0, dst_mode: str = "uniform",
                      **kw) -> Tuple[HeteroGraph, torch.Tensor]:
    """Block-diagonal batch of ``real_schema_graph`` slides + labels.  Slides differ in which of the up to 72 relations occur and
    ``dgl.batch`` needs one schema, so the batch uses the UNION: a relation missing from a slide is an EMPTY relation of that slide -
    exactly what ``dgl.batch`` would hold after the reference's node-dropping transforms."""
    gs = [real_schema_graph(num_nodes, in_dim, seed=611 + 1000 * rank + i, dst_mode=dst_mode, **kw) for i in range(batch_size)]
    rels = sorted({r for g in gs for r in g.canonical_etypes})
    empty = torch.empty(0, dtype=torch.int64)
    gs = [HeteroGraph.from_coo(OrderedDict((t, g.num_nodes(t)) for t in g.ntypes),
                               OrderedDict((r, g.edges(r) if r in g.canonical_etypes else (empty, empty)) for r in rels),
                               feat={t: g.nodes[t].data["feat"] for t in g.ntypes},
                               sim={r: (g.edata["sim"][r] if r in g.canonical_etypes else torch.empty(0)) for r in rels}) for g in gs]
    labels = torch.randint(0, 2, (batch_size,), generator=torch.Generator().manual_seed(611 + 1000 * rank + 999))
    return batch(gs), labels


def knn_slide(num_nodes: int = 10000, in_dim: int = 1024, seed: int = 611, device="cuda", n_types: int = 3, clusters: int = 40,
              locality: bool = True):
    """A WSI-LIKE slide, built the way the reference builds its graphs (construct_graph/graph_constructor.py:256-303): patch
    features clustered in feature space (tissue types), every patch linked to its 8 nearest OTHER patches under L2 (radius 9),
    the edge typed 'pos' / 'neg' by the sign of the Pearson correlation of the two feature vectors - unlike ``hetero_graph``,
    whose sources and destinations are uniformly random.  Needs the GPU (``construct.construct_graph`` runs the kNN / Pearson
    kernels).  ``locality``: renumber the patches with ``graph.apply_locality_order`` (reverse Cuthill-McKee per slide), so that
    the attention kernels walk neighbourhoods XCD-contiguously.  Returns the graph on the CPU (like the other generators)."""
    from . import construct
    from .graph import apply_locality_order
    g = torch.Generator().manual_seed(seed)
    centres = torch.rand(clusters, in_dim, generator=g)
    x = (centres[torch.randint(0, clusters, (num_nodes,), generator=g)] + 0.12 * torch.randn(num_nodes, in_dim, generator=g)).clamp_(min=0).float()
    x[:, ::2] -= 0.4                                   # so that Pearson signs of both kinds occur
    nt = torch.randint(0, n_types, (num_nodes,), generator=g)
    het, _, _ = construct.construct_graph(x.to(device), nt.tolist(), 9, n_types)
    het = het.to("cpu")
    return apply_locality_order(het) if locality else het
