"""Graph construction edge step on the GPU — the producer of the hot path's input.

Mirrors ``GraphConstructor.construct_graph`` (construct_graph/graph_constructor.py:256-303) from the point where the
per-patch features and node types exist (the CNN encoders / HoVer-Net / openslide tiling upstream are out of scope):

* reference :265-273  ``Hnsw(space='l2').fit(features)`` + one ``knnQuery(features[v], k=radius)`` per patch, first hit (the
  patch itself) dropped: ``a = repeat(range(N), radius-1)``, ``b = neighbours``; edge ``a -> b``;
* reference :276-282  ``scipy.stats.pearsonr(features[a], features[b])[0]`` in a Python loop over all E pairs; edge type
  ``1 if corr > 0 else 0`` (names ``['neg', 'pos']``), ``edata['sim'] = corr``;
* reference :285-301  ``dgl.to_heterogeneous(graph, ['0'..str(T-1)], ['neg', 'pos'])`` and a homogeneous twin.

Here: ``X X^T`` row blocks on the matrix cores (the grouped GEMM of the hot path), a streaming top-(k+pad) shortlist per row
(``wsi_knn_select``) and one gather kernel that computes, for the shortlisted pairs, the EXACT squared distance and the
Pearson correlation and keeps the ``radius-1`` nearest (``wsi_pair_stats``).  HNSW is an approximate index; this is the
exact k-NN it approximates (recall 1.0 by construction), ties broken towards the smaller index.
There is no CPU fallback: the kernels live in libwsi_hgnn.so.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional, Tuple

import torch

from . import _native as N
from . import ops
from .graph import HeteroGraph

EDGE_TYPE_NAMES = ["neg", "pos"]          # graph_constructor.py:295: index = (corr > 0)


def knn_pearson(features: torch.Tensor, radius: int, pad: int = 8, block_rows: Optional[int] = None
                ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """For every row of ``features`` [N, F] (fp32, CUDA) its ``radius - 1`` nearest OTHER rows under L2.

    Returns ``(nbr [N, radius-1] int64, corr [N, radius-1] float32, dist2 [N, radius-1] float32)``, neighbours ascending by
    (distance, index).  ``pad`` extra candidates are shortlisted from the GEMM form of the distance and re-ranked exactly.
    """
    N.require_cuda(features)
    if features.dim() != 2 or features.dtype != torch.float32:
        raise ValueError("features must be a 2-D float32 tensor")
    x = features.contiguous()
    n, F = x.shape
    keep = int(radius) - 1
    if keep < 1:
        raise ValueError("radius must be >= 2 (the first neighbour is the patch itself and is dropped)")
    if n < radius:
        raise ValueError(f"{n} patches cannot have {keep} distinct neighbours each (the reference's np.fromiter fails here too)")
    kc = min(keep + int(pad), n - 1, 32)
    if kc < keep:
        raise ValueError(f"radius - 1 = {keep} exceeds the 32-candidate limit of wsi_knn_select")
    lib = N.load()
    dev = x.device
    st = N.stream()
    sqn = torch.empty(n, dtype=torch.float32, device=dev)
    N.check(lib.wsi_row_sqnorm(N.ptr(x), F, n, F, N.ptr(sqn), st), "wsi_row_sqnorm")
    if block_rows is None:      # <= 1 GiB of dot products in flight
        block_rows = max(128, min(n, (1 << 28) // max(n, 1)))
    cand = torch.empty((n, kc), dtype=torch.int32, device=dev)
    dots = torch.empty((min(block_rows, n), n), dtype=torch.float32, device=dev)
    for r0 in range(0, n, block_rows):
        rows = min(block_rows, n - r0)
        ops._gemm(N.WSI_GEMM_NT, 0, [dict(A=N.ptr(x, r0 * F * 4), lda=F, B=N.ptr(x), ldb=F, C=N.ptr(dots), ldc=n,
                                          M=rows, N=n, K=F)], dev)
        N.check(lib.wsi_knn_select(N.ptr(dots), n, N.ptr(sqn), r0, rows, n, kc, N.ptr(cand, r0 * kc * 4), st), "wsi_knn_select")
    nbr = torch.full((n, keep), -1, dtype=torch.int32, device=dev)
    dist2 = torch.empty((n, keep), dtype=torch.float32, device=dev)
    corr = torch.empty((n, keep), dtype=torch.float32, device=dev)
    N.check(lib.wsi_pair_stats(N.ptr(x), F, n, F, N.ptr(cand), kc, keep, N.ptr(nbr), N.ptr(dist2), N.ptr(corr), st),
            "wsi_pair_stats")
    return nbr.long(), corr, dist2


def to_heterogeneous(num_nodes: int, src: torch.Tensor, dst: torch.Tensor, node_type: torch.Tensor, edge_type: torch.Tensor,
                     ntypes, etypes, feat: Optional[torch.Tensor] = None, sim: Optional[torch.Tensor] = None) -> HeteroGraph:
    """``dgl.to_heterogeneous`` as graph_constructor.py:292-296 uses it: nodes of each type are renumbered in increasing
    homogeneous id (``ndata['_ID']`` keeps the original id), one relation per (source type, edge type, destination type)
    triple that occurs, edges in their original order.  Relations are ordered lexicographically by (source type id, edge
    type id, destination type id) — DGL's own order is not verifiable here (third-party, absent); no model on the path
    depends on it beyond the summation order of the cross-relation mean."""
    dev = src.device
    node_type = node_type.to(dev).long()
    edge_type = edge_type.to(dev).long()
    T = len(ntypes)
    order = torch.argsort(node_type, stable=True)
    counts = torch.bincount(node_type, minlength=T)
    starts = torch.cumsum(counts, 0) - counts
    local = torch.empty(num_nodes, dtype=torch.int64, device=dev)
    local[order] = torch.arange(num_nodes, device=dev) - starts[node_type[order]]
    counts_h = counts.tolist()
    nn_ = OrderedDict((str(t), counts_h[i]) for i, t in enumerate(ntypes))
    st, dt = node_type[src], node_type[dst]
    key = (st * len(etypes) + edge_type) * T + dt
    present = torch.unique(key).tolist()
    edges, sims = OrderedDict(), {}
    for kk in present:
        m = (key == kk).nonzero(as_tuple=True)[0]
        s_id, rest = divmod(kk, len(etypes) * T)
        e_id, d_id = divmod(rest, T)
        r = (str(ntypes[s_id]), str(etypes[e_id]), str(ntypes[d_id]))
        edges[r] = (local[src[m]], local[dst[m]])
        if sim is not None:
            sims[r] = sim[m]
    g = HeteroGraph.from_coo(nn_, edges, sim=sims if sim is not None else None)
    off = 0
    for i, t in enumerate(ntypes):
        ids = order[off:off + counts_h[i]]
        off += counts_h[i]
        g.nodes[str(t)].data["_ID"] = ids
        if feat is not None:
            g.nodes[str(t)].data["feat"] = feat[ids]
    return g


def construct_graph(features: torch.Tensor, node_type, radius: int, n_node_type: int, pad: int = 8):
    """``GraphConstructor.construct_graph`` (graph_constructor.py:256-303) from (features, node types) on:
    returns ``(het_graph, homo_graph, node_type)`` like the reference.  ``features``: [N, F] fp32 on the GPU;
    ``node_type``: N ints in [0, n_node_type)."""
    n = features.shape[0]
    dev = features.device
    nbr, corr, _ = knn_pearson(features, radius, pad)
    keep = nbr.shape[1]
    a = torch.arange(n, device=dev).repeat_interleave(keep, output_size=n * keep)      # :263 np.repeat(range(N), radius-1)
    b = nbr.reshape(-1)
    sim = corr.reshape(-1)
    etype = (sim > 0).long()                                                            # :279  1 if corr > 0 else 0
    nt = torch.as_tensor(node_type, dtype=torch.int64, device=dev)
    het = to_heterogeneous(n, a, b, nt, etype, [str(t) for t in range(n_node_type)], EDGE_TYPE_NAMES, feat=features, sim=sim)
    homo = HeteroGraph.homogeneous(n, a, b, feat=features)
    return het, homo, node_type
