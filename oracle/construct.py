"""CPU restatement of the reference's graph-construction edge step — TEST INFRASTRUCTURE ONLY (never imported by the
product; see oracle/__init__.py).

Follows construct_graph/graph_constructor.py:256-303:
  :263      a = np.repeat(range(n_patches), radius - 1)
  :264-270  b = knn_model.query(features[v], topn=radius)[1:]  for every patch v
  :276-282  corr = scipy.stats.pearsonr(features[a], features[b])[0]; edge_type = 1 if corr > 0 else 0; edge_sim = corr
  :292-296  dgl.to_heterogeneous(graph, ['0'..], ['neg', 'pos'])

PARITY UNPINNED for the neighbour search: the reference queries an nmslib HNSW index (graph_constructor.py:43-81,
M=16, efConstruction=400, ef=90; nmslib is a third-party wheel, absent here, no version pinned), which is an APPROXIMATE
nearest-neighbour structure — its output is not a function of the input alone (insertion order, level draws).  What it
approximates is restated here exactly: brute-force L2 in float64, neighbours ascending by (distance, index), the query point
itself dropped.  scipy IS present, so the Pearson step calls the very function the reference calls.
"""
from collections import OrderedDict

import numpy as np
from scipy.stats import pearsonr


def knn_bruteforce(features: np.ndarray, radius: int):
    """b [N, radius-1]: the radius-1 nearest other rows under L2 (float64), ties -> smaller index; and their distances^2."""
    x = np.asarray(features, dtype=np.float64)
    n = x.shape[0]
    keep = radius - 1
    nbr = np.empty((n, keep), dtype=np.int64)
    d2o = np.empty((n, keep), dtype=np.float64)
    for i in range(n):
        d2 = ((x - x[i]) ** 2).sum(1)
        d2[i] = -1.0                                   # the query itself is the first hit the reference drops ([1:])
        order = np.lexsort((np.arange(n), d2))[1:keep + 1]
        nbr[i] = order
        d2o[i] = d2[order]
    return nbr, d2o


def edge_lists(features: np.ndarray, radius: int):
    """(a, b, edge_type, edge_sim) exactly as graph_constructor.py:263-282 builds them (pearsonr on the float32 rows)."""
    n = features.shape[0]
    nbr, _ = knn_bruteforce(features, radius)
    a = np.repeat(range(n), radius - 1)
    b = nbr.reshape(-1)
    edge_type, edge_sim = [], []
    for ia, ib in zip(a, b):
        corr = pearsonr(features[ia], features[ib])[0]
        edge_type.append(1 if corr > 0 else 0)
        edge_sim.append(corr)
    return a, b, np.asarray(edge_type), np.asarray(edge_sim, dtype=np.float64)


def to_heterogeneous(n, a, b, node_type, edge_type, ntypes, etypes):
    """Per-type renumbering in increasing homogeneous id, one relation per occurring (src type, etype, dst type) triple,
    edges in original order; relations ordered lexicographically by type ids (documented DGL behaviour, unverifiable here)."""
    node_type = np.asarray(node_type)
    local = np.empty(n, dtype=np.int64)
    ids = OrderedDict()
    for ti, t in enumerate(ntypes):
        m = np.nonzero(node_type == ti)[0]
        local[m] = np.arange(m.size)
        ids[t] = m
    rels = OrderedDict()
    triples = sorted({(int(node_type[s]), int(e), int(node_type[d])) for s, d, e in zip(a, b, edge_type)})
    for (si, ei, di) in triples:
        m = np.nonzero((node_type[a] == si) & (np.asarray(edge_type) == ei) & (node_type[b] == di))[0]
        rels[(ntypes[si], etypes[ei], ntypes[di])] = (local[a[m]], local[b[m]], m)
    return ids, rels
