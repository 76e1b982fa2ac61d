"""NTPoolGCN — drop-in for the reference's ``models/GCN_NTPool.py:16-123``: a homogeneous GCN over ALL nodes of the
heterogeneous graph (``dgl.to_homogeneous`` + ``dgl.add_self_loop``, :90-91) whose per-layer readout is done PER NODE
TYPE through the ``'_ID'`` index maps (``alloc_features``, :76-87) — the only real "node-type pooling" in the
reference (``pooling/nt_pooling.py`` is an empty stub).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from ..graph import to_homogeneous
from ..pooling.readout import all_types_plan
from ..pooling import GlobalAttentionPooling
from .GCN import GraphConv
from .heat_net import make_pool


class NTPoolGCN(nn.Module):
    def __init__(self, in_dim, hidden_dim, out_dim, node_dict, n_layers, activation, dropout, graph_pooling_type="att"):
        super().__init__()
        self.in_feats = in_dim
        self.n_layers = n_layers
        self.layers = nn.ModuleList()
        self.node_dict = node_dict
        self.num_node_types = len(node_dict)
        self.layers.append(GraphConv(in_dim, hidden_dim, activation=activation))
        for _ in range(n_layers - 1):
            self.layers.append(GraphConv(hidden_dim, hidden_dim, activation=activation))
        self.dropout = nn.Dropout(p=dropout)
        self.classify = nn.Linear(hidden_dim, out_dim)
        self.linears_prediction = nn.ModuleDict({k: nn.ModuleList() for k in node_dict})
        self.pools = nn.ModuleList()
        for layer in range(n_layers + 1):
            for k in self.linears_prediction:
                self.linears_prediction[k].append(nn.Linear(in_dim if layer == 0 else hidden_dim, out_dim))
            self.pools.append(make_pool(graph_pooling_type, layer, in_dim, hidden_dim))

    def alloc_features(self, g, h):
        """GCN_NTPool.py:76-87: h_dict[k] = h[g.ndata['_ID'][k]] (rows of the homogeneous table picked by the stored ids)."""
        ids = g.ndata["_ID"]
        if not isinstance(ids, dict):
            ids = {g.ntypes[0]: ids}
        return {k: h.index_select(0, v.reshape(-1).to(h.device)) for k, v in ids.items()}

    def forward(self, g):
        """GCN_NTPool.py:89-123.  Per layer, BEFORE the layer is applied: readout of the current homogeneous states per
        (node type, graph), one Linear per node type, everything summed and finally divided by the number of terms.
        The reference does this with a Python loop over node types; here the rows of all types are gathered once into
        type-major order (the concatenation of ``alloc_features``' dict), reduced by one segmented kernel and projected by
        one grouped GEMM."""
        homo = to_homogeneous(g, add_self_loop=True)                                     # :90-91
        x = homo.ndata["feat"].to(torch.float32)
        dev = x.device
        ntypes = g.ntypes
        B, T = g.batch_size, len(ntypes)
        cache = g.__dict__.setdefault("_ntpool_cache", {})
        key = ("ids", str(dev))
        if key not in cache:
            ids = g.ndata["_ID"]
            if not isinstance(ids, dict):
                ids = {ntypes[0]: ids}
            cache[key] = torch.cat([ids[t].reshape(-1).to(dev) for t in ntypes])
        rows_of = cache[key]
        present = [g.num_nodes(t) > 0 for t in ntypes]                                   # :102 h[k].shape[0] > 0
        total, terms = 0, 0
        for i, conv in enumerate(self.layers):                                           # :95-109
            if i != 0:
                x = self.dropout(x)
            rows = x.index_select(0, rows_of)                                            # == cat(alloc_features(g, x).values())
            pool = self.pools[i]
            if isinstance(pool, GlobalAttentionPooling):
                off, parts = 0, []
                for t in ntypes:
                    n_t = g.num_nodes(t)
                    parts.append(pool(g, rows[off:off + n_t], ntype=t))
                    off += n_t
                pooled = torch.cat(parts, dim=0)
            else:
                pooled = ops.segment_reduce(rows, all_types_plan(g, dev), pool.op)       # [T*B, F]
            skey = ("spec", B, i)
            if skey not in cache:
                width = self.linears_prediction[ntypes[0]][i].weight.shape[0]
                cache[skey] = ops.LinearSpec([(j * B, (j + 1) * B) for j in range(T)], [0] * T, width, T * B)
            out = ops.grouped_linear(pooled, cache[skey], [self.linears_prediction[t][i].weight for t in ntypes],
                                     [self.linears_prediction[t][i].bias for t in ntypes])
            for j in range(T):                                                           # :116-121
                if present[j]:
                    total = total + out[j * B:(j + 1) * B]
                    terms += 1
            x = conv(homo, x)
        return total / terms
