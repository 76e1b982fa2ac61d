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
        g_homo = to_homogeneous(g, add_self_loop=True)                                   # :90-91
        h_homo = g_homo.ndata["feat"].to(torch.float32)
        B = g.batch_size
        ntypes = g.ntypes
        h_list = []
        for i, layer in enumerate(self.layers):                                          # :95-109
            if i != 0:
                h_homo = self.dropout(h_homo)
            h = self.alloc_features(g, h_homo)
            out_h = {}
            for k in ntypes:
                if h[k].shape[0] > 0:
                    pooled = self.pools[i](g, h, ntype=k)
                    lin = self.linears_prediction[k][i]
                    out_h[k] = ops.linear(pooled, lin.weight, lin.bias)
                else:
                    out_h[k] = h[k]
            h_list.append(out_h)
            h_homo = layer(g_homo, h_homo)
        hg, count = 0, 0
        for hh in h_list:                                                                # :116-121
            for nt in ntypes:
                if hh[nt].shape[0] > 0:
                    hg = hg + hh[nt]
                    count += 1
        return hg / count
