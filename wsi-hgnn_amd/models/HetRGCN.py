"""HeteroRGCN — drop-in for the reference's ``models/HetRGCN.py`` (layer :13-46, model :49-125).

The reference layer touches no edge: for every source type it averages ``W_rel(h_src)`` over the
relations leaving that type.  mean_r (h W_r^T + b_r) == h (mean_r W_r)^T + mean_r b_r, so the weights are averaged
first (tiny) and ONE grouped MFMA GEMM per layer does the work of R per-relation GEMMs + stack/mean.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .heat_layer import heat_context
from .heat_net import make_pool
from .HGT import _readout_sum_dead_parameters, _readout_sum_forward


class HeteroRGCNLayer(nn.Module):
    def __init__(self, in_size, out_size, etype_dict):
        super().__init__()
        self.etype_dict = etype_dict
        self.weight = nn.ModuleDict({name: nn.Linear(in_size, out_size) for name in etype_dict.values()})

    def forward_cat(self, G, hctx, h: torch.Tensor) -> torch.Tensor:
        per_src = {}
        for (s, e, d) in G.canonical_etypes:                                   # HetRGCN.py:25-37
            per_src.setdefault(s, []).append(self.weight[self.etype_dict[(s, e, d)]])
        types = [i for i, t in enumerate(hctx.ntypes) if t in per_src]
        if not types:
            return h
        ws, bs = [], []
        for i in types:
            lins = per_src[hctx.ntypes[i]]
            ws.append(torch.stack([l.weight for l in lins]).mean(0))            # == torch.stack(Wh).mean(0) of :43
            bs.append(torch.stack([l.bias for l in lins]).mean(0))
        spec = hctx.cache.get(("rgcn", tuple(types)))
        if spec is None:
            spec = hctx.cache[("rgcn", tuple(types))] = ops.LinearSpec([hctx.rows[i] for i in types], [0] * len(types),
                                                                       ws[0].shape[0], hctx.num_nodes)
        y = ops.grouped_linear(h, spec, ws, bs)
        if len(types) < len(hctx.ntypes):                                       # :40-41 types without outgoing relation keep their features
            keep = torch.zeros(hctx.num_nodes, 1, dtype=torch.bool, device=h.device)
            for i in types:
                a, b = hctx.rows[i]
                keep[a:b] = True
            y = torch.where(keep, y, h)
        return y

    def forward(self, G, feat_dict):
        dev = next(iter(feat_dict.values())).device
        node_dict = {t: i for i, t in enumerate(G.ntypes)}
        hctx = heat_context(G, node_dict, next(iter(self.weight.values())).out_features, dev)
        x = torch.cat([feat_dict[t] for t in hctx.ntypes], dim=0)
        out = self.forward_cat(G, hctx, x)
        return {t: out[a:b] for t, (a, b) in zip(hctx.ntypes, hctx.rows)}


class HeteroRGCN(nn.Module):
    def __init__(self, in_dim, hidden_dim, out_dim, n_layers, etypes, node_dict, graph_pooling_type="sum"):
        super().__init__()
        self.node_dict = node_dict
        self.n_layers = n_layers
        self.n_hid = hidden_dim
        self.adapt_ws = nn.ModuleList()
        for _ in range(len(node_dict)):
            self.adapt_ws.append(nn.Linear(in_dim, hidden_dim))
        self.layers = nn.ModuleList()
        for _ in range(n_layers):
            self.layers.append(HeteroRGCNLayer(hidden_dim, hidden_dim, etypes))
        self.out = nn.Linear(hidden_dim, out_dim)
        self.pools = nn.ModuleList()
        self.linears_prediction = nn.ModuleDict({k: nn.ModuleList() for k in node_dict})
        for layer in range(n_layers + 1):
            for k in self.linears_prediction:
                self.linears_prediction[k].append(nn.Linear(hidden_dim, out_dim))
            self.pools.append(make_pool(graph_pooling_type, layer, in_dim, hidden_dim))

    def dead_parameter_names(self):
        return _readout_sum_dead_parameters(self, "layers")

    def forward(self, G, h=None):
        return _readout_sum_forward(self, G, h, lambda i, hctx, x: self.layers[i].forward_cat(G, hctx, x))
