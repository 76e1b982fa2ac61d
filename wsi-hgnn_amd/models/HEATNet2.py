"""HEATNet2 — drop-in for the reference's ``models/HEATNet2.py:116-196`` on the MI355X kernels.

Same trunk as HEATNet4; the readout is the sum over node types of ``linears_prediction[k]`` applied
to the per-type mean/sum/max pooled states (HEATNet2.py:180-194).
"""
from __future__ import annotations

import torch.nn as nn

from .heat_layer import HEATLayer
from .heat_net import HEATTrunk, make_pool


class HEATNet2(HEATTrunk):
    def __init__(self, in_dim, hidden_dim, out_dim, n_layers, n_heads, node_dict, dropuout, graph_pooling_type='mean'):
        super().__init__()
        self.node_dict = node_dict
        self.gcs = nn.ModuleList()
        self.n_inp, self.n_hid, self.n_out = in_dim, hidden_dim, out_dim
        self.n_layers, self.n_heads = n_layers, n_heads
        self.adapt_ws = nn.ModuleList()
        self.pools = nn.ModuleList()
        self.linears_prediction = nn.ModuleDict({k: nn.Linear(hidden_dim, out_dim) for k in node_dict})
        for _ in range(len(node_dict)):
            self.adapt_ws.append(nn.Linear(in_dim, hidden_dim))
        for _ in range(n_layers):
            self.gcs.append(HEATLayer(hidden_dim, hidden_dim, node_dict, n_heads, dropuout))
        for layer in range(n_layers + 1):
            self.pools.append(make_pool(graph_pooling_type, layer, in_dim, hidden_dim))

    def forward(self, G, h=None):
        ctx, hcat, out, B = self.encode(G, h)
        hg = 0
        for i, (a, b) in enumerate(ctx.rows):                                # HEATNet2.py:189-194
            if b - a > 0:
                hg = hg + out[i * B:(i + 1) * B]
        return hg
