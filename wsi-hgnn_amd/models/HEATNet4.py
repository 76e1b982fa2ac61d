"""HEATNet4 — drop-in for the reference's ``models/HEATNet4.py:141-247`` on the MI355X kernels.

Constructor signature (including the ``dropuout`` spelling), ``forward(G, h=None) -> [B, out_dim]``,
``n_layers`` and every ``state_dict`` key of the reference are kept.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .heat_layer import HEATLayer
from .heat_net import HEATTrunk, make_pool


class _IdentityZeroGrad(torch.autograd.Function):
    """y = l (no kernel); backward: dl = dy, dw = 0 — what `softmax` over a length-1 axis followed by `a * l` amounts to."""

    @staticmethod
    def forward(ctx, l, w):
        ctx.save_for_backward(w)
        return l.view_as(l)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        return g, torch.zeros_like(w)


class LinearAttentionBlock(nn.Module):
    """models/HEATNet4.py:20-42.  For [N,C] inputs the softmax is over a length-1 axis, so the block
    returns ``l`` unchanged and ``op.weight`` gets an exactly-zero gradient; this class
    keeps the parameter (state_dict key ``attn.{k}.op.weight``) and reproduces that behaviour
    without launching the dead Conv1d/softmax."""

    def __init__(self, in_features, normalize_attn=True):
        super().__init__()
        self.normalize_attn = normalize_attn
        self.op = nn.Conv1d(in_channels=in_features, out_channels=1, kernel_size=1, padding=0, bias=False)

    def forward(self, l, g):
        if not self.normalize_attn:
            a = torch.sigmoid(((l + g) * self.op.weight.view(1, -1)).sum(dim=1, keepdim=True))
            return a * l
        return _IdentityZeroGrad.apply(l, self.op.weight)   # identity; zero (not None) gradient for op.weight


class HEATNet4(HEATTrunk):
    def __init__(self, in_dim, hidden_dim, out_dim, n_layers, n_heads, node_dict, dropuout, graph_pooling_type='mean'):
        super().__init__()
        self.node_dict = node_dict
        self.gcs = nn.ModuleList()
        self.n_inp, self.n_hid, self.n_out = in_dim, hidden_dim, out_dim
        self.n_layers, self.n_heads = n_layers, n_heads
        self.adapt_ws = nn.ModuleList()
        self.pools = nn.ModuleList()
        self.linears_prediction = nn.ModuleDict({k: nn.Linear(hidden_dim, 256) for k in node_dict})
        for _ in range(len(node_dict)):
            self.adapt_ws.append(nn.Linear(in_dim, hidden_dim))
        for _ in range(n_layers):
            self.gcs.append(HEATLayer(hidden_dim, hidden_dim, node_dict, n_heads, dropuout))
        self.attn = nn.ModuleDict({a: LinearAttentionBlock(in_features=256, normalize_attn=True) for a in node_dict})
        for layer in range(n_layers + 1):
            self.pools.append(make_pool(graph_pooling_type, layer, in_dim, hidden_dim))
        self.head_2 = nn.Linear(256 * len(node_dict), 256)
        self.head_1 = nn.Linear(256, 64)
        self.head = nn.Linear(64, out_dim)

    def forward(self, G, h=None):
        ctx, hcat, out, B = self.encode(G, h)
        present = [(b - a) > 0 for (a, b) in ctx.rows]                       # HEATNet4.py:218,230 h[k].shape[0] > 0
        hg = 0
        for i, t in enumerate(ctx.ntypes):                                   # :229-232
            if present[i]:
                hg = hg + out[i * B:(i + 1) * B]
        parts = []
        for i, t in enumerate(ctx.ntypes):                                   # :236-240
            if present[i]:
                parts.append(self.attn[t](out[i * B:(i + 1) * B], hg))
            else:
                parts.append(torch.zeros(B, 256, dtype=out.dtype, device=out.device))
        g = torch.cat(parts, dim=1)                                          # :242
        g = ops.linear(g, self.head_2.weight, self.head_2.bias)              # :243
        g = ops.linear(g, self.head_1.weight, self.head_1.bias)              # :244
        return ops.linear(g, self.head.weight, self.head.bias)               # :245
