"""GCN — drop-in for the reference's ``models/GCN.py:15-79`` (DGL ``GraphConv(norm='both')`` + glob poolings).

GraphConv = degree-normalised neighbour sum (HIP ``wsi_spmm_sum``: CSR gather forward, CSC gather backward, bias and
ReLU fused) around a dense projection on the MFMA GEMM; as DGL does, the projection runs first when in > out and
after the aggregation otherwise.  ``weight`` is [in,out] (DGL layout) for state_dict parity.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops
from .heat_net import make_pool


class HomoPlan:
    """CSR/CSC + GraphConv degree norms of a homogeneous (single node type) graph."""

    def __init__(self, g):
        p = g.plan()
        self.rowptr, self.src, self.colptr, self.csc_dst = p.rowptr, p.src, p.colptr, p.csc_dst
        n = p.num_nodes
        indeg = (p.rowptr[1:n + 1] - p.rowptr[:n]).to(torch.float32)
        outdeg = (p.colptr[1:] - p.colptr[:-1]).to(torch.float32)
        self.in_norm = indeg.clamp(min=1).pow(-0.5).contiguous()
        self.out_norm = outdeg.clamp(min=1).pow(-0.5).contiguous()
        self.num_nodes = n


def homo_plan(g) -> HomoPlan:
    if "_homo_plan" not in g.__dict__:
        if len(g.ntypes) != 1 or len(g.canonical_etypes) != 1:
            raise ValueError("GraphConv needs a homogeneous graph (use wsi_hgnn_amd.graph.to_homogeneous)")
        g.__dict__["_homo_plan"] = HomoPlan(g)
    return g.__dict__["_homo_plan"]


class _MatmulNN(torch.autograd.Function):
    """y = x @ w with w stored [in,out] (DGL GraphConv layout) on the MFMA GEMM."""

    @staticmethod
    def forward(ctx, x, w):
        from .. import _native as N
        N.require_cuda(x, w)
        x = x.contiguous()
        M, K = x.shape
        y = torch.empty((M, w.shape[1]), dtype=torch.float32, device=x.device)
        ops._gemm(N.WSI_GEMM_NN, 0, [dict(A=N.ptr(x), lda=K, B=N.ptr(w), ldb=w.stride(0), C=N.ptr(y), ldc=w.shape[1],
                                          M=M, N=w.shape[1], K=K)], x.device)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        from .. import _native as N
        x, w = ctx.saved_tensors
        gy = gy.contiguous()
        M, K = x.shape
        n_out = w.shape[1]
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            ops._gemm(N.WSI_GEMM_NT, 0, [dict(A=N.ptr(gy), lda=n_out, B=N.ptr(w), ldb=w.stride(0), C=N.ptr(gx), ldc=K,
                                              M=M, N=K, K=n_out)], x.device)
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(w)
            ops._gemm(N.WSI_GEMM_TN, 0, [dict(A=N.ptr(x), lda=K, B=N.ptr(gy), ldb=n_out, C=N.ptr(gw), ldc=n_out,
                                              M=K, N=n_out, K=M)], x.device)
        return gx, gw


class GraphConv(nn.Module):
    """dgl.nn.pytorch.GraphConv(in, out, norm='both', weight=True, bias=True, activation) — parameters ``weight`` [in,out]
    (xavier-uniform) and ``bias`` (zeros), as DGL creates them."""

    def __init__(self, in_feats, out_feats, activation=None):
        super().__init__()
        self._in_feats, self._out_feats = in_feats, out_feats
        self.weight = nn.Parameter(torch.Tensor(in_feats, out_feats))
        self.bias = nn.Parameter(torch.Tensor(out_feats))
        nn.init.xavier_uniform_(self.weight)
        nn.init.zeros_(self.bias)
        self._activation = activation

    def forward(self, g, x):
        hp = homo_plan(g)
        relu = self._activation is not None and getattr(self._activation, "__name__", "") == "relu"
        if self._in_feats > self._out_feats:
            y = ops.graph_conv_aggregate(_MatmulNN.apply(x, self.weight), self.bias, hp, relu)
            fused_act = relu
        else:
            y = _MatmulNN.apply(ops.graph_conv_aggregate(x, None, hp, False), self.weight) + self.bias
            fused_act = False
        if self._activation is not None and not fused_act:
            y = self._activation(y)
        return y


class GCN(nn.Module):
    def __init__(self, in_dim, hidden_dim, out_dim, n_layers, activation, dropout, graph_pooling_type="att"):
        super().__init__()
        self.in_feats = in_dim
        self.n_layers = n_layers
        self.layers = nn.ModuleList()
        self.layers.append(GraphConv(in_dim, hidden_dim, activation=activation))
        for _ in range(n_layers - 1):
            self.layers.append(GraphConv(hidden_dim, hidden_dim, activation=activation))
        self.dropout = nn.Dropout(p=dropout)
        self.classify = nn.Linear(hidden_dim, out_dim)
        self.linears_prediction = nn.ModuleList()
        self.pools = nn.ModuleList()
        for layer in range(n_layers + 1):
            self.linears_prediction.append(nn.Linear(in_dim if layer == 0 else hidden_dim, out_dim))
            self.pools.append(make_pool(graph_pooling_type, layer, in_dim, hidden_dim))

    def dead_parameter_names(self):
        """``linears_prediction[n_layers]`` is created (GCN.py:41-62) and never applied (:75 uses ``classify``)."""
        return [n for n, _ in self.named_parameters() if n.startswith(f"linears_prediction.{self.n_layers}.")]

    def forward(self, g, h=None):
        if h is None:
            h = g.ndata["feat"]                                                         # GCN.py:65-66
        h = h.to(torch.float32)
        h_list = []
        for i, layer in enumerate(self.layers):                                         # :69-73
            if i != 0:
                h = self.dropout(h)
            p = self.pools[i](g, h)
            h_list.append(ops.linear(p, self.linears_prediction[i].weight, self.linears_prediction[i].bias))
            h = layer(g, h)
        p = self.pools[-1](g, h)
        h_list.append(ops.linear(p, self.classify.weight, self.classify.bias))          # :75
        return torch.stack(h_list).mean(0)                                              # :77
