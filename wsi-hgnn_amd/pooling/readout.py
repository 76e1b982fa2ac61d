"""Per-(graph, node type) readout on the segmented-reduction HIP kernel.

Mirrors pooling/avg_pooling.py:6-19 (``AvgPooling``), pooling/sum_pooling.py (``SumPooling``),
pooling/max_pooling.py (``MaxPooling``): ``forward(graph, feat, ntype=None)`` where ``feat`` is a
tensor (one node type) or a ``{ntype: tensor}`` dict, returning ``[B, D]`` over the graph's
``batch_num_nodes(ntype)`` segments (empty segment -> 0).  ``NTPooling`` is an empty stub in the
reference (pooling/nt_pooling.py:4-10, ``forward`` returns None); here it is the node-type readout
the models actually perform: every node type pooled at once, ``{ntype: [B, D]}``.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import nn

from .. import ops


def _rows_plan(graph, ntype: str, device, cache_key: str):
    cache = graph.__dict__.setdefault("_readout_plans", {})
    key = (cache_key, ntype, str(device))
    if key not in cache:
        bnn = graph.batch_num_nodes(ntype).tolist()
        ptr = [0]
        for c in bnn:
            ptr.append(ptr[-1] + int(c))
        cache[key] = ops.ReducePlan.from_ptr(ptr, device)
    return cache[key]


def all_types_plan(graph, device):
    """ReducePlan over the type-major concatenated node table: segment t*B+b = rows of (ntype t, graph b)."""
    cache = graph.__dict__.setdefault("_readout_plans", {})
    key = ("all", str(device))
    if key not in cache:
        ptr = [0]
        type_rows, type_segs = [], []
        for t in graph.ntypes:
            r0, s0 = ptr[-1], len(ptr) - 1
            for c in graph.batch_num_nodes(t).tolist():
                ptr.append(ptr[-1] + int(c))
            type_rows.append((r0, ptr[-1]))
            type_segs.append((s0, len(ptr) - 1))
        rp = ops.ReducePlan.from_ptr(ptr, device)
        rp.type_rows, rp.type_segments = type_rows, type_segs      # which segments are whose: an empty (type, graph) segment on a type boundary is not ambiguous
        cache[key] = rp
    return cache[key]


class _Readout(nn.Module):
    op = "mean"

    def forward(self, graph, feat, ntype: Optional[str] = None) -> torch.Tensor:
        if isinstance(feat, dict):
            if ntype is None:
                if len(feat) != 1:
                    raise ValueError("ntype is required when feat holds several node types")
                ntype = next(iter(feat))
            x = feat[ntype]
        else:
            x = feat
            if ntype is None:
                ntype = graph.ntypes[0]
        if x.dim() != 2:
            x = x.reshape(x.shape[0], -1)
        return ops.segment_reduce(x, _rows_plan(graph, ntype, x.device, "one"), self.op)


class AvgPooling(_Readout):
    op = "mean"


class SumPooling(_Readout):
    op = "sum"


class MaxPooling(_Readout):
    op = "max"


class NTPooling(nn.Module):
    """Node-type pooling: mean/sum/max readout of every node type in one launch.

    The reference's ``pooling/nt_pooling.py:4-10`` is an empty stub (``forward`` is ``pass`` and returns None) that
    no model calls — the per-node-type readout the north star names is done by ``NTPoolGCN.alloc_features`` + the
    ``*Pooling`` modules.  This class keeps the constructor/forward signature (``NTPooling()``, ``forward(g, h)``) and
    returns what such a module would have to return: ``{ntype: [B, F]}``."""

    def __init__(self, op: str = "mean"):
        super().__init__()
        self.op = op

    def forward(self, g, h) -> Dict[str, torch.Tensor]:
        if isinstance(h, dict):
            x = torch.cat([h[t] for t in g.ntypes], dim=0) if len(g.ntypes) > 1 else h[g.ntypes[0]]
        else:
            x = h
        B = g.batch_size
        out = ops.segment_reduce(x, all_types_plan(g, x.device), self.op)
        return {t: out[i * B:(i + 1) * B] for i, t in enumerate(g.ntypes)}


class GlobalAttentionPooling(nn.Module):
    """dgl.nn.pytorch.glob.GlobalAttentionPooling (graph_pooling_type='att', models/HEATNet4.py:182-187):
    gate = softmax over the graph's nodes of gate_nn(h); readout = sum(gate * h)."""

    def __init__(self, gate_nn: nn.Module):
        super().__init__()
        self.gate_nn = gate_nn

    def forward(self, graph, feat, ntype: Optional[str] = None) -> torch.Tensor:
        if isinstance(feat, dict):
            x = feat[ntype if ntype is not None else next(iter(feat))]
        else:
            x = feat
        nt = ntype if ntype is not None else graph.ntypes[0]
        rp = _rows_plan(graph, nt, x.device, "one")
        bnn = graph.batch_num_nodes(nt).to(x.device)
        seg = torch.repeat_interleave(torch.arange(bnn.numel(), device=x.device), bnn, output_size=x.shape[0])
        gate = ops.linear(x, self.gate_nn.weight, self.gate_nn.bias)            # [N,1]
        mx = ops.segment_reduce(gate, rp, "max")                                  # [B,1]
        ex = torch.exp(gate - mx[seg])
        den = ops.segment_reduce(ex, rp, "sum")
        return ops.segment_reduce(x * (ex / den[seg]), rp, "sum")
