"""Shared trunk of HEATNet2 / HEATNet4: input projection, L x HEATLayer, per-node-type readout.

Follows models/HEATNet4.py:195-221 (== models/HEATNet2.py:159-185) but runs on the type-major
concatenated node table: one grouped GEMM for all ``adapt_ws``, the layers' fused kernels, one
segmented-reduce launch for all (node type, graph) readouts and one grouped GEMM for all
``linears_prediction``.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import ops
from ..pooling import AvgPooling, SumPooling, MaxPooling, GlobalAttentionPooling
from ..pooling.readout import all_types_plan
from ..graph import _resolve_device
from .heat_layer import HEATLayer, heat_context


def make_pool(kind: str, layer: int, in_dim: int, hidden_dim: int) -> nn.Module:
    # models/HEATNet4.py:175-189
    if kind == "sum":
        return SumPooling()
    if kind == "mean":
        return AvgPooling()
    if kind == "max":
        return MaxPooling()
    if kind == "att":
        return GlobalAttentionPooling(nn.Linear(in_dim if layer == 0 else hidden_dim, 1))
    raise NotImplementedError


class HEATTrunk(nn.Module):
    """Not a reference class: holds what HEATNet2 and HEATNet4 share.  Subclasses create the
    parameters in the reference's order."""

    fuse_readout = True      # class default; set the attribute on the class or on an instance to keep the last layer's readout apart (A/B measurements)

    def dead_parameter_names(self) -> List[str]:
        """Parameters ``forward`` never reaches on ANY input (they exist for state_dict parity): the ``weight`` Linear of
        every HEATLayer (reference HEATNet4.py:54 / HEATNet2.py:29 creates it and never calls it) and the readouts after
        ``pools[0]`` (only ``pools[0]`` is ever applied, :219).  ``dist.GradBucket.from_model`` leaves them out."""
        dead = []
        for n, _ in self.named_parameters():
            parts = n.split(".")
            if parts[0] == "gcs" and parts[2] == "weight":
                dead.append(n)
            elif parts[0] == "pools" and parts[1] != "0":
                dead.append(n)
        return dead

    def _input_features(self, G, h, ctx) -> torch.Tensor:
        if h is None:
            return G.cat_ndata("feat")                                       # HEATNet4.py:202
        parts = [h[t] for t in ctx.ntypes]                                   # :204-206
        return (torch.cat(parts, dim=0) if len(parts) > 1 else parts[0]).to(torch.float32)

    def encode(self, G, h=None, predict=True):
        """Returns (ctx, node states [N, hidden] - None when the last layer returned its readout directly -, per-type readout
        features [T*B, out_pred] - or, with ``predict=False``, the readout rows [T*B, hidden] they are projected from -, B)."""
        dev = self.adapt_ws[0].weight.device
        if _resolve_device(G.device) != _resolve_device(dev):
            raise RuntimeError(f"graph is on {G.device} but the model is on {dev}: call G.to(device) first "
                               "(trainer/train_gnn.py:60 does the same)")
        ctx = heat_context(G, self.node_dict, self.n_hid, dev)
        x = self._input_features(G, h, ctx)
        if h is None:
            ops.remember_constant_rows(x, G)         # fp16x3 / auto: the features of a resident graph are scanned for their scales once
            if ops.want_col_stats(x.shape[0], self.n_hid, x.shape[1]):
                ops.remember_constant_cols(x, ctx.rows)      # ... and for the column scales of the input projection's weight gradient
        hcat = ops.grouped_linear(x, ctx.all_spec,
                                  [self.adapt_ws[n].weight for n in ctx.nid],
                                  [self.adapt_ws[n].bias for n in ctx.nid])
        B = G.batch_size
        T = len(ctx.ntypes)
        pool = self.pools[0]
        # the last layer's output is only ever read by the readout (:219; HEATNet2.py:183): for a sum / mean readout the layer returns
        # the pooled rows directly (mean over nodes commutes with its affine output stage - ops._HeatLayerFused) and the [N, hidden]
        # output is never formed.  ``fuse_readout = False`` keeps the two steps apart (A/B measurements, the parity tests of both forms).
        rp = all_types_plan(G, dev) if not isinstance(pool, GlobalAttentionPooling) else None
        fuse = (self.fuse_readout and self.n_layers > 0 and rp is not None and pool.op in ("sum", "mean") and self.gcs[-1].can_pool()
                and rp.num_rows == hcat.shape[0] and rp.segments_of(ctx.rows) is not None)
        if rp is not None and pool.op in ("sum", "mean") and rp._row_seg is None:
            rp.prepare_broadcast()
        for i in range(self.n_layers):                                       # :213-214
            last = i == self.n_layers - 1
            hcat = self.gcs[i].forward_cat(ctx, hcat, pool=(rp, pool.op) if (fuse and last) else None, first_layer=(i == 0))
        if fuse:
            pooled, hcat = hcat, None
        elif isinstance(pool, GlobalAttentionPooling):
            pooled = torch.cat([pool(G, hcat[a:b], ntype=t) for t, (a, b) in zip(ctx.ntypes, ctx.rows)], dim=0)
        else:
            pooled = ops.segment_reduce(hcat, rp, pool.op)        # :219 pools[0](G, h, ntype=k), all k at once
        if not predict:
            return ctx, hcat, pooled, B
        pred_out = self.linears_prediction[ctx.ntypes[0]].weight.shape[0]
        spec = ctx.cache.get(("pred", B, pred_out))
        if spec is None:
            spec = ctx.cache[("pred", B, pred_out)] = ops.LinearSpec([(i * B, (i + 1) * B) for i in range(T)], [0] * T, pred_out, T * B)
        out = ops.grouped_linear(pooled, spec,
                                 [self.linears_prediction[t].weight for t in ctx.ntypes],
                                 [self.linears_prediction[t].bias for t in ctx.ntypes])           # :219
        return ctx, hcat, out, B
