"""HEATLayer on the HIP kernels: one grouped K|Q|V GEMM, one fused relation-attention launch, one
grouped output GEMM per layer — instead of the reference's per-relation Python loop of 3 GEMMs +
~7 DGL kernels (models/HEATNet4.py:85-138; identical copy at models/HEATNet2.py:24-113).

Same constructor signature, parameter creation order and ``state_dict`` keys as the reference
(``weight``, ``k/q/v/a_linears.{t}``, ``e_linear``, ``skip`` — SURVEY Appendix A.7), so reference
checkpoints load.  ``self.weight`` is unused there too (HEATNet4.py:54) and is kept only for that.
"""
from __future__ import annotations

import math
import weakref
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..graph import _resolve_device, host_to_device


class HeatContext:
    """Per-(graph batch, node_dict) static data shared by all layers: kernel plan, CSR-ordered sim,
    grouped-GEMM row specs, row -> node-type index."""

    def __init__(self, G, node_dict: Dict[str, int], hidden: int, device):
        self.plan = G.plan()
        device = _resolve_device(device)
        if _resolve_device(self.plan.device) != device:
            raise RuntimeError(f"graph lives on {self.plan.device}, model on {device}; call G.to(device) first")
        self.ntypes: List[str] = G.ntypes
        for t in self.ntypes:
            if t not in node_dict:
                raise KeyError(f"node type {t!r} of the graph is not in the model's node_dict")
        self.nid = [node_dict[t] for t in self.ntypes]
        off = self.plan.type_off
        self.rows = [(off[i], off[i + 1]) for i in range(len(self.ntypes))]
        n = self.plan.num_nodes
        self.num_nodes = n
        self._graph = weakref.ref(G)        # (the context lives in G's own cache: a strong reference would be a cycle)
        D = hidden
        # K at column 0, Q at D, V at 2D of the fused table
        kqv_rows, kqv_cols = [], []
        for r in self.rows:
            kqv_rows += [r, r, r]
            kqv_cols += [0, D, 2 * D]
        self.kqv_spec = ops.LinearSpec(kqv_rows, kqv_cols, 3 * D, n)
        self.incoming = [self.plan.rel_slots[i] > 0 for i in range(len(self.ntypes))]
        self.a_types = [i for i in range(len(self.ntypes)) if self.incoming[i]]
        self.a_spec = ops.LinearSpec([self.rows[i] for i in self.a_types], [0] * len(self.a_types), D, n)
        self.all_spec = ops.LinearSpec(self.rows, [0] * len(self.rows), D, n)
        # row -> node-type tables built ON THE DEVICE with one fill per node type: no host tensor math (a CPU
        # repeat_interleave / pinned copy of 80k elements wakes the whole OpenMP pool: sporadic 50-90 ms stalls when a
        # new batch arrives every step) and no host->device transfer
        self.row_nid = torch.empty(n, dtype=torch.int64, device=device)                                # [N] index into skip
        self.row_incoming = torch.empty((n, 1), dtype=torch.float32, device=device)                     # [N,1]
        for i, (a, b) in enumerate(self.rows):
            if b > a:
                self.row_nid[a:b].fill_(self.nid[i])
                self.row_incoming[a:b].fill_(1.0 if self.incoming[i] else 0.0)
        self._type_rplan = None
        self.device = device
        self.cache = {}     # per-graph-batch static objects of the model (specs with device-side tables)

    @property
    def sim_csr(self) -> torch.Tensor:
        """CSR-ordered ``edata['sim']``, fetched at use time: ``cat_edata_csr`` caches it and notices re-assignment or
        in-place edits of the per-relation tensors, so a context never serves a stale copy."""
        return self._graph().cat_edata_csr("sim")

    def row_gate(self, skip: torch.Tensor) -> torch.Tensor:
        """[N,1] per-row gate sigmoid(skip[type of row]) (0 on passthrough types), built from one broadcast per node type.
        (``sigmoid(skip)[row_nid]`` computes the same values but its backward is a sort-based ``index_put`` over N rows:
        0.64 ms per layer at 80k nodes; the backward of ``expand`` is a plain column reduction.)"""
        s = torch.sigmoid(skip)
        parts = []
        for i, (a, b) in enumerate(self.rows):
            if b > a:
                g = s[self.nid[i]] if self.incoming[i] else s.new_zeros(())
                parts.append(g.reshape(1, 1).expand(b - a, 1))
        return torch.cat(parts, dim=0)

    def type_rplan(self):
        """ReducePlan whose segments are the node types' row ranges (bias / skip-gate gradients)."""
        if self._type_rplan is None:
            self._type_rplan = ops.ReducePlan.from_ranges(self.rows, self.device, chunk=512)
        return self._type_rplan


def heat_context(G, node_dict, hidden: int, device) -> HeatContext:
    cache = G.__dict__.setdefault("_heat_ctx", {})
    key = (tuple(sorted(node_dict.items())), int(hidden), str(_resolve_device(device)))
    if key not in cache:
        cache[key] = HeatContext(G, node_dict, hidden, device)
    return cache[key]


class HEATLayer(nn.Module):
    def __init__(self, in_size, out_size, node_dict, n_heads, dropout=0.2):
        super().__init__()
        self.weight = nn.Linear(in_size, out_size)   # unused (reference HEATNet4.py:54); state_dict parity
        self.in_size, self.out_size = in_size, out_size
        self.node_dict = node_dict
        self.num_node_types = len(node_dict)
        self.n_heads = n_heads
        self.d_k = out_size // n_heads
        self.sqrt_dk = math.sqrt(self.d_k)
        self.k_linears = nn.ModuleList()
        self.q_linears = nn.ModuleList()
        self.v_linears = nn.ModuleList()
        self.a_linears = nn.ModuleList()
        self.e_linear = nn.Linear(1, 1)
        self.skip = nn.Parameter(torch.ones(self.num_node_types))
        self.drop = nn.Dropout(dropout)
        self.fused = True   # False forces the composed (unfused) path; used by tests
        self.counter_dropout = True     # False: the train-mode dropout mask as a torch-drawn [N, D] tensor (A/B measurements; the composed path always does)
        for _ in range(self.num_node_types):
            self.k_linears.append(nn.Linear(in_size, out_size))
            self.q_linears.append(nn.Linear(in_size, out_size))
            self.v_linears.append(nn.Linear(in_size, out_size))
            self.a_linears.append(nn.Linear(out_size, out_size))

    # type-major concatenated fast path used by HEATNet2/4
    def can_pool(self) -> bool:
        """True when ``forward_cat(..., pool=)`` may fold a sum / mean readout into this layer: the fused path, and no dropout draw
        between the output projection and the readout (eval, or p = 0)."""
        return self.fused and not (self.training and self.drop.p > 0.0)

    def forward_cat(self, ctx: HeatContext, h: torch.Tensor, pool=None, first_layer: bool = True) -> torch.Tensor:
        """``pool`` = (ops.ReducePlan, "sum" | "mean"): return the readout of the layer's output ([segments, D]) instead of the output
        - only the last layer of HEATNet2 / HEATNet4 is asked to, see ops._HeatLayerFused; requires ``can_pool()``."""
        if pool is not None and not self.can_pool():
            raise RuntimeError("HEATLayer.forward_cat(pool=...) needs the fused path without an active dropout")
        if self.in_size != self.out_size:
            raise NotImplementedError("HEATLayer kernels assume in_size == out_size (as every reference config)")
        D = self.out_size
        if self.fused:
            # fused layer: gating (and, in training, the dropout mask of :135) in the GEMM epilogue, hand-written backward
            params = []
            for nid in ctx.nid:
                params += [self.k_linears[nid].weight, self.q_linears[nid].weight, self.v_linears[nid].weight,
                           self.a_linears[nid].weight, self.k_linears[nid].bias, self.q_linears[nid].bias,
                           self.v_linears[nid].bias, self.a_linears[nid].bias]
            mask = None
            if self.training and self.drop.p > 0.0:          # nn.Dropout: keep with probability 1-p, scale kept values by 1/(1-p)
                keep = 1.0 - self.drop.p
                if self.counter_dropout and keep > 0.0:
                    # the draw as a function of (seed, row, column), applied inside the projection's epilogue and regenerated in the backward:
                    # no [N, D] mask is generated, stored or read (ops.CounterDropout)
                    mask = ops.CounterDropout(self.drop.p, ops.next_dropout_seed(), ops.current_dropout_seed_base(h.device))
                else:
                    mask = torch.empty_like(h).bernoulli_(keep).mul_(1.0 / keep) if keep > 0.0 else torch.zeros_like(h)
            return ops.heat_layer_fused(h, ctx, self.n_heads, self.skip, self.e_linear.weight, self.e_linear.bias, params, mask, pool,
                                        background_dw=not first_layer)
        # training with dropout > 0: dropout sits between the output projection and the gate (:134), so the
        # projection cannot carry the gate in its epilogue; composed from the individual ops instead
        ws, bs = [], []
        for nid in ctx.nid:
            for lin in (self.k_linears[nid], self.q_linears[nid], self.v_linears[nid]):
                ws.append(lin.weight)
                bs.append(lin.bias)
        kqv = ops.grouped_linear(h, ctx.kqv_spec, ws, bs)                                  # HEATNet4.py:100-102 (dedup per type)
        t = ops.heat_attention(kqv, self.e_linear.weight, self.e_linear.bias, ctx.plan,
                               ctx.sim_csr, D, self.n_heads)                               # :103-119
        if not ctx.a_types:
            return h                                                                       # no relation at all: passthrough (:129-133)
        y = ops.grouped_linear(t, ctx.a_spec,
                               [self.a_linears[ctx.nid[i]].weight for i in ctx.a_types],
                               [self.a_linears[ctx.nid[i]].bias for i in ctx.a_types])     # :134
        y = self.drop(y)
        alpha = ctx.row_gate(self.skip)                                                    # :128 ; 0 on passthrough types
        return torch.lerp(h, y, alpha)                                                     # :135  a*y + (1-a)*h

    # reference signature: dict of per-type features in, dict out (models/HEATNet4.py:85)
    def forward(self, G, feat_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        dev = next(iter(feat_dict.values())).device
        ctx = heat_context(G, self.node_dict, self.out_size, dev)
        h = torch.cat([feat_dict[t] for t in ctx.ntypes], dim=0) if len(ctx.ntypes) > 1 else feat_dict[ctx.ntypes[0]]
        out = self.forward_cat(ctx, h)
        return {t: out[a:b] for t, (a, b) in zip(ctx.ntypes, ctx.rows)}
