"""HGT — drop-in for the reference's ``models/HGT.py`` (HGTLayer :21-127, HGT :130-209) on the MI355X kernels.

Per relation r = (s, e, d) the reference transforms k and v by per-head d_k x d_k matrices
(``einsum("bij,ijk->bik", k, relation_att[e_id])`` :92-93) and scales the logits by ``relation_pri[e_id]``
(:100).  Here those per-relation maps are folded into the projection weights
(``W'_k = blockdiag(att_r * pri_r)^T W_k`` — a tiny [D,D] op left to autograd), so one grouped MFMA GEMM
writes a stacked K'|V' table with one row block per (relation, source node), and the SAME relation-attention
kernels as HEAT run on it (logit scale 1/sqrt(d_k): e_weight = 0, e_bias = 1).  Gated skip in the output GEMM's
epilogue, LayerNorm and GELU as row-wise HIP kernels.  Constructor signatures / state_dict keys as the reference.
The last layer's output is never read by the reference, so it is not computed.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..pooling import GlobalAttentionPooling
from ..pooling.readout import all_types_plan
from ..graph import _resolve_device
from .heat_layer import heat_context
from .heat_net import make_pool


_FAST_D, _FAST_H = (128, 256, 512), (1, 2, 4, 8, 16)


def padded_head_dim(D: int, H: int) -> int:
    """Per-head width the attention tables are laid out with.  The specialised attention kernels (16-byte lane loads, DPP
    head reductions, hub kernels) exist for D in {128,256,512} x H in {1,2,4,8,16}; any other width (the reference's HGT
    configs use hidden 200 = 4 heads x 50) is zero-padded PER HEAD to the next such shape — zeros add nothing to q.k nor to
    the weighted sum of v, and the projection GEMMs compute whole 128-column tiles anyway — instead of taking the slower
    generic kernels.  Returns d_k itself when no padding applies."""
    dk = D // H
    if H in _FAST_H:
        for Dp in _FAST_D:
            if Dp % H == 0 and Dp // H >= dk:
                return Dp // H
    return dk


class HgtContext:
    """Static per-(graph batch, edge_dict) data of the HGT layers."""

    def __init__(self, G, hctx, edge_dict, D: int, device, H: int = 1):
        self.h = hctx
        self.dkp = padded_head_dim(D, H)
        true_D, D = D, self.dkp * H                  # below, D is the (padded) row width of the q / k|v / t tables
        self.plan = G.plan(per_relation_src=True)
        rels = G.canonical_etypes
        tindex = {t: i for i, t in enumerate(hctx.ntypes)}
        self.rels = rels
        self.e_ids = [edge_dict[r] for r in rels]                   # KeyError for a relation missing from edge_dict, as HGT.py:86
        self.src_t = [tindex[r[0]] for r in rels]
        kv_rows, kv_out, kv_cols = [], [], []
        for ri, r in enumerate(rels):
            rows = hctx.rows[self.src_t[ri]]
            out = self.plan.rel_rows[ri]
            kv_rows += [rows, rows]
            kv_out += [out, out]
            kv_cols += [0, D]
        self.kv_spec = ops.LinearSpec(kv_rows, kv_cols, 2 * D, hctx.num_nodes, out_rows=kv_out, num_out_rows=self.plan.num_src_rows)
        self.q_types = [i for i in range(len(hctx.ntypes)) if hctx.incoming[i]]
        self.q_spec = ops.LinearSpec([hctx.rows[i] for i in self.q_types], [0] * len(self.q_types), D, hctx.num_nodes)
        self.a_rows = [hctx.rows[i] for i in hctx.a_types]
        self.a_nids = [hctx.nid[i] for i in hctx.a_types]
        self.a_rplan = ops.ReducePlan.from_ranges(_fill(self.a_rows, hctx.num_nodes)[0], device, chunk=512) if self.a_rows else None
        self.a_seg = _fill(self.a_rows, hctx.num_nodes)[1] if self.a_rows else []
        self.zero_w = torch.zeros(1, 1, device=device)
        # the kernels scale logits by (w*sim + b)/sqrt(row width / H): w = 0, b = sqrt(padded d_k / d_k) gives HGT's 1/sqrt(d_k)
        self.one_b = torch.full((1,), math.sqrt(self.dkp * H / true_D), device=device)
        self.Dp = D
        self.sim0 = torch.zeros(max(self.plan.num_edges, 1), device=device)
        self.row_type = torch.empty(hctx.num_nodes, dtype=torch.int32, device=device)
        for i, (a, b) in enumerate(hctx.rows):
            if b > a:
                self.row_type[a:b].fill_(i)
        self.all_incoming = all(hctx.incoming)
        self.cache = {}
        # index tables for folding all relations' d_k x d_k maps into the projections in ONE batched einsum
        from ..graph import host_to_device
        self.e_ids_t = host_to_device(self.e_ids, torch.int64, device)
        self.src_nid_t = host_to_device([hctx.nid[t] for t in self.src_t], torch.int64, device)
        # the same selection as a constant one-hot matrix [R, node types of the model]: "rows of the per-type stack by source type" as a product
        # (exact: ones and zeros) whose backward is a product too - advanced indexing with this repeated index costs, per use, an index kernel forward and
        # a sort + five tiny index-arithmetic launches + an accumulating index_put backward (14 such uses per HGT step: ~0.35 ms of launch-sized kernels)
        n_model_types = max(hctx.nid) + 1 if hctx.nid else 1
        oh = [[0.0] * n_model_types for _ in self.src_t]
        for r, t in enumerate(self.src_t):
            oh[r][hctx.nid[t]] = 1.0
        self.src_onehot = host_to_device(oh, torch.float32, device) if self.src_t else None


def _fill(rows, n):
    """Sorted, gap-free ranges covering [first, last] + index of each original range (ReducePlan needs contiguity)."""
    order = sorted(range(len(rows)), key=lambda i: rows[i])
    filled, seg = [], [0] * len(rows)
    pos = rows[order[0]][0] if order else 0
    for i in order:
        a, b = rows[i]
        if a > pos:
            filled.append((pos, a))
        seg[i] = len(filled)
        filled.append((a, b))
        pos = b
    return filled, seg


def hgt_context(G, hctx, edge_dict, D, device, H: int = 1) -> HgtContext:
    cache = G.__dict__.setdefault("_hgt_ctx", {})
    key = (tuple(sorted(edge_dict.items())), D, H, str(device))     # by value: id() of a dead dict can be reused
    if key not in cache:
        cache[key] = HgtContext(G, hctx, edge_dict, D, device, H)
    return cache[key]


class HGTLayer(nn.Module):
    def __init__(self, in_dim, out_dim, node_dict, edge_dict, n_heads, dropout=0.2, use_norm=False):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.node_dict, self.edge_dict = node_dict, edge_dict
        self.num_types = len(node_dict)
        self.num_relations = len(edge_dict)
        self.total_rel = self.num_types * self.num_relations * self.num_types
        self.n_heads = n_heads
        self.d_k = out_dim // n_heads
        self.sqrt_dk = math.sqrt(self.d_k)
        self.att = None
        self.k_linears = nn.ModuleList()
        self.q_linears = nn.ModuleList()
        self.v_linears = nn.ModuleList()
        self.a_linears = nn.ModuleList()
        self.norms = nn.ModuleList()
        self.use_norm = use_norm
        for _ in range(self.num_types):
            self.k_linears.append(nn.Linear(in_dim, out_dim))
            self.q_linears.append(nn.Linear(in_dim, out_dim))
            self.v_linears.append(nn.Linear(in_dim, out_dim))
            self.a_linears.append(nn.Linear(out_dim, out_dim))
            if use_norm:
                self.norms.append(nn.LayerNorm(out_dim))
        self.relation_pri = nn.Parameter(torch.ones(self.num_relations, self.n_heads))
        self.relation_att = nn.Parameter(torch.Tensor(self.num_relations, n_heads, self.d_k, self.d_k))
        self.relation_msg = nn.Parameter(torch.Tensor(self.num_relations, n_heads, self.d_k, self.d_k))
        self.skip = nn.Parameter(torch.ones(self.num_types))
        self.drop = nn.Dropout(dropout)
        nn.init.xavier_uniform_(self.relation_att)
        nn.init.xavier_uniform_(self.relation_msg)

    def forward_cat(self, hctx, gctx: HgtContext, h: torch.Tensor) -> torch.Tensor:
        D = self.out_dim
        if self.in_dim != self.out_dim:
            raise NotImplementedError("HGTLayer kernels assume in_dim == out_dim (as every reference config)")
        if not gctx.e_ids:
            return h
        # HGT.py:75-97: k = einsum('bij,ijk->bik', K_s(h), relation_att[e]) (and the :100 prior), v likewise with relation_msg:
        # folded into the projection weights, all relations in one batched einsum (R = 18 at T=3: one launch, not 36)
        H, dk, R = self.n_heads, self.d_k, len(gctx.e_ids)
        def by_src(params):                       # [R, ...]: the per-type parameter of every relation's SOURCE type (one-hot product: HgtContext)
            st = torch.stack(params)
            oh = gctx.src_onehot
            if oh.shape[1] != st.shape[0]:        # (a model with more node types than this graph's context has seen)
                return st[gctx.src_nid_t]
            return (oh @ st.reshape(st.shape[0], -1)).view(R, *st.shape[1:])
        Wk_src = by_src([l.weight for l in self.k_linears]).view(R, H, dk, -1)
        bk_src = by_src([l.bias for l in self.k_linears]).view(R, H, dk)
        Wv_src = by_src([l.weight for l in self.v_linears]).view(R, H, dk, -1)
        bv_src = by_src([l.bias for l in self.v_linears]).view(R, H, dk)
        # (the relation ids of a graph are distinct: index_select's backward - an index_add_ - has nothing to accumulate out of order)
        rel_k = self.relation_att.index_select(0, gctx.e_ids_t) * self.relation_pri.index_select(0, gctx.e_ids_t).view(R, H, 1, 1)
        rel_v = self.relation_msg.index_select(0, gctx.e_ids_t)
        pad = gctx.dkp - dk                       # per-head zero padding of the attention tables (padded_head_dim)
        Dp = gctx.Dp
        Wk = F.pad(torch.einsum("rhjk,rhjc->rhkc", rel_k, Wk_src), (0, 0, 0, pad)).reshape(R, Dp, -1)
        bk = F.pad(torch.einsum("rhjk,rhj->rhk", rel_k, bk_src), (0, pad)).reshape(R, Dp)
        Wv = F.pad(torch.einsum("rhjk,rhjc->rhkc", rel_v, Wv_src), (0, 0, 0, pad)).reshape(R, Dp, -1)
        bv = F.pad(torch.einsum("rhjk,rhj->rhk", rel_v, bv_src), (0, pad)).reshape(R, Dp)
        ws = [w for pair in zip(Wk.unbind(0), Wv.unbind(0)) for w in pair]      # unbind: backward is one stack, not R slices
        bs = [b for pair in zip(bk.unbind(0), bv.unbind(0)) for b in pair]
        kv = ops.grouped_linear(h, gctx.kv_spec, ws, bs)
        T = len(self.q_linears)
        Wq = F.pad(torch.stack([l.weight for l in self.q_linears]).view(T, H, dk, -1), (0, 0, 0, pad)).reshape(T, Dp, -1).unbind(0)
        bq = F.pad(torch.stack([l.bias for l in self.q_linears]).view(T, H, dk), (0, pad)).reshape(T, Dp).unbind(0)
        q = ops.grouped_linear(h, gctx.q_spec, [Wq[hctx.nid[i]] for i in gctx.q_types], [bq[hctx.nid[i]] for i in gctx.q_types])   # :84
        t = ops.relation_attention(q, kv, gctx.zero_w, gctx.one_b, gctx.plan, gctx.sim0, Dp, self.n_heads)   # :99-106
        # a_linears read the padded t: zero COLUMNS at the pad positions
        Wa = F.pad(torch.stack([l.weight for l in self.a_linears]).view(T, D, H, dk), (0, pad)).reshape(T, D, Dp).unbind(0)
        aw = [Wa[n] for n in gctx.a_nids]
        ab = [self.a_linears[n].bias for n in gctx.a_nids]
        mask = None
        if self.training and self.drop.p > 0.0:                                                # :121 nn.Dropout: keep mask / (1 - p), applied in the
            keep = 1.0 - self.drop.p                                                           # projection's epilogue (as the HEAT layer does)
            mask = torch.empty_like(h).bernoulli_(keep).mul_(1.0 / keep) if keep > 0.0 else torch.zeros_like(h)
        z = ops.gated_linear(t, h, self.skip, gctx.a_rows, gctx.a_nids, gctx.a_rplan, gctx.a_seg, aw, ab, drop_mask=mask)   # :121-122
        if not self.use_norm:
            return z
        gamma = torch.stack([self.norms[n].weight for n in hctx.nid])
        beta = torch.stack([self.norms[n].bias for n in hctx.nid])
        out = ops.layer_norm(z, gamma, beta, gctx.row_type, hctx.type_rplan(), list(range(len(hctx.nid))), self.norms[0].eps)  # :124
        if not gctx.all_incoming:
            out = torch.where(hctx.row_incoming > 0, out, z)                                   # passthrough types skip the norm (:116-120)
        return out

    def forward(self, G, h: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        dev = next(iter(h.values())).device
        hctx = heat_context(G, self.node_dict, self.out_dim, dev)
        gctx = hgt_context(G, hctx, self.edge_dict, self.out_dim, dev, self.n_heads)
        x = torch.cat([h[t] for t in hctx.ntypes], dim=0) if len(hctx.ntypes) > 1 else h[hctx.ntypes[0]]
        out = self.forward_cat(hctx, gctx, x)
        return {t: out[a:b] for t, (a, b) in zip(hctx.ntypes, hctx.rows)}


class HGT(nn.Module):
    def __init__(self, node_dict, edge_dict, in_dim, hidden_dim, out_dim, n_layers, n_heads,
                 use_norm=True, graph_pooling_type="mean"):
        super().__init__()
        self.node_dict, self.edge_dict = node_dict, edge_dict
        self.gcs = nn.ModuleList()
        self.n_layers = n_layers
        self.n_hid = hidden_dim
        self.adapt_ws = nn.ModuleList()
        self.pools = nn.ModuleList()
        self.linears_prediction = nn.ModuleDict({k: nn.ModuleList() for k in node_dict})
        for _ in range(len(node_dict)):
            self.adapt_ws.append(nn.Linear(in_dim, hidden_dim))
        for _ in range(n_layers):
            self.gcs.append(HGTLayer(hidden_dim, hidden_dim, node_dict, edge_dict, n_heads, use_norm=use_norm))
        self.out = nn.Linear(hidden_dim, out_dim)
        for layer in range(n_layers + 1):
            for k in self.linears_prediction:
                self.linears_prediction[k].append(nn.Linear(hidden_dim, out_dim))
            self.pools.append(make_pool(graph_pooling_type, layer, in_dim, hidden_dim))

    def dead_parameter_names(self) -> List[str]:
        return _readout_sum_dead_parameters(self, "gcs")

    def forward(self, G, h=None):
        return _readout_sum_forward(self, G, h, lambda i, hctx, x: self.gcs[i].forward_cat(
            hctx, hgt_context(G, hctx, self.edge_dict, self.n_hid, x.device, self.gcs[i].n_heads), x))


def _readout_sum_dead_parameters(model, layers_attr: str) -> List[str]:
    """Parameters the readout-sum forward (HGT.py:173-209 / HetRGCN.py:91-125) never reaches: the LAST layer (its output is
    never read), ``out``, and the prediction heads / readouts of index ``n_layers``."""
    L = model.n_layers
    dead = []
    for n, _ in model.named_parameters():
        parts = n.split(".")
        if parts[0] == layers_attr and parts[1] == str(L - 1):
            dead.append(n)
        elif parts[0] == "out":
            dead.append(n)
        elif parts[0] == "linears_prediction" and parts[2] == str(L):
            dead.append(n)
        elif parts[0] == "pools" and parts[1] == str(L):
            dead.append(n)
    return dead


def _readout_sum_forward(model, G, h, layer_fn, need_last: bool = False):
    """Shared by HGT and HeteroRGCN (models/HGT.py:173-209, models/HetRGCN.py:91-125): GELU(input projection), then for
    every layer i: hg += sum_k linears_prediction[k][i](pool_i(h)) BEFORE applying layer i; the output of the last
    layer is never read by the reference, so it is not computed — unless ``need_last`` (a caller that consumes the final
    node states, models/HGT_ASAP.py), in which case ``(hg, node states [N, hidden], hctx)`` is returned."""
    dev = model.adapt_ws[0].weight.device
    if _resolve_device(G.device) != _resolve_device(dev):
        raise RuntimeError(f"graph is on {G.device} but the model is on {dev}: call G.to(device) first")
    hctx = heat_context(G, model.node_dict, model.n_hid, dev)
    if h is None:
        x = G.cat_ndata("feat")
        ops.remember_constant_rows(x, G)             # fp16x3 / auto: a resident graph's features are scanned for their scales once
    else:
        x = torch.cat([h[t] for t in hctx.ntypes], dim=0).to(torch.float32)
    x = ops.gelu(ops.grouped_linear(x, hctx.all_spec, [model.adapt_ws[n].weight for n in hctx.nid],
                                    [model.adapt_ws[n].bias for n in hctx.nid]))
    B, T = G.batch_size, len(hctx.ntypes)
    present = [(b - a) > 0 for a, b in hctx.rows]
    hg = 0
    for i in range(model.n_layers):
        pool = model.pools[i]
        if isinstance(pool, GlobalAttentionPooling):
            pooled = torch.cat([pool(G, x[a:b], ntype=t) for t, (a, b) in zip(hctx.ntypes, hctx.rows)], dim=0)
        else:
            pooled = ops.segment_reduce(x, all_types_plan(G, dev), pool.op)
        out_dim = model.linears_prediction[hctx.ntypes[0]][i].weight.shape[0]
        spec = hctx.cache.get(("pred", B, out_dim))
        if spec is None:
            spec = hctx.cache[("pred", B, out_dim)] = ops.LinearSpec([(j * B, (j + 1) * B) for j in range(T)], [0] * T, out_dim, T * B)
        out = ops.grouped_linear(pooled, spec, [model.linears_prediction[t][i].weight for t in hctx.ntypes],
                                 [model.linears_prediction[t][i].bias for t in hctx.ntypes])
        for j in range(T):
            if present[j]:
                hg = hg + out[j * B:(j + 1) * B]
        if need_last or i + 1 < model.n_layers:
            x = layer_fn(i, hctx, x)
    return (hg, x, hctx) if need_last else hg
