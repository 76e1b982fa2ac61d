"""Restatement of the DGL primitives the reference calls (TEST INFRASTRUCTURE — see oracle/__init__.py).

DGL is third-party, un-vendored and un-pinned (API usage implies >= 0.8); these functions restate
its *documented* behaviour as listed in SURVEY.md Appendix A.1, on plain COO tensors.
Each function names the reference call site that reaches the primitive.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch


def v_dot_u(q_dst: torch.Tensor, k_src: torch.Tensor, src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """``fn.v_dot_u('q','k','t')`` (models/HEATNet4.py:109, models/HGT.py:99).

    q_dst [N_d,H,dk], k_src [N_s,H,dk] -> [E,H,1]: dot over the last dim of dst 'q' and src 'k'.
    """
    return (q_dst[dst] * k_src[src]).sum(-1, keepdim=True)


def edge_softmax_dst(score: torch.Tensor, dst: torch.Tensor, num_dst: int) -> torch.Tensor:
    """``dgl.nn.edge_softmax(g, score)`` with the default ``norm_by='dst'`` (models/HEATNet4.py:113).

    Softmax over the in-edges of every dst node of THIS relation, independently per trailing dim;
    max-subtracted, no epsilon.  score [E,H] -> [E,H].
    """
    E = score.shape[0]
    if E == 0:
        return score.clone()
    H = score.shape[1:]
    idx = dst.view(-1, *([1] * len(H))).expand_as(score)
    mx = torch.full((num_dst, *H), float("-inf"), dtype=score.dtype, device=score.device)
    mx = mx.scatter_reduce(0, idx, score, reduce="amax", include_self=True)
    ex = torch.exp(score - mx[dst])
    den = torch.zeros((num_dst, *H), dtype=score.dtype, device=score.device).index_add_(0, dst, ex)
    return ex / den[dst]


def u_mul_e_sum(v_src: torch.Tensor, a: torch.Tensor, src: torch.Tensor, dst: torch.Tensor, num_dst: int) -> torch.Tensor:
    """``(fn.u_mul_e('v','t','m'), fn.sum('m','t'))`` (models/HEATNet4.py:118): dst nodes without in-edges get 0.

    v_src [N_s,H,dk], a [E,H,1] -> [N_d,H,dk].
    """
    out = torch.zeros((num_dst, *v_src.shape[1:]), dtype=v_src.dtype, device=v_src.device)
    if src.numel():
        out.index_add_(0, dst, v_src[src] * a)
    return out


def cross_reduce_mean(per_rel: List[torch.Tensor]) -> torch.Tensor:
    """``multi_update_all(..., cross_reducer='mean')`` (models/HEATNet4.py:119).

    One relation -> used as is; else ``torch.stack(list, 0).mean(0)``: the denominator is the number
    of relations in the dict with this dst type, whether or not they have edges (Appendix A.1.4).
    """
    if len(per_rel) == 1:
        return per_rel[0]
    return torch.stack(per_rel, 0).mean(0)


def segment_readout(feat: torch.Tensor, batch_num_nodes: torch.Tensor, op: str) -> torch.Tensor:
    """``dgl.readout.{mean,sum,max}_nodes(graph,'h',ntype=t)`` (pooling/avg_pooling.py:15-17, ...).

    feat [N_t,D], batch_num_nodes [B] -> [B,D]; mean of an empty segment = 0 (sum / clamp(count,1));
    max of an empty segment = 0 as well (DGL pads empty segments before the max, A.1.6).
    """
    B = int(batch_num_nodes.numel())
    counts = batch_num_nodes.to(feat.device)
    seg = torch.repeat_interleave(torch.arange(B, device=feat.device), counts)
    D = feat.shape[1:]
    if op in ("sum", "mean"):
        out = torch.zeros((B, *D), dtype=feat.dtype, device=feat.device).index_add_(0, seg, feat)
        if op == "mean":
            out = out / counts.clamp(min=1).to(feat.dtype).view(-1, *([1] * len(D)))
        return out
    if op == "max":
        out = torch.full((B, *D), float("-inf"), dtype=feat.dtype, device=feat.device)
        idx = seg.view(-1, *([1] * len(D))).expand_as(feat)
        out = out.scatter_reduce(0, idx, feat, reduce="amax", include_self=True)
        return torch.where(torch.isinf(out) & (out < 0), torch.zeros_like(out), out)
    raise ValueError(op)


def graph_conv_both(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                    src: torch.Tensor, dst: torch.Tensor, num_nodes: int, activation=None) -> torch.Tensor:
    """``dgl.nn.pytorch.GraphConv(in,out,norm='both')`` forward (models/GCN.py:30-33).

    ``weight`` is [in,out].  x~ = x * outdeg.clamp(1)^-1/2; if in>out: Aggr(x~ W) else Aggr(x~) W;
    y = y * indeg.clamp(1)^-1/2 + b; activation.
    """
    outdeg = torch.bincount(src, minlength=num_nodes).clamp(min=1).to(x.dtype)
    indeg = torch.bincount(dst, minlength=num_nodes).clamp(min=1).to(x.dtype)
    xs = x * outdeg.pow(-0.5).unsqueeze(-1)

    def aggr(z):
        return torch.zeros((num_nodes, z.shape[1]), dtype=z.dtype, device=z.device).index_add_(0, dst, z[src])

    if weight.shape[0] > weight.shape[1]:
        y = aggr(xs @ weight)
    else:
        y = aggr(xs) @ weight
    y = y * indeg.pow(-0.5).unsqueeze(-1)
    if bias is not None:
        y = y + bias
    if activation is not None:
        y = activation(y)
    return y
