"""Second, independent formulation of the HEAT layer (TEST INFRASTRUCTURE — see oracle/__init__.py).

``oracle/models.py`` uses scatter/index_add restatements of the DGL primitives; this file computes
the same layer (models/HEATNet4.py:85-138) with a dense ``[N_dst, E_r]``
incidence mask and ``torch.softmax`` / ``einsum`` — no scatter, no shared helper — so a mistake in
one formulation shows up as a disagreement (tests/test_oracle.py).  Quadratic memory: tiny graphs only.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def heat_layer_dense(layer, G, feat_dict: Dict[str, torch.Tensor], sim: Dict) -> Dict[str, torch.Tensor]:
    """``layer`` is an ``oracle.models.HEATLayer`` (only its parameters are read)."""
    H, dk = layer.n_heads, layer.d_k
    nd = layer.node_dict
    sums: Dict[str, torch.Tensor] = {}
    counts: Dict[str, int] = {}
    for rel in G.canonical_etypes:
        s, _, d = rel
        src, dst = G.edges(rel)
        hs, hd = feat_dict[s], feat_dict[d]
        Wk, bk = layer.k_linears[nd[s]].weight, layer.k_linears[nd[s]].bias
        Wv, bv = layer.v_linears[nd[s]].weight, layer.v_linears[nd[s]].bias
        Wq, bq = layer.q_linears[nd[d]].weight, layer.q_linears[nd[d]].bias
        k = (hs @ Wk.t() + bk).reshape(-1, H, dk)
        v = (hs @ Wv.t() + bv).reshape(-1, H, dk)
        q = (hd @ Wq.t() + bq).reshape(-1, H, dk)
        Nd, E = q.shape[0], int(src.numel())
        ea = layer.e_linear.weight.reshape(()) * sim[rel].to(k.dtype) + layer.e_linear.bias.reshape(())   # [E]
        ke = k[src]                                                     # [E,H,dk]
        logits = torch.einsum("whc,ehc->weh", q, ke) * ea.view(1, E, 1) / math.sqrt(dk)   # [Nd,E,H]
        mask = (dst.view(1, E) == torch.arange(Nd).view(Nd, 1))         # [Nd,E]
        logits = logits.masked_fill(~mask.unsqueeze(-1), float("-inf"))
        has = mask.any(dim=1)
        p = torch.zeros_like(logits)
        if E:
            p[has] = torch.softmax(logits[has], dim=1)
        m = torch.einsum("weh,ehc->whc", p, v[src]).reshape(Nd, H * dk)  # rows without in-edges stay 0
        sums[d] = m if d not in sums else sums[d] + m
        counts[d] = counts.get(d, 0) + 1
    out = {}
    for nt in G.ntypes:
        if nt not in sums:
            out[nt] = feat_dict[nt]
            continue
        t = sums[nt] / counts[nt]
        A = layer.a_linears[nd[nt]]
        alpha = torch.sigmoid(layer.skip[nd[nt]])
        out[nt] = alpha * (t @ A.weight.t() + A.bias) + (1 - alpha) * feat_dict[nt]
    return out
