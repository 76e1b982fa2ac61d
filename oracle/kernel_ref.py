"""Kernel-level CPU references on the *plan* layout (TEST INFRASTRUCTURE — see oracle/__init__.py).

``heat_attention_ref`` restates models/HEATNet4.py:103-119 directly on the
two-level CSR the HIP kernel consumes, so a kernel can be checked in isolation (any dtype, autograd
gives the reference gradients).  ``dgl_semantics.edge_softmax_dst`` is reused with the (node,
relation) segment id as the softmax group — exactly "softmax over in-edges of each dst within this
relation".
"""
from __future__ import annotations

import math

import torch

from . import dgl_semantics as S


def heat_attention_ref(kqv: torch.Tensor, e_weight: torch.Tensor, e_bias: torch.Tensor, plan, sim_csr: torch.Tensor,
                       D: int, H: int) -> torch.Tensor:
    """kqv [N,3D] (K|Q|V), plan tensors on CPU.  Returns t [N,D]."""
    dk = D // H
    n = plan.num_nodes
    src = plan.src.long()
    dst, seg = plan_edge_tables(plan)
    k = kqv[:, 0:D].reshape(n, H, dk)
    q = kqv[:, D:2 * D].reshape(n, H, dk)
    v = kqv[:, 2 * D:3 * D].reshape(n, H, dk)
    ea = e_weight.reshape(()) * sim_csr.to(kqv.dtype) + e_bias.reshape(())
    score = (q[dst] * k[src]).sum(-1) * ea.unsqueeze(-1) / math.sqrt(dk)      # [E,H]
    a = S.edge_softmax_dst(score, seg, plan.num_segs)
    msg = v[src] * a.unsqueeze(-1)                                             # [E,H,dk]
    t = torch.zeros(n, H, dk, dtype=kqv.dtype).index_add_(0, dst, msg)
    return t.reshape(n, D) * plan.inv_rd.to(kqv.dtype).unsqueeze(-1)


def plan_edge_tables(plan):
    """(dst [E], seg_of_edge [E]) in the plan's CSR edge order, derived from the two-level CSR (the product plan does not
    carry per-edge destination / segment ids: the kernels never need them)."""
    rowptr = plan.rowptr.long().cpu()
    node_seg = plan.node_seg.long().cpu()
    S, n = rowptr.numel() - 1, node_seg.numel() - 1
    seg = torch.repeat_interleave(torch.arange(S, dtype=torch.int64), rowptr[1:] - rowptr[:-1])
    dst = torch.repeat_interleave(torch.arange(n, dtype=torch.int64), rowptr[node_seg[1:]] - rowptr[node_seg[:-1]])
    return dst, seg


def plan_to_cpu(plan):
    """Shallow CPU copy of a GraphPlan (tensors moved to host)."""
    import copy
    p = copy.copy(plan)
    for name, val in vars(plan).items():
        if isinstance(val, torch.Tensor):
            setattr(p, name, val.cpu())
    p.device = torch.device("cpu")
    return p
