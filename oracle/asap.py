"""CPU restatement of the reference's ``pooling/ASAP.py`` (TEST INFRASTRUCTURE — see oracle/__init__.py).

Follows ASAP.py:20-199 with the PyG-2.0.x / torch_scatter / torch_sparse calls replaced by DENSE linear algebra
(dense adjacency, dense S, ``S.T @ A @ S``), i.e. deliberately a different formulation from the product's sparse one.
PARITY UNPINNED: torch_geometric / torch_scatter / torch_sparse are absent; semantics per SURVEY Appendix A.6.
Tiny graphs only (O(N^2) memory).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def asap_forward(mod, x, edge_index, batch=None, edge_weight=None):
    """``mod`` holds the parameters (product ``ASAPPooling`` or anything with the same attribute names).
    ``edge_weight`` None is the reference's only use; explicit weights follow PyG 2.0.x: ``add_remaining_self_loops`` (ASAP.py:151-152) keeps the
    weight of a loop the input already holds and gives the added ones 1, GCNConv normalises ``dis[row] * w * dis[col]`` with ``deg = sum of w`` per
    target, the fitness LEConv (:45-61, called at :183 WITHOUT edge_weight) drops the loops and uses unit weights, and A of ``S^T A S`` (:68-81) holds the weights.
    Returns (x', dense E [kN,kN] incl. structural mask, batch', perm)."""
    N, Fd = x.shape
    if batch is None:
        batch = torch.zeros(N, dtype=torch.long)
    # adjacency as COUNT matrix over (i = edge_index[0], j = edge_index[1]) after add_remaining_self_loops (ASAP.py:151)
    i0, j0 = edge_index
    nl = i0 != j0
    ei = torch.cat([edge_index[:, nl], torch.arange(N).repeat(2, 1)], dim=1)
    i, j = ei
    # --- GCNConv (ASAP.py:157): self loops already present; messages j_src=ei[0] -> ei[1]
    if edge_weight is None:
        ew = torch.ones(ei.shape[1], dtype=x.dtype)
    else:
        loop_w = torch.ones(N, dtype=x.dtype)
        loop_w[i0[~nl]] = edge_weight[~nl].to(x.dtype)                    # an existing loop keeps its weight (the last one wins, as index assignment does)
        ew = torch.cat([edge_weight[nl].to(x.dtype), loop_w])
    deg = torch.zeros(N, dtype=x.dtype).index_add_(0, j, ew)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0
    h = x @ mod.gnn_intra_cluster.lin.weight.t()
    An = torch.zeros(N, N, dtype=x.dtype)
    An.index_put_((j, i), dis[i] * ew * dis[j], accumulate=True)         # row = target ei[1], col = source ei[0]
    x_pool = An @ h + mod.gnn_intra_cluster.bias
    # --- master query (:163-167): X_q[i] = max over neighbours j of x_pool[j]
    M = torch.zeros(N, N, dtype=torch.bool)
    M[i, j] = True
    big = x_pool.unsqueeze(0).expand(N, N, Fd).masked_fill(~M.unsqueeze(-1), float("-inf"))
    X_q = big.max(dim=1).values
    M_q = X_q @ mod.lin_q.weight.t() + mod.lin_q.bias
    # --- attention (:169-171), per edge (parallel edges keep separate entries)
    sc = torch.cat([M_q[i], x_pool[j]], dim=-1) @ mod.gat_att.weight.t() + mod.gat_att.bias
    sc = F.leaky_relu(sc, mod.negative_slope).view(-1)
    mx = torch.full((N,), float("-inf"), dtype=x.dtype).scatter_reduce(0, i, sc, reduce="amax")
    ex = torch.exp(sc - mx[i])
    score = ex / (torch.zeros(N, dtype=x.dtype).index_add_(0, i, ex)[i] + 1e-16)
    out = torch.zeros_like(x).index_add_(0, i, x[j] * score.view(-1, 1))           # :176-179
    # --- LEConv fitness (:183, LEConv :45-61): self loops removed
    g = mod.gnn_score
    hh = out @ g.weight
    k2 = i != j
    # (ASAP.py:183 passes NO edge_weight to gnn_score: unit weights here even when the pooling itself got explicit ones)
    degl = torch.zeros(N, dtype=x.dtype).index_add_(0, i[k2], torch.ones_like(ew[k2]))
    aggr = torch.zeros(N, hh.shape[1], dtype=x.dtype).index_add_(0, i[k2], hh[j[k2]])
    fit = degl.view(-1, 1) * (out @ g.lin1.weight.t() + g.lin1.bias) + aggr + (out @ g.lin2.weight.t() + g.lin2.bias)
    fitness = torch.sigmoid(fit).view(-1)
    # --- top-k per graph (:184)
    perm = []
    for b in range(int(batch.max()) + 1):
        idx = (batch == b).nonzero().view(-1)
        k = int(math.ceil(mod.ratio * idx.numel()))
        order = torch.sort(fitness[idx], descending=True, stable=True).indices[:k]
        perm.append(idx[order])
    perm = torch.cat(perm)
    x_new = out[perm] * fitness[perm].view(-1, 1)
    # --- connectivity (:84-117): S[j, c] = score of edge (i=perm[c], j); A = edge weights (ones) incl. loops; E = S^T A S
    kN = perm.numel()
    n_idx = torch.full((N,), -1, dtype=torch.long)
    n_idx[perm] = torch.arange(kN)
    S = torch.zeros(N, kN, dtype=x.dtype)
    Sm = torch.zeros(N, kN, dtype=torch.bool)
    sel = n_idx[i] >= 0
    S.index_put_((j[sel], n_idx[i[sel]]), score[sel].detach(), accumulate=True)
    Sm[j[sel], n_idx[i[sel]]] = True
    A = torch.zeros(N, N, dtype=x.dtype).index_put_((i, j), ew, accumulate=True)
    Am = torch.zeros(N, N, dtype=torch.bool)
    Am[i, j] = True
    E = S.t() @ A @ S
    Em = (Sm.t().to(x.dtype) @ Am.to(x.dtype) @ Sm.to(x.dtype)) > 0            # structural non-zeros of the sparse product
    eye = torch.eye(kN, dtype=torch.bool)
    E = E.masked_fill(eye, 0.0) + torch.eye(kN, dtype=x.dtype)                # remove self loops, add remaining with weight 1
    Em = (Em & ~eye) | eye
    return x_new, E, Em, batch[perm], perm
