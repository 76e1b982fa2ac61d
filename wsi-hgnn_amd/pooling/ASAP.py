"""ASAPPooling — mirror of the reference's ``pooling/ASAP.py`` (LEConv :20-61, StAS + graph_connectivity :68-117,
ASAPPooling :120-199) without torch_scatter / torch_sparse / torch_geometric.

The reference never constructs this class (commented out of ``pooling/__init__.py:1,7``) and it needs three
packages that are absent here, so its semantics follow PyG 2.0.x as written down in SURVEY Appendix A.6.  Dense work
(``lin_q``, ``gat_att`` split into two per-node GEMVs, GCNConv / LEConv projections) runs on the MFMA GEMM; the edge
work of the forward (:158-179: scatter_max of gathered rows, per-target softmax of the attention logits, weighted
neighbour sum) and its backward run on the CSR/CSC kernels of csrc/asap.hip, and the neighbour sums of GCNConv / LEConv on
``wsi_spmm_sum`` whenever the edge weights are all one (always, the way the class is called: ``edge_weight=None``); the
per-graph top-k is ``wsi_graph_topk`` (rank by counting, no sort) and the S^T A S product of ``graph_connectivity`` is
``wsi_stas`` (path walk + fixed-point integer-atomic accumulation); explicit edge weights ride on the same kernels as per-edge
constants (``ops.EdgeCSR.with_weights``); only a row wider than the kernel's hash table, attention dropout in training and CPU tensors
take the PyTorch formulation.  One CSR/CSC of the self-looped edge list serves all of them.
Same constructor / forward signature and parameter names as the reference (``lin_q``, ``gat_att``, ``gnn_score.{lin1,
lin2,weight}``, ``gnn_intra_cluster.{lin.weight,bias}``).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


# ----------------------------------------------------------------------------- small PyG-utility restatements
def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask], (None if edge_attr is None else edge_attr[mask])


def add_remaining_self_loops(edge_index, edge_attr=None, fill_value=1.0, num_nodes=None):
    """PyG 2.0: non-loop edges (original order) followed by one loop per node 0..N-1; an existing loop keeps its weight."""
    n = int(num_nodes)
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop = torch.arange(n, dtype=row.dtype, device=row.device)
    if edge_attr is not None:
        loop_attr = edge_attr.new_full((n,), fill_value)
        inv = ~mask
        loop_attr[row[inv]] = edge_attr[inv]
        edge_attr = torch.cat([edge_attr[mask], loop_attr], dim=0)
    edge_index = torch.cat([edge_index[:, mask], torch.stack([loop, loop])], dim=1)
    return edge_index, edge_attr


def segment_softmax(src, index, num_nodes):
    """PyG ``softmax(src, index)``: exp(src - max) / (sum + 1e-16) per group."""
    mx = torch.full((num_nodes,) + src.shape[1:], float("-inf"), dtype=src.dtype, device=src.device)
    mx = mx.scatter_reduce(0, index.view(-1, *([1] * (src.dim() - 1))).expand_as(src), src, reduce="amax", include_self=True)
    out = torch.exp(src - mx[index])
    den = torch.zeros((num_nodes,) + src.shape[1:], dtype=src.dtype, device=src.device).index_add_(0, index, out)
    return out / (den[index] + 1e-16)


def topk(x, ratio, batch, num_per_graph=None):
    """PyG ``topk``: per graph the ceil(ratio*n) highest scores, graphs in order, descending score inside a graph.
    On the GPU: the rank-by-counting kernel ``wsi_graph_topk`` (no sort).  ``num_per_graph`` (host list of node counts per
    graph) avoids the one device->host read the output size otherwise needs."""
    if x.is_cuda:
        from .. import _native as N
        from ..graph import host_to_device
        if num_per_graph is None:
            B = int(batch.max().item()) + 1 if batch.numel() else 0
            num_per_graph = torch.bincount(batch, minlength=B).tolist()
        n_per = torch.tensor([int(c) for c in num_per_graph], dtype=torch.int64)
        k = torch.ceil(ratio * n_per.to(x.dtype)).to(torch.int64)           # the arithmetic PyG uses (fp32 product, then ceil)
        start = [0]
        for kk in k.tolist():
            start.append(start[-1] + int(kk))
        perm = torch.empty(start[-1], dtype=torch.int64, device=x.device)
        xs = x.detach().to(torch.float32).contiguous()
        bt = batch.to(torch.int64).contiguous()
        rank_ws = torch.empty(max(xs.numel(), 1), dtype=torch.int32, device=x.device)
        N.check(N.load().wsi_graph_topk(N.ptr(xs), N.ptr(bt), xs.numel(), len(start) - 1,
                                        N.ptr(host_to_device(start, torch.int64, x.device)), N.ptr(rank_ws), N.ptr(perm), N.stream()), "wsi_graph_topk")
        return perm
    B = int(batch.max().item()) + 1 if batch.numel() else 0
    n_per = torch.bincount(batch, minlength=B)
    k = torch.ceil(ratio * n_per.to(x.dtype)).to(torch.long)
    # sort by (graph asc, score desc); stable so equal scores keep node order
    order = torch.sort(x, descending=True, stable=True).indices
    order = order[torch.sort(batch[order], stable=True).indices]
    start = torch.cumsum(n_per, 0) - n_per
    rank = torch.arange(x.numel(), device=x.device) - start[batch[order]]
    return order[rank < k[batch[order]]]


class LEConv(nn.Module):
    """pooling/ASAP.py:20-61."""

    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin1 = nn.Linear(in_channels, out_channels, bias=bias)
        self.lin2 = nn.Linear(in_channels, out_channels, bias=bias)
        self.weight = nn.Parameter(torch.Tensor(in_channels, out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.in_channels)                          # torch_geometric.nn.inits.uniform
        self.weight.data.uniform_(-bound, bound)
        self.lin1.reset_parameters()
        self.lin2.reset_parameters()

    def forward(self, x, edge_index, edge_weight=None, size=None, looped_csr=None):
        """``looped_csr`` (optional, not in the reference signature): ops.EdgeCSR of ``edge_index`` when the caller knows it holds
        every non-loop edge plus EXACTLY one self loop per node (what ASAPPooling passes on): the sum over the non-loop edges is
        then the sum over all edges minus the node's own row, and no second edge sort is needed."""
        n = x.shape[0]
        if x.is_cuda:
            # x W (:48), lin1(x) and lin2(x) (:59) read the same rows: ONE projection onto [W^T; lin1.weight; lin2.weight] (3 output
            # columns when out_channels == 1, the fitness score of ASAPPooling) instead of a GEMV through rocBLAS and two more launches
            oc = self.out_channels
            wcat = torch.cat([self.weight.t(), self.lin1.weight, self.lin2.weight], dim=0)
            zero = torch.zeros(oc, dtype=x.dtype, device=x.device)
            bcat = torch.cat([zero, self.lin1.bias if self.lin1.bias is not None else zero, self.lin2.bias if self.lin2.bias is not None else zero])
            y = ops.linear(x, wcat, bcat)
            h, l1, l2 = y[:, :oc], y[:, oc:2 * oc], y[:, 2 * oc:]
        else:
            h = torch.matmul(x, self.weight)                                                                               # :48
            l1 = l2 = None
        if looped_csr is not None and x.is_cuda:
            # all edges minus the node's own loop (weight 1, or what ``loop_weight`` says when the caller's edge list came with weighted loops)
            lw = getattr(looped_csr, "loop_weight", None)
            lw = torch.ones(n, 1, dtype=x.dtype, device=x.device) if lw is None else lw.view(-1, 1).to(x.dtype)
            deg = _weighted_degree(looped_csr, n, x.dtype, x.device).view(-1, 1) - lw                                       # :54-55
            aggr = ops.graph_conv_aggregate(h.contiguous(), None, looped_csr, False) - lw * h                               # :57-58
            return (deg * l1 + aggr) + l2                                                                                   # :59
        unit = edge_weight is None
        if edge_weight is None:
            edge_weight = torch.ones(edge_index.size(1), dtype=x.dtype, device=x.device)
        edge_index, edge_weight = remove_self_loops(edge_index, edge_weight)                                                # :54
        if x.is_cuda:
            ec = ops.EdgeCSR(edge_index[0], edge_index[1], n)
            if not unit:
                ec = ec.with_weights(edge_weight)
            deg = _weighted_degree(ec, n, x.dtype, x.device)                                                                # :55
            aggr = ops.graph_conv_aggregate(h.contiguous(), None, ec, False)                                                # :57-58
        else:
            deg = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, edge_index[0], edge_weight)                  # :55
            aggr = torch.zeros(n, h.shape[1], dtype=x.dtype, device=x.device).index_add_(0, edge_index[0], edge_weight.view(-1, 1) * h[edge_index[1]])  # :57-58
        if l1 is None:
            l1 = ops.linear(x, self.lin1.weight, self.lin1.bias)
            l2 = ops.linear(x, self.lin2.weight, self.lin2.bias)
        return (deg.view(-1, 1) * l1 + aggr) + l2                                                                           # :59


class _CsrView:
    """rowptr/src/colptr/csc_dst (+ norms) as ops.graph_conv_aggregate reads them."""
    __slots__ = ("rowptr", "src", "colptr", "csc_dst", "in_norm", "out_norm", "edge_w", "edge_w_csc", "num_nodes", "num_edges")


def _transposed(ec):
    """The same edge list grouped the other way round: no sort, the CSR and CSC sides of an ops.EdgeCSR swap roles."""
    v = _CsrView()
    v.rowptr, v.src, v.colptr, v.csc_dst = ec.colptr, ec.csc_dst, ec.rowptr, ec.src
    v.in_norm = v.out_norm = None
    v.edge_w, v.edge_w_csc = getattr(ec, "edge_w_csc", None), getattr(ec, "edge_w", None)      # per-edge weights swap sides with the grouping
    v.num_nodes, v.num_edges = ec.num_nodes, ec.num_edges
    return v


def _weighted_degree(ec, n, dtype, device):
    """sum of the edge weights of every group of ``ec`` (its plain in-degree when it carries none): the weighted sum on ``wsi_spmm_sum`` over a
    column of ones - a fixed summation order, where an index_add_ of the weights would add them atomically."""
    if getattr(ec, "edge_w", None) is None:
        return (ec.rowptr[1:] - ec.rowptr[:-1]).to(dtype)
    plain = _CsrView()
    plain.rowptr, plain.src, plain.colptr, plain.csc_dst = ec.rowptr, ec.src, ec.colptr, ec.csc_dst
    plain.in_norm = plain.out_norm = None
    plain.edge_w, plain.edge_w_csc = ec.edge_w, ec.edge_w_csc
    plain.num_nodes, plain.num_edges = ec.num_nodes, ec.num_edges
    with torch.no_grad():
        return ops.graph_conv_aggregate(torch.ones(n, 1, dtype=torch.float32, device=device), None, plain, False).view(-1).to(dtype)


class GCNConv(nn.Module):
    """torch_geometric.nn.GCNConv (2.0.x): ``lin`` (no bias, glorot) + ``bias`` (zeros); symmetric normalisation with
    remaining self loops; messages flow edge_index[0] -> edge_index[1]."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.lin.weight)
        nn.init.zeros_(self.bias)

    def forward(self, x, edge_index, edge_weight=None, looped_csr=None):
        """``looped_csr`` (optional): ops.EdgeCSR(edge_index[0], edge_index[1]) of an edge list that already holds its remaining
        self loops; its CSC side IS the grouping by target this convolution needs, so it is reused instead of sorting again."""
        n = x.shape[0]
        if looped_csr is not None and x.is_cuda:
            # (explicit edge weights ride on the CSR - ops.EdgeCSR.with_weights -: norm = dis[row] * w * dis[col] with deg = sum of w per target)
            ec = _transposed(looped_csr)                                  # group by target (edge_index[1]), gather source
            deg = _weighted_degree(ec, n, x.dtype, x.device)
            dis = deg.pow(-0.5)
            ec.in_norm = ec.out_norm = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis).contiguous()
            return ops.graph_conv_aggregate(ops.linear(x, self.lin.weight, None), self.bias, ec, False)
        unit = edge_weight is None
        if edge_weight is None:
            edge_weight = torch.ones(edge_index.size(1), dtype=x.dtype, device=x.device)
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, 1.0, n)
        row, col = edge_index[0], edge_index[1]
        h = ops.linear(x, self.lin.weight, None)
        if x.is_cuda:                 # norm = dis[row] * w * dis[col]: node scales + per-edge weights -> the GraphConv gather kernel
            ec = ops.EdgeCSR(col, row, n)
            if not unit:
                ec = ec.with_weights(edge_weight)
            deg = _weighted_degree(ec, n, x.dtype, x.device)
            dis = deg.pow(-0.5)
            ec.in_norm = ec.out_norm = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis).contiguous()
            return ops.graph_conv_aggregate(h, self.bias, ec, False)
        deg = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, col, edge_weight)
        dis = deg.pow(-0.5)
        dis = torch.where(torch.isinf(dis), torch.zeros_like(dis), dis)
        norm = dis[row] * edge_weight * dis[col]
        out = torch.zeros_like(h).index_add_(0, col, norm.view(-1, 1) * h[row])
        return out + self.bias


def graph_connectivity(device, perm, edge_index, edge_weight, score, ratio, batch, N):
    """pooling/ASAP.py:84-117 (with ``StAS`` :68-81 folded in) as three sparse-matrix products — the general path (explicit edge
    weights, or a row too wide for ``wsi_stas``).  S [N, kN] holds the detached attention score of every edge whose centre was
    selected (row = neighbour, column = pooled index of the centre), A [N, N] the edge weights (1 when none are given);
    duplicates sum on ``coalesce``.  E = S^T (A S); its diagonal is dropped and one unit self loop per pooled node appended."""
    kN = int(perm.size(0))
    dev = edge_index.device
    pooled = torch.full((N,), -1, dtype=torch.long, device=dev)
    pooled[perm] = torch.arange(kN, device=dev)
    centre, nbr = edge_index[0], edge_index[1]
    chosen = pooled[centre] >= 0                                                                   # :91-102
    vals = score.detach().reshape(-1)                                                              # :97
    S = torch.sparse_coo_tensor(torch.stack([nbr[chosen], pooled[centre[chosen]]]), vals[chosen], (N, kN)).coalesce()
    w = vals.new_ones(edge_index.size(1)) if edge_weight is None else edge_weight.reshape(-1).to(vals.dtype)
    A = torch.sparse_coo_tensor(edge_index, w, (N, N)).coalesce()
    E = torch.sparse.mm(S.t().coalesce(), torch.sparse.mm(A, S)).coalesce()                        # :71-78
    idx, val = E.indices(), E.values()
    off = idx[0] != idx[1]                                                                         # :113
    loop = torch.arange(kN, device=dev)
    return (torch.cat([idx[:, off], torch.stack([loop, loop])], dim=1),                            # :114-115
            torch.cat([val[off], val.new_ones(kN)]))


def graph_connectivity_native(ec, score, perm, N):
    """pooling/ASAP.py:84-117 on the HIP kernel ``wsi_stas`` (csrc/asap.hip): E = S^T A S walked off the
    CSR/CSC of the edge list ``ec`` (ops.EdgeCSR of the self-looped edges; A = the weights it carries, 1 without) with ``score`` [E] in ORIGINAL edge order (as
    ``ops.asap_attend`` returns it).  Returns (index_E, value_E) in the reference's order — coalesced non-loop entries, then one
    unit self loop per pooled node — or None when a row has more distinct neighbours than the kernel's hash table holds
    (the caller then takes the sparse-matrix path).  One device->host read (the output size), as the reference's spspmm."""
    from .. import _native as Nn
    lib = Nn.load()
    dev = perm.device
    kN = int(perm.numel())
    n_idx = torch.full((N,), -1, dtype=torch.int32, device=dev)
    n_idx[perm] = torch.arange(kN, dtype=torch.int32, device=dev)
    score_csr = score.detach().reshape(-1).to(torch.float32)[ec.perm].contiguous()                    # :97 value_S is detached
    row_count = torch.empty(max(kN, 1), dtype=torch.int32, device=dev)
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    permc = perm.to(torch.int64).contiguous()
    args = (kN, Nn.ptr(permc), Nn.ptr(n_idx), Nn.ptr(ec.rowptr), Nn.ptr(ec.src), Nn.ptr(score_csr), Nn.ptr(getattr(ec, "edge_w", None)),
            Nn.ptr(ec.colptr), Nn.ptr(ec.csc_eid), Nn.ptr(ec.csc_dst))
    Nn.check(lib.wsi_stas(0, *args, Nn.ptr(row_count), None, None, None, Nn.ptr(overflow), Nn.stream()), "wsi_stas(count)")
    counts = row_count[:kN].to(torch.int64)
    total, over = torch.stack([counts.sum(), overflow[0].to(torch.int64)]).tolist()
    if over:
        return None
    row_start = torch.zeros(kN + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=row_start[1:])
    out_col = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    out_val = torch.empty(max(total, 1), dtype=torch.float32, device=dev)
    Nn.check(lib.wsi_stas(1, *args, Nn.ptr(row_count), Nn.ptr(row_start), Nn.ptr(out_col), Nn.ptr(out_val), Nn.ptr(overflow), Nn.stream()),
             "wsi_stas(fill)")
    rows = torch.repeat_interleave(torch.arange(kN, device=dev), counts, output_size=total)
    loop = torch.arange(kN, dtype=torch.int64, device=dev)
    index_E = torch.cat([torch.stack([rows, out_col[:total]]), torch.stack([loop, loop])], dim=1)      # :113-115
    value_E = torch.cat([out_val[:total], torch.ones(kN, dtype=torch.float32, device=dev)])
    return index_E, value_E


class _TakeRows(torch.autograd.Function):
    """``x[perm]`` for a ``perm`` WITHOUT repeated entries (topk's output): the backward is a plain scatter of rows into zeros (index_copy_) instead of
    the sorting, accumulating index_put_ autograd uses for an arbitrary index (a radix sort + the accumulate kernel per use)."""

    @staticmethod
    def forward(ctx, x, perm):
        ctx.save_for_backward(perm)
        ctx.shape = x.shape
        return x.index_select(0, perm)

    @staticmethod
    def backward(ctx, g):
        (perm,) = ctx.saved_tensors
        return g.new_zeros(ctx.shape).index_copy_(0, perm, g.contiguous()), None


class ASAPPooling(nn.Module):
    def __init__(self, in_channels, ratio=0.8, dropout_att=0, negative_slope=0.2):
        super().__init__()
        self.in_channels = in_channels
        self.ratio = ratio
        self.negative_slope = negative_slope
        self.dropout_att = dropout_att
        self.lin_q = nn.Linear(in_channels, in_channels)
        self.gat_att = nn.Linear(2 * in_channels, 1)
        self.gnn_score = LEConv(self.in_channels, 1)
        self.gnn_intra_cluster = GCNConv(self.in_channels, self.in_channels)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin_q.reset_parameters()
        self.gat_att.reset_parameters()
        self.gnn_score.reset_parameters()
        self.gnn_intra_cluster.reset_parameters()

    def _edge_csr(self, i, j, N, edge_index):
        """CSR/CSC of the (self-looped) edge list, kept while the caller passes the SAME edge_index tensor object, unmodified
        (a resident batch is pooled every step: two device sorts per call otherwise).  The cache holds a reference to that
        tensor, so its storage cannot be recycled for another edge list behind our back."""
        hit = self.__dict__.get("_ec_cache")
        if hit is not None and hit[0] is edge_index and hit[1] == edge_index._version and hit[2] == int(N):
            return hit[3]
        ec = ops.EdgeCSR(i, j, N)
        self.__dict__["_ec_cache"] = (edge_index, edge_index._version, int(N), ec)
        return ec

    def forward(self, x, edge_index, edge_weight=None, batch=None, num_per_graph=None, need_connectivity=True):
        """``num_per_graph`` (not in the reference signature, optional): host list of node counts per graph; saves the
        device->host read ``topk`` otherwise needs for its output size.  ``need_connectivity=False`` (optional, not in the
        reference signature): the caller consumes only the pooled features - the pooled graph's edges (:189-197, E = S^T A S: the
        most expensive part of the layer, and a device->host read for its size) are not computed and the second and third
        results are None."""
        if batch is None:
            batch = edge_index.new_zeros(x.size(0))
        x = x.unsqueeze(-1) if x.dim() == 1 else x
        N = x.size(0)
        unit = edge_weight is None
        edge_index_in = edge_index
        edge_index, edge_weight = add_remaining_self_loops(edge_index, edge_weight, 1.0, N)        # ASAP.py:151-152
        i, j = edge_index[0], edge_index[1]
        F_ = self.in_channels
        native = x.is_cuda and not (self.training and self.dropout_att > 0)
        ec = self._edge_csr(i, j, N, edge_index_in) if x.is_cuda else None        # one CSR/CSC of the looped edge list serves every step below
        shared = ec
        if ec is not None and not unit:
            shared = ec.with_weights(edge_weight)                        # explicit weights ride on the same CSR / CSC (constants: ASAP.py:97 detaches them)
            shared.loop_weight = edge_weight[-N:]                        # add_remaining_self_loops appends the N loops last, in node order
        x_pool = self.gnn_intra_cluster(x, edge_index, None if unit else edge_weight, looped_csr=shared)   # :157 (fill value 1 == default weight)
        if native:
            X_q = ops.csr_gather_max(x_pool, ec)                                                   # :158,163 scatter_max(x_pool[j], i)
        else:
            x_pool_j = x_pool[j]                                                                   # :158
            X_q = torch.full((N, x.size(1)), float("-inf"), dtype=x.dtype, device=x.device)
            X_q = X_q.scatter_reduce(0, i.view(-1, 1).expand_as(x_pool_j), x_pool_j, reduce="amax", include_self=True)   # :163
        M_q = ops.linear(X_q, self.lin_q.weight, self.lin_q.bias)                                  # :165
        # gat_att(cat(M_q[i], x_pool[j])) = M_q[i].w1 + x_pool[j].w2 + b : two per-node projections, then a per-edge add (:167-169)
        a = ops.linear(M_q, self.gat_att.weight[:, :F_].contiguous(), self.gat_att.bias)
        b = ops.linear(x_pool, self.gat_att.weight[:, F_:].contiguous(), None)
        if native:
            out, score = ops.asap_attend(a, b, x, ec, self.negative_slope)                         # :170-179 fused
            score = score.view(-1, 1)
        else:
            score = F.leaky_relu(a[i] + b[j], self.negative_slope)                                 # :170
            score = segment_softmax(score, i, N)                                                   # :171
            score = F.dropout(score, p=self.dropout_att, training=self.training)                   # :174
            out = torch.zeros_like(x).index_add_(0, i, x[j] * score.view(-1, 1))                   # :176-179
        # :183 calls gnn_score(x=out, edge_index=edge_index) WITHOUT edge_weight (so does PyG's ASAPooling): the fitness LEConv sees unit weights
        # even when the pooling got explicit ones - the unweighted CSR (loop weight 1), not ``shared``
        fitness = torch.sigmoid(self.gnn_score(out, edge_index, None, looped_csr=ec)).view(-1)                                     # :183
        perm = topk(fitness, self.ratio, batch, num_per_graph)                                     # :184
        x = _TakeRows.apply(out, perm) * _TakeRows.apply(fitness, perm).view(-1, 1)                # :185
        batch = batch[perm]                                                                        # :188
        if not need_connectivity:
            return x, None, None, batch, perm
        conn = graph_connectivity_native(shared, score, perm, N) if native else None                     # :189-197 on wsi_stas
        if conn is None:
            conn = graph_connectivity(x.device, perm, edge_index, None if unit else edge_weight, score, self.ratio, batch, N)
        edge_index, edge_weight = conn
        return x, edge_index, edge_weight, batch, perm

    def __repr__(self):
        return "{}({}, ratio={})".format(self.__class__.__name__, self.in_channels, self.ratio)
