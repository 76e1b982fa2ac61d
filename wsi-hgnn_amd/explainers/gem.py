"""GEM-style causal node attribution (https://arxiv.org/abs/2104.06643) as the reference implements it: the contribution of a
node is the change of the loss when the model is run on the graph WITHOUT that node — one forward per node
(``explainers/GEM.py:22-55`` for homogeneous graphs, ``explainers/gem_het.py:25-43`` for heterogeneous ones).

The reference removes one node, rebuilds a DGL graph and calls the model, N times (GEM.py batches 10 altered graphs per call).
Here the N altered graphs go through the same batched engine as training: ``graph.remove_nodes`` + ``graph.batch`` build
block-diagonal batches of ``batch_size`` altered graphs and every batch is ONE forward under ``no_grad`` — an inference stress of
N forwards' worth of work in N / batch_size launches sequences (second half).  Same constructor signatures,
method names and return values as the reference classes.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

from ..graph import HeteroGraph, batch as batch_graphs, remove_nodes


class GemExplainer:
    """explainers/GEM.py:14-55 (homogeneous graph: one node type, one relation)."""

    def __init__(self, graph: HeteroGraph, model: nn.Module, label, batch_size: int = 10):
        self.graph = graph
        self.label = label
        self.gnn = model
        self.batch_size = int(batch_size)                   # GEM.py:29
        self.loss_fcn = nn.CrossEntropyLoss()

    def explain_node(self):
        temp = 40                                           # GEM.py:24 temperature (only used for the reference's unused `loss`)
        g = self.graph
        ntype = g.ntypes[0]
        n = g.num_nodes()
        with torch.no_grad():
            pred = self.gnn(g)                              # :25
            _ = self.loss_fcn(pred / temp, self.label)      # :26 (computed and never used there either)
            node_mask = torch.zeros(n)
            lf = nn.CrossEntropyLoss(reduction="none")
            for start in range(0, n, self.batch_size):      # :31-50
                end = min(start + self.batch_size, n)
                bg = batch_graphs([remove_nodes(g, torch.tensor([nid]), ntype) for nid in range(start, end)])   # :38-40
                pred_alt = self.gnn(bg)                     # :43
                lb = torch.ones(end - start, dtype=torch.long, device=pred.device) * int(self.label)            # :46
                node_mask[start:end] = lf(pred - pred_alt, lb).cpu()                                            # :48-50
        m = node_mask.numpy()
        return (m - m.min()) / (m.max() - m.min())          # :53-54


class HetGemExplainer:
    """explainers/gem_het.py:12-43.  The reference first collapses every relation into one edge type 'pos' per (source type,
    destination type) pair (``to_homogeneous`` -> ``edata['_TYPE'] *= 0`` -> ``to_heterogeneous(etypes=['pos'])``, :15-18); the
    same collapse is applied here, so a model explained with this class must have been built for that schema, as there."""

    def __init__(self, graph: HeteroGraph, model: nn.Module, label, batch_size: int = 16):
        self.graph = collapse_relations(graph)
        self.label = label
        self.gnn = model
        self.batch_size = int(batch_size)
        self.loss_fcn = nn.CrossEntropyLoss()

    def explain_node(self) -> Dict[str, torch.Tensor]:
        g = self.graph
        node_mask = {t: torch.zeros(g.num_nodes(t)) for t in g.ntypes}                                  # :28
        with torch.no_grad():
            loss = self.loss_fcn(self.gnn(g), self.label)                                               # :26-27
            lf = nn.CrossEntropyLoss(reduction="none")
            for t in g.ntypes:                                                                          # :30
                n = g.num_nodes(t)
                for start in range(0, n, self.batch_size):                                              # :31 (one node per forward there)
                    end = min(start + self.batch_size, n)
                    bg = batch_graphs([remove_nodes(g, torch.tensor([i]), t) for i in range(start, end)])   # :35
                    pred_alt = self.gnn(bg)                                                             # :36
                    lb = self.label.to(pred_alt.device).reshape(-1)[:1].expand(end - start)
                    node_mask[t][start:end] = (loss - lf(pred_alt, lb)).cpu()                           # :37-39 (CE of a single graph = its row)
        return node_mask


def collapse_relations(g: HeteroGraph) -> HeteroGraph:
    """One relation ('pos') per (source type, destination type) pair holding the edges of every original relation between the
    two types, original relations in canonical order (explainers/gem_het.py:15-18)."""
    from collections import OrderedDict
    edges, sims = OrderedDict(), {}
    for (s, e, d) in g.canonical_etypes:
        u, v = g.edges((s, e, d))
        key = (s, "pos", d)
        edges.setdefault(key, ([], []))
        edges[key][0].append(u)
        edges[key][1].append(v)
        if "sim" in g._eframes[(s, e, d)]:
            sims.setdefault(key, []).append(g._eframes[(s, e, d)]["sim"])
    out = HeteroGraph(OrderedDict((t, g.num_nodes(t)) for t in g.ntypes),
                      OrderedDict((k, (torch.cat(us), torch.cat(vs))) for k, (us, vs) in edges.items()))
    for t in g.ntypes:
        for k, x in g._nframes[t].items():
            out._nframes[t][k] = x
    for k, parts in sims.items():
        out._eframes[k]["sim"] = torch.cat(parts)
    return out
