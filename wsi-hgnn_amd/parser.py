"""Config -> objects, mirroring the reference's ``parser.py`` (same function names, same config keys, same error behaviour):

  parse_gnn_model(cfg["GNN"])      parser.py:48-174   the in-scope branches: GCN, GCN_NTPool, HetRGCN, HGT, HEAT2, HEAT4
  parse_optimizer(cfg["optim"], m) parser.py:15-46    adagrad / adadelta / adam / anything else -> SGD
  parse_loss(cfg["train"])         parser.py:176-184  BCE / CE

``node_dict`` and the etype-major enumeration of ``edge_dict`` / ``etypes`` are built exactly as :107-113,122-129; a config
without a key the reference reads raises the same ``KeyError``; names the reference does not know (``HEAT``, ``HEAT3`` of
configs/COAD/HEAT*_staging.yml: orphan configs) raise its ``NotImplementedError``.  ``GAT`` and ``GIN`` are the
reference's homogeneous baselines, outside the hot path: they raise ``NotImplementedError`` here too, with a message that
says so.  tests/test_parser.py replays the calls the REFERENCE function makes on its own configs (tests/golden/
reference_surface.json, produced by executing parser.py:48-174 with recording stand-ins for the classes).
"""
from __future__ import annotations

import torch.nn.functional as F
from torch import nn, optim

from .models import GCN, HGT, HEATNet2, HEATNet4, HeteroRGCN, NTPoolGCN


def parse_optimizer(config_optim, model):
    opt_method = config_optim["opt_method"].lower()
    alpha = config_optim["lr"]
    weight_decay = config_optim["weight_decay"]
    if opt_method == "adagrad":
        return optim.Adagrad(model.parameters(), lr=alpha, lr_decay=weight_decay, weight_decay=weight_decay)    # (lr_decay = weight_decay: parser.py:23)
    if opt_method == "adadelta":
        return optim.Adadelta(model.parameters(), lr=alpha, weight_decay=weight_decay)
    if opt_method == "adam":
        return optim.Adam(model.parameters(), lr=alpha, weight_decay=weight_decay)
    return optim.SGD(model.parameters(), lr=alpha, weight_decay=weight_decay)


def _typed_schema(config_gnn):
    n_node_types = config_gnn["n_node_types"]
    etypes = config_gnn["edge_types"]
    canonical_etypes = [(str(s), r, str(t)) for r in etypes for s in range(n_node_types) for t in range(n_node_types)]   # etype-major
    node_dict = {str(i): i for i in range(n_node_types)}
    return node_dict, canonical_etypes


def parse_gnn_model(config_gnn):
    gnn_name = config_gnn["name"]
    if gnn_name in ("GAT", "GIN"):
        raise NotImplementedError(f"{gnn_name} is one of the reference's homogeneous baselines, outside the hot path this package rebuilds")
    if gnn_name == "GCN":
        return GCN(in_dim=config_gnn["in_dim"], hidden_dim=config_gnn["hidden_dim"], out_dim=config_gnn["out_dim"],
                   n_layers=config_gnn["num_layers"], activation=F.relu, dropout=config_gnn["feat_drop"],
                   graph_pooling_type=config_gnn["graph_pooling_type"])
    if gnn_name == "GCN_NTPool":
        node_dict = {str(i): i for i in range(config_gnn["n_node_types"])}
        return NTPoolGCN(in_dim=config_gnn["in_dim"], hidden_dim=config_gnn["hidden_dim"], out_dim=config_gnn["out_dim"],
                         node_dict=node_dict, n_layers=config_gnn["num_layers"], activation=F.relu, dropout=config_gnn["feat_drop"],
                         graph_pooling_type=config_gnn["graph_pooling_type"])
    if gnn_name == "HetRGCN":
        node_dict, canonical_etypes = _typed_schema(config_gnn)
        etypes = {et: str(i) for i, et in enumerate(canonical_etypes)}
        return HeteroRGCN(in_dim=config_gnn["in_dim"], hidden_dim=config_gnn["hidden_dim"], out_dim=config_gnn["out_dim"],
                          n_layers=config_gnn["num_layers"], etypes=etypes, node_dict=node_dict,
                          graph_pooling_type=config_gnn["graph_pooling_type"])
    if gnn_name == "HGT":
        node_dict, canonical_etypes = _typed_schema(config_gnn)
        edge_dict = {et: i for i, et in enumerate(canonical_etypes)}
        return HGT(node_dict, edge_dict, in_dim=config_gnn["in_dim"], hidden_dim=config_gnn["hidden_dim"], out_dim=config_gnn["out_dim"],
                   n_layers=config_gnn["num_layers"], n_heads=config_gnn["num_heads"])
    if gnn_name in ("HEAT2", "HEAT4"):
        node_dict = {str(i): i for i in range(config_gnn["n_node_types"])}
        cls = HEATNet2 if gnn_name == "HEAT2" else HEATNet4
        return cls(in_dim=config_gnn["in_dim"], hidden_dim=config_gnn["hidden_dim"], out_dim=config_gnn["out_dim"],
                   n_layers=config_gnn["num_layers"], n_heads=config_gnn["n_heads"], node_dict=node_dict,
                   dropuout=config_gnn["feat_drop"], graph_pooling_type=config_gnn["graph_pooling_type"])
    raise NotImplementedError("This GNN model is not implemented")


def parse_loss(config_train):
    loss_name = config_train["loss"]
    if loss_name == "BCE":
        return nn.BCELoss()
    if loss_name == "CE":
        return nn.CrossEntropyLoss()
    raise NotImplementedError("This Loss is not implemented")
