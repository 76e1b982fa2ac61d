#!/usr/bin/env python
"""Generate the oracle golden vectors under tests/golden/ (tiny seeded graphs).

PARITY UNPINNED: these vectors come from the build's own CPU oracle (oracle/models.py), because the
reference's arithmetic lives in DGL which cannot be installed here.  They pin the
ORACLE against drift and give the GPU tests fixed expected values; if a DGL wheel ever becomes
available, re-generate the same cases through the reference and diff.
Cases cover: hub + isolated destination nodes, parallel edges, an empty-but-present relation, an empty
node type, R_d in {1,2}, batch of 2.  Re-run: ``python tests/golden/make_golden.py``.
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import wsi_hgnn_amd as W  # noqa: E402
from wsi_hgnn_amd import synthetic  # noqa: E402
from oracle import models as OM  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ND = {"0": 0, "1": 1, "2": 2}


def graph_to_arrays(g, prefix, out):
    out[prefix + "ntypes"] = np.array(g.ntypes)
    out[prefix + "num_nodes"] = np.array([g.num_nodes(t) for t in g.ntypes])
    for i, r in enumerate(g.canonical_etypes):
        u, v = g.edges(r)
        out[f"{prefix}rel{i}_name"] = np.array(list(r))
        out[f"{prefix}rel{i}_src"] = u.numpy()
        out[f"{prefix}rel{i}_dst"] = v.numpy()
        if "sim" in g._eframes[r]:
            out[f"{prefix}rel{i}_sim"] = g._eframes[r]["sim"].numpy()
    out[prefix + "num_rels"] = np.array(len(g.canonical_etypes))
    for t in g.ntypes:
        out[f"{prefix}feat_{t}"] = g.nodes[t].data["feat"].numpy()
        out[f"{prefix}bnn_{t}"] = g.batch_num_nodes(t).numpy()


def special_graph(seed):
    """Parallel edges, an isolated dst, an empty relation, and node type '2' with zero nodes."""
    gen = torch.Generator().manual_seed(seed)
    nn_ = OrderedDict([("0", 7), ("1", 5), ("2", 0)])
    edges = OrderedDict()
    sim = {}
    edges[("0", "pos", "0")] = (torch.tensor([0, 0, 1, 2, 2, 6]), torch.tensor([1, 1, 1, 3, 3, 0]))      # parallel edges 0->1, 2->3
    edges[("1", "pos", "0")] = (torch.tensor([0, 4, 4]), torch.tensor([1, 5, 5]))
    edges[("0", "neg", "1")] = (torch.tensor([3, 5]), torch.tensor([0, 2]))
    edges[("1", "neg", "1")] = (torch.tensor([], dtype=torch.int64), torch.tensor([], dtype=torch.int64))  # empty relation (present in metagraph)
    for r, (u, v) in edges.items():
        mag = torch.rand(u.numel(), generator=gen)
        sim[r] = mag if r[1] == "pos" else -mag
    feat = {t: torch.rand(nn_[t], 16, generator=gen) for t in nn_}
    return W.HeteroGraph.from_coo(nn_, edges, feat=feat, sim=sim)


def run_case(name, model_cls, g, labels, hidden=32, heads=4, layers=2, in_dim=16, pooling="mean", grad_prefixes=None):
    torch.manual_seed(611)
    m = model_cls(in_dim, hidden, 2, layers, heads, ND, 0.0, pooling)
    with torch.no_grad():
        for layer in m.gcs:
            layer.skip.copy_(torch.tensor([0.3, 1.0, -0.7]))
            layer.e_linear.weight.fill_(0.8)
            layer.e_linear.bias.fill_(0.25)
    out = m(g)
    loss = torch.nn.functional.cross_entropy(out, labels)
    loss.backward()
    arr = {}
    graph_to_arrays(g, "g_", arr)
    arr["labels"] = labels.numpy()
    arr["logits"] = out.detach().numpy()
    arr["loss"] = np.array(loss.item(), dtype=np.float64)
    arr["config"] = np.array([in_dim, hidden, 2, layers, heads])
    arr["pooling"] = np.array(pooling)
    for k, v in m.state_dict().items():
        arr["sd_" + k] = v.numpy()
    for k, p in m.named_parameters():
        if p.grad is not None and (grad_prefixes is None or k.startswith(grad_prefixes)):
            arr["grad_" + k] = p.grad.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arr)
    print(f"{name}: logits {out.detach().numpy().round(6).tolist()} loss {loss.item():.8f} -> {os.path.getsize(path) // 1024} KiB")


def main():
    g1 = W.batch([synthetic.hetero_graph(60, 16, seed=7, dst_mode="hub"), synthetic.hetero_graph(45, 16, seed=8, dst_mode="uniform")])
    run_case("heatnet4_hub_batch2", OM.HEATNet4, g1, torch.tensor([1, 0]))
    run_case("heatnet2_hub_batch2", OM.HEATNet2, g1, torch.tensor([1, 0]))
    g2 = special_graph(3)
    # HEATNet4's fixed 256-wide head dominates the file size: keep full gradients only in the first case
    run_case("heatnet4_special", OM.HEATNet4, g2, torch.tensor([1]), grad_prefixes=("gcs.", "adapt_ws.", "head."))
    run_case("heatnet2_special_sum", OM.HEATNet2, g2, torch.tensor([0]), pooling="sum")
    run_case("heatnet2_special_max", OM.HEATNet2, g2, torch.tensor([0]), pooling="max")


if __name__ == "__main__":
    main()
