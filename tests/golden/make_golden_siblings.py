#!/usr/bin/env python
"""Oracle golden vectors for the sibling models of the path: HGT, HeteroRGCN, GCN, NTPoolGCN.
Same status as make_golden.py: PARITY UNPINNED (the build's own CPU oracle; DGL cannot be installed here).
Re-run: ``python tests/golden/make_golden_siblings.py``."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import wsi_hgnn_amd as W  # noqa: E402
from wsi_hgnn_amd import synthetic  # noqa: E402
from oracle import models as OM  # noqa: E402
from make_golden import graph_to_arrays, special_graph  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ND = {"0": 0, "1": 1, "2": 2}
RELS = [(str(s), r, str(t)) for r in ("pos", "neg") for s in range(3) for t in range(3)]      # parser.py:127-134


def build(pkg, kind):
    """The four sibling constructors with the fixture hyper-parameters (shared with tests/test_golden.py)."""
    if kind == "hgt":
        return pkg.HGT(ND, {et: i for i, et in enumerate(RELS)}, 16, 24, 2, 3, 4, use_norm=True)       # d_k = 6: padded head layout
    if kind == "hetrgcn":
        return pkg.HeteroRGCN(16, 24, 2, 3, {r: str(i) for i, r in enumerate(RELS)}, ND, "sum")      # parser.py:106-113
    if kind == "gcn":
        return pkg.GCN(16, 24, 2, 2, F.relu, 0.0, "att")
    if kind == "ntpool":
        return pkg.NTPoolGCN(16, 24, 2, ND, 2, F.relu, 0.0, "mean")
    raise KeyError(kind)


def run(name, kind, g, labels):
    torch.manual_seed(611)
    m = build(OM, kind).eval()                       # HGTLayer hard-codes Dropout(0.2): eval mode for a deterministic vector
    if kind == "hgt":
        with torch.no_grad():
            for layer in m.gcs:
                layer.skip.copy_(torch.tensor([0.3, 1.0, -0.7]))
                layer.relation_pri.uniform_(0.5, 1.5)
    out = m(g)
    loss = F.cross_entropy(out, labels)
    loss.backward()
    arr = {}
    graph_to_arrays(g, "g_", arr)
    if "_ID" in g.nodes[g.ntypes[0]].data:
        for t in g.ntypes:
            arr[f"g_id_{t}"] = g.nodes[t].data["_ID"].numpy()
    arr["kind"] = np.array(kind)
    arr["labels"] = labels.numpy()
    arr["logits"] = out.detach().numpy()
    arr["loss"] = np.array(loss.item(), dtype=np.float64)
    for k, v in m.state_dict().items():
        arr["sd_" + k] = v.numpy()
    for k, p in m.named_parameters():
        if p.grad is not None:
            arr["grad_" + k] = p.grad.numpy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arr)
    print(f"{name}: logits {out.detach().numpy().round(6).tolist()} loss {loss.item():.8f} -> {os.path.getsize(path) // 1024} KiB")


def main():
    gb = W.batch([synthetic.hetero_graph(60, 16, seed=7, dst_mode="hub"), synthetic.hetero_graph(45, 16, seed=8, dst_mode="uniform")])
    run("sibling_hgt_hub_batch2", "hgt", gb, torch.tensor([1, 0]))
    run("sibling_hgt_special", "hgt", special_graph(3), torch.tensor([1]))
    run("sibling_hetrgcn_hub_batch2", "hetrgcn", gb, torch.tensor([0, 1]))
    gh = W.batch([synthetic.homogeneous_graph(50, 16, seed=3), synthetic.homogeneous_graph(31, 16, seed=4)])
    run("sibling_gcn_att_batch2", "gcn", gh, torch.tensor([0, 1]))
    gn = W.batch([synthetic.hetero_graph(40, 16, seed=70), synthetic.hetero_graph(33, 16, seed=71, dst_mode="hub")])
    off = gn.type_offsets()
    gen = torch.Generator().manual_seed(5)
    gn.ndata["_ID"] = {t: off[i] + torch.randperm(gn.num_nodes(t), generator=gen) for i, t in enumerate(gn.ntypes)}
    run("sibling_ntpool_batch2", "ntpool", gn, torch.tensor([1, 0]))


if __name__ == "__main__":
    main()
