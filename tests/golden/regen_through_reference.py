#!/usr/bin/env python
"""Pin the oracle's golden vectors to the REFERENCE ITSELF - the one command that lifts "parity unpinned".

The vectors under tests/golden/ (heatnet*.npz, sibling_*.npz) were produced by the build's own CPU oracle because the reference's arithmetic
lives in DGL, which cannot be installed in the build container.  This script replays every stored case through the
reference's own modules wherever ``import dgl`` succeeds (a workstation with ``pip install dgl``; never the GPU box, never the test suite):

  * the graph of each fixture is rebuilt as a ``dgl.heterograph`` (``dgl.graph`` for the homogeneous GCN case) from the arrays the fixture stores:
    per-relation COO, ``feat`` per node type, ``sim`` per relation, the batch's per-type node counts (``set_batch_num_nodes`` /
    ``set_batch_num_edges``: per-graph edge counts follow from the destination's graph);
  * ``models/HEATNet4.py``, ``HEATNet2.py``, ``HGT.py``, ``HetRGCN.py``, ``GCN.py``, ``GCN_NTPool.py`` are loaded FROM THEIR FILES under
    ``--reference`` (default /root/reference) - ``models/__init__.py`` cannot be imported as shipped (:10 pulls in HAN / EfficientNet with their
    own dependencies), and ``pooling/__init__.py`` (which those files import) needs DGL only;
  * the stored ``state_dict`` is loaded strictly, the model is run as the fixture was (eval mode for the HGT family: HGTLayer hard-codes
    Dropout(0.2)), logits / cross-entropy loss / every stored parameter gradient are compared with the stored oracle values;
  * the report (one line per case: max |logit difference|, |loss difference|, worst relative gradient difference) is printed and, with ``--write``,
    saved as tests/golden/reference_pin.json - committing that file, produced by the real reference, is what changes DESIGN 0 from
    "unpinned" to "pinned"; ``--regenerate`` additionally rewrites the expected values in the .npz files from the reference's outputs.

Exit status: 0 = every case within --tol (default 1e-5 in float32 arithmetic), 1 = a difference (the oracle's reading of DGL is wrong somewhere:
fix oracle/dgl_semantics.py, not the tolerance), 2 = DGL or the reference is not available (nothing was checked).

    python tests/golden/regen_through_reference.py [--reference /root/reference] [--tol 1e-5] [--write] [--regenerate]
"""
import argparse
import importlib.util
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ND = {"0": 0, "1": 1, "2": 2}
RELS = [(str(s), r, str(t)) for r in ("pos", "neg") for s in range(3) for t in range(3)]      # parser.py:127-134 (shared with make_golden_siblings.py)

HEAT_CASES = {   # fixture -> reference class
    "heatnet4_hub_batch2": "HEATNet4", "heatnet4_special": "HEATNet4",
    "heatnet2_hub_batch2": "HEATNet2", "heatnet2_special_sum": "HEATNet2", "heatnet2_special_max": "HEATNet2",
}
SIBLING_CASES = ["sibling_hgt_hub_batch2", "sibling_hgt_special", "sibling_hetrgcn_hub_batch2", "sibling_gcn_att_batch2", "sibling_ntpool_batch2"]


def load_reference_module(ref_root, name):
    """models/<name>.py as a module of its own (bypasses models/__init__.py); `from pooling import ...` resolves through sys.path."""
    if ref_root not in sys.path:
        sys.path.insert(0, ref_root)
    spec = importlib.util.spec_from_file_location(f"wsi_reference_{name}", os.path.join(ref_root, "models", f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def dgl_graph_of(z, dgl):
    """The fixture's graph as DGL builds it.  Returns (graph, is_homogeneous)."""
    ntypes = [str(t) for t in z["g_ntypes"]]
    counts = {t: int(c) for t, c in zip(ntypes, z["g_num_nodes"])}
    n_rel = int(z["g_num_rels"])
    bnn = {t: torch.as_tensor(z[f"g_bnn_{t}"]).long() for t in ntypes}
    B = int(next(iter(bnn.values())).numel())
    starts = {t: torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(bnn[t], 0)]) for t in ntypes}

    def graph_of_node(t, ids):           # which graph of the batch a node of type t belongs to
        return torch.bucketize(ids, starts[t][1:], right=True)

    if ntypes == ["_N"]:
        src, dst = torch.as_tensor(z["g_rel0_src"]).long(), torch.as_tensor(z["g_rel0_dst"]).long()
        g = dgl.graph((src, dst), num_nodes=counts["_N"])
        g.ndata["feat"] = torch.as_tensor(z["g_feat__N"])
        g.set_batch_num_nodes(bnn["_N"])
        g.set_batch_num_edges(torch.bincount(graph_of_node("_N", dst), minlength=B))
        return g, True
    data, sims = {}, {}
    for i in range(n_rel):
        r = tuple(str(x) for x in z[f"g_rel{i}_name"])
        data[r] = (torch.as_tensor(z[f"g_rel{i}_src"]).long(), torch.as_tensor(z[f"g_rel{i}_dst"]).long())
        if f"g_rel{i}_sim" in z.files:
            sims[r] = torch.as_tensor(z[f"g_rel{i}_sim"])
    g = dgl.heterograph(data, num_nodes_dict=counts)
    for t in ntypes:
        g.nodes[t].data["feat"] = torch.as_tensor(z[f"g_feat_{t}"])
        if f"g_id_{t}" in z.files:
            g.nodes[t].data["_ID"] = torch.as_tensor(z[f"g_id_{t}"]).long()
    for r, s in sims.items():
        g.edges[r].data["sim"] = s
    g.set_batch_num_nodes(bnn)
    g.set_batch_num_edges({r: torch.bincount(graph_of_node(r[2], v), minlength=B) for r, (u, v) in data.items()})
    return g, False


def build_model(case, z, ref_root):
    """(model, eval_mode) for a fixture, constructed exactly as the generating scripts construct the oracle's."""
    if case in HEAT_CASES:
        cls = getattr(load_reference_module(ref_root, HEAT_CASES[case]), HEAT_CASES[case])
        in_dim, hidden, out_dim, layers, heads = (int(x) for x in z["config"])
        return cls(in_dim, hidden, out_dim, layers, heads, ND, 0.0, str(z["pooling"])), False     # (in_dim, hidden_dim, out_dim, n_layers, n_heads, node_dict, dropuout, pooling)
    kind = str(z["kind"])
    if kind == "hgt":
        return load_reference_module(ref_root, "HGT").HGT(ND, {et: i for i, et in enumerate(RELS)}, 16, 24, 2, 3, 4, use_norm=True), True
    if kind == "hetrgcn":
        return load_reference_module(ref_root, "HetRGCN").HeteroRGCN(16, 24, 2, 3, {r: str(i) for i, r in enumerate(RELS)}, ND, "sum"), True
    if kind == "gcn":
        return load_reference_module(ref_root, "GCN").GCN(16, 24, 2, 2, F.relu, 0.0, "att"), True
    if kind == "ntpool":
        return load_reference_module(ref_root, "GCN_NTPool").NTPoolGCN(16, 24, 2, ND, 2, F.relu, 0.0, "mean"), True
    raise KeyError(kind)


def run_case(case, ref_root, dgl, regenerate):
    path = os.path.join(HERE, case + ".npz")
    z = np.load(path, allow_pickle=False)
    g, _ = dgl_graph_of(z, dgl)
    model, eval_mode = build_model(case, z, ref_root)
    sd = {k[3:]: torch.as_tensor(z[k]) for k in z.files if k.startswith("sd_")}
    model.load_state_dict(sd, strict=True)
    if eval_mode:
        model.eval()
    labels = torch.as_tensor(z["labels"]).long()
    out = model(g)
    loss = F.cross_entropy(out, labels)
    loss.backward()
    rep = {"logits": float(np.abs(out.detach().numpy() - z["logits"]).max()), "loss": abs(float(loss.item()) - float(z["loss"])), "grad": 0.0, "worst_grad": None}
    new = {}
    for k, p in model.named_parameters():
        key = "grad_" + k
        if key not in z.files:
            continue
        if p.grad is None:
            rep["grad"], rep["worst_grad"] = float("inf"), k + " (no gradient in the reference)"
            continue
        want = z[key]
        rel = float(np.abs(p.grad.numpy() - want).max() / max(float(np.abs(want).max()), 1e-30))
        if rel > rep["grad"]:
            rep["grad"], rep["worst_grad"] = rel, k
        new[key] = p.grad.numpy()
    if regenerate:
        arr = {k: z[k] for k in z.files}
        arr.update(new)
        arr["logits"] = out.detach().numpy()
        arr["loss"] = np.array(loss.item(), dtype=np.float64)
        np.savez_compressed(path, **arr)
    return rep


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--tol", type=float, default=1e-5)
    ap.add_argument("--write", action="store_true", help="save the report as tests/golden/reference_pin.json")
    ap.add_argument("--regenerate", action="store_true", help="rewrite logits / loss / gradients of the fixtures from the reference's outputs")
    args = ap.parse_args()
    try:
        import dgl
    except ImportError as exc:
        print(f"regen_through_reference: DGL is not importable here ({exc}); nothing checked.  Install dgl next to this torch and re-run.", file=sys.stderr)
        return 2
    if not os.path.isfile(os.path.join(args.reference, "models", "HEATNet4.py")):
        print(f"regen_through_reference: no reference checkout at {args.reference}; nothing checked.", file=sys.stderr)
        return 2
    torch.manual_seed(611)
    report, bad = {}, []
    for case in list(HEAT_CASES) + SIBLING_CASES:
        rep = run_case(case, args.reference, dgl, args.regenerate)
        report[case] = rep
        ok = rep["logits"] <= args.tol and rep["loss"] <= args.tol and rep["grad"] <= 10 * args.tol
        print(f"{'ok  ' if ok else 'DIFF'} {case:32s} |dlogit| {rep['logits']:.3e}  |dloss| {rep['loss']:.3e}  grad rel {rep['grad']:.3e} ({rep['worst_grad']})")
        if not ok:
            bad.append(case)
    if args.write:
        meta = {"dgl": getattr(dgl, "__version__", "?"), "torch": torch.__version__, "reference": os.path.abspath(args.reference), "tol": args.tol,
                "cases": report, "all_within_tol": not bad}
        with open(os.path.join(HERE, "reference_pin.json"), "w") as f:
            json.dump(meta, f, indent=1)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
