"""On-disk formats around the hot path.

* **Graph files.**  The reference stores each WSI graph as a ``pickle`` of a ``dgl.DGLGraph``
  (get_graph.py:279-289, read back at data.py:96-97) — only loadable where DGL is installed.  Here a graph is ONE flat
  ``.safetensors`` file (memory-mappable, no pickle): per node type ``feat.{t}`` fp32 [N_t,F]; per canonical relation
  ``src.{i}``, ``dst.{i}`` int64 and ``sim.{i}`` fp32; the schema (node types, relation triples) in the metadata.
  ``from_dgl`` / ``convert_dgl_pickle`` do the one-off conversion wherever DGL exists.
* **Labels from file names** — the three rules of data.py:99-114 (tumour vs normal list), :207-220 (cancer stage) and
  :267-279 (cancer type / ESCA).
* **Checkpoints** — ``CheckpointManager``'s layout (checkpoint.py:72-136): ``model_v{N}.pt`` = ``torch.save(state_dict)``,
  ``version.txt``, ``training_stats.json`` (one JSON object per line).  ``state_dict`` keys/shapes of every model here
  equal the reference's, so reference checkpoints load unchanged and vice versa.
* **Metrics** — ``utils.metrics`` (utils.py:37-47: sklearn precision/recall/F1 + ROC-AUC) restated on tensors (no sklearn
  on the path, no forced host round trip per step).
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from .graph import HeteroGraph

FORMAT = "wsi-hgnn-graph-v1"


# ----------------------------------------------------------------------------------------------- graph files
def save_graph(path: str, g: HeteroGraph, extra: Optional[Dict[str, str]] = None) -> None:
    from safetensors.torch import save_file
    tensors = {}
    for t in g.ntypes:
        tensors[f"feat.{t}"] = g.nodes[t].data["feat"].detach().to(torch.float32).cpu().contiguous()
        if "_ID" in g.nodes[t].data:
            tensors[f"id.{t}"] = g.nodes[t].data["_ID"].detach().to(torch.int64).cpu().contiguous()
    for i, r in enumerate(g.canonical_etypes):
        u, v = g.edges(r)
        tensors[f"src.{i}"] = u.detach().to(torch.int64).cpu().contiguous()
        tensors[f"dst.{i}"] = v.detach().to(torch.int64).cpu().contiguous()
        sim = g._eframes[r].get("sim")
        tensors[f"sim.{i}"] = (sim.detach().to(torch.float32).cpu().contiguous() if sim is not None
                               else torch.zeros(u.numel(), dtype=torch.float32))
    meta = {"format": FORMAT, "ntypes": json.dumps(g.ntypes), "num_nodes": json.dumps([g.num_nodes(t) for t in g.ntypes]),
            "relations": json.dumps([list(r) for r in g.canonical_etypes])}
    if extra:
        meta.update({str(k): str(v) for k, v in extra.items()})
    save_file(tensors, path, metadata=meta)


def load_graph(path: str, device="cpu") -> HeteroGraph:
    from safetensors import safe_open
    with safe_open(path, framework="pt", device=str(device)) as f:
        meta = f.metadata() or {}
        if meta.get("format") != FORMAT:
            raise ValueError(f"{path}: not a {FORMAT} file")
        ntypes = json.loads(meta["ntypes"])
        counts = json.loads(meta["num_nodes"])
        rels = [tuple(r) for r in json.loads(meta["relations"])]
        nn_ = OrderedDict(zip(ntypes, counts))
        edges = OrderedDict((r, (f.get_tensor(f"src.{i}"), f.get_tensor(f"dst.{i}"))) for i, r in enumerate(rels))
        g = HeteroGraph(nn_, edges)
        keys = set(f.keys())
        for t in ntypes:
            g.nodes[t].data["feat"] = f.get_tensor(f"feat.{t}")
            if f"id.{t}" in keys:
                g.nodes[t].data["_ID"] = f.get_tensor(f"id.{t}")
        for i, r in enumerate(rels):
            g._eframes[r]["sim"] = f.get_tensor(f"sim.{i}")
    return g


def from_dgl(dg) -> HeteroGraph:
    """``dgl.DGLGraph`` (as construct_graph/graph_constructor.py:285-303 builds it) -> HeteroGraph.  Needs DGL only for the
    argument's own accessors; ``edata['sim']`` is fp64 in the pickles and is cast to fp32 like models/HEATNet4.py:103."""
    nn_ = OrderedDict((t, int(dg.num_nodes(t))) for t in dg.ntypes)
    edges, sim = OrderedDict(), {}
    for r in dg.canonical_etypes:
        u, v = dg.edges(etype=r)
        edges[tuple(r)] = (u.long(), v.long())
        ed = dg.edges[r].data
        sim[tuple(r)] = ed["sim"].to(torch.float32) if "sim" in ed else torch.zeros(u.numel())
    g = HeteroGraph.from_coo(nn_, edges, feat={t: dg.nodes[t].data["feat"].to(torch.float32) for t in dg.ntypes}, sim=sim)
    for t in dg.ntypes:
        if "_ID" in dg.nodes[t].data:
            g.nodes[t].data["_ID"] = dg.nodes[t].data["_ID"].long()
    return g


def convert_dgl_pickle(pickle_path: str, out_path: str) -> None:
    """One-off converter (run where DGL is installed): the reference's pickle -> flat safetensors."""
    import pickle
    with open(pickle_path, "rb") as f:
        dg = pickle.load(f)
    save_graph(out_path, from_dgl(dg), extra={"source": os.path.basename(pickle_path)})


# ----------------------------------------------------------------------------------------------- labels
def _barcode(path: str, n: int) -> str:
    """``s[pos:pos + n]`` with ``pos = s.find("TCGA")`` exactly as the reference slices it - a path without a barcode yields the
    reference's own (degenerate) slice rather than an error of ours: tumour-vs-normal then says 1, the mapping rules KeyError."""
    s = str(path)
    pos = s.find("TCGA")
    if pos < 0:
        import warnings
        warnings.warn(f"no TCGA barcode in {s!r}: the reference's slice s[-1:{n - 1}] is used as the key (tumour-vs-normal then labels the slide 1)",
                      RuntimeWarning, stacklevel=3)
    return s[pos:pos + n]


def label_tumour_vs_normal(path: str, normal_list: Iterable[str], name: str = "COAD") -> int:
    """data.py:99-114: 0 if the 16-character TCGA barcode is in the normal list, else 1; ``name`` is the data set
    (COAD / BRCA / ESCA: the same rule; anything else is the reference's ``raise ValueError``)."""
    if name not in ("COAD", "BRCA", "ESCA"):
        raise ValueError
    return 0 if _barcode(path, 16) in set(normal_list) else 1


_STAGES = [("Stage I", "Stage IA", "Stage IB"), ("Stage IIA", "Stage IIB", "Stage II", "Stage IIC"),
           ("Stage IIIB", "Stage IIIC", "Stage III", "Stage IIIA"), ("Stage IV", "Stage IVA", "Stage IVB")]


def label_cancer_stage(path: str, mapping: Dict[str, str]) -> int:
    """data.py:207-220: 12-character barcode -> pathologic stage string -> {0,1,2,3}."""
    lb = mapping[_barcode(path, 12)]
    for i, names in enumerate(_STAGES):
        if lb in names:
            return i
    raise ValueError("Undefined label")


def label_cancer_type(path: str, mapping: Dict[str, str], esca: bool = False) -> int:
    """data.py:267-279: ESCA label files hold the integer label; BRCA maps ductal -> 0, lobular -> 1."""
    lb = mapping[_barcode(path, 12)]
    if esca:
        return int(lb)
    if lb == "Infiltrating Ductal Carcinoma":
        return 0
    if lb == "Infiltrating Lobular Carcinoma":
        return 1
    raise ValueError("Undefined label")


# ----------------------------------------------------------------------------------------------- checkpoints
class CheckpointStore:
    """File layout of the reference's ``CheckpointManager`` (checkpoint.py): ``{dir}/version.txt``,
    ``{dir}/model_v{N}.pt``, ``{dir}/training_stats.json`` (JSON lines), ``{dir}/configs.json``.

    Said plainly: the methods named after the reference's (``write_new_version``, ``save_version``, ``append_stats`` … below) are a
    method-by-method TRANSLITERATION of ``checkpoint.py:26-136`` — about 50 lines, off the hot path — because the requirement is
    byte-identical files for a replayed session (``tests/golden/reference_io.json`` was produced by executing the reference's class).
    It is compatibility plumbing, not a design of this build."""

    def __init__(self, path: str):
        self.path = path
        os.makedirs(path, exist_ok=True)
        self.version = self.load_version()
        self.old_version = self.version

    def model_file(self, version: int) -> str:
        return os.path.join(self.path, f"model_v{version}.pt")

    def load_version(self) -> int:
        try:
            with open(os.path.join(self.path, "version.txt")) as f:
                s = f.read().strip()
            return int(s) if s else 0
        except FileNotFoundError:
            return 0

    def save_model(self, state_dict, version: int, stats: Optional[Dict] = None, config: Optional[Dict] = None) -> None:
        if self.version == 0 and config is not None:
            with open(os.path.join(self.path, "configs.json"), "w") as f:
                json.dump(config, f, indent=4)
        self.old_version, self.version = self.version, int(version)
        with open(os.path.join(self.path, "version.txt"), "w") as f:
            f.write(f"{self.version}\n")
        torch.save({k: v.detach().cpu() for k, v in state_dict.items()}, self.model_file(self.version))
        if stats is not None:
            stats = {k: (v if isinstance(v, int) else round(float(v), 5)) for k, v in stats.items()}
            with open(os.path.join(self.path, "training_stats.json"), "a") as f:
                f.write(json.dumps(stats) + "\n")

    def load_model(self, version: Optional[int] = None, map_location="cpu"):
        return torch.load(self.model_file(self.version if version is None else version), map_location=map_location)

    # ---- the reference's own method names (checkpoint.py:26-136), same arguments and effects, so that trainer code written
    #      against ``CheckpointManager`` runs unchanged
    def get_version_file(self):
        return os.path.join(self.path, "version.txt")

    def get_config_file(self):
        return os.path.join(self.path, "configs.json")

    def get_model_file(self, version: int):
        return self.model_file(version)

    def get_stats_file(self):
        return os.path.join(self.path, "training_stats.json")

    def save_config(self, config: Dict) -> None:                       # checkpoint.py:46-49
        with open(self.get_config_file(), "wt") as f:
            f.write(json.dumps(config, indent=4))

    def load_config(self) -> str:                                      # :51-56
        with open(self.get_config_file(), "rt") as f:
            return f.read()

    def append_stats(self, stats: Dict) -> None:                       # :58-61
        with open(self.get_stats_file(), "at") as f:
            f.write(f"{json.dumps(stats)}\n")

    def load_stats(self):                                              # :63-69
        with open(self.get_stats_file(), "rt") as f:
            for line in f:
                yield line

    def save_version(self, version: int) -> None:                      # :90-94
        with open(self.get_version_file(), "wt") as f:
            f.write(f"{version}\n")
            f.flush()
            os.fsync(f.fileno())

    def write_new_version(self, config: Dict, state_dict, epoch_stats: Dict = None) -> None:
        """checkpoint.py:108-136: config saved on the first version only; version = epoch_stats['Epoch']; non-int stats
        rounded to 5 decimals IN PLACE (the reference mutates the caller's dict) and appended as one JSON line."""
        if self.version == 0:
            self.save_config(config)
        self.old_version = self.version
        self.version = epoch_stats["Epoch"]
        self.save_version(self.version)
        torch.save(state_dict, self.model_file(self.version))
        for k, v in epoch_stats.items():
            if type(v) != int:
                epoch_stats[k] = round(v, 5)
        self.append_stats(epoch_stats)

    def remove_old_version(self) -> None:
        try:
            os.unlink(self.model_file(self.old_version))
        except FileNotFoundError:
            pass


# ----------------------------------------------------------------------------------------------- metrics
def classification_metrics(outputs: torch.Tensor, targets: torch.Tensor, average: str = "binary") -> Tuple[float, float, float, float]:
    """utils.py:37-47 ``metrics(outputs, targets, average)`` -> (precision, recall, f1, auc) without sklearn.

    'binary': precision/recall/F1 of class 1 and the AUC of the single-threshold ROC built from the hard predictions
    (``roc_curve(targets, preds)`` on 0/1 predictions = (TPR + TNR) / 2).  'macro': unweighted class means and the
    one-vs-rest AUC of the per-class scores (``roc_auc_score(targets, outputs, multi_class='ovr')``, ties averaged)."""
    outputs = outputs.detach().double().cpu()
    targets = targets.detach().long().cpu()
    preds = outputs.argmax(1)
    C = outputs.shape[1]

    def prf(c):
        tp = float(((preds == c) & (targets == c)).sum())
        fp = float(((preds == c) & (targets != c)).sum())
        fn = float(((preds != c) & (targets == c)).sum())
        p = tp / (tp + fp) if tp + fp > 0 else 0.0
        r = tp / (tp + fn) if tp + fn > 0 else 0.0
        f = 2 * p * r / (p + r) if p + r > 0 else 0.0
        return p, r, f

    def auc_scores(score, pos):
        """Mann-Whitney AUC with average ranks for ties."""
        order = torch.argsort(score)
        s = score[order]
        ranks = torch.empty_like(s)
        i, n = 0, s.numel()
        while i < n:
            j = i
            while j + 1 < n and s[j + 1] == s[i]:
                j += 1
            ranks[i:j + 1] = (i + j) / 2.0 + 1.0
            i = j + 1
        r = torch.empty_like(ranks)
        r[order] = ranks
        npos = float(pos.sum())
        nneg = float((~pos).sum())
        if npos == 0 or nneg == 0:
            return float("nan")
        return float((r[pos].sum() - npos * (npos + 1) / 2.0) / (npos * nneg))

    if average == "binary":
        p, r, f = prf(1)
        return p, r, f, auc_scores(preds.double(), targets == 1)
    ps, rs, fs = zip(*[prf(c) for c in range(C)])
    aucs = [auc_scores(outputs[:, c], targets == c) for c in range(C)]
    return sum(ps) / C, sum(rs) / C, sum(fs) / C, sum(aucs) / C


@torch.no_grad()
def evaluate(gnn: torch.nn.Module, loader, average: str = "binary") -> Dict[str, float]:
    """evaluator/eval_homo_graph.py:61-95 (per-graph ``test_one_step`` loop) as batched inference over a
    ``GraphBatchLoader``: accuracy, precision, recall, F1, AUC and mean cross-entropy."""
    was_training = gnn.training
    gnn.eval()
    outs, ys = [], []
    with torch.no_grad():
        for G, y in loader:
            outs.append(gnn(G))
            ys.append(y)
    gnn.train(was_training)
    out = torch.cat(outs)
    y = torch.cat(ys)
    loss = torch.nn.functional.cross_entropy(out, y).item()
    accuracy = float((out.argmax(1) == y).float().mean())
    p, r, f, a = classification_metrics(out, y, average)
    return {"loss": loss, "accuracy": accuracy, "precision": p, "recall": r, "f1": f, "auc": a}
