#!/usr/bin/env python
"""Generates tests/golden/reference_surface.json FROM THE REFERENCE ITSELF (run in the build container; /root/reference is read, never
copied).  DGL is absent, so nothing DGL-dependent is imported; what CAN be pinned is read with ``ast`` and, for DGL-free
fragments, executed:

  * constructor signatures, forward signatures and the ORDER in which ``__init__`` creates its sub-modules / parameters for the
    seven model classes and the pooling classes on the path (models/*.py, pooling/*.py) - read from the syntax tree;
  * ``parser.parse_gnn_model`` (parser.py:48-174) - its source is executed with RECORDER callables bound to the class names,
    on the ``GNN:`` block of every in-scope config under configs/COAD (and on the orphan names HEAT / HEAT3): the fixture holds
    the class each config constructs and the exact arguments (incl. node_dict and the etype-major edge_dict enumeration);
  * ``parser.parse_optimizer`` / ``parse_loss`` (parser.py:15-46,176-184) - executed against a one-parameter model: optimizer
    class and param-group defaults for the four branches, loss class for the two names, the error for an unknown one;
  * the label-from-filename rules of the three dataset classes (data.py:99-114,207-220,267-279) - the statements between
    ``s = str(graph_path)`` and the transform are compiled from the syntax tree and executed on a list of TCGA barcodes.

Only inputs and observed outputs are stored.  Usage: python tests/golden/make_reference_surface_fixture.py
"""
import ast
import json
import os
import types

import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_surface.json")


def parse(rel):
    src = open(os.path.join(REF, rel)).read()
    return src, ast.parse(src)


def find_class(tree, name):
    for n in tree.body:
        if isinstance(n, ast.ClassDef) and n.name == name:
            return n
    raise KeyError(name)


def find_func(node, name):
    for n in node.body:
        if isinstance(n, ast.FunctionDef) and n.name == name:
            return n
    return None


def signature(fn):
    a = fn.args
    names = [x.arg for x in a.args]
    defaults = [None] * (len(names) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
    return [{"name": n, "default": d} for n, d in zip(names, defaults)]


def created_attributes(init):
    """``self.X = <call>`` statements of __init__ in source order (nested blocks included): what X is and the call text."""
    out = []
    for node in ast.walk(init):
        pass
    def visit(stmts):
        for st in stmts:
            if isinstance(st, ast.Assign) and len(st.targets) == 1:
                t = st.targets[0]
                if isinstance(t, ast.Attribute) and isinstance(t.value, ast.Name) and t.value.id == "self" and isinstance(st.value, ast.Call):
                    out.append({"attr": t.attr, "call": ast.unparse(st.value.func), "line": st.lineno})
            for field in ("body", "orelse"):
                sub = getattr(st, field, None)
                if isinstance(sub, list):
                    visit(sub)
    visit(init.body)
    return out


fixture = {"source": "HKU-MedAI/WSI-HGNN at /root/reference (ast + exec of DGL-free fragments)", "classes": {}}

CLASSES = [("models/HEATNet4.py", ["HEATLayer", "HEATNet4", "LinearAttentionBlock"]), ("models/HEATNet2.py", ["HEATLayer", "HEATNet2"]),
           ("models/HGT.py", ["HGTLayer", "HGT"]), ("models/HetRGCN.py", ["HeteroRGCNLayer", "HeteroRGCN"]), ("models/GCN.py", ["GCN"]),
           ("models/GCN_NTPool.py", ["NTPoolGCN"]), ("pooling/avg_pooling.py", ["AvgPooling"]), ("pooling/sum_pooling.py", ["SumPooling"]),
           ("pooling/max_pooling.py", ["MaxPooling"]), ("pooling/nt_pooling.py", ["NTPooling"]), ("pooling/ASAP.py", ["LEConv", "ASAPPooling"])]
for rel, names in CLASSES:
    _, tree = parse(rel)
    for name in names:
        cls = find_class(tree, name)
        init, fwd = find_func(cls, "__init__"), find_func(cls, "forward")
        fixture["classes"][f"{rel}:{name}"] = {
            "line": cls.lineno,
            "init": signature(init) if init else None,
            "forward": [a.arg for a in fwd.args.args] if fwd else None,
            "created": created_attributes(init) if init else [],
        }

# ---- parser.py: execute the three DGL-free functions
psrc, ptree = parse("parser.py")
funcs = {n.name: ast.get_source_segment(psrc, n) for n in ptree.body if isinstance(n, ast.FunctionDef)}

import torch
from torch import nn, optim
import torch.nn.functional as F


class Recorder:
    def __init__(self, name):
        self.name = name

    def __call__(self, *args, **kwargs):
        def enc(v):
            if callable(v):
                return "callable:" + getattr(v, "__name__", type(v).__name__)
            if isinstance(v, dict):
                return {"__dict__": [[list(k) if isinstance(k, tuple) else k, enc(x)] for k, x in v.items()]}     # insertion order kept
            return v
        return {"class": self.name, "args": [enc(a) for a in args], "kwargs": {k: enc(v) for k, v in kwargs.items()}}


ns = {"F": F, "nn": nn, "optim": optim}
for cname in ("GCN", "GAT", "NTPoolGCN", "GIN", "HGT", "HEATNet2", "HEATNet4", "HeteroRGCN"):
    ns[cname] = Recorder(cname)
exec(funcs["parse_gnn_model"], ns)
exec(funcs["parse_optimizer"], ns)
exec(funcs["parse_loss"], ns)

fixture["parse_gnn_model"] = []
cfg_dir = os.path.join(REF, "configs", "COAD")
IN_SCOPE = {"HEAT2", "HEAT4", "HGT", "HetRGCN", "GCN", "GCN_NTPool", "HEAT", "HEAT3"}
for fn in sorted(os.listdir(cfg_dir)):
    cfg = yaml.safe_load(open(os.path.join(cfg_dir, fn)))
    gnn = cfg.get("GNN") if isinstance(cfg, dict) else None
    if not gnn or gnn.get("name") not in IN_SCOPE:
        continue
    rec = {"config": f"configs/COAD/{fn}", "GNN": gnn}
    try:
        rec["constructs"] = ns["parse_gnn_model"](gnn)
    except NotImplementedError as e:
        rec["raises"] = ["NotImplementedError", str(e)]
    except KeyError as e:
        rec["raises"] = ["KeyError", str(e)]
    fixture["parse_gnn_model"].append(rec)

model = nn.Linear(2, 1)
fixture["parse_optimizer"] = []
for method in ("adagrad", "Adadelta", "ADAM", "sgd", "anything-else"):
    opt = ns["parse_optimizer"]({"opt_method": method, "lr": 0.005, "weight_decay": 0.0001}, model)
    d = {k: v for k, v in opt.defaults.items() if isinstance(v, (int, float, bool, tuple, type(None)))}
    fixture["parse_optimizer"].append({"opt_method": method, "lr": 0.005, "weight_decay": 0.0001, "class": type(opt).__name__,
                                       "defaults": {k: (list(v) if isinstance(v, tuple) else v) for k, v in d.items()}})
fixture["parse_loss"] = []
for name in ("BCE", "CE", "MSE"):
    try:
        fixture["parse_loss"].append({"loss": name, "class": type(ns["parse_loss"]({"loss": name})).__name__})
    except NotImplementedError as e:
        fixture["parse_loss"].append({"loss": name, "raises": ["NotImplementedError", str(e)]})

# ---- data.py: the label rules, compiled from the statements between `s = str(graph_path)` and the transform
dsrc, dtree = parse("data.py")


def label_fragment(cls):
    get = find_func(cls, "__getitem__")
    start = end = None
    for i, st in enumerate(get.body):
        if start is None and isinstance(st, ast.Assign) and ast.unparse(st) == "s = str(graph_path)":
            start = i
        if start is not None and isinstance(st, ast.If) and "self.type_" in ast.unparse(st.test):
            end = i
            break
    mod = ast.Module(body=get.body[start:end], type_ignores=[])
    return compile(ast.fix_missing_locations(mod), f"data.py:{cls.name}.__getitem__[{get.body[start].lineno}-{get.body[end - 1].end_lineno}]", "exec"), \
        [get.body[start].lineno, get.body[end - 1].end_lineno]


datasets = [c for c in dtree.body if isinstance(c, ast.ClassDef) and find_func(c, "__getitem__") is not None
            and "s = str(graph_path)" in ast.unparse(find_func(c, "__getitem__"))]
paths = ["/d/COAD/graphs/TCGA-AA-3489-01Z-00-DX1.abc.pkl", "/d/COAD/graphs/TCGA-AA-3489-11A-01-TS1.pkl", "/x/TCGA-A6-2671-01A-01-BS1.pkl",
         "/x/TCGA-E2-A1B1-11A-02-TSB.pkl", "relative/TCGA-3C-AALI-01Z-00-DX2.pkl", "/no_barcode_here.pkl"]
normal = ["TCGA-AA-3489-11A", "TCGA-E2-A1B1-11A"]
stage_map = {"TCGA-AA-3489": "Stage IIIA", "TCGA-A6-2671": "Stage I", "TCGA-E2-A1B1": "Stage IV", "TCGA-3C-AALI": "Stage IIB"}
type_map = {"TCGA-AA-3489": "Infiltrating Ductal Carcinoma", "TCGA-A6-2671": "Infiltrating Lobular Carcinoma", "TCGA-E2-A1B1": "Mucinous", "TCGA-3C-AALI": "1"}
fixture["labels"] = []
for cls in datasets:
    code, lines = label_fragment(cls)
    uses = ast.unparse(find_func(cls, "__getitem__"))
    variants = []
    if "normal_list" in uses:
        variants = [({"name_": nm, "normal_list": normal}, "tumour_vs_normal") for nm in ("COAD", "BRCA", "ESCA", "OTHER")]
    elif "Stage" in uses:
        variants = [({"mapping": stage_map}, "cancer_stage")]
    else:
        variants = [({"mapping": type_map, "label_path": "data/BRCA_types.json"}, "cancer_type"), ({"mapping": type_map, "label_path": "data/ESCA_types.json"}, "cancer_type_esca")]
    for attrs, rule in variants:
        rows = []
        for pth in paths:
            env = {"self": types.SimpleNamespace(**attrs), "graph_path": pth}
            try:
                exec(code, env)
                rows.append({"path": pth, "label": env["label"]})
            except (ValueError, KeyError) as e:
                rows.append({"path": pth, "raises": type(e).__name__})
        fixture["labels"].append({"class": cls.name, "lines": lines, "rule": rule, "self": attrs, "cases": rows})

json.dump(fixture, open(OUT, "w"), indent=1, sort_keys=False)
print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(fixture["classes"]), "classes,", len(fixture["parse_gnn_model"]), "configs,", len(fixture["labels"]), "label rule sets")
