"""The reference's config -> object layer (parser.py) and its label rules (data.py), checked against what the REFERENCE'S OWN CODE
does: tests/golden/reference_surface.json holds the calls ``parser.parse_gnn_model`` makes on its own COAD configs (recorded by
executing parser.py:48-174 with recording stand-ins for the model classes), the optimizers / losses ``parse_optimizer`` /
``parse_loss`` return, and the labels the three dataset classes compute from file names (tests/golden/
make_reference_surface_fixture.py).  No GPU."""
import json
import os

import pytest
import torch

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_surface.json")))


def _enc(v):
    if callable(v):
        return "callable:" + getattr(v, "__name__", type(v).__name__)
    if isinstance(v, dict):
        return {"__dict__": [[list(k) if isinstance(k, tuple) else k, _enc(x)] for k, x in v.items()]}
    return v


class _Recorder:
    def __init__(self, name):
        self.name = name

    def __call__(self, *args, **kwargs):
        return {"class": self.name, "args": [_enc(a) for a in args], "kwargs": {k: _enc(v) for k, v in kwargs.items()}}


@pytest.mark.parametrize("rec", FIX["parse_gnn_model"], ids=lambda r: r["config"].split("/")[-1])
def test_parse_gnn_model_makes_the_reference_calls(rec, monkeypatch):
    """Same class, same positional and keyword arguments (node_dict, the etype-major edge_dict / etypes enumeration, the
    ``dropuout`` spelling, F.relu), same KeyError for configs that lack a key the reference reads, same NotImplementedError for the
    orphan names (HEAT, HEAT3)."""
    from wsi_hgnn_amd import parser as P
    for cname in ("GCN", "NTPoolGCN", "HGT", "HEATNet2", "HEATNet4", "HeteroRGCN"):
        monkeypatch.setattr(P, cname, _Recorder(cname))
    if "raises" in rec:
        exc = {"KeyError": KeyError, "NotImplementedError": NotImplementedError}[rec["raises"][0]]
        with pytest.raises(exc) as ei:
            P.parse_gnn_model(rec["GNN"])
        assert str(ei.value) == rec["raises"][1]
    else:
        got = P.parse_gnn_model(rec["GNN"])
        assert json.loads(json.dumps(got)) == rec["constructs"]


def test_from_config_builds_real_modules():
    """The factory on the real classes: every in-scope config that the reference can construct yields a module of that class
    whose constructor-visible attributes match the config."""
    from wsi_hgnn_amd import models
    seen = set()
    for rec in FIX["parse_gnn_model"]:
        if "constructs" not in rec or rec["constructs"]["class"] in seen:
            continue
        seen.add(rec["constructs"]["class"])
        m = models.from_config(rec["GNN"])
        assert type(m).__name__ == rec["constructs"]["class"]
        assert m.n_layers == rec["GNN"]["num_layers"]
        assert sum(p.numel() for p in m.parameters()) > 0
    assert seen == {"GCN", "NTPoolGCN", "HEATNet2", "HEATNet4", "HGT", "HeteroRGCN"}
    with pytest.raises(NotImplementedError):
        models.from_config({"name": "GAT"})


@pytest.mark.parametrize("rec", FIX["parse_optimizer"], ids=lambda r: r["opt_method"])
def test_parse_optimizer(rec):
    from wsi_hgnn_amd.parser import parse_optimizer
    opt = parse_optimizer({"opt_method": rec["opt_method"], "lr": rec["lr"], "weight_decay": rec["weight_decay"]}, torch.nn.Linear(2, 1))
    assert type(opt).__name__ == rec["class"]
    for k in ("lr", "weight_decay", "lr_decay"):
        if k in rec["defaults"]:
            assert opt.defaults[k] == rec["defaults"][k], k


@pytest.mark.parametrize("rec", FIX["parse_loss"], ids=lambda r: r["loss"])
def test_parse_loss(rec):
    from wsi_hgnn_amd.parser import parse_loss
    if "raises" in rec:
        with pytest.raises(NotImplementedError) as ei:
            parse_loss({"loss": rec["loss"]})
        assert str(ei.value) == rec["raises"][1]
    else:
        assert type(parse_loss({"loss": rec["loss"]})).__name__ == rec["class"]


@pytest.mark.parametrize("rules", FIX["labels"], ids=lambda r: f"{r['class']}-{r['rule']}-{r['self'].get('name_', '')}")
def test_label_rules_match_the_reference(rules):
    """data.py:99-114, 207-220, 267-279 executed on TCGA file names (incl. a path without a barcode and unmapped cases)."""
    from wsi_hgnn_amd import io as wio
    s = rules["self"]
    for case in rules["cases"]:
        if rules["rule"] == "tumour_vs_normal":
            f = lambda: wio.label_tumour_vs_normal(case["path"], s["normal_list"], name=s["name_"])
        elif rules["rule"] == "cancer_stage":
            f = lambda: wio.label_cancer_stage(case["path"], s["mapping"])
        else:
            f = lambda: wio.label_cancer_type(case["path"], s["mapping"], esca="ESCA" in s["label_path"])
        if "raises" in case:
            with pytest.raises({"ValueError": ValueError, "KeyError": KeyError}[case["raises"]]):
                f()
        else:
            assert f() == case["label"], case
