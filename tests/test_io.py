"""Rows n2/n3: graph file round trip, label rules (data.py:99-114,207-220,267-279), checkpoint layout, sklearn-free metrics."""
import json
import os

import pytest
import torch

import wsi_hgnn_amd as W
from wsi_hgnn_amd import io as wio, synthetic


def test_graph_file_round_trip(tmp_path):
    g = synthetic.hetero_graph(120, 16, seed=3, dst_mode="hub")
    p = str(tmp_path / "TCGA-AA-0001-01Z-00-DX1.safetensors")
    wio.save_graph(p, g, extra={"slide": "x"})
    h = wio.load_graph(p)
    assert h.ntypes == g.ntypes and h.canonical_etypes == g.canonical_etypes
    for t in g.ntypes:
        assert torch.equal(h.nodes[t].data["feat"], g.nodes[t].data["feat"])
    for r in g.canonical_etypes:
        assert torch.equal(h.edges(r)[0], g.edges(r)[0]) and torch.equal(h.edges(r)[1], g.edges(r)[1])
        assert torch.equal(h.edata["sim"][r], g.edata["sim"][r])
    assert torch.equal(h.plan().src, g.plan().src)


def test_checkpoint_layout_and_reference_state_dict(tmp_path):
    from oracle import models as OM
    nd = {"0": 0, "1": 1}
    torch.manual_seed(0)
    ref = OM.HEATNet4(8, 16, 2, 1, 2, nd, 0.0)       # stands in for a reference-trained model: identical state_dict keys
    st = wio.CheckpointStore(str(tmp_path / "ckpt"))
    assert st.version == 0
    st.save_model(ref.state_dict(), 3, stats={"Epoch": 3, "loss": 0.123456789}, config={"GNN": {"name": "HEAT4"}})
    assert sorted(os.listdir(st.path)) == ["configs.json", "model_v3.pt", "training_stats.json", "version.txt"]
    assert open(os.path.join(st.path, "version.txt")).read() == "3\n"
    assert json.loads(open(os.path.join(st.path, "training_stats.json")).readline()) == {"Epoch": 3, "loss": 0.12346}
    st2 = wio.CheckpointStore(st.path)
    assert st2.version == 3
    from wsi_hgnn_amd import models
    m = models.HEATNet4(8, 16, 2, 1, 2, nd, 0.0)
    missing = m.load_state_dict(st2.load_model(), strict=True)        # reference checkpoint loads unchanged
    assert not missing.missing_keys and not missing.unexpected_keys
    st2.save_model(m.state_dict(), 4)
    st2.remove_old_version()
    assert not os.path.exists(st2.model_file(3)) and os.path.exists(st2.model_file(4))


def test_metrics_match_sklearn():
    sk = pytest.importorskip("sklearn.metrics")
    import numpy as np
    torch.manual_seed(1)
    out = torch.randn(40, 2)
    y = (torch.rand(40) > 0.4).long()
    p, r, f, a = wio.classification_metrics(out, y, "binary")
    preds = out.argmax(1).numpy()
    fpr, tpr, _ = sk.roc_curve(y.numpy(), preds)
    want = (sk.precision_score(y, preds), sk.recall_score(y, preds), sk.f1_score(y, preds), sk.auc(fpr, tpr))
    assert np.allclose([p, r, f, a], want, atol=1e-12)
    out3 = torch.softmax(torch.randn(60, 4), 1)
    y3 = torch.randint(0, 4, (60,))
    p, r, f, a = wio.classification_metrics(out3, y3, "macro")
    pr3 = out3.argmax(1).numpy()
    want = (sk.precision_score(y3, pr3, average="macro", zero_division=0), sk.recall_score(y3, pr3, average="macro", zero_division=0),
            sk.f1_score(y3, pr3, average="macro", zero_division=0), sk.roc_auc_score(y3.numpy(), out3.numpy(), multi_class="ovr"))
    assert np.allclose([p, r, f, a], want, atol=1e-9)


# ---------------------------------------------------------------------------------------------- fixtures from the executed reference
def _ref_io():
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_io.json")))


def _snapshot(path):
    return {f: (open(os.path.join(path, f)).read() if f.endswith((".txt", ".json")) else None) for f in sorted(os.listdir(path))}


def test_checkpoint_store_replays_reference_checkpoint_manager_session(tmp_path):
    """tests/golden/reference_io.json was produced by running /root/reference/checkpoint.py::CheckpointManager through a
    scripted session; CheckpointStore must leave byte-identical text files and the same versions after every call."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_reference_io_fixture import CKPT_SCRIPT
    states = _ref_io()["checkpoint"]
    path = str(tmp_path / "run" / "ckpt")
    st = wio.CheckpointStore(path)
    assert st.version == states[0]["version"] and _snapshot(path) == states[0]["files"]
    for i, ((call, kw), want) in enumerate(zip(CKPT_SCRIPT, states[1:])):
        assert want["call"] == call
        if call == "write_new_version":
            stats = dict(kw["epoch_stats"])
            st.write_new_version(kw["config"], {"w": torch.full((2, 2), float(i))}, stats)
            assert stats == want["stats_after"]                      # rounded in place, ints untouched
            assert st.old_version == want["old_version"]
            assert float(st.load_model()["w"].sum()) == want["model_sum"]
        else:
            getattr(st, call)()
        assert st.version == want["version"], (i, call)
        assert _snapshot(path) == want["files"], (i, call)
    again = wio.CheckpointStore(path)
    assert again.version == states[-1]["version"]
    assert again.load_config() == states[-1]["config_text"]
    assert list(again.load_stats()) == states[-1]["stats_lines"]


def test_metrics_reproduce_executed_reference_utils():
    """utils.py::metrics / acc were EXECUTED on these seeded logits (make_reference_io_fixture.py); the sklearn-free
    re-implementation must return the same numbers (binary AUC is taken over the hard predictions, as the reference does)."""
    for case in _ref_io()["metrics"]:
        logits = torch.tensor(case["logits"])
        y = torch.tensor(case["targets"])
        out = torch.softmax(logits, 1) if case["softmaxed"] else logits
        p, r, f, a = wio.classification_metrics(out, y, case["average"])
        for got, key in ((p, "precision"), (r, "recall"), (f, "f1"), (a, "auc")):
            assert abs(got - case[key]) < 1e-6, (key, got, case[key], case["average"])
        from wsi_hgnn_amd.trainer import acc
        assert abs(float(acc(logits, y)) - case["acc"]) < 1e-7


def test_label_rules_directly():
    """The label-from-filename rules of data.py:99-114,207-220,267-279 on hand-written cases (tests/test_parser.py replays the statements of the
    reference itself; this is the direct unit test): the 16- / 12-character barcode slices, the stage and type tables, and the guard the
    reference lacks - a path WITHOUT a barcode still yields the reference's degenerate slice (same label), but not silently."""
    import warnings
    from wsi_hgnn_amd import io as wio
    p = "/data/COAD/graphs/TCGA-AA-3561-11A-01-TS1.9d2a.pkl"
    assert wio.label_tumour_vs_normal(p, ["TCGA-AA-3561-11A"]) == 0 and wio.label_tumour_vs_normal(p, ["TCGA-AA-3561-01A"]) == 1
    with pytest.raises(ValueError):
        wio.label_tumour_vs_normal(p, [], name="LUAD")
    stages = {"TCGA-AA-3561": "Stage IIB", "TCGA-AA-0001": "Stage IVA", "TCGA-AA-0002": "Stage X"}
    assert wio.label_cancer_stage(p, stages) == 1
    assert wio.label_cancer_stage(p.replace("3561", "0001"), stages) == 3
    with pytest.raises(ValueError):
        wio.label_cancer_stage(p.replace("3561", "0002"), stages)
    with pytest.raises(KeyError):
        wio.label_cancer_stage(p.replace("3561", "9999"), stages)
    assert wio.label_cancer_type(p, {"TCGA-AA-3561": "Infiltrating Lobular Carcinoma"}) == 1
    assert wio.label_cancer_type(p, {"TCGA-AA-3561": "Infiltrating Ductal Carcinoma"}) == 0
    assert wio.label_cancer_type(p, {"TCGA-AA-3561": "1"}, esca=True) == 1
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert wio.label_tumour_vs_normal("/data/slide_without_barcode.pkl", ["TCGA-AA-3561-11A"]) == 1      # the reference's behaviour ...
    assert any("no TCGA barcode" in str(x.message) for x in w)                                               # ... but said aloud
