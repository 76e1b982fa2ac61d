#!/usr/bin/env python
"""Fixtures produced by EXECUTING the reference's own code (rows n3 of SURVEY 8f), in this container:

* /root/reference/checkpoint.py  ``CheckpointManager``: a scripted sequence of calls in a scratch directory; the fixture records
  the resulting file names and the text of version.txt / configs.json / training_stats.json after every call.
* /root/reference/utils.py  ``metrics`` and ``acc`` on seeded logits/labels (binary and 3-class).

Only inputs and observed outputs are stored (reference_io.json); no reference source travels.  Re-run here (needs
/root/reference): ``python tests/golden/make_reference_io_fixture.py``."""
import importlib.util
import json
import os
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, f"/root/reference/{name}.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def snapshot(path):
    out = {}
    for f in sorted(os.listdir(path)):
        out[f] = open(os.path.join(path, f)).read() if f.endswith((".txt", ".json")) else None
    return out


# the scripted checkpoint session (mirrored by tests/test_io.py on CheckpointStore)
CKPT_SCRIPT = [
    ("write_new_version", {"config": {"GNN": {"name": "HEAT4", "n_layers": 2}, "optimizer": {"lr": 1e-05}},
                           "epoch_stats": {"Epoch": 1, "Train Loss": 0.6931471805599453, "Train Acc": 0.5, "N": 7}}),
    ("write_new_version", {"config": {"ignored": True}, "epoch_stats": {"Epoch": 2, "Train Loss": 0.123456789, "Train Acc": 1.0}}),
    ("remove_old_version", {}),
    ("write_new_version", {"config": {}, "epoch_stats": {"Epoch": 5, "Val AUC": 0.9999949}}),
    ("remove_old_version", {}),
]


def main():
    ck, ut = load("checkpoint"), load("utils")
    fix = {"checkpoint": [], "metrics": []}
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "run", "ckpt")
        mgr = ck.CheckpointManager(path)
        fix["checkpoint"].append({"call": "init", "version": mgr.version, "files": snapshot(path)})
        for i, (call, kw) in enumerate(CKPT_SCRIPT):
            if call == "write_new_version":
                stats = dict(kw["epoch_stats"])
                mgr.write_new_version(kw["config"], {"w": torch.full((2, 2), float(i))}, stats)
                fix["checkpoint"].append({"call": call, "version": mgr.version, "old_version": mgr.old_version,
                                          "stats_after": stats, "files": snapshot(path),
                                          "model_sum": float(mgr.load_model()["w"].sum())})
            else:
                getattr(mgr, call)()
                fix["checkpoint"].append({"call": call, "version": mgr.version, "files": snapshot(path)})
        reopened = ck.CheckpointManager(path)
        fix["checkpoint"].append({"call": "reopen", "version": reopened.version, "config_text": reopened.load_config(),
                                  "stats_lines": list(reopened.load_stats())})
    gen = torch.Generator().manual_seed(611)
    for n, c, avg in [(40, 2, "binary"), (64, 2, "binary"), (90, 3, "macro"), (5, 2, "binary")]:
        logits = torch.randn(n, c, generator=gen)
        y = torch.randint(0, c, (n,), generator=gen)
        if c == 2:
            y[0], y[1] = 0, 1
        else:
            y[:3] = torch.arange(3)
        if c == 2:
            p, r, f, a = ut.metrics(logits, y, avg)
        else:                                     # utils.metrics feeds raw logits to roc_auc_score, which wants probabilities
            p, r, f, a = ut.metrics(torch.softmax(logits, 1), y, avg)
        fix["metrics"].append({"logits": logits.tolist(), "targets": y.tolist(), "average": avg, "softmaxed": c != 2,
                               "precision": float(p), "recall": float(r), "f1": float(f), "auc": float(a),
                               "acc": float(ut.acc(logits, y))})
    with open(os.path.join(HERE, "reference_io.json"), "w") as f:
        json.dump(fix, f, indent=1)
    print("wrote reference_io.json:", len(fix["checkpoint"]), "checkpoint states,", len(fix["metrics"]), "metric cases")


if __name__ == "__main__":
    main()
