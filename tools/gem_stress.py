#!/usr/bin/env python
"""Inference stress of SURVEY row n4 (second half): the GEM explainer's N-forward loop (explainers/gem_het.py:30-39) on one
10k-node slide — 10 000 leave-one-node-out forwards, batched.  Reports forwards/s.  GPU."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import models, synthetic
from wsi_hgnn_amd.explainers import HetGemExplainer

dev = torch.device("cuda:0")
n = int(os.environ.get("NODES", "10000"))
limit = int(os.environ.get("LIMIT", "640"))       # nodes actually explained (the full loop is n of them)
bs = int(os.environ.get("BATCH", "16"))
nd = {"0": 0, "1": 1, "2": 2}
torch.manual_seed(611)
m = models.HEATNet4(1024, 512, 2, 2, 4, nd, 0.0, "mean").to(dev).eval()
g = synthetic.hetero_graph(n, 1024, seed=611).to(dev)
ex = HetGemExplainer(g, m, torch.tensor([1], device=dev), batch_size=bs)
# time a slice of the loop: first `limit` nodes of type '0'
from wsi_hgnn_amd.graph import batch as batch_graphs, remove_nodes
gc = ex.graph
torch.cuda.synchronize()
t0 = time.perf_counter()
t_build = 0.0
with torch.no_grad():
    for start in range(0, limit, bs):
        b0 = time.perf_counter()
        bg = batch_graphs([remove_nodes(gc, torch.tensor([i]), "0") for i in range(start, min(start + bs, limit))])
        bg.plan()
        torch.cuda.synchronize()
        t_build += time.perf_counter() - b0
        m(bg)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"workload": f"HEATNet4 leave-one-node-out forwards on a {n}-node slide, {bs} altered graphs per forward",
                  "forwards": limit, "seconds": round(dt, 3), "forwards_per_s": round(limit / dt, 1),
                  "graph_and_plan_build_s": round(t_build, 3), "model_s": round(dt - t_build, 3),
                  "full_slide_estimate_s": round(dt * n / limit, 1)}))
