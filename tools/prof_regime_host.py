#!/usr/bin/env python
"""Host-side profile of the REFERENCE'S OWN regime (trainer/train_gnn.py:59-65): a NEW batch of 2 slides every step, feat_drop 0.2, loader-fed from
an HBM-resident data set - where the Python time per step goes (cProfile, autograd on the calling thread), and the split loader / forward /
backward / optimizer.   python tools/prof_regime_host.py [--batch 2] [--steps 200] [--top 45] [--no-profile]   (GPU)"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.autograd.set_multithreading_enabled(False)
from wsi_hgnn_amd import models, ops, synthetic
from wsi_hgnn_amd.data import GraphBatchLoader
from wsi_hgnn_amd.optim import Adam
from wsi_hgnn_amd.trainer import apply_loss
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=2); ap.add_argument("--nodes", type=int, default=10000); ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--top", type=int, default=45); ap.add_argument("--no-profile", action="store_true"); ap.add_argument("--dropout", type=float, default=0.2)
a = ap.parse_args()
dev = torch.device("cuda:0")
ops.set_gemm_precision("auto")
torch.manual_seed(611)
m = models.HEATNet4(1024, 512, 2, 2, 4, {"0": 0, "1": 1, "2": 2}, a.dropout, "mean").to(dev).train()
opt = Adam(m.parameters(), lr=1e-5, weight_decay=5e-3)
pool = [synthetic.hetero_graph(a.nodes, 1024, seed=7000 + i) for i in range(8 * a.batch)]
loader = GraphBatchLoader(pool, [i % 2 for i in range(len(pool))], a.batch, dev, shuffle=True, drop_last=True, resident=True, passes=8)
lf = torch.nn.CrossEntropyLoss()
split = {"loader": 0.0, "forward": 0.0, "backward": 0.0, "optimizer": 0.0}


def run(nsteps):
    done = 0
    while done < nsteps:
        it = iter(loader)
        while done < nsteps:
            h0 = time.perf_counter()
            try:
                G, y = next(it)
            except StopIteration:
                break
            h1 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            l = apply_loss(lf, m(G), y)
            h2 = time.perf_counter()
            l.backward()
            h3 = time.perf_counter()
            opt.step()
            h4 = time.perf_counter()
            split["loader"] += h1 - h0; split["forward"] += h2 - h1; split["backward"] += h3 - h2; split["optimizer"] += h4 - h3
            done += 1


run(120)
torch.cuda.synchronize()
for k in split:
    split[k] = 0.0
t0 = time.perf_counter()
run(a.steps)
host_ms = (time.perf_counter() - t0) / a.steps * 1e3
torch.cuda.synchronize()
wall_ms = (time.perf_counter() - t0) / a.steps * 1e3
print(f"issue time {host_ms:.3f} ms/step, wall {wall_ms:.3f} ms/step; host split (ms/step): " + ", ".join(f"{k} {v / a.steps * 1e3:.3f}" for k, v in split.items()))
if not a.no_profile:
    pr = cProfile.Profile()
    pr.enable()
    run(a.steps)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats("tottime").print_stats(a.top)
    st.sort_stats("cumtime").print_stats(35)
    st.print_callers("host_to_device")
    st.print_callers("torch.empty")
