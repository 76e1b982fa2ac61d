#!/usr/bin/env python
"""Host-side profile (cProfile, autograd on the calling thread) of the eager one-slide step - the launch-bound regime of the reference (one 10k-node
slide per step): where the Python time per step goes.  usage: python tools/prof_step_host.py [--wsi-adam] [--nodes 10000] [--top 40]   (GPU)"""
import argparse, cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.autograd.set_multithreading_enabled(False)
import wsi_hgnn_amd as W
from wsi_hgnn_amd import models, ops, synthetic
ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=10000); ap.add_argument("--top", type=int, default=40); ap.add_argument("--wsi-adam", action="store_true")
ap.add_argument("--steps", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda:0")
ops.set_gemm_precision("auto")
torch.manual_seed(611)
m = models.HEATNet4(1024, 512, 2, 2, 4, {"0": 0, "1": 1, "2": 2}, 0.0, "mean").to(dev).train()
if a.wsi_adam:
    from wsi_hgnn_amd.optim import Adam
    opt = Adam(m.parameters(), lr=1e-4)
else:
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
G = synthetic.hetero_graph(a.nodes, 1024, seed=611).to(dev)
y = torch.zeros(1, dtype=torch.long, device=dev)
lf = torch.nn.CrossEntropyLoss()
from wsi_hgnn_amd.trainer import apply_loss
def step():
    opt.zero_grad(set_to_none=True)
    l = apply_loss(lf, m(G), y)
    l.backward()
    opt.step()
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
host_ms = (time.perf_counter() - t0) / a.steps * 1e3          # time to ISSUE a step (no sync inside)
torch.cuda.synchronize()
wall_ms = (time.perf_counter() - t0) / a.steps * 1e3
print(f"issue time {host_ms:.3f} ms/step, wall {wall_ms:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(a.top)
