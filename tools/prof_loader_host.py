#!/usr/bin/env python
"""cProfile of loader-fed training steps (a NEW batch of 8 slides per step, resident data set): where the host time of a step with a graph it
has not seen goes.  `python tools/prof_loader_host.py [steps]`  (GPU)"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.autograd.set_multithreading_enabled(False)
import __graft_entry__
__graft_entry__.build()
from wsi_hgnn_amd import models, ops, synthetic
from wsi_hgnn_amd.data import GraphBatchLoader
from wsi_hgnn_amd.optim import Adam
from wsi_hgnn_amd.trainer import apply_loss
dev = torch.device("cuda:0")
ops.set_gemm_precision("auto")
torch.manual_seed(611)
m = models.HEATNet4(1024, 512, 2, 2, 4, {"0": 0, "1": 1, "2": 2}, 0.0, "mean").to(dev)
opt = Adam(m.parameters(), lr=1e-5)
lf = torch.nn.CrossEntropyLoss()
pool = [synthetic.hetero_graph(10000, 1024, seed=7000 + i) for i in range(32)]
loader = GraphBatchLoader(pool, [i % 2 for i in range(32)], batch_size=8, device=dev, shuffle=True, resident=True)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24


def run(n):
    k = 0
    while k < n:
        for G, y in loader:
            opt.zero_grad(set_to_none=True)
            apply_loss(lf, m(G), y).backward()
            opt.step()
            k += 1
            if k >= n:
                break


run(8)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
run(steps)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
