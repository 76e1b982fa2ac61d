#!/usr/bin/env python
"""cProfile of loader-fed training steps (a NEW batch of slides per step, resident data set): where the host time of a step with a graph it
has not seen goes.  `[DROPOUT=0.2] python tools/prof_loader_host.py [steps] [batch]`  (GPU)"""
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
m = models.HEATNet4(1024, 512, 2, 2, 4, {"0": 0, "1": 1, "2": 2}, float(os.environ.get("DROPOUT", "0.0")), "mean").to(dev)
opt = Adam(m.parameters(), lr=1e-5)
lf = torch.nn.CrossEntropyLoss()
pool = [synthetic.hetero_graph(10000, 1024, seed=7000 + i) for i in range(32)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 24
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
loader = GraphBatchLoader(pool, [i % 2 for i in range(32)], batch_size=batch, device=dev, shuffle=True, resident=True)


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
import time
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
print(f'batch {batch}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per loader-fed step (wall, unprofiled)')
pr = cProfile.Profile()
pr.enable()
run(steps)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
