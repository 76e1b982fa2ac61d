"""Find hidden host<->device synchronisations in the loader + first step on a new batch (diagnostic)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__; __graft_entry__.build()
from wsi_hgnn_amd import models, synthetic
from wsi_hgnn_amd.data import GraphBatchLoader
dev = torch.device("cuda:0")
nd = {"0": 0, "1": 1, "2": 2}
torch.manual_seed(611)
model = models.HEATNet4(256, 128, 2, 2, 4, nd, 0.0, "mean").to(dev)
pool = [synthetic.hetero_graph(2000, 256, seed=7000 + i) for i in range(8)]
opt = torch.optim.Adam(model.parameters(), lr=1e-5)
loss_fn = torch.nn.CrossEntropyLoss()
loader = GraphBatchLoader(pool, [i % 2 for i in range(8)], 4, dev, shuffle=True, drop_last=True, resident=True)
def run(n):
    d = 0
    while d < n:
        for G, y in loader:
            opt.zero_grad(set_to_none=True); loss_fn(model(G), y).backward(); opt.step(); d += 1
            if d >= n: break
run(4)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    run(2)
    torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("default")
import collections
c = collections.Counter()
for x in w:
    c[(os.path.basename(x.filename), x.lineno, str(x.message)[:80])] += 1
for k, v in c.most_common(20):
    print(v, k)
import traceback
