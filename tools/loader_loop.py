import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__; __graft_entry__.build()
from wsi_hgnn_amd import models, synthetic
from wsi_hgnn_amd.data import GraphBatchLoader
dev = torch.device("cuda:0")
nd = {"0": 0, "1": 1, "2": 2}
torch.manual_seed(611)
model = models.HEATNet4(1024, 512, 2, 2, 4, nd, 0.0, "mean").to(dev)
pool = [synthetic.hetero_graph(10000, 1024, seed=7000 + i) for i in range(16)]
opt = torch.optim.Adam(model.parameters(), lr=1e-5)
loss_fn = torch.nn.CrossEntropyLoss()
loader = GraphBatchLoader(pool, [i % 2 for i in range(16)], 8, dev, shuffle=True, drop_last=True, resident=True)
def run(n, per_step=None):
    d = 0
    while d < n:
        for G, y in loader:
            t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True); loss_fn(model(G), y).backward(); opt.step(); d += 1
            if per_step is not None:
                torch.cuda.synchronize(); per_step.append(round((time.perf_counter() - t0) * 1e3, 1))
            if d >= n: break
run(6); torch.cuda.synchronize()
ps = []
run(8, ps)
print("per-step ms with sync:", ps)
t0 = time.perf_counter(); run(10); torch.cuda.synchronize(); print("loop ms/step", (time.perf_counter() - t0) / 10 * 1e3)
ms0 = torch.cuda.memory_stats()
ps = []
run(8, ps)
ms1 = torch.cuda.memory_stats()
print("per-step:", ps)
for k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "segment.all.allocated", "segment.all.freed", "reserved_bytes.all.peak", "reserved_bytes.all.current", "allocated_bytes.all.peak"):
    print(k, ms0.get(k), "->", ms1.get(k))
print("---- detailed")
rows = []
d = 0
while d < 12:
    it = iter(loader)
    while True:
        t0 = time.perf_counter()
        try:
            G, y = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        opt.zero_grad(set_to_none=True); out = model(G); t2 = time.perf_counter()
        loss = loss_fn(out, y); loss.backward(); t3 = time.perf_counter()
        opt.step(); t4 = time.perf_counter()
        torch.cuda.synchronize(); t5 = time.perf_counter()
        rows.append(tuple(round((b - a) * 1e3, 1) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))))
        d += 1
print("(next, fwd_host, bwd_host, opt_host, sync_wait):")
for r in rows: print(r)
