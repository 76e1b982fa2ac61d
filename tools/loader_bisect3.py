import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__; __graft_entry__.build()
from wsi_hgnn_amd import models, synthetic, batch
from wsi_hgnn_amd.graph import host_to_device
dev = torch.device("cuda:0")
nd = {"0": 0, "1": 1, "2": 2}
torch.manual_seed(611)
model = models.HEATNet4(1024, 512, 2, 2, 4, nd, 0.0, "mean").to(dev)
G = batch([synthetic.hetero_graph(10000, 1024, seed=7000 + i) for i in range(8)]).to(dev)
y = torch.zeros(8, dtype=torch.int64, device=dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-5)
loss_fn = torch.nn.CrossEntropyLoss()
def timeit(label, fn, n=12):
    for _ in range(4): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    print(f"{label:64s} {ts}", flush=True)
small = list(range(600))
pinned = torch.arange(600, dtype=torch.int32).pin_memory()
def variant(k, how):
    def f():
        opt.zero_grad(set_to_none=True)
        out = model(G)
        keep = []
        for _ in range(k):
            if how == "arena": keep.append(host_to_device(small, torch.int32, dev))
            elif how == "pinned": keep.append(pinned.to(dev, non_blocking=True))
            elif how == "event": e = torch.cuda.Event(); e.record(); keep.append(e)
            elif how == "fill": keep.append(torch.full((600,), 3, dtype=torch.int32, device=dev))
        loss_fn(out, y).backward(); opt.step()
    return f
timeit("baseline", variant(0, ""))
timeit("8 x arena host_to_device between fwd and bwd", variant(8, "arena"))
timeit("8 x persistent-pinned .to(non_blocking) between fwd and bwd", variant(8, "pinned"))
timeit("8 x Event().record() between fwd and bwd", variant(8, "event"))
timeit("8 x torch.full on device between fwd and bwd", variant(8, "fill"))
timeit("1 x persistent-pinned .to(non_blocking)", variant(1, "pinned"))
