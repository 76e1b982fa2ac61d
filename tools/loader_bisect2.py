import os, sys, time, gc
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
G0, y0, _ = loader._assemble(list(range(8)), 0)
def step(G, y):
    opt.zero_grad(set_to_none=True); loss_fn(model(G), y).backward(); opt.step()
def timeit(label, fn, n=12):
    for _ in range(4): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    print(f"{label:64s} {ts}", flush=True)
step(G0, y0)
def g1():
    G0.__dict__.pop("_heat_ctx", None); G0.__dict__.pop("_readout_plans", None); step(G0, y0)
timeit("G1: same graph, contexts rebuilt every step", g1)
def g2():
    G, y, _ = loader._assemble(list(range(8)), 0)
    G.__dict__["_heat_ctx"] = G0.__dict__["_heat_ctx"]; G.__dict__["_readout_plans"] = G0.__dict__["_readout_plans"]
    step(G, y)
step(G0, y0)
timeit("G2: new graph tensors, contexts shared from G0", g2)
def g3():
    G0.__dict__.pop("_heat_ctx", None); G0.__dict__.pop("_readout_plans", None)
    with torch.no_grad():
        model(G0)
timeit("G3: contexts rebuilt, forward only (no autograd)", g3)
from wsi_hgnn_amd.models.heat_layer import heat_context
def g4():
    G0.__dict__.pop("_heat_ctx", None)
    heat_context(G0, nd, 512, dev); step(G0, y0) if False else None
timeit("G4: only heat_context() rebuild, no model", g4)
