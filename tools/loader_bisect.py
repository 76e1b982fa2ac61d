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
G0, y0, _ = loader._assemble(list(range(8)), 0)
def step(G, y):
    opt.zero_grad(set_to_none=True); loss_fn(model(G), y).backward(); opt.step()
def timeit(label, fn, n=12):
    for _ in range(4): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
    print(f"{label:60s} {ts}")
timeit("A: same graph every step", lambda: step(G0, y0))
def b():
    loader._assemble(list(range(8)), 0); step(G0, y0)
timeit("B: assemble (discarded) + step on the same graph", b)
def c():
    G, y, _ = loader._assemble(list(range(8)), 0); step(G, y)
timeit("C: assemble + step on the NEW graph", c)
keep = []
def d():
    G, y, _ = loader._assemble(list(range(8)), 0); keep.append(G); step(G, y)
    if len(keep) > 3: keep.pop(0)
timeit("D: as C but keep the last 3 graphs alive", d)
import gc
def e():
    G, y, _ = loader._assemble(list(range(8)), 0); step(G, y); del G; gc.collect()
timeit("E: as C + explicit gc.collect()", e)
gc.collect(); gc.freeze()
timeit("F: as C after gc.freeze()", c)
print("gc counts", gc.get_count(), "frozen", gc.get_freeze_count())
print("---- gc stats")
import collections
def g():
    G, y, _ = loader._assemble(list(range(8)), 0); step(G, y)
for _ in range(3): g()
torch.cuda.synchronize()
before = collections.Counter(type(o).__name__ for o in gc.get_objects())
g(); g(); g()
torch.cuda.synchronize()
after = collections.Counter(type(o).__name__ for o in gc.get_objects())
diff = after - before
print("new tracked objects after 3 steps:", diff.most_common(12))
t0 = time.perf_counter(); n = gc.collect(); print("gc.collect() ms", (time.perf_counter() - t0) * 1e3, "collected", n)
t0 = time.perf_counter(); n = gc.collect(); print("gc.collect() again ms", (time.perf_counter() - t0) * 1e3, "collected", n)
print(gc.get_stats())
