"""Host/GPU timing of the batch loader vs the training step (diagnostic)."""
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

def sync_time(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return r, (t1 - t0) * 1e3, (t2 - t0) * 1e3

for resident in (True, False):
    loader = GraphBatchLoader(pool, [i % 2 for i in range(16)], 8, dev, shuffle=True, drop_last=True, resident=resident)
    for rep in range(3):
        (G, y, ready), h_ms, tot_ms = sync_time(lambda: loader._assemble(list(range(8)), 0))
        def step():
            opt.zero_grad(set_to_none=True)
            loss_fn(model(G), y).backward()
            opt.step()
        _, sh, st = sync_time(step)      # first step on a new graph object: builds contexts
        _, sh2, st2 = sync_time(step)    # second step on the same graph: contexts cached
        print(f"resident={resident} rep{rep}: assemble host {h_ms:.2f} ms, host+gpu {tot_ms:.2f} ms | first step host {sh:.2f} total {st:.2f} | cached step host {sh2:.2f} total {st2:.2f}")

print("---- loop timing")
import gc
for resident in (True, False):
    loader = GraphBatchLoader(pool, [i % 2 for i in range(16)], 8, dev, shuffle=True, drop_last=True, resident=resident)
    def run(nsteps, gc_off=False):
        done = 0
        while done < nsteps:
            for G, y in loader:
                opt.zero_grad(set_to_none=True)
                loss_fn(model(G), y).backward()
                opt.step()
                done += 1
                if done >= nsteps: break
    run(6); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(12); torch.cuda.synchronize(); print(f"resident={resident} loop: {(time.perf_counter()-t0)/12*1e3:.2f} ms/step")
    gc.disable()
    t0 = time.perf_counter(); run(12); torch.cuda.synchronize(); print(f"resident={resident} loop, gc disabled: {(time.perf_counter()-t0)/12*1e3:.2f} ms/step")
    gc.enable()
    # host time per iteration
    ts = []
    it = iter(loader)
    for _ in range(2):
        t0 = time.perf_counter(); G, y = next(it); t1 = time.perf_counter()
        opt.zero_grad(set_to_none=True); loss_fn(model(G), y).backward(); opt.step(); t2 = time.perf_counter()
        ts.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2)))
    print("   host ms (next(), step):", ts)
    print("   mem allocated GB", torch.cuda.memory_allocated() / 1e9, "reserved GB", torch.cuda.memory_reserved() / 1e9)
