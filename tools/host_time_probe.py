#!/usr/bin/env python
"""Host time to ENQUEUE one training step (no sync inside the loop) against the GPU time of the step: the margin by which the host runs ahead.
`python tools/host_time_probe.py [--model HEATNet2 --hidden 256 --nodes 5000]`  (GPU)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
import wsi_hgnn_amd as W
from wsi_hgnn_amd import models, ops, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="HEATNet4"); ap.add_argument("--hidden", type=int, default=512); ap.add_argument("--nodes", type=int, default=10000)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--torch-adam", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
ops.set_gemm_precision("auto")
nd = {"0": 0, "1": 1, "2": 2}
torch.manual_seed(611)
m = getattr(models, a.model)(1024, a.hidden, 2, 2, 4, nd, 0.0, "mean").to(dev)
if a.torch_adam:
    opt = torch.optim.Adam(m.parameters(), lr=1e-5, fused=True)
else:
    from wsi_hgnn_amd.optim import Adam
    opt = Adam(m.parameters(), lr=1e-5)
lf = torch.nn.CrossEntropyLoss()
G = W.batch([synthetic.hetero_graph(a.nodes, 1024, seed=611 + i) for i in range(8)]).to(dev)
y = torch.arange(8, device=dev) % 2


def step():
    opt.zero_grad(set_to_none=True)
    lf(m(G), y).backward()
    opt.step()


for _ in range(int(os.environ.get("WARM", "5"))):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
per = []
for _ in range(a.steps):
    s0 = time.perf_counter()
    step()
    per.append(time.perf_counter() - s0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{a.model} hidden {a.hidden} nodes {a.nodes}: host enqueue {1e3 * (t1 - t0) / a.steps:.2f} ms/step, wall {1e3 * (t2 - t0) / a.steps:.2f} ms/step "
      f"(queue drained {1e3 * (t2 - t1):.1f} ms after the last enqueue); slowest host steps [ms]: {[round(1e3 * x, 1) for x in sorted(per)[-4:]]}, at step {max(range(len(per)), key=lambda i: per[i])}")
