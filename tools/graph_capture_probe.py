#!/usr/bin/env python
"""Does a whole training step (forward + CE + backward + Adam) of the HIP path capture into ONE hipGraph, and what does replaying it buy where the
step is host-bound?  `python tools/graph_capture_probe.py [--model HEATNet2 --hidden 256 --nodes 5000 --batch 1]`  (GPU)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
import wsi_hgnn_amd as W
from wsi_hgnn_amd import models, ops, synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="HEATNet2"); ap.add_argument("--hidden", type=int, default=256); ap.add_argument("--nodes", type=int, default=5000)
ap.add_argument("--batch", type=int, default=1); ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--dropout", type=float, default=0.0, help="feat_drop of the layers (train mode): the captured step then draws its masks through a device word (ops.dropout_seed_base)")
ap.add_argument("--no-packed-cache", action="store_true", help="ops.set_packed_weight_cache(False): every projection packs its weights itself (A/B)")
ap.add_argument("--json", action="store_true", help="print one JSON object (bench.py's single_graph_step leg runs this script in a subprocess)")
a = ap.parse_args()
dev = torch.device("cuda:0")
ops.set_gemm_precision("auto")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _knobs
_knobs.apply()                                       # WSI_BACKGROUND_DW=0 etc. of tools/r04_probe_a.sh (tools/_knobs.py)
if a.no_packed_cache:
    ops.set_packed_weight_cache(False)
nd = {"0": 0, "1": 1, "2": 2}


def build():
    torch.manual_seed(611)
    m = getattr(models, a.model)(1024, a.hidden, 2, 2, 4, nd, a.dropout, "mean").to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True, capturable=True)
    return m, opt


gs = [synthetic.hetero_graph(a.nodes, 1024, seed=611 + i) for i in range(a.batch)]
G = (W.batch(gs) if a.batch > 1 else gs[0]).to(dev)
y = (torch.arange(a.batch, device=dev) % 2)
lf = torch.nn.CrossEntropyLoss()

# eager
m, opt = build()
def step(m, opt):
    opt.zero_grad(set_to_none=True)
    l = lf(m(G), y)
    l.backward()
    opt.step()
    return l
for _ in range(5):
    step(m, opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
el = [step(m, opt) for _ in range(a.steps)]
torch.cuda.synchronize()
eager_ms = (time.perf_counter() - t0) / a.steps * 1e3
eager_losses = [x.item() for x in el]

# captured: trainer.CapturedStep (side-stream warm-up of 5 steps, then one capture)
from wsi_hgnn_amd.trainer import CapturedStep
m, opt = build()
cs = CapturedStep(m, opt, lf, G, y, warmup=5)
torch.cuda.synchronize()
losses = []
t0 = time.perf_counter()
for _ in range(a.steps):
    losses.append(cs().clone())
torch.cuda.synchronize()
graph_ms = (time.perf_counter() - t0) / a.steps * 1e3
graph_losses = [x.item() for x in losses]
# the captured run took one more step (the capture itself is a real step only in effect of the replay): compare trajectories loosely
if a.json:
    import json
    print(json.dumps({"eager_ms_per_step": round(eager_ms, 4), "hipgraph_ms_per_step": round(graph_ms, 4), "edges": G.num_edges(),
                      "trajectories_equal": eager_losses == graph_losses, "dropout": a.dropout}))
    sys.exit(0)
print(f"{a.model} hidden {a.hidden}, {a.batch} x {a.nodes} nodes: eager {eager_ms:.3f} ms/step, one hipGraph per step {graph_ms:.3f} ms/step")
print("eager losses  ", [round(x, 6) for x in eager_losses[:4]], "...", round(eager_losses[-1], 6))
print("graph losses  ", [round(x, 6) for x in graph_losses[:4]], "...", round(graph_losses[-1], 6))
