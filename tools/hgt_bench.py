#!/usr/bin/env python
"""configs[4]-shaped timing: HGT (hidden 200, 4 heads, 2 layers) on a batch of 20k-node synthetic graphs, one GPU,
fwd + CE + bwd + Adam, with the per-kernel HIP-event breakdown of ops.py."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
if os.environ.get("WSI_TN_AUTO_GFLOP"):                 # A/B of auto's weight-gradient threshold: the switch exists in the measurement build only
    from wsi_hgnn_amd import _native as _N
    _N.use_measurement_library()
import __graft_entry__
__graft_entry__.build()
from wsi_hgnn_amd import models, synthetic, ops
ops.set_gemm_precision(os.environ.get("GEMM", "auto"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _knobs
_knobs.apply()                                       # WSI_GEMM_PRECISION / WSI_BACKGROUND_DW / ... of the old command lines (tools/_knobs.py)
if os.environ.get("WSI_TN_AUTO_GFLOP"):
    ops._TN_AUTO["flop"] = float(os.environ["WSI_TN_AUTO_GFLOP"]) * 1e9


dev = torch.device("cuda:0")
ND = {"0": 0, "1": 1, "2": 2}
rels = [(str(s), r, str(t)) for r in ("pos", "neg") for s in range(3) for t in range(3)]
ed = {et: i for i, et in enumerate(rels)}
torch.manual_seed(611)
B = int(os.environ.get("B", "4"))
ASAP = os.environ.get("ASAP", "0") == "1"          # configs[4] as BASELINE.json words it: HGT + ASAP pooling (models/HGT_ASAP.py)
EDGES = os.environ.get("EDGES", "0") == "1"        # HGT + ASAP: also build the pooled graph's edges (E = S^T A S), which this composition does not consume
m = (models.HGTASAP(ND, ed, 1024, 200, 2, 2, 4, pooled_edges=EDGES) if ASAP else models.HGT(ND, ed, 1024, 200, 2, 2, 4)).to(dev).train()
G, y = synthetic.hetero_batch(B, 20000, 1024, rank=0, dst_mode=os.environ.get("DST", "uniform"), edges_per_dst=3)
G = G.to(dev); y = y.to(dev)
from wsi_hgnn_amd.optim import Adam
opt = Adam([p for p in m.parameters()], lr=1e-5)        # (the reference's Adam in one launch: wsi_adam_step, as bench.py)
lf = torch.nn.CrossEntropyLoss()

def step():
    opt.zero_grad(set_to_none=True)
    l = lf(m(G), y)
    l.backward()
    opt.step()
    return l

if os.environ.get("PER_STEP"):                       # wall time of every step from the first (synchronised): how long the start-up transient lasts
    ts = []
    for _ in range(int(os.environ["PER_STEP"])):
        torch.cuda.synchronize(); t_ = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t_) * 1e3, 2))
    print("per-step ms:", ts, file=sys.stderr)
for _ in range(int(os.environ.get("WARM", "3"))):
    step()
torch.cuda.synchronize()
import gc
gc.collect(); gc.freeze()          # (as bench.py: a full cyclic collection - ~80 ms in this process, once around the 50th step - must not land in the timed steps)
t0 = time.perf_counter()
K = int(os.environ.get("STEPS", "10"))
for _ in range(K):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
ops.enable_kernel_timing(True)
for _ in range(5):
    step()
torch.cuda.synchronize()
st = ops.kernel_timing_summary()
print(json.dumps({"model": (("HGT + ASAPPooling" + (" (+ pooled edges)" if EDGES else "")) if ASAP else "HGT") + " hidden 200, 4 heads, 2 layers", "graphs": B, "nodes": G.num_nodes(), "edges": G.num_edges(),
                  "gemm": os.environ.get("GEMM", "auto"), "ms_per_step": round(ms, 3), "edges_per_s": round(G.num_edges() / (ms * 1e-3)),
                  "kernels_ms_per_step": {k: round(v["ms"] / 5, 3) for k, v in st.items()}}))
