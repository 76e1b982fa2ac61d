#!/usr/bin/env python
"""Per-call timing of the projection GEMMs inside the bench's HEATNet4 step (HIP events around every wsi_gemm_grouped call), grouped by
(op, epilogue, groups, M-sum, N, K): which launches of the step a GEMM kernel change actually moves.  GPU.
usage: python tools/gemm_calls.py [w]     (w: the register-fragment scaled-fp16 kernel instead of the LDS-DMA one)"""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import _native as N
N.use_measurement_library()        # the WSI_* kernel switches below exist only in the -DWSI_ABLATE build (csrc/common.h::knob)
import __graft_entry__
__graft_entry__.build()
import wsi_hgnn_amd as W
from wsi_hgnn_amd import models, synthetic, ops
if len(sys.argv) > 1 and sys.argv[1] == "w":
    os.environ["WSI_GEMM_F16_KERNEL"] = "w"
dev = torch.device("cuda:0")
torch.manual_seed(611)
m = models.HEATNet4(1024, 512, 2, 2, 4, {"0": 0, "1": 1, "2": 2}, 0.0, "mean").to(dev).train()
G, y = synthetic.hetero_batch(8, 10000, 1024, rank=0)
G, y = G.to(dev), y.to(dev)
ops.set_gemm_precision("auto")
recs = []
orig = ops._gemm
def timed(op, epilogue, groups, device):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(op, epilogue, groups, device); e1.record()
    gs = [g for g in groups if g["M"] > 0 and g["N"] > 0]
    recs.append((("NT", "NN", "TN")[op], epilogue, len(gs), sum(g["M"] for g in gs), gs[0]["N"], gs[0]["K"], sum(2.0 * g["M"] * g["N"] * g["K"] for g in gs), e0, e1))
ops._gemm = timed
def step():
    m.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(m(G), y).backward()
for _ in range(3): step()
torch.cuda.synchronize(); recs.clear()
R = 5
for _ in range(R): step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for op, epi, ng, M, Nn, K, fl, e0, e1 in recs:
    k = (op, epi, ng, M, Nn, K)
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = 0.0
for (op, epi, ng, M, Nn, K), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += ms / R
    print(f"{op} epi={epi:3d} groups={ng:2d} Msum={M:7d} N={Nn:5d} K={K:6d}  calls/step={n / R:4.1f}  ms/step={ms / R:7.3f}  TF-eq={fl / ms / 1e9:7.1f}")
print("total GEMM ms/step", round(tot, 3))
