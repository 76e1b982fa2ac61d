#!/usr/bin/env python
"""Which lines of the package launch the eager (at::native / Tensile) kernels of the HGT + ASAP step (BASELINE configs[4] shape): GPU time of every
torch operator attributed to the innermost wsi-hgnn_amd source line on its Python stack.  GPU.  usage: python tools/prof_hgt_ops.py [hgt]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
from wsi_hgnn_amd import models, synthetic, ops
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
ND = {"0": 0, "1": 1, "2": 2}
rels = [(str(s), r, str(t)) for r in ("pos", "neg") for s in range(3) for t in range(3)]
ed = {et: i for i, et in enumerate(rels)}
torch.manual_seed(611)
ops.set_gemm_precision("auto")
cls = models.HGT if (len(sys.argv) > 1 and sys.argv[1] == "hgt") else models.HGTASAP
m = cls(ND, ed, 1024, 200, 2, 2, 4).to(dev).train()
G, y = synthetic.hetero_batch(4, 20000, 1024, rank=0, dst_mode="uniform", edges_per_dst=3)
G = G.to(dev); y = y.to(dev)
opt = torch.optim.Adam([p for p in m.parameters()], lr=1e-5)
lf = torch.nn.CrossEntropyLoss()
def step():
    opt.zero_grad(set_to_none=True)
    l = lf(m(G), y); l.backward(); opt.step(); return l
for _ in range(3): step()
torch.cuda.synchronize()
R = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(R): step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.self_device_time_total <= 0:
        continue
    site = "(autograd / optimizer)"
    for fr in (e.stack or []):
        if "wsi-hgnn_amd" in fr or "wsi_hgnn_amd" in fr:
            site = fr.split("wsi-hgnn_amd/")[-1]
            break
    a = agg[(site, e.name)]
    a[0] += e.self_device_time_total; a[1] += 1
tot = sum(v[0] for v in agg.values())
print(f"eager operator GPU time per step: {tot / R / 1e3:.3f} ms")
for (site, name), (us, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{us / R:9.1f} us/step {n / R:6.1f} calls  {name:28s} {site}")
