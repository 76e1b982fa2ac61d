#!/usr/bin/env python
"""K | Q | V against K | V | Q as the column order of the fused layer's [n, 3D] table (ops.set_kqv_layout): steps of the full-depth configuration
(every layer forms V: --dropout 0.2 style training config and the default with the last layer at full depth), interleaved in one process, and bit
equality of the gradients (the arithmetic does not depend on the layout).  GPU.  usage: python tools/layout_probe.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wsi_hgnn_amd as W
from wsi_hgnn_amd import models, synthetic, ops
from wsi_hgnn_amd.optim import Adam
dev = torch.device("cuda:0")
ops.set_gemm_precision("auto")
G, y = synthetic.hetero_batch(8, 10000, 1024, rank=0, dst_mode="uniform")
G, y = G.to(dev), y.to(dev)
nd = {"0": 0, "1": 1, "2": 2}
res, grads = {}, {}
for rep in range(3):
    for order in ("kqv", "kvq"):
        ops.set_kqv_layout(order)
        torch.manual_seed(611)
        m = models.HEATNet4(1024, 512, 2, 2, 4, nd, 0.0, "mean").to(dev)
        m.fuse_readout = False                      # both layers at full depth: two full attention forward / backward pairs per step
        opt = Adam(m.parameters(), lr=1e-5)
        def step():
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(m(G), y).backward()
            opt.step()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            step()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(order, []).append(e0.elapsed_time(e1) / 20)
        if rep == 0:
            torch.manual_seed(611)
            m2 = models.HEATNet4(1024, 512, 2, 2, 4, nd, 0.0, "mean").to(dev)
            m2.fuse_readout = False
            torch.nn.functional.cross_entropy(m2(G), y).backward()
            grads[order] = [p.grad.clone() for p in m2.parameters() if p.grad is not None]
same = all(torch.equal(a, b) for a, b in zip(grads["kqv"], grads["kvq"]))
out = {"ms_per_step_full_depth": {k: round(min(v), 4) for k, v in res.items()}, "all": res, "gradients_bit_identical": same}
print(json.dumps(out))
