#!/usr/bin/env python
"""Experiment: the bench step (8 graphs) as ONE batch vs as 2 / 4 concurrent micro-batches on separate HIP streams (the
memory-bound attention kernels of one micro-batch run beside the matrix-bound GEMMs of the other).  GPU."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wsi_hgnn_amd as W
from wsi_hgnn_amd import models, synthetic, ops

dev = torch.device("cuda:0")
torch.manual_seed(611)
nd = {"0": 0, "1": 1, "2": 2}
model = models.HEATNet4(1024, 512, 2, 2, 4, nd, 0.0, "mean").to(dev).train()
gs = [synthetic.hetero_graph(10000, 1024, seed=611 + i, dst_mode=os.environ.get("DST", "uniform")) for i in range(8)]
labels = torch.randint(0, 2, (8,), generator=torch.Generator().manual_seed(1)).to(dev)
loss_fn = torch.nn.CrossEntropyLoss()
opt = torch.optim.Adam([p for n, p in model.named_parameters() if n not in set(model.dead_parameter_names())], lr=1e-5, weight_decay=5e-3, fused=True)
res = {}
for mode in ("fp32", "bf16x6"):
    ops.set_gemm_precision(mode)
    for k in (1, 2, 4):
        per = 8 // k
        Gs = [W.batch(gs[i * per:(i + 1) * per]).to(dev) for i in range(k)]
        streams = [torch.cuda.Stream() for _ in range(k)]

        def step():
            opt.zero_grad(set_to_none=True)
            cur = torch.cuda.current_stream()
            outs = []
            if k == 1:
                outs.append(model(Gs[0]))
            else:
                for G, s in zip(Gs, streams):
                    s.wait_stream(cur)
                    with torch.cuda.stream(s):
                        outs.append(model(G))
                for s in streams:
                    cur.wait_stream(s)
            out = torch.cat(outs) if k > 1 else outs[0]
            loss = loss_fn(out, labels)
            loss.backward()
            opt.step()
            return loss
        for _ in range(5):
            l = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            l = step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        res[f"{mode}_k{k}"] = {"ms_per_step": round(ms, 3), "loss": float(l)}
        print(mode, k, round(ms, 3), float(l), flush=True)
print(json.dumps(res))
