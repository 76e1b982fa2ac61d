#!/usr/bin/env python
"""Row n4 timing: GPU kNN + Pearson graph construction vs the CPU restatement of graph_constructor.py:263-282
(brute-force L2 + scipy.stats.pearsonr loop; nmslib is unavailable) on a bounded sample.  Run on the GPU box."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__
__graft_entry__.build()
from wsi_hgnn_amd import construct

n, F, radius, T = 10000, 1024, 9, 3
g = torch.Generator().manual_seed(611)
centres = torch.rand(32, F, generator=g)
x = (centres[torch.randint(0, 32, (n,), generator=g)] + 0.15 * torch.randn(n, F, generator=g)).clamp_(min=0).float()
node_type = torch.randint(0, T, (n,), generator=g).tolist()
xd = x.cuda()
for _ in range(2):
    construct.knn_pearson(xd, radius)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
e0.record()
for _ in range(reps):
    construct.knn_pearson(xd, radius)
e1.record()
torch.cuda.synchronize()
knn_ms = e0.elapsed_time(e1) / reps
construct.construct_graph(xd, node_type, radius, T)      # first call pays torch's lazy init of unique/nonzero
torch.cuda.synchronize()
t0 = time.perf_counter()
het, homo, _ = construct.construct_graph(xd, node_type, radius, T)
torch.cuda.synchronize()
full_ms = (time.perf_counter() - t0) * 1e3

from scipy.stats import pearsonr
xs = x.numpy()
rows = 200
t0 = time.perf_counter()
x64 = xs.astype(np.float64)
for i in range(rows):
    d2 = ((x64 - x64[i]) ** 2).sum(1)
    np.lexsort((np.arange(n), d2))[1:radius]
knn_cpu_s = (time.perf_counter() - t0) / rows * n
pairs = 4000
nb = torch.randint(0, n, (pairs, 2)).numpy()
t0 = time.perf_counter()
for a, b in nb:
    pearsonr(xs[a], xs[b])[0]
pear_cpu_s = (time.perf_counter() - t0) / pairs * n * (radius - 1)
out = {"config": f"N={n} patches, F={F}, radius={radius} ({n * (radius - 1)} edges), {T} node types",
       "gpu_knn_pearson_ms": round(knn_ms, 3), "gpu_construct_graph_ms_wall": round(full_ms, 2),
       "edges_per_s_gpu": round(n * (radius - 1) / (knn_ms * 1e-3)),
       "cpu_bruteforce_knn_s_extrapolated": round(knn_cpu_s, 1), "cpu_pearsonr_loop_s_extrapolated": round(pear_cpu_s, 1),
       "cpu_sample": f"{rows} of {n} brute-force float64 rows, {pairs} scipy.stats.pearsonr calls, scaled to the full graph; "
                     f"the reference's nmslib HNSW (unavailable) is faster than brute force, its pearsonr loop is this one",
       "relations": [list(r) for r in het.canonical_etypes]}
print(json.dumps(out))
