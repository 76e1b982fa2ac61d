import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
m = models.HGTASAP(ND, ed, 1024, 200, 2, 2, 4).to(dev).train()
G, y = synthetic.hetero_batch(4, 20000, 1024, rank=0, dst_mode="uniform", edges_per_dst=3)
G = G.to(dev); y = y.to(dev)
opt = torch.optim.Adam([p for p in m.parameters()], lr=1e-5)
lf = torch.nn.CrossEntropyLoss()
def step():
    opt.zero_grad(set_to_none=True)
    l = lf(m(G), y); l.backward(); opt.step(); return l
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=60, max_name_column_width=60, max_src_column_width=110))
