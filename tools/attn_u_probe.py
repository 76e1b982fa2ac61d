"""Rows in flight per wave of the SHIPPED attention forward (review, round 5: "U = 2 ... no in-flight sweep is recorded"): the full-depth forward of the
bench batch (8 x 10k nodes, D = 512, H = 4) with U = 1 / 2 (shipped) / 3 / 4 gathered row pairs in flight per wave (measurement build: WSI_ATTN_U).  GPU."""
import json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd import _native as N
N.use_measurement_library()
from wsi_hgnn_amd import ops, synthetic
lib = N.load()
dev = torch.device("cuda:0")
g, _ = synthetic.hetero_batch(8, 10000, in_dim=8)
g = g.to(dev)
plan = g.plan()
sim = g.cat_edata_csr("sim")
D, H = 512, 4
n, E, S = plan.num_nodes, plan.num_edges, plan.num_segs
torch.manual_seed(3)
kqv = torch.randn(n, 3 * D, device=dev) * 0.5
ew, eb = torch.tensor([0.7], device=dev), torch.tensor([0.3], device=dev)
t = torch.empty(n, D, device=dev); sc = torch.empty(E, H, device=dev); ls = torch.zeros(S, H, device=dev)


def fwd():
    N.check(lib.wsi_heat_attn_fwd(N.ptr(kqv, D * 4), 3 * D, N.ptr(kqv), 3 * D, N.ptr(kqv, 8 * D), 3 * D, n, D, H, N.ptr(plan.node_seg), N.ptr(plan.rowptr), N.ptr(plan.src),
                                  N.ptr(sim), N.ptr(plan.order_dst), plan.num_heavy, ops._attn_flags(plan), N.ptr(ew), N.ptr(eb), N.ptr(t), D, N.ptr(sc), N.ptr(ls), None,
                                  N.context(), N.stream()), "fwd")


out, ref = {}, None
for u in ("2", "1", "3", "4", "2"):
    os.environ["WSI_ATTN_U"] = u
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fwd(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    if ref is None:
        ref = t.clone()
    out.setdefault(f"U={u}", []).append(round(statistics.median(ts), 1))
    print(f"U={u}: {statistics.median(ts):.1f} us, max |t - t(U=2)| = {(t - ref).abs().max().item():.2e}", flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
