"""Is a small async H2D from pinned memory ever slow on the host while the GPU is busy? (diagnostic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wsi_hgnn_amd.graph import host_to_device, _PinnedArena
dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev); b = torch.randn(8192, 8192, device=dev)
torch.mm(a, b); torch.cuda.synchronize()
vals = list(range(1000))
big = torch.arange(80000)
def trial(label, fn, busy):
    ts = []
    for i in range(30):
        if busy:
            for _ in range(3): torch.mm(a, b)      # ~20 ms of queued GPU work
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
    ts.sort()
    print(f"{label:55s} busy={busy}: median {ts[15]:.3f} ms  max {ts[-1]:.3f} ms")
for busy in (False, True):
    trial("host_to_device(list of 1000 ints)", lambda: host_to_device(vals, torch.int32, dev), busy)
    trial("host_to_device(80k int64 tensor)", lambda: host_to_device(big, torch.int64, dev), busy)
    trial("torch.tensor(list, device=cuda) [pageable]", lambda: torch.tensor(vals, dtype=torch.int32, device=dev), busy)
    trial("torch.cuda.Event().record()", lambda: torch.cuda.Event().record(), busy)
    trial("torch.empty(492MB) + free", lambda: torch.empty(123_000_000, device=dev), busy)
    x = torch.zeros(100, device=dev)
    trial("x[5] = 3 (scalar setitem)", lambda: x.__setitem__(5, 3), busy)
    trial("x[5:6].fill_(3)", lambda: x[5:6].fill_(3), busy)
