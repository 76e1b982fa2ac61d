#!/usr/bin/env python
"""Where a loader-fed training step spends its HOST time: per step, the wall time of next(loader) / forward / backward / optimizer as the
host sees them (asynchronous launches: a phase that takes long on the host is one that blocked), the time spent waiting inside the
pinned staging arena, and the GPU time of the step.  `python tools/loader_step_probe.py [steps]`  (GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__
__graft_entry__.build()
import wsi_hgnn_amd as W
from wsi_hgnn_amd import models, ops, synthetic, graph as G_
from wsi_hgnn_amd.data import GraphBatchLoader

dev = torch.device("cuda:0")
ops.set_gemm_precision("auto")
nd = {"0": 0, "1": 1, "2": 2}
torch.manual_seed(611)
m = models.HEATNet4(1024, 512, 2, 2, 4, nd, 0.0, "mean").to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-5, fused=True)
lf = torch.nn.CrossEntropyLoss()
pool = [synthetic.hetero_graph(10000, 1024, seed=7000 + i) for i in range(16)]
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12

wait = {"t": 0.0, "n": 0}
real_sync = torch.cuda.Event.synchronize


def timed_sync(self):
    t0 = time.perf_counter()
    real_sync(self)
    wait["t"] += time.perf_counter() - t0
    wait["n"] += 1


torch.cuda.Event.synchronize = timed_sync
real_stage = G_._PinnedArena.stage


def timed_stage(self, t, device):
    """_PinnedArena.stage restated with a timer around each of its three steps."""
    nbytes = t.numel() * t.element_size()
    need = (nbytes + 255) // 256 * 256
    if nbytes == 0 or need > self.size:
        return real_stage(self, t, device)
    if self.off + need > self.size:
        self.off = 0
        self.lap += 1
    start, end = self.off, self.off + need
    t0 = time.perf_counter()
    while self.pending:
        lap, a, b, evt = self.pending[0]
        if lap == self.lap or (lap == self.lap - 1 and a >= end):
            break
        evt.synchronize()
        self.pending.popleft()
    t1 = time.perf_counter()
    view = self.buf[start:start + nbytes].view(t.dtype).view(t.shape)
    view.copy_(t)
    t2 = time.perf_counter()
    out = view.to(device, non_blocking=True)
    t3 = time.perf_counter()
    evt = torch.cuda.Event()
    evt.record(torch.cuda.current_stream(device))
    t4 = time.perf_counter()
    self.off = end
    self.pending.append((self.lap, start, end, evt))
    if t4 - t0 > 4e-3:
        import traceback
        print(f"   slow stage: ring wait {1e3 * (t1 - t0):.1f} | host memcpy {1e3 * (t2 - t1):.1f} | H2D call {1e3 * (t3 - t2):.1f} | event {1e3 * (t4 - t3):.1f} ms; "
              f"{nbytes} B, stream {torch.cuda.current_stream().cuda_stream:#x}, from " +
              " <- ".join(f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-5:-1]))
    return out


G_._PinnedArena.stage = timed_stage

# every C call or Python function of the main thread that takes longer than 4 ms, with where it was called from
slow = []
_stack = []


def prof(frame, event, arg):
    if event in ("c_call", "call"):
        _stack.append(time.perf_counter())
    elif event in ("c_return", "c_exception", "return"):
        if _stack:
            dt = time.perf_counter() - _stack.pop()
            if dt > 4e-3 and event != "return":
                slow.append((dt, getattr(arg, "__qualname__", repr(arg)), f"{frame.f_code.co_filename.split('/')[-1]}:{frame.f_lineno}"))


for resident in (False, True):
    loader = GraphBatchLoader(pool, [i % 2 for i in range(len(pool))], 8, dev, shuffle=True, drop_last=True, resident=resident)
    done = 0
    rows = []
    it = iter(loader)
    torch.cuda.synchronize()
    wall0 = time.perf_counter()
    while done < steps:
        if done == 2:
            sys.setprofile(prof)
        t0 = time.perf_counter()
        try:
            Gb, yb = next(it)
        except StopIteration:
            it = iter(loader)
            Gb, yb = next(it)
        t1 = time.perf_counter()
        w0 = (wait["t"], wait["n"])
        opt.zero_grad(set_to_none=True)
        loss = lf(m(Gb), yb)
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        opt.step()
        t4 = time.perf_counter()
        rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, wait["t"] - w0[0], wait["n"] - w0[1]))
        done += 1
    sys.setprofile(None)
    torch.cuda.synchronize()
    wall = time.perf_counter() - wall0
    print(f"resident={resident}: {wall / steps * 1e3:.2f} ms/step wall;  per step [ms]: next / forward / backward / optimizer | arena wait ms (events)")
    for r in rows:
        print("   " + " ".join(f"{x * 1e3:7.2f}" for x in r[:5]) + f" ({r[5]})")
    for dt, name, where in slow:
        print(f"   slow C call: {dt * 1e3:7.2f} ms  {name}  at {where}")
    slow.clear()
