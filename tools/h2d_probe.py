"""Probe: host->device copy rates from pinned memory (one big copy vs many pieces) and host-side staging cost."""
import time
import torch

n = 82_000_000
x = torch.empty(n, dtype=torch.float32).pin_memory()
d = torch.empty_like(x, device="cuda")


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


dt = t(lambda: d.copy_(x, non_blocking=True))
print(f"H2D one 328MB copy: {dt*1e3:.2f} ms {n*4/dt/1e9:.1f} GB/s")
for pieces in (8, 24, 96):
    parts = [torch.empty(n // pieces, dtype=torch.float32).pin_memory() for _ in range(pieces)]
    def many():
        o = 0
        for p in parts:
            d[o:o + p.numel()].copy_(p, non_blocking=True); o += p.numel()
    dt = t(many)
    print(f"H2D {pieces} pieces: {dt*1e3:.2f} ms {n*4/dt/1e9:.1f} GB/s  ({dt/pieces*1e6:.0f} us/piece)")
    for th in (8, 32):
        torch.set_num_threads(th)
        def stage():
            o = 0
            for p in parts:
                x[o:o + p.numel()].copy_(p); o += p.numel()
        t0 = time.perf_counter(); stage(); stage(); dt = (time.perf_counter() - t0) / 2
        print(f"   host staging into one pinned buffer, {th} threads: {dt*1e3:.2f} ms {n*4/dt/1e9:.1f} GB/s")
s2 = torch.cuda.Stream()
def two_streams():
    half = n // 2
    with torch.cuda.stream(s2):
        d[:half].copy_(x[:half], non_blocking=True)
    d[half:].copy_(x[half:], non_blocking=True)
dt = t(two_streams)
print(f"H2D two halves on two streams: {dt*1e3:.2f} ms {n*4/dt/1e9:.1f} GB/s")
