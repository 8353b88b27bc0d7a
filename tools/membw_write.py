"""HBM write bandwidth as seen by plain kernels (torch fill / copy) at the sizes the training sweeps write: is ~2.6 TB/s of
stores a property of the device or of our kernels' store pattern?"""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e-3
for mb in (100, 400, 1600):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    y = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    tf = t(lambda: x.fill_(1.0))
    tc = t(lambda: y.copy_(x))
    tr = t(lambda: x.float().sum()) if mb <= 400 else float("nan")
    print("%5d MB: fill %.2f TB/s | copy %.2f TB/s (r+w) | " % (mb, n * 2 / tf / 1e12, 2 * n * 2 / tc / 1e12))
