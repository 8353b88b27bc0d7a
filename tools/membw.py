"""HBM streaming ceilings with torch's own kernels (device-to-device copy, fill, read-reduce)."""
import torch, time
def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for mb in (32, 128, 1024):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device="cuda").to(torch.bfloat16); y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x)); print("copy  %5d MB -> %7.1f us  %6.0f GB/s (read+write)" % (mb, t * 1e6, 2 * mb * 1.048576e6 / t / 1e9))
    t = timeit(lambda: y.zero_()); print("fill  %5d MB -> %7.1f us  %6.0f GB/s (write)" % (mb, t * 1e6, mb * 1.048576e6 / t / 1e9))
    xf = x.view(torch.int16)
    t = timeit(lambda: xf.sum()); print("sum   %5d MB -> %7.1f us  %6.0f GB/s (read)" % (mb, t * 1e6, mb * 1.048576e6 / t / 1e9))
