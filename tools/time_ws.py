"""Times the 100-wide backward layers of the 3D critic's top (gemm_nt_ws_kernel<7>): 3B x 112 -> 112 with a bf16 mask and a skip,
3B x 112 -> 512 with the mask as an image / as sign bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops, fused
M = int(os.environ.get("M", 196608))
bf = lambda t: t.to(torch.bfloat16)
g = bf(torch.randn(M, 112, device="cuda")); g[:, 100:] = 0
W = bf(torch.randn(112, 112, device="cuda") / 10)
Wm = bf(torch.randn(512, 112, device="cuda") / 10)
y = bf(torch.randn(M, 112, device="cuda"))
cat = bf(torch.randn(M, 512, device="cuda"))
catb = cat.clone()
nb = (M + 127) // 128 * 4 * 256
catb._dhaug_bits_cols = [torch.randint(-2**31, 2**31 - 1, (nb,), dtype=torch.int32, device="cuda") for _ in range(2)]
skip = bf(torch.randn(M, 112, device="cuda"))
cases = (("112 -> 100, mask", lambda: ops.gemm_nt_dmask(g, W, 100, 112, y, 1, 0.0), M * 112 * 2 * 3),
         ("112 -> 100, mask + skip", lambda: ops.gemm_nt_dmask(g, W, 100, 112, y, 1, 0.0, res_bf16=skip), M * 112 * 2 * 4),
         ("112 -> 512, mask image", lambda: ops.gemm_nt_dmask(g, Wm, 512, 112, cat, 1, 0.0), M * (112 + 512 + 512) * 2),
         ("112 -> 512, sign bits", lambda: ops.gemm_nt_dmask(g, Wm, 512, 112, catb, 1, 0.0), M * (112 + 512) * 2))
for name, fn, by in cases:
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 30 * 1e3)
    print("%-26s %.1f us  (%.0f MB -> %.2f TB/s)" % (name, best, by / 1e6, by / best / 1e6), flush=True)
