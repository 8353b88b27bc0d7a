"""Times the weight-gradient contraction dW = gz^T x at the explicit critic step's shape (3B = 196 608 rows, 256 x 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops

M = int(os.environ.get("M", 196608))
SH = ((256, 256),) if os.environ.get('ONLY256') else ((256, 256), (100, 512), (256, 64), (256, 32), (256, 48), (100, 100), (1, 100), (1, 256))
for N1, N2 in SH:
    c16 = lambda n: (n + 15) // 16 * 16
    g = (torch.randn(M, c16(N1), device="cuda") * 0.1).bfloat16()
    x = (torch.randn(M, c16(N2), device="cuda") * 0.1).bfloat16()
    g[:, N1:] = 0
    x[:, N2:] = 0
    out = torch.zeros(N1, N2, device="cuda")
    cs = torch.zeros(N1, device="cuda")
    fn = lambda: ops.gemm_tn(g, x, N1, N2, colsum=cs, out=out, accumulate=True, colsum_rows=M // 3 * 2)
    for tn256 in ((True, False) if ((N1, N2) == (256, 256) and not os.environ.get('ONLY256')) else (True,)):     # whole-output kernel vs 64 x 64 tiles
        ops.TN256 = tn256
        for _ in range(20): fn()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(50): fn()
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 50 * 1e3)
        fl = 2.0 * M * N1 * N2
        by = M * (N1 + N2) * 2
        print("N1=%d N2=%d tn256=%s: %.1f us  %.0f TFLOP/s  %.2f TB/s of operand bytes" % (N1, N2, tn256, best, fl / best / 1e6, by / best / 1e6), flush=True)
    ops.TN256 = True
