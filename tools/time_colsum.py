"""Development aid: dhaug_colsum_f32 on a 3B x 256 fp32 cotangent (the parity-grade step's bias gradients)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops
for M, N in ((196608, 256), (131072, 256), (196608, 100)):
    X = torch.randn(M, N, device="cuda")
    out = torch.zeros(N, device="cuda")
    for _ in range(5): ops.colsum(X, out=out, accumulate=True)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): ops.colsum(X, out=out, accumulate=True)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 50.0
    print("%d x %d: %.1f us = %.2f TB/s" % (M, N, us, M * N * 4 / us / 1e6))
