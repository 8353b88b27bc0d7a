"""Times the two-layer block launch (dhaug_gemm_block2_bf16) against the two dhaug_gemm_bf16_dbits launches it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops, fused
for M in (196608, 65536):
    bf = lambda t: t.to(torch.bfloat16)
    X = bf(torch.randn(M, 256, device="cuda"))
    W1 = bf(torch.randn(256, 256, device="cuda") / 16); W2 = bf(torch.randn(256, 256, device="cuda") / 16)
    t1 = torch.zeros(M, 256, dtype=torch.bfloat16, device="cuda"); t2 = torch.zeros_like(t1)
    nb = (M + 127) // 128 * 4 * 256
    t1._dhaug_bits = torch.randint(-2**31, 2**31 - 1, (nb,), dtype=torch.int32, device="cuda")
    t2._dhaug_bits = torch.randint(-2**31, 2**31 - 1, (nb,), dtype=torch.int32, device="cuda")
    y1 = torch.empty(M, 256, dtype=torch.bfloat16, device="cuda"); y2 = torch.empty_like(y1)

    def two():
        a = ops.gemm_nt_dmask(X, W1, 256, 256, t1, 1, 0.0, out=y1)
        ops.gemm_nt_dmask(a, W2, 256, 256, t2, 1, 0.0, res_bf16=X, out=y2)

    def one():
        ops.gemm_block2(X, W1, W2, t1, t2, 1, 0.0, out1=y1, out2=y2)

    for name, fn, mb in (("two launches", two, 5 * M * 512 / 1e6), ("block2", one, 3 * M * 512 / 1e6)):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(30): fn()
            e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 30 * 1e3)
        print("M=%d %-13s %.1f us  (%.0f MB -> %.2f TB/s; %.0f TFLOP/s)" % (M, name, best, mb, mb / best, 4.0 * M * 65536 / best / 1e6), flush=True)
