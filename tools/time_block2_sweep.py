"""Development aid: the two-layer block launch over batch sizes -- its fixed cost (launch, 256 KB of weight registers per workgroup,
pipeline fill and drain) against its cost per 32-row tile; and a three-block stack in one launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops
bf = lambda t: t.to(torch.bfloat16)
for tiles in (1, 2, 4, 8, 16, 24):
    M = 32 * 256 * tiles
    X = bf(torch.randn(M, 256, device="cuda"))
    Ws = [bf(torch.randn(256, 256, device="cuda") / 16) for _ in range(6)]
    ts = []
    nb = (M + 127) // 128 * 4 * 256
    for _ in range(6):
        t = torch.zeros(M, 256, dtype=torch.bfloat16, device="cuda")
        t._dhaug_bits = torch.randint(-2**31, 2**31 - 1, (nb,), dtype=torch.int32, device="cuda")
        ts.append(t)
    y1 = torch.empty(M, 256, dtype=torch.bfloat16, device="cuda"); y2 = torch.empty_like(y1)
    one = lambda: ops.gemm_block2(X, Ws[0], Ws[1], ts[0], ts[1], 1, 0.0, out1=y1, out2=y2)
    stack = lambda: ops.gemm_block2_stack(X, [(Ws[2 * b], Ws[2 * b + 1], ts[2 * b], ts[2 * b + 1], None, None) for b in range(3)], 1, 0.0)
    line = []
    for name, fn in (("one block", one), ("stack of 3", stack)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                for _ in range(20): fn()
        g.replay(); torch.cuda.synchronize()
        best = 1e9
        for rep in range(4):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); g.replay(); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 20 * 1e3)
        line.append("%s %.1f us" % (name, best))
    print("tiles per workgroup %2d (M = %6d): %s" % (tiles, M, " | ".join(line)), flush=True)
