"""Development aid: the DenseDim-1000 layer GEMMs of the video workload at its row counts (motion critics 3 x 512 rows, frame
critics 3 x 4608, generator 512)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops

def t(fn, n=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best

N = K = 1000
Kp = 1008
for M in ((13824,) if os.environ.get('ONLYBIG') else (512, 1536, 4608, 13824)):
    x = (torch.randn(M, Kp, device="cuda") * 0.1).bfloat16(); x[:, K:] = 0
    w = (torch.randn(N, Kp, device="cuda") * 0.03).bfloat16(); w[:, K:] = 0
    b = torch.zeros(N, device="cuda")
    y = (torch.randn(M, Kp, device="cuda") * 0.1).bfloat16()
    fl = 2.0 * M * N * K
    a = t(lambda: ops.gemm_nt(x, w, N, Kp, bias=b, act=1, out_bf16=True, n_pad=Kp))
    r = t(lambda: ops.gemm_nt(x, w, N, Kp, bias=b, res_bf16=y, act=1, out_bf16=True, n_pad=Kp))
    m = t(lambda: ops.gemm_nt_dmask(x, w, N, Kp, y, 1, 0.0))
    g = torch.zeros(N, K, device="cuda")
    tn = t(lambda: ops.gemm_tn(x, y, N, K, out=g, accumulate=True), 20)
    print("M=%5d: fwd %.1f us (%.0f TF/s) | +res %.1f | dmask %.1f | TN %.1f us" % (M, a, fl / a / 1e6, r, m, tn), flush=True)
