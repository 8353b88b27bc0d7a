"""Development aid: the split-operand layer product of the parity-grade training step (3B = 196 608 rows, 256 -> 256, fp32 residual and mask)
with the activation side as the mode 0 operand (six segments, 12 bytes per value) against the three planes (dhaug_gemm_bf16x6_planes): same
bits, and the time of split + GEMM each way."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops
M, N, K = int(os.environ.get("M", 196608)), 256, 256
g = torch.Generator().manual_seed(3)
x = (torch.randn(M, K, generator=g)).cuda()
W = (torch.randn(N, K, generator=g) / 16).cuda()
res = torch.randn(M, N, generator=g).cuda()
msk = torch.randn(M, N, generator=g).cuda()
bias = torch.randn(N, generator=g).cuda()
B6 = ops.split_bf16(W, 1, 6, K)
def old():
    A6 = ops.split_bf16(x, 0, 6, K)
    return ops.gemm_nt_dmask_f32(A6, B6, N, 6 * K, msk, 1, 0.0, res_f32=res)
def new():
    A3 = ops.split_bf16(x, 2, 6, K)
    return ops.gemm_nt_planes(A3, B6, N, K, res_f32=res, dmask_f32=msk, dmask_act=1)
a, b = old(), new()
print("bit-identical:", torch.equal(a, b), (a - b).abs().max().item())
A6, A3 = ops.split_bf16(x, 0, 6, K), ops.split_bf16(x, 2, 6, K)
cases = {"split mode 0": lambda: ops.split_bf16(x, 0, 6, K), "split planes": lambda: ops.split_bf16(x, 2, 6, K),
         "gemm mode 0": lambda: ops.gemm_nt_dmask_f32(A6, B6, N, 6 * K, msk, 1, 0.0, res_f32=res),
         "gemm planes": lambda: ops.gemm_nt_planes(A3, B6, N, K, res_f32=res, dmask_f32=msk, dmask_act=1),
         "both mode 0": old, "both planes": new}
for name, fn in cases.items():
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 100.0)
    print("%-14s %7.1f us" % (name, best))
