"""Development aid: the video step's grouped weight-gradient launch of WIDE layers alone -- L layers (M, 1000, 1000), each 4 x 4 blocks of
256 x 256, one workgroup per block over all M rows, adding into the gradient slots (env M, L; with an ablation library: DHAUG_TN256_ABL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops
M, L, N = int(os.environ.get("M", 4608)), int(os.environ.get("L", 12)), int(os.environ.get("N", 1000))
ld = (N + 15) // 16 * 16
items = []
for _ in range(L):
    g = (torch.randn(M, ld, device="cuda") * 0.1).bfloat16()
    x = (torch.randn(M, ld, device="cuda") * 0.1).bfloat16()
    items.append((g, x, N, N, torch.zeros(N, N, device="cuda"), torch.zeros(N, device="cuda"), M, True, None, None, None))
fn = lambda: ops.gemm_tn_group(items)
for _ in range(5): fn()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 10 * 1e3)
flops = 2.0 * M * N * N * L
print("M %d, %d layers of %d x %d: %.1f us = %.2f of the MFMA peak, %.0f clocks per 32-row stage at 2.4 GHz (abl %s)" % (
    M, L, N, N, best, flops / (best * 1e-6) / 2.5e15, best * 2400.0 / (M / 32) / max(1, -(-16 * L // 256)), os.environ.get("DHAUG_TN256_ABL", "-")))
