"""Times sweep 4 of the 3D motion critic at the video configuration (B = 512 clips: 1 536 rows, DenseDim 1000: every layer 4 x 4
blocks of 256 x 256, TN_GROUP_MAX blocks per launch) with the layers handed over whole (one workgroup per block, adding into the
gradient itself) and block by block (DHAUG_TN_WIDE_MIN_BLOCKS is read once per process: this script re-runs itself per value)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--one" not in sys.argv:
    for f in (sys.argv[1:] or ["128", "100000"]):
        env = dict(os.environ, DHAUG_TN_WIDE_MIN_BLOCKS=f)
        print("wide layers from", f, "blocks:", subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
    sys.exit(0)
import torch
import dhaug_amd
from dhaug_amd import ops
M, D, L = int(os.environ.get("M", 1536)), 1000, int(os.environ.get("LAYERS", 24))
c16 = lambda n: (n + 15) // 16 * 16
items, total = [], 0
for l in range(L):
    g = (torch.randn(M, c16(D), device="cuda") * 0.1).bfloat16()
    x = (torch.randn(M, c16(D), device="cuda") * 0.1).bfloat16()
    out = torch.zeros(D, D, device="cuda")
    cs = torch.zeros(D, device="cuda")
    items.append((g, x, D, D, out, cs, M // 3 * 2, True, M, None, None))
    total += M * 2 * c16(D) * 2
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
print("%.1f us for %d layers of %d x %d at %d rows (%.2f GB of distinct operands; DHAUG_TN_WIDE_MIN_BLOCKS = %d)" % (
    best, L, D, D, M, total / 1e9, ops.TN_WIDE_MIN_BLOCKS))
