"""Times the 3D critic's whole sweep 4 as ONE grouped launch (the step's real layer shapes at 3B = 196 608 rows) for several
values of the dealing floor (DHAUG_TN_FLOOR is read once per process: this script re-runs itself per value)."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--one" not in sys.argv:
    for f in (sys.argv[1:] or ["0", "224", "320", "448", "640"]):
        env = dict(os.environ, DHAUG_TN_FLOOR=f)
        print("floor", f, subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True).stdout.strip(), flush=True)
    sys.exit(0)
import torch
import dhaug_amd
from dhaug_amd import ops
M = int(os.environ.get("M", 196608))
c16 = lambda n: (n + 15) // 16 * 16
# (N, K) of the 3D critic's 19 contractions at D = 256: two branches (input layer + 6), the merge layer in two column blocks,
# the 100-wide block, the logit layer
shapes = [(256, 30)] + [(256, 256)] * 6 + [(256, 48)] + [(256, 256)] * 6 + [(100, 256), (100, 256), (100, 100), (100, 100), (1, 100)]
items, total = [], 0
for N1, N2 in shapes:
    g = (torch.randn(M, c16(N1), device="cuda") * 0.1).bfloat16()
    x = (torch.randn(M, c16(N2), device="cuda") * 0.1).bfloat16()
    out = torch.zeros(N1, N2, device="cuda")
    cs = torch.zeros(N1, device="cuda")
    items.append((g, x, N1, N2, out, cs, M // 3 * 2, True, None, None, None))
    total += M * (c16(N1) + c16(N2)) * 2
fn = lambda: ops.gemm_tn_group(items)
for _ in range(10): fn()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fn()
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 20 * 1e3)
print("%.1f us for %.2f GB of operands = %.2f TB/s (launch + reduce)" % (best, total / 1e9, total / best / 1e6))
