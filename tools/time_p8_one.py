"""Development aid: one library variant (DHAUG_LIB) of the ping-pong NT kernel at the frame critics' row count, as a replayed hipGraph;
with a stamps build (P8_TIMING) also prints workgroup 0's phase stamps (waves 0 and 4)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops, _lib
from tools.time_p8 import graph_time  # noqa

N = K = 1000
Kp = 1008
name = os.path.basename(os.environ.get("DHAUG_LIB", "product"))
res = []
for M in (13824, 65536):
    x = (torch.randn(M, Kp, device="cuda") * 0.1).bfloat16(); x[:, K:] = 0
    w = (torch.randn(N, Kp, device="cuda") * 0.03).bfloat16(); w[:, K:] = 0
    b = torch.zeros(N, device="cuda")
    out = torch.empty(M, Kp, device="cuda", dtype=torch.bfloat16)
    t = graph_time(lambda: ops.gemm_nt(x, w, N, Kp, bias=b, act=1, n_pad=Kp, c_bf16=out))
    res.append("M=%d %.1f us (%.0f TF/s)" % (M, t, 2.0 * M * N * K / t / 1e6))
print("%-16s %s" % (name, " | ".join(res)), flush=True)
lib = _lib.lib()
if hasattr(lib, "dhaug_debug_p8_stamps"):
    M = 13824
    ops.gemm_nt(x[:M], w, N, Kp, bias=b, act=1, n_pad=Kp, c_bf16=out[:M]); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 320)()
    lib.dhaug_debug_p8_stamps(buf, 320)
    for g in (0, 1):
        s = list(buf[160 * g:160 * g + 160])
        t0 = s[0]
        print("wave %d: prologue %d, stagger %d" % (4 * g, s[1] - t0, 0))
        ph = [(s[2 + 2 * j] - (s[2 + 2 * j - 1] if j else s[1]), s[3 + 2 * j] - s[2 + 2 * j]) for j in range(64)]
        print("  load segments (reads + copy issue + wait):", [p[0] for p in ph])
        print("  barrier + matrix segments               :", [p[1] for p in ph])
        print("  loop total %d, epilogue %d" % (s[150] - s[1], s[151] - s[150]))
