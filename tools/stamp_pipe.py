"""Development aid: shader-clock stamps of workgroup 0 / thread 0 of gemm_nt_pipe_kernel (lib built with -DDHAUG_PIPE_TIMING): per
k-stage wait | barrier | copy issue | compute."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dhaug_amd
from dhaug_amd import ops, _lib
M, N, K, Kp = int(os.environ.get("M", 1536)), 1000, 1000, 1008
x = (torch.randn(M, Kp, device="cuda") * 0.1).bfloat16(); w = (torch.randn(N, Kp, device="cuda") * 0.03).bfloat16()
b = torch.zeros(N, device="cuda")
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 256)()
for _ in range(5):
    ops.gemm_nt(x, w, N, Kp, bias=b, act=1, out_bf16=True, n_pad=Kp)
torch.cuda.synchronize()
L.dhaug_debug_pipe_stamps(buf, 256)
st = list(buf)
print("prologue -> loop start: -, loop %d clk, epilogue to C tile %d, store %d" % (st[1] - st[0], st[2] - st[1], st[3] - st[2]))
for kt in range(16):
    a, b_, c, d = st[4 + 4 * kt:8 + 4 * kt]
    nxt = st[4 + 4 * (kt + 1)] if kt < 15 else st[1]
    print("stage %2d: wait %5d | barrier %5d | copy issue %5d | compute %5d" % (kt, b_ - a, c - b_, d - c, nxt - d))
