"""Development aid: shader-clock stamps of workgroup 0 of gemm_block2_kernel (lib built with -DDHAUG_PIPE_TIMING; thread 0 = stage A,
thread 256 = stage B): launch prologue, weights, every iteration's phases, the drain."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dhaug_amd
from dhaug_amd import ops, _lib
tiles = int(os.environ.get("TILES", 1))
M = 32 * 256 * tiles
bf = lambda t: t.to(torch.bfloat16)
X = bf(torch.randn(M, 256, device="cuda"))
W1 = bf(torch.randn(256, 256, device="cuda") / 16); W2 = bf(torch.randn(256, 256, device="cuda") / 16)
nb = (M + 127) // 128 * 4 * 256
t1 = torch.zeros(M, 256, dtype=torch.bfloat16, device="cuda"); t2 = torch.zeros_like(t1)
t1._dhaug_bits = torch.randint(-2**31, 2**31 - 1, (nb,), dtype=torch.int32, device="cuda")
t2._dhaug_bits = torch.randint(-2**31, 2**31 - 1, (nb,), dtype=torch.int32, device="cuda")
y1 = torch.empty(M, 256, dtype=torch.bfloat16, device="cuda"); y2 = torch.empty_like(y1)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_longlong * 256)()
for _ in range(5):
    ops.gemm_block2(X, W1, W2, t1, t2, 1, 0.0, out1=y1, out2=y2)
torch.cuda.synchronize()
L.dhaug_debug_pipe_stamps(buf, 256)
st = list(buf)
for name, o in (("stage A", 0), ("stage B", 128)):
    s = st[o:o + 128]
    print(name, "entry->weights issued %d | wait for them %d | barrier %d | loop %d | drain %d | last barrier %d   (clocks of s_memtime: 100 MHz? see below)" %
          (s[1] - s[0], s[2] - s[1], s[3] - s[2], s[4] - s[3], s[5] - s[4], s[6] - s[5]))
    for i in range(tiles + 2):
        b = s[8 + 8 * i:15 + 8 * i]
        print("   it %d: copies %d | B stream %d | compute %d | A stream %d | vmcnt wait %d | barrier %d" % (i, b[1] - b[0], b[2] - b[1], b[3] - b[2], b[4] - b[3], b[5] - b[4], b[6] - b[5]))
print("whole kernel (stage A thread):", st[6] - st[0])
