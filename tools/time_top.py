"""Development aid: the fused top-of-the-critic backward launch (dhaug_critic_top_backward_bf16) against the four launches it replaces,
and the tangent launch, at M = 3B = 196 608 rows (env M); twenty calls replayed as one hipGraph.  STAMPS=1 with a -DDHAUG_TOP_TIMING
build (DHAUG_LIB, DHAUG_ABLATION_BUILD=1): the phases of one tile of the backward launch in clocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops
M, n0 = int(os.environ.get("M", 196608)), 100
bf = lambda t: t.to(torch.bfloat16)
act = lambda: bf(torch.cat([torch.relu(torch.randn(M, n0, device="cuda")), torch.zeros(M, 12, device="cuda")], 1))
m1, mh, m0 = act(), act(), act()
seed = bf(torch.cat([torch.randn(M, 1, device="cuda") * 0.01, torch.zeros(M, 15, device="cuda")], 1))
mk = lambda rows, cols, pad: bf(torch.cat([torch.randn(rows, cols, device="cuda") / cols ** 0.5, torch.zeros(rows, pad - cols, device="cuda")], 1))
W2, W1, Wm, wout = mk(n0, n0, 112), mk(n0, n0, 112), mk(512, n0, 112), mk(n0, 1, 16)
nb = (M + 127) // 128 * 4 * 256
bits = [torch.randint(-2**31, 2**31 - 1, (nb,), dtype=torch.int32, device="cuda") for _ in range(2)]
cat = torch.zeros(M, 512, dtype=torch.bfloat16, device="cuda"); cat._dhaug_bits_cols = bits
gc = torch.empty(M, 512, dtype=torch.bfloat16, device="cuda")


def four():
    r2 = ops.rank1_mask(seed, wout[:, 0], m1, n0, 1, 0.0)
    r1 = ops.gemm_nt_dmask(r2, W2, n0, 112, mh, 1, 0.0)
    r0 = ops.gemm_nt_dmask(r1, W1, n0, 112, m0, 1, 0.0, res_bf16=r2)
    ops.gemm_nt_dmask(r0, Wm, 512, 112, cat, 1, 0.0, out=gc)


one = lambda: ops.critic_top_backward(seed, wout[:, 0], m1, mh, m0, W2, W1, Wm, bits, n0, 1, 0.0, gcat=gc)
# the tangent sweep (M rows here; the step runs it over the B interpolated rows), in place over copies of the activations
ucat = bf(torch.randn(M, 512, device="cuda"))
WmT, t0, th, t1 = mk(n0, 512, 512), m0.clone(), mh.clone(), m1.clone()
tan = lambda: ops.critic_top_tangent(ucat, t0, th, t1, WmT, W1, W2, n0, 1, 0.0)
for name, fn in (("four launches", four), ("one launch", one), ("tangent launch", tan)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(20): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 20 * 1e3)
    print("%-14s %.1f us" % (name, best), flush=True)
if os.environ.get("STAMPS"):
    import ctypes
    from dhaug_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    buf = (ctypes.c_longlong * 64)()
    one(); torch.cuda.synchronize()
    L.dhaug_debug_top_stamps(buf, 64)
    st = list(buf)
    names = ["-", "logit layer", "barrier", "store g2", "fc2", "barrier + store g1", "fc1", "barrier + store g0 + masks -> LDS + requests", "merge", "barrier", "store gcat", "-"]
    for i, n in enumerate(names):
        print("  %-20s %6d clocks" % (n, st[i + 1] - st[i]))
    print("  tile total %d" % (st[12] - st[0]))
    print("  inside phase 7: barrier %d, store g0 %d, masks -> LDS %d, requests %d" % (st[13] - st[7], st[14] - st[13], st[15] - st[14], st[8] - st[15]))
