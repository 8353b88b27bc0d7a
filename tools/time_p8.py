"""Development aid: the ping-pong NT kernel (csrc/dhaug_gemm_p8.hip) against the kernels it replaces, as replayed hipGraphs (a Python
loop of 20 us kernels measures the host): single launches at the frame critics' / the forward workload's row counts, grouped launches at
the motion critics'."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops

N = K = 1000
Kp = 1008


def graph_time(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / reps * 1e3)
    return best


def env(name, val, fn):
    os.environ[name] = val
    try:
        return fn()
    finally:
        del os.environ[name]


def main():
    warm = torch.randn(4096, 4096, device="cuda")
    for _ in range(50): warm @ warm
    for M in (4608, 13824, 65536):
        x = (torch.randn(M, Kp, device="cuda") * 0.1).bfloat16(); x[:, K:] = 0
        w = (torch.randn(N, Kp, device="cuda") * 0.03).bfloat16(); w[:, K:] = 0
        b = torch.zeros(N, device="cuda")
        y = (torch.randn(M, Kp, device="cuda") * 0.1).bfloat16()
        out = torch.empty(M, Kp, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * M * N * K
        fns = dict(fwd=lambda: ops.gemm_nt(x, w, N, Kp, bias=b, act=1, n_pad=Kp, c_bf16=out),
                   res=lambda: ops.gemm_nt(x, w, N, Kp, bias=b, res_bf16=y, act=1, n_pad=Kp, c_bf16=out),
                   dmask=lambda: ops.gemm_nt_dmask(x, w, N, Kp, y, 1, 0.0, out=out))
        line = []
        for name, fn in fns.items():
            new = graph_time(fn)
            old = env("DHAUG_GEMM_NOP8", "1", lambda: graph_time(fn))
            line.append("%s %.1f us (%.0f TF/s = %.3f; before %.1f)" % (name, new, fl / new / 1e6, fl / new / 1e6 / 2500, old))
        print("M=%5d: " % M + " | ".join(line), flush=True)
    for M in (512, 1024, 1536, 4608, 13824):
        for n in (2, 4):
            xs = [(torch.randn(M, Kp, device="cuda") * 0.1).bfloat16() for _ in range(n)]
            ws = [(torch.randn(N, Kp, device="cuda") * 0.03).bfloat16() for _ in range(n)]
            for t in xs + ws: t[:, K:] = 0
            b = torch.zeros(N, device="cuda")
            outs = [torch.empty(M, Kp, device="cuda", dtype=torch.bfloat16) for _ in range(n)]
            fn = lambda: ops.gemm_nt_group([dict(A=xs[i], B=ws[i], N=N, K=Kp, bias=b, act=1, out=outs[i], n_pad=Kp) for i in range(n)])
            new = env("DHAUG_NT_GROUP_P8_ROWS", "1", lambda: graph_time(fn))
            old = env("DHAUG_NT_GROUP_P8_ROWS", "0", lambda: graph_time(fn))
            fl = 2.0 * M * N * K * n
            print("group of %d x %5d rows: ping-pong %.1f us (%.0f TF/s) | 128 x 128 tiles %.1f us" % (n, M, new, fl / new / 1e6, old), flush=True)



if __name__ == "__main__":
    main()
