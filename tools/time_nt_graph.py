"""Development aid: GPU time of the short DenseDim-1000 layer GEMMs (the host issues a C-ABI call in ~10 us: a timing loop of 13 us
kernels measures the host).  Fifty calls are captured into one hipGraph and the replay is timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops

N = K = 1000
Kp = 1008
for M in (512, 1536, 4608):
    x = (torch.randn(M, Kp, device="cuda") * 0.1).bfloat16(); x[:, K:] = 0
    w = (torch.randn(N, Kp, device="cuda") * 0.03).bfloat16(); w[:, K:] = 0
    b = torch.zeros(N, device="cuda")
    y = (torch.randn(M, Kp, device="cuda") * 0.1).bfloat16()
    out = torch.empty(M, Kp, device="cuda", dtype=torch.bfloat16)
    xs = [x.clone() for _ in range(4)]; ws = [w.clone() for _ in range(4)]; os_ = [out.clone() for _ in range(4)]
    grp = lambda n: (lambda: ops.gemm_nt_group([dict(A=xs[i], B=ws[i], N=N, K=Kp, bias=b, act=1, out=os_[i], n_pad=Kp) for i in range(n)]))
    fns = dict(fwd=lambda: ops.gemm_nt(x, w, N, Kp, bias=b, act=1, out_bf16=True, n_pad=Kp, c_bf16=out), group2=grp(2), group4=grp(4),
               res=lambda: ops.gemm_nt(x, w, N, Kp, bias=b, res_bf16=y, act=1, out_bf16=True, n_pad=Kp, c_bf16=out),
               dmask=lambda: ops.gemm_nt_dmask(x, w, N, Kp, y, 1, 0.0, out=out))
    line = []
    for name, fn in fns.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        st = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(st):
            with torch.cuda.graph(g, stream=st):
                for _ in range(50): fn()
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); g.replay(); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 50 * 1e3)
        line.append("%s %.1f us (%.0f TF/s)" % (name, best, 2.0 * M * N * K / best / 1e6))
    print("M=%5d: " % M + " | ".join(line), flush=True)
