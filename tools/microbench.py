"""Kernel micro-benchmarks (HIP-event timed).  python tools/microbench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops

def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

dev = "cuda"
for N in (65536, 1 << 20, 1 << 22):
    a = (torch.rand(N, 37, device=dev) * 2 - 1) * 180; bl = torch.rand(N, 15, device=dev) * .4 + .1; rt = torch.randn(N, 3, device=dev)
    g = torch.randn(N, 48, device=dev); head = torch.randn(N, 35, device=dev); sc = torch.randint(-200, 200, (N, 8), device=dev) / 1000.
    t = timeit(lambda: ops.fk_forward(a, bl, rt)); print("fk_forward   N=%8d %8.1f us  %7.1f Mposes/s  %6.1f GB/s (412 B/pose)" % (N, t * 1e6, N / t / 1e6, 412 * N / t / 1e9))
    t = timeit(lambda: ops.gen_tail_forward(head, bl, sc)); print("gen_tail_fwd N=%8d %8.1f us  %7.1f Mposes/s  %6.1f GB/s (424 B/pose)" % (N, t * 1e6, N / t / 1e6, 424 * N / t / 1e9))
    cam = ([0.7, 0.1, -0.1, 0.7], [0.1, 0.2, 5.0], [1.1, 1.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    t = timeit(lambda: ops.gen_tail_forward_critics(head, bl, None, True, cam, (1234, 0))); print("tail+critics N=%8d %8.1f us  %7.1f Mposes/s" % (N, t * 1e6, N / t / 1e6))
    t = timeit(lambda: ops.fk_backward(a, bl, g)); print("fk_backward  N=%8d %8.1f us  %7.1f Mposes/s" % (N, t * 1e6, N / t / 1e6))
    t = timeit(lambda: ops.kcs_forward(g, True, False, 32)); print("kcs_fwd bf16 N=%8d %8.1f us  %6.1f GB/s" % (N, t * 1e6, (192 + 64) * N / t / 1e9))
for (M, N, K) in ((65536, 256, 256), (65536, 256, 128), (65536, 256, 48), (65536, 100, 512), (65536, 35, 256), (65536, 1, 112), (65536, 256, 768)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = (torch.randn(N, K, device=dev) / 16).to(torch.bfloat16)
    bias = torch.randn(N, device=dev); res = torch.randn(M, (N + 7) // 8 * 8, device=dev).to(torch.bfloat16)
    t = timeit(lambda: ops.gemm_nt(A, B, N, K, bias=bias, res_bf16=res, act=1, out_bf16=True))
    by = M * K * 2 + M * N * 2 * 2 + N * K * 2
    print("gemm_nt M=%d N=%4d K=%4d %8.1f us  %7.1f TFLOP/s  %6.1f GB/s" % (M, N, K, t * 1e6, 2 * M * N * K / t / 1e12, by / t / 1e9))
    if N == 256 and K <= 256:
        t = timeit(lambda: ops.gemm_nt(A, B, N, K, out_bf16=True))
        print("gemm_nt M=%d N=%4d K=%4d %8.1f us  (no bias / residual / activation: %6.1f GB/s)" % (M, N, K, t * 1e6, (M * K * 2 + M * N * 2) / t / 1e9))
for (M, N1, N2) in ((65536, 256, 256), (65536, 256, 48), (65536, 100, 512)):
    A = torch.randn(M, (N1 + 7) // 8 * 8, device=dev).to(torch.bfloat16); B = torch.randn(M, (N2 + 7) // 8 * 8, device=dev).to(torch.bfloat16)
    t = timeit(lambda: ops.gemm_tn(A, B, N1, N2))
    print("gemm_tn M=%d N1=%4d N2=%4d %8.1f us  %7.1f TFLOP/s  %6.1f GB/s" % (M, N1, N2, t * 1e6, 2 * M * N1 * N2 / t / 1e12, (M * (N1 + N2) * 2) / t / 1e9))
