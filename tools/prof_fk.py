"""Profiling target: the FK kernel alone at N = 4 Mi poses (the launch bench.py's roofline_fk times) and the generator tail
at B = 65 536."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops

n = 1 << 22
a = (torch.rand(n, 37, device="cuda") * 2 - 1) * 180
b = torch.rand(n, 15, device="cuda") * 0.4 + 0.1
r = torch.randn(n, 3, device="cuda")
for _ in range(30):
    ops.fk_forward(a, b, r)
torch.cuda.synchronize()
B = 65536
head = torch.randn(B, 35, device="cuda")
cam = ([0.5, 0.5, -0.5, 0.5], [0.0, 0.0, 5.0], [2.3, 2.3, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
for _ in range(200):
    ops.gen_tail_forward_critics(head, b[:B], None, True, cam, rng=(1, 0), inputs_bf16=True)
torch.cuda.synchronize()
