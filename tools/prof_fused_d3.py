"""Profiling target: the 3D critic's fused launch alone (the three networks share one kernel name), B = 65 536, D = 256.
    python tools/prof_fused_d3.py [bf16|f16x3]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import fused, ops
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import Fk_discriminator

mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, D = 65536, 256
args = synth_args(B, D)
D3 = Fk_discriminator.Fk_3D_Discriminator("cuda", args).cuda()
x3 = torch.randn(B, 48, device="cuda") * 0.3
kcs = ops.kcs_forward(x3, True, f32=True)[0] if mode == "f16x3" else ops.kcs_forward(x3, True, f32=False, bf16_ld=32)[1]
# clocks up first (a cold device needs a few hundred ms of load: the same pre-warm bench.py applies), with a kernel of another
# name so that the profile's average of the fused kernel is a steady-state average
import time
w = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    for _ in range(20):
        w @ w
    torch.cuda.synchronize()
with torch.no_grad():
    for _ in range(10):                                    # (weight packing, first-touch: outside the steady state too)
        fused.critic3d(D3, x3, kcs=kcs, mode=mode)
    torch.cuda.synchronize()
    for _ in range(400 if mode == "bf16" else 150):
        fused.critic3d(D3, x3, kcs=kcs, mode=mode)
torch.cuda.synchronize()
