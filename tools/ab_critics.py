"""Development aid: both critics as two launches vs one merged launch (same process, alternating)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dhaug_amd
from dhaug_amd import ops
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import Fk_discriminator as FD
B = 65536
args = synth_args(B, 256)
D3 = FD.Fk_3D_Discriminator("cuda", args).cuda(); D2 = FD.Fk_2D_Discriminator(args, 16).cuda()
x3 = torch.randn(B, 16, 3, device="cuda") * 0.3; x3 = x3 - x3[:, :1]
p2 = torch.randn(B, 16, 2, device="cuda") * 0.3
_, kcs = ops.kcs_forward(x3.reshape(B, 48), True, f32=False, bf16_ld=32)
def timeit(fn, iters=50, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
with torch.no_grad():
    for r in range(3):
        a = timeit(lambda: (D3(x3, kcs=kcs), D2(p2)))
        b = timeit(lambda: FD.score_fake_pair(D3, D2, x3, kcs, p2))
        print("two launches %.1f us   one launch %.1f us" % (a, b))
