"""Development aid: one optimizer step (FusedAdam.step: count + Adam + the weights' bf16 operand copies) of the 3D critic at D = 256
and of the 3D motion critic at D = 1000, as the two streaming launches of dhaug_adam_repack_step and as the four launches they replace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import optim


def net(widths):
    return torch.nn.Sequential(*[torch.nn.Linear(a, b) for a, b in widths]).cuda()


shapes = dict(D3_256=[(48, 256)] + [(256, 256)] * 6 + [(225, 256)] + [(256, 256)] * 6 + [(512, 100), (100, 100), (100, 100), (100, 1)],
              M3_1000=[(135, 1000), (120, 1000), (432, 1000), (384, 1000)] + [(1000, 1000)] * 24 + [(4000, 100), (100, 100), (100, 100), (100, 1)])
for name, w in shapes.items():
    for fused in (True, False):
        optim.FUSED_STEP = fused
        n = net(w)
        opt = optim.FusedAdam(n.parameters(), lr=1e-4, betas=(0.5, 0.9))
        opt.flat_grad.normal_()
        for _ in range(5):
            opt.step()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            opt.step()
        e.record()
        torch.cuda.synchronize()
        print("%-8s %s: %7.1f us per step (%.1f M parameters)" % (name, "two launches " if fused else "four launches", s.elapsed_time(e) / 50 * 1e3, opt.flat_param.numel() / 1e6))
