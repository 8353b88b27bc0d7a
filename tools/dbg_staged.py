"""Development aid: fused critics against the oracle's bf16 emulation at a few batch sizes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dhaug_amd
from dhaug_amd import fused
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import Fk_discriminator
from oracle import dhaug_oracle as O
import golden_util as GU
for B in (128, 256, 1000):
    args = synth_args(B, 256)
    torch.manual_seed(5)
    D3 = Fk_discriminator.Fk_3D_Discriminator("cuda", args).cuda()
    D2 = Fk_discriminator.Fk_2D_Discriminator(args, 16).cuda()
    x3 = GU.synth_pose16(B, seed=8); x3 = (x3 - x3[:, :1]).cuda()
    g = torch.Generator().manual_seed(6)
    x2 = ((torch.rand(B, 16, 2, generator=g) - 0.5) * 1.6).cuda()
    with torch.no_grad():
        l3 = fused.critic3d(D3, x3).cpu(); l2 = fused.critic2d(D2, x2).cpu()
    r3 = O.d3_forward(x3.cpu(), {k: v.detach().cpu() for k, v in D3.state_dict().items()}, precision="bf16")
    r2 = O.d2_forward(x2.cpu().reshape(B, 32), {k: v.detach().cpu() for k, v in D2.state_dict().items()}, precision="bf16") if hasattr(O, "d2_forward") else None
    e3 = (l3 - r3).abs().reshape(-1)
    print(B, "d3 max err", e3.max().item(), "scale", r3.abs().max().item(), "bad rows", (e3 > 0.02 * r3.abs().max()).nonzero().reshape(-1)[:20].tolist(), "count", int((e3 > 0.02 * r3.abs().max()).sum()))
    if r2 is not None:
        e2 = (l2 - r2).abs().reshape(-1)
        print(B, "d2 max err", e2.max().item(), "scale", r2.abs().max().item(), "count", int((e2 > 0.02 * r2.abs().max()).sum()))
