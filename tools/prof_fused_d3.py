import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import fused
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import Fk_discriminator
B, D = 65536, 256
args = synth_args(B, D)
D3 = Fk_discriminator.Fk_3D_Discriminator("cuda", args).cuda()
x3 = torch.randn(B, 16, 3, device="cuda") * 0.3
with torch.no_grad():
    for _ in range(400):
        fused.critic3d(D3, x3)
torch.cuda.synchronize()
