"""Debug: fused forward vs layerwise vs oracle bf16 emulation / fp32 for the 3D critic."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dhaug_amd
from dhaug_amd import fused
import golden_util as GU
from oracle import dhaug_oracle as O
import test_gpu_models as T
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
args = T.make_args(batch_size=B, Gen_DenseDim=256, Dis_DenseDim_3D=256, Dis_DenseDim_2D=256)
from dhaug_amd.models_Fk_GAN import Fk_discriminator, Fk_generator, forward_kinematics_DH_model as fkm
fk = fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
torch.manual_seed(5)
G = Fk_generator.Fk_Generator(fk, args, "cuda").cuda()
D3 = Fk_discriminator.Fk_3D_Discriminator("cuda", args).cuda()
x3 = GU.synth_pose16(B, seed=8); x3 = (x3 - x3[:, :1]).cuda()
with torch.no_grad():
    os.environ["DHAUG_MLP_NOSTACK"] = "1"
    l = fused.critic3d(D3, x3).reshape(-1).cpu()
    os.environ.pop("DHAUG_MLP_NOSTACK")
    f = fused.critic3d(D3, x3).reshape(-1).cpu()
sd3 = {k: v.detach().cpu() for k, v in D3.state_dict().items()}
rb = O.d3_forward(x3.cpu(), sd3, precision="bf16").reshape(-1)
rf = O.d3_forward(x3.cpu(), sd3).reshape(-1)
sc = rf.abs().max().item()
def e(a, b): return (a.double() - b.double()).abs().max().item() / sc
print("scale %.4g | fused-layer %.4g fused-bf16emu %.4g layer-bf16emu %.4g | fused-fp32 %.4g layer-fp32 %.4g emu-fp32 %.4g"
      % (sc, e(f, l), e(f, rb), e(l, rb), e(f, rf), e(l, rf), e(rb, rf)))
print("test relerr fused-emu %.4g  layer-emu %.4g" % (T.relerr(f, rb), T.relerr(l, rb)))
