import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import fused
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model
B, D = 65536, 256
args = synth_args(B, D)
d = T.my_get_poseFk_model(args, None, Forward_Kinematics_DH_Model(args, ["S1"], None))
G = d["model_G"]
z = torch.randn(B, 128, device="cuda")
with torch.no_grad():
    for _ in range(5):
        fused.generator_head(G, z)
torch.cuda.synchronize()
