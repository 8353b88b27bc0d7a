"""A/B of the parity programs: KCS features handed in / computed by a launch in front / computed inside the program (DHAUG_MLP_LOAD_KCS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import fused, ops
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model
B, D = 65536, 256
args = synth_args(B, D)
fk = Forward_Kinematics_DH_Model(args, ["S1"], None)
m = T.my_get_poseFk_model(args, None, fk)
D3, D2 = m["model_d3d"], m["model_d2d"]
x3 = torch.randn(B, 48, device="cuda") * 0.3
x2 = torch.randn(B, 32, device="cuda") * 0.5
kf = ops.kcs_forward(x3, True, f32=True)[0]
def t(fn, n=50):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best
with torch.no_grad():
    a = t(lambda: fused.critics(D3, D2, x3, kf, x2, "f16x3"))
    b = t(lambda: (ops.kcs_forward(x3, True, f32=True), fused.critics(D3, D2, x3, kf, x2, "f16x3")))
    c = t(lambda: fused.critics(D3, D2, x3, None, x2, "f16x3"))
print("critics with features given %.1f us | KCS launch + critics %.1f us | critics computing KCS %.1f us" % (a, b, c))
