"""Times the three fused bf16 programs at B = 65 536, D = 256 (clocks up, 300 launches each, best of 3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops, fused
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model

B, D = 65536, 256
args = synth_args(B, D)
d = T.my_get_poseFk_model(args, None, Forward_Kinematics_DH_Model(args, ["S1"], None))
G, D3, D2 = d["model_G"], d["model_d3d"], d["model_d2d"]
z = torch.randn(B, 128, device="cuda")
x3 = (torch.randn(B, 48, device="cuda") * .3).bfloat16()
x2 = (torch.rand(B, 32, device="cuda") - .5).bfloat16()
kcs = ops.kcs_forward(x3.float(), True, f32=False, bf16_ld=32)[1]
mac = dict(G=128 * D + 6 * D * D + 35 * D, D3=78 * D + 12 * D * D + 200 * D + 2 * 100 * 100 + 100, D2=32 * D + 4 * D * D + D)
fns = dict(G=lambda: fused.generator_head(G, z), D3=lambda: fused.critic3d(D3, x3, kcs=kcs), D2=lambda: fused.critic2d(D2, x2))
with torch.no_grad():
    for _ in range(300):
        fns["D3"]()                                        # clocks up
    for name in ("G", "D3", "D2"):
        fn = fns[name]
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(300):
                fn()
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 300 * 1e-3)
        fl = 2.0 * mac[name] * B
        print("%-3s %8.1f us  %7.1f TFLOP/s  %.3f of 2.5 PF" % (name, best * 1e6, fl / best / 1e12, fl / best / 2.5e15))
