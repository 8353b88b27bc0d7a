"""Times the forward-with-save programs of the two critics at 3B = 196 608 rows (the explicit critic step's sweep 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops, fused
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model

M, D = 3 * 65536, 256
args = synth_args(65536, D)
d = T.my_get_poseFk_model(args, None, Forward_Kinematics_DH_Model(args, ["S1"], None))
D3, D2 = d["model_d3d"], d["model_d2d"]
x3 = torch.randn(M, 48, device="cuda") * .3
x2 = torch.rand(M, 32, device="cuda") - .5
kf, kb = ops.kcs_forward(x3, True, f32=True, bf16_ld=32)
B2 = 2 * 65536
fns = dict(D3save=lambda: fused.critic3d_forward_save(D3, x3, kb), D2save=lambda: fused.critic2d_forward_save(D2, x2),
           D3save2B=lambda: fused.critic3d_forward_save(D3, x3, kb, save_rows=B2), D2save2B=lambda: fused.critic2d_forward_save(D2, x2, save_rows=B2),
           D3infer=lambda: fused.critic3d(D3, x3, kcs=kb), D2infer=lambda: fused.critic2d(D2, x2))
with torch.no_grad():
    for _ in range(100):
        fns["D3infer"]()
    for name, fn in fns.items():
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(30):
                fn()
            e.record()
            torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / 30 * 1e3)
        print("%-8s %8.1f us" % (name, best))
