"""fused one-launch forward vs layer-by-layer.  python tools/microbench_fused.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops, fused
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

B, D = 65536, 256
args = synth_args(B, D)
d = T.my_get_poseFk_model(args, None, Forward_Kinematics_DH_Model(args, ["S1"], None))
G, D3, D2 = d["model_G"], d["model_d3d"], d["model_d2d"]
z = torch.randn(B, 128, device="cuda"); x3 = torch.randn(B, 48, device="cuda") * .3; x2 = torch.rand(B, 32, device="cuda") - .5
mac = dict(G=128*D+6*D*D+35*D, D3=78*D+12*D*D+200*D+2*100*100+100, D2=32*D+4*D*D+D)
with torch.no_grad():
    for name, f_fused, f_layer in (("G", lambda: fused.generator_head(G, z), lambda: G.trunk(z)),
                                   ("D3", lambda: fused.critic3d(D3, x3), lambda: D3(x3)),
                                   ("D2", lambda: fused.critic2d(D2, x2), lambda: D2(x2))):
        tf, tl = timeit(f_fused), timeit(f_layer)
        fl = 2.0 * mac[name] * B
        print("%-3s fused %8.1f us %7.1f TFLOP/s (%.1f%% of 2.5 PF) | layerwise %8.1f us %6.1f TFLOP/s | x%.1f" %
              (name, tf * 1e6, fl / tf / 1e12, fl / tf / 2.5e13, tl * 1e6, fl / tl / 1e12, tl / tf))
