"""Development aid: time the fused programs (G trunk, D3, D2, both critics) in both arithmetics at B = 65 536."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import fused, ops
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model

B, D = int(os.environ.get("B", 65536)), 256
args = synth_args(B, D)
fk = Forward_Kinematics_DH_Model(args, ["S1"], None)
m = T.my_get_poseFk_model(args, None, fk)
G, D3, D2 = m["model_G"], m["model_d3d"], m["model_d2d"]
z = torch.randn(B, 128, device="cuda")
x3 = torch.randn(B, 48, device="cuda") * 0.3
x2 = torch.randn(B, 32, device="cuda") * 0.5
kf, kb = ops.kcs_forward(x3, True, f32=True)[0], ops.kcs_forward(x3, True, f32=False, bf16_ld=32)[1]


def t(fn, n=50, warm=10):
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


mac = dict(G=128 * D + 6 * D * D + 35 * D, D3=78 * D + 12 * D * D + 200 * D + 2 * 100 * 100 + 100, D2=32 * D + 4 * D * D + D)
with torch.no_grad():
    for mode, k in (("bf16", kb), ("f16x3", kf)):
        r = dict(G=t(lambda: fused.generator_head(G, z, mode)), D3=t(lambda: fused.critic3d(D3, x3, kcs=k, mode=mode)),
                 D2=t(lambda: fused.critic2d(D2, x2, mode)), both=t(lambda: fused.critics(D3, D2, x3, k, x2, mode)))
        print(mode, " ".join("%s %.1f us (%.0f TF alg)" % (n, v, 2 * (mac[n] if n in mac else mac["D3"] + mac["D2"]) * B / v / 1e6)
                             for n, v in r.items()), flush=True)
