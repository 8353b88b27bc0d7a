"""Development aid: the forward workload (FK + Gen + D3 + D2) at the reference's DEFAULT width, DenseDim 1000 (R/function_aug/config.py:
101-109), B = 65 536, bf16, layer by layer (the fused programs cover 64 / 128 / 256) -- per network and in all, eager and as one
hipGraph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model

B, D = int(os.environ.get("PB", "65536")), 1000
args = synth_args(B, D)
d = T.my_get_poseFk_model(args, None, Forward_Kinematics_DH_Model(args, ["S1"], None))
G, D3, D2 = d["model_G"], d["model_d3d"], d["model_d2d"]
if os.environ.get("PREC"):                                # PREC=f16x3: the compliant arithmetic (f16x3 layer GEMMs at this width)
    for net in (G, D3, D2):
        net.precision = os.environ["PREC"]
z = torch.randn(B, 128, device="cuda")
x3 = torch.randn(B, 16, 3, device="cuda") * .3
x2 = torch.rand(B, 16, 2, device="cuda") - .5
mac = lambda D: dict(G=128 * D + 6 * D * D + 35 * D, D3=78 * D + 12 * D * D + 200 * D + 2 * 100 * 100 + 100, D2=32 * D + 4 * D * D + D)
fns = dict(G=lambda: G.trunk(z), D3=lambda: D3(x3), D2=lambda: D2(x2))      # (G: the trunk; its FK tail is 14 us)


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n)
    return best


with torch.no_grad():
    tot = 0.0
    for name, fn in fns.items():
        ms = t(fn)
        fl = 2.0 * mac(D)[name] * B
        tot += ms
        print("%-3s %8.3f ms  %7.1f TFLOP/s (%.3f of 2.5 PF)" % (name, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500))
    fl = 2.0 * sum(mac(D).values()) * B
    print("sum %7.3f ms = %.1f M poses/s, %.1f TFLOP/s = %.3f of 2.5 PF (DenseDim %d, B = %d, bf16, layer by layer)" % (
        tot, B / tot / 1e3, fl / tot / 1e9, fl / tot / 1e9 / 2500, D, B))
