"""development: what the fused INFERENCE programs (fused_mlp_kernel<false>, fused_mlp_x3_kernel) do with non-finite inputs and
weights -- the facts tests/test_gpu_edge.py::test_nonfinite_in_the_fused_inference_programs pins.  DHAUG_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dhaug_amd
from dhaug_amd import ops, fused, autograd_ops as A
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model

B, D = 512, 256
args = synth_args(B, D)
torch.manual_seed(3)
d = T.my_get_poseFk_model(args, None, Forward_Kinematics_DH_Model(args, ["S1"], None))
G, D3, D2 = d["model_G"], d["model_d3d"], d["model_d2d"]
z = torch.randn(B, 128, device="cuda")
x3 = torch.randn(B, 48, device="cuda") * .3
x2 = torch.rand(B, 32, device="cuda") - .5


def run(mode, x3, x2, z):
    kf, kb = ops.kcs_forward(x3, True, f32=True, bf16_ld=32)
    with torch.no_grad():
        l3 = fused.critic3d(D3, x3 if mode == "f16x3" else x3.bfloat16(), kcs=kf if mode == "f16x3" else kb, mode=mode).float().reshape(-1)
        l2 = fused.critic2d(D2, x2 if mode == "f16x3" else x2.bfloat16(), mode=mode).float().reshape(-1)
        h = fused.generator_head(G, z, mode).float()
    return l3.clone(), l2.clone(), h.clone()


for mode in ("bf16", "f16x3"):
    c3, c2, ch = run(mode, x3, x2, z)
    for name, val in (("nan", float("nan")), ("+inf", float("inf")), ("-inf", float("-inf")), ("1e30", 1e30)):
        a3, a2, az = x3.clone(), x2.clone(), z.clone()
        a3[7, 20] = val; a2[11, 3] = val; az[5, 9] = val
        l3, l2, h = run(mode, a3, a2, az)
        k3 = torch.arange(B, device="cuda") != 7; k2 = torch.arange(B, device="cuda") != 11; kz = torch.arange(B, device="cuda") != 5
        print("%-5s input %-5s: D3 row -> %s (others equal: %s) | D2 row -> %s (others equal: %s) | G head row nan-count %d / %d, inf %d (others equal: %s)" % (
            mode, name, l3[7].item(), torch.equal(l3[k3], c3[k3]), l2[11].item(), torch.equal(l2[k2], c2[k2]),
            torch.isnan(h[5]).sum().item(), h.shape[1], torch.isinf(h[5]).sum().item(), torch.equal(h[kz], ch[kz])))
    # one NaN weight in a first layer
    with torch.no_grad():
        w3, w2, wg = D3.previous[0].weight[5, 9].item(), D2.pose_layer_1.weight[5, 9].item(), G.preprocess[0].weight[5, 9].item()
        D3.previous[0].weight[5, 9] = float("nan"); D2.pose_layer_1.weight[5, 9] = float("nan"); G.preprocess[0].weight[5, 9] = float("nan")
    A.bump_weight_epoch()
    l3, l2, h = run(mode, x3, x2, z)
    print("%-5s NaN weight  : D3 nan logits %d / %d | D2 %d / %d | G head nan %d / %d" % (
        mode, torch.isnan(l3).sum().item(), B, torch.isnan(l2).sum().item(), B, torch.isnan(h).sum().item(), h.numel()))
    with torch.no_grad():
        D3.previous[0].weight[5, 9] = w3; D2.pose_layer_1.weight[5, 9] = w2; G.preprocess[0].weight[5, 9] = wg
    A.bump_weight_epoch()
