"""Development aid: per-launch times of the forward step's kernels with the clocks up."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dhaug_amd
from dhaug_amd import ops, fused
from dhaug_amd.function_aug.config import synth_args
from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model
from dhaug_amd.models_Fk_GAN.Fk_discriminator import score_fake_pair
B = 65536
args = synth_args(B, 256)
fk = Forward_Kinematics_DH_Model(args, ["S1"], None)
m = T.my_get_poseFk_model(args, None, fk)
G, D3, D2 = m["model_G"], m["model_d3d"], m["model_d2d"]
x = torch.randn(B, 16, 3, device="cuda") * 0.3
G.GAN_generator_get_bone_length(x)
z = torch.randn(B, 128, device="cuda")
cam = ([0.7, 0.1, -0.1, 0.7], [0.1, 0.2, 5.0], [1.1, 1.1, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
with torch.no_grad():
    fw, xc, kcs, p2 = G.sample_for_critics(z, cam)
    head = G.trunk(z)
def ev(fn, iters=200):
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        for _ in range(20): fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
with torch.no_grad():
    print("G trunk            %.1f us" % ev(lambda: G.trunk(z)))
    print("tail + critic inputs %.1f us" % ev(lambda: ops.gen_tail_forward_critics(head, G.boneLength, None, True, cam, (1, 0))))
    print("D3 (fused only)    %.1f us" % ev(lambda: fused.critic3d(D3, xc, kcs=kcs)))
    print("D2                 %.1f us" % ev(lambda: fused.critic2d(D2, p2)))
    print("both critics       %.1f us" % ev(lambda: score_fake_pair(D3, D2, xc, kcs, p2)))
    print("whole step         %.1f us" % ev(lambda: score_fake_pair(D3, D2, *G.sample_for_critics(z, cam)[1:])))
