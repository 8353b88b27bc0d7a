"""Timing-perturbed repeat of one critic step: the same step (same weights, same rows) is run REP times while a second
stream keeps the memory system and the matrix pipes busy with unrelated work of varying length; every run's gradient
bucket and scalars must equal the first run's bit for bit (every kernel of the step sums in a fixed order -- except the
logit layer's bias, one value, added with atomics).  A stale LDS stage, a missing wait or a missing barrier in one of the
hand-scheduled kernels would show up here as a rare mismatch.
    python tools/stress_determinism.py [B] [D] [REP]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import argparse
import torch
import golden_util as GU
import test_gpu_graphs as T
import dhaug_amd
from dhaug_amd import graphs, critic_step as CS
from dhaug_amd.models_Fk_GAN import forward_kinematics_DH_model as fkm, model_fk_gan_train as train
from test_gpu_models import make_args

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
D = int(sys.argv[2]) if len(sys.argv) > 2 else 256
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 300
M = argparse.Namespace(graphs=graphs, fkm=fkm, train=train, cam=None)
args = make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
gen = torch.Generator().manual_seed(5)
x3 = GU.synth_pose16(B, seed=3).cuda()
x3 = (x3 - x3[:, :1]).reshape(B, 48).contiguous()
x2 = ((torch.rand(B, 32, generator=gen) - 0.5) * 1.2).cuda()
al = torch.rand(B, 1, generator=torch.Generator().manual_seed(3)).cuda()
side = torch.cuda.Stream()
big = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
buf = torch.empty(64 << 20, device="cuda", dtype=torch.uint8)
rng = torch.Generator().manual_seed(0)

d = T._build(M, args, D)
for split in (False, True):
    CS.TN_SPLIT = split
    for key, okey, real in (("model_d3d", "optimizer_d3d", x3), ("model_d2d", "optimizer_d2d", x2)):
        net, opt = d[key], d[okey]
        fake = (real.roll(1, 0) * 1.05 + 0.01).contiguous()
        p0 = opt.flat_param.clone(); m0 = opt.exp_avg.clone(); v0 = opt.exp_avg_sq.clone(); s0 = opt.step_dev.clone()
        # the logit layer's bias slot (atomics): excluded from the comparison
        names = [n for n, _ in net.named_parameters()]
        skip = [p for n, p in net.named_parameters() if p.numel() == 1]
        ref, bad = None, torch.zeros(1, device="cuda", dtype=torch.int64)
        for r in range(REP):
            with torch.no_grad():
                opt.flat_param.copy_(p0); opt.exp_avg.copy_(m0); opt.exp_avg_sq.copy_(v0); opt.step_dev.copy_(s0)
            from dhaug_amd import autograd_ops as A
            A.bump_weight_epoch()
            if r:
                k = int(torch.randint(0, 4, (1,), generator=rng))
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):              # unrelated load, a different amount every time
                    for _ in range(k):
                        buf.add_(1)
                        torch.mm(big, big)
            sc = CS.critic_step(net, opt, real, fake, al, 10.0)
            g = opt.flat_grad.clone()
            for p in skip:
                off = (p.grad.data_ptr() - opt.flat_grad.data_ptr()) // 4
                g[off] = 0
            cur = torch.cat((g, sc[:5].reshape(-1).float(), opt.flat_param.masked_fill(torch.zeros_like(g, dtype=torch.bool).index_fill_(0, torch.tensor([ (p.grad.data_ptr() - opt.flat_grad.data_ptr()) // 4 for p in skip], device="cuda"), True), 0)))
            if ref is None:
                ref = cur
            else:
                bad += (cur != ref).any().long()
        torch.cuda.synchronize()
        print(f"B={B} D={D} split={split} {key}: {int(bad.item())} of {REP - 1} repeats differ from the first", flush=True)
