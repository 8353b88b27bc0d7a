#!/usr/bin/env python3
"""Golden vectors of ONE critic step at the benchmark's width (DenseDim 256), captured from the reference's own
train_Fk_discriminator (R/models_Fk_GAN/model_fk_gan_train.py:177-230) on torch CPU in THIS container:
tests/golden/critic_step_{d3,d2}_D256.npz.  The critics have 0.9 M / 0.27 M parameters: gradients and post-Adam weights are
kept as compact records (golden_util.compact: strided samples + seeded +-1 projections; biases and narrow layers whole).
Weights come from the closed-form seeded_state_dict both sides evaluate.  Run:  python tests/golden/make_golden_d256.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import _ref_import as RI          # noqa: E402
import golden_util as GU          # noqa: E402


def main():
    torch.set_num_threads(8)
    M = RI.load_reference()
    dis, train = M["dis"], M["train"]
    B, D = 64, 256
    args = RI.make_args(batch_size=B, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
    cpu = torch.device("cpu")
    proxy = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    proxy.device = lambda *a, **k: cpu                       # train_Fk_discriminator hard-codes "cuda"
    train.torch = proxy
    for tag, make, shapes, xr, xf, seed in (
            ("d3", lambda: dis.Fk_3D_Discriminator("cpu", args), GU.shapes_d3(D), GU.synth_pose16(B, seed=51), GU.synth_pose16(B, seed=52), 703),
            ("d2", lambda: dis.Fk_2D_Discriminator(args, 16), GU.shapes_d2(D),
             (torch.rand(B, 16, 2, generator=torch.Generator().manual_seed(53)) - 0.5) * 1.6,
             (torch.rand(B, 16, 2, generator=torch.Generator().manual_seed(54)) - 0.5) * 1.6, 704)):
        if tag == "d3":
            xr = xr - xr[:, :1]; xf = xf - xf[:, :1]
        net = make()
        sd = GU.seeded_state_dict(shapes, seed)
        assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v) for k, v in shapes.items()}
        net.load_state_dict(sd)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.5, 0.9))
        summary = types.SimpleNamespace(train_discrim_iter_num=1, train_iter_num=1)
        one = torch.tensor(1, dtype=torch.float32)
        torch.manual_seed(8765)
        alpha = torch.rand(B, 1)
        torch.manual_seed(8765)                              # calc_gradient_penalty draws alpha first (:210)
        W, C = train.train_Fk_discriminator(net, xr.clone(), xf.clone(), summary, M["Writer"](), "Fk_" + tag, opt, args, one, one * -1)
        out = dict(real=xr.numpy(), fake=xf.numpy(), alpha=alpha.numpy(), Wasserstein_D=W.detach().numpy(), D_cost=C.detach().numpy(),
                   weight_seed=np.array(seed))
        for i, (k, p) in enumerate(net.named_parameters()):
            for kind, t in (("grad", p.grad), ("delta", p.detach() - sd[k])):
                for part, v in GU.compact(t, 100 + i).items():
                    out["%s__%s__%s" % (kind, part, k)] = v.numpy()
        path = os.path.join(HERE, "critic_step_%s_D256.npz" % tag)
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB", "W %.6f C %.6f" % (W.item(), C.item()))


def forward_second_seed():
    """a SECOND set of forward goldens at DenseDim 256 (different weights, noise, poses, jitter): the parity-grade fused
    forward holds north_star's tolerances (pose 1e-5 m, logits 1e-4 rel) on more than one vector --
    tests/golden/{gen,critics}_D256_s2.npz, same fields as gen_D256 / critics_D256 of make_golden.py."""
    M = RI.load_reference()
    fkm, gen, dis = M["fkm"], M["gen"], M["dis"]
    D, B = 256, 256
    args = RI.make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
    fk = fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
    G = gen.Fk_Generator(fk, args, "cpu")
    G.load_state_dict(GU.seeded_state_dict(GU.shapes_generator(D), 1357))
    real = GU.synth_pose16(B, seed=71)
    G.GAN_generator_get_bone_length(real)
    z = torch.randn(B, 128, generator=torch.Generator().manual_seed(72))
    heads = []
    hk = G.deconv_out.register_forward_hook(lambda m, i, o: heads.append(o.detach().clone()))
    torch.manual_seed(4321)
    scaler = torch.randint(-200, 200, size=(B, 8)) / 1000.0      # what forward draws first
    torch.manual_seed(4321)
    fake = G(z)
    hk.remove()
    path = os.path.join(HERE, "gen_D256_s2.npz")
    np.savez_compressed(path, z=z.numpy(), real16=real.numpy(), bone_len=G.boneLength.detach().numpy(), scaler=scaler.numpy(),
                        head=heads[0].numpy(), angle37=G.distribute_angle[-1].detach().numpy(), fake=fake.detach().numpy(),
                        weight_seed=np.array(1357))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    D3, D2 = dis.Fk_3D_Discriminator("cpu", args), dis.Fk_2D_Discriminator(args, 16)
    D3.load_state_dict(GU.seeded_state_dict(GU.shapes_d3(D), 2468)); D2.load_state_dict(GU.seeded_state_dict(GU.shapes_d2(D), 3579))
    x3 = GU.synth_pose16(B, seed=73); x3 = x3 - x3[:, :1]
    x2 = (torch.rand(B, 16, 2, generator=torch.Generator().manual_seed(74)) - 0.5) * 1.6
    path = os.path.join(HERE, "critics_D256_s2.npz")
    np.savez_compressed(path, x3=x3.numpy(), x2=x2.numpy(), logit3=D3(x3).detach().numpy(), logit2=D2(x2).detach().numpy(),
                        weight_seed3=np.array(2468), weight_seed2=np.array(3579))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "forward":
        forward_second_seed()
    else:
        main()
