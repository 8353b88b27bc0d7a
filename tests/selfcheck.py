"""smoke(): one small invocation of the hot path on cuda:0, checked against the oracle (test infrastructure)."""
import os
import sys

import torch


def smoke():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import dhaug_oracle as O           # checker only
    import dhaug_amd  # noqa: F401
    from dhaug_amd import _lib, ops
    from dhaug_amd.function_aug.config import synth_args
    from dhaug_amd.models_Fk_GAN import model_fk_gan_train as T
    from dhaug_amd.models_Fk_GAN.forward_kinematics_DH_model import Forward_Kinematics_DH_Model

    _lib.lib()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(0)
    N = 1000
    ang = (torch.rand(N, 37, generator=g) * 2 - 1) * 180
    bl = torch.rand(N, 15, generator=g) * 0.4 + 0.1
    rt = torch.randn(N, 3, generator=g)
    out = ops.fk_forward(ang.to(dev), bl.to(dev), rt.to(dev))
    err = (out.cpu() - O.fk_forward16(ang, bl, rt)).abs().max().item()
    assert err <= 1e-5, "FK parity %g" % err
    # generator + critics, fp32-grade mode, against the oracle on the same weights
    B, D = 256, 64
    args = synth_args(B, D)
    fk = Forward_Kinematics_DH_Model(args, ["S1"], None)
    d = T.my_get_poseFk_model(args, None, fk)
    G, D3, D2 = d["model_G"], d["model_d3d"], d["model_d2d"]
    for m in (G, D3, D2):
        m.precision = "bf16x6"
    z = torch.randn(B, 128, generator=g)
    real = torch.randn(B, 16, 3, generator=g) * 0.3
    sc = torch.randint(-200, 200, (B, 8), generator=g) / 1000.0
    G.GAN_generator_get_bone_length(real.to(dev))
    fake = G(z.to(dev), bone_len_scaler=sc)
    sdG = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    ref, _, _ = O.generator_forward(z, sdG, O.bone_lengths(real), sc)
    e = (fake.detach().cpu() - ref).abs().max().item()
    assert e <= 5e-5, "generator parity %g" % e
    fw = fake.detach().reshape(-1, 16, 3)
    x3 = ops.center_flip(fw, True, False)
    l3 = D3(x3)
    r3 = O.d3_forward(x3.cpu(), {k: v.detach().cpu() for k, v in D3.state_dict().items()})
    rel = ((l3.detach().cpu() - r3).abs() / r3.abs().clamp_min(0.1 * r3.abs().mean())).max().item()
    assert rel <= 1e-3, "D3 parity %g" % rel
    # one full GAN iteration in the default (bf16) arithmetic: runs, finite, parameters move
    for m in (G, D3, D2):
        m.precision = "bf16"
    before = D3.output.weight.detach().clone()
    cam_param = torch.zeros(B, 16)
    cam_param[:, 9] = 1.0
    r = T.gan_iteration(args, d, real, cam_param, torch.rand(B, 16, 2) - 0.5, ["S1"], summary=None, writer=None,
                        do_g_step=True)
    torch.cuda.synchronize()
    assert torch.isfinite(r["Wasserstein_D_3D"]).item() and torch.isfinite(r["G_cost"]).item()
    assert (D3.output.weight.detach() - before).abs().max().item() > 0
    print("smoke: FK err %.2e, generator err %.2e, D3 rel %.2e, W3 %.4f" % (err, e, rel, r["Wasserstein_D_3D"].item()))
