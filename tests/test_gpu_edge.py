"""GPU: degenerate inputs through the kernels, against what the reference's ATen arithmetic does with them (the oracle).

  * zero-length bones: the KCS cosines divide by the bone lengths (R/models_Fk_GAN/Fk_discriminator.py:36-146) -> 0/0 = NaN
    in exactly the entries that touch the degenerate bone, finite everywhere else; same through bone_length / KCS-VJP.
  * NaN / inf through the training-path dense layers (IEEE semantics kept in dhaug_gemm.hip / dhaug_elem.hip): a NaN
    activation reaches the logit, as with nn.Linear + ReLU.
  * angles far outside the joint range (+-1e4 deg, 55 turns): the Cody-Waite reduction of the FK kernel stays within
    1e-5 of the reference arithmetic.
  * FK is compiled with -ffinite-math-only (documented in __graft_entry__.py): a NaN angle gives an unspecified value for
    THAT pose only; other poses of the launch are untouched."""
import pytest
import torch

import golden_util as GU
from oracle import dhaug_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dhaug_amd
    dhaug_amd._lib.lib()
    from dhaug_amd import ops
    return ops


def same_nan_pattern(a, b, tol):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert torch.equal(torch.isnan(a), torch.isnan(b)), (torch.isnan(a).sum().item(), torch.isnan(b).sum().item())
    m = ~torch.isnan(b)
    assert ((a[m] - b[m]).abs().max().item() if m.any() else 0.0) <= tol


def test_kcs_zero_length_bone_is_nan_like_the_reference(ops):
    x = GU.synth_pose16(70, seed=3)
    x[3, 6] = x[3, 5]               # left shin collapsed (bone 0): cosine 0 involves it
    x[17, 9] = x[17, 8]             # neck collapsed (bone 14): cosine 8
    x[40] = 0.0                     # every bone degenerate
    ref = O.kcs_features(x)
    assert torch.isnan(ref[3]).sum().item() == 1 and torch.isnan(ref[40, :15]).all()
    got, _ = ops.kcs_forward(x.cuda(), True, f32=True)
    same_nan_pattern(got, ref, 5e-6)
    same_nan_pattern(ops.kcs_forward(x.cuda(), False, f32=True)[0], O.kcs_features(x, with_lengths=False), 5e-6)
    same_nan_pattern(ops.bone_length(x.cuda()), O.bone_lengths(x), 1e-6)
    # the bf16 operand the fused 3D critic reads carries the same NaNs
    _, kb = ops.kcs_forward(x.cuda(), True, f32=False, bf16_ld=32)
    assert torch.equal(torch.isnan(kb[:, :30].float().cpu()), torch.isnan(ref))
    # VJP: NaN rows stay confined to the degenerate poses
    gf = torch.ones(70, 30)
    gx = ops.kcs_backward(x.cuda(), gf.cuda(), True).cpu()
    bad = torch.isnan(gx).reshape(70, -1).any(1)
    assert bad[3] and bad[17] and bad[40] and bad.sum().item() == 3


def test_nan_and_inf_reach_the_logit_on_the_training_path(ops):
    from dhaug_amd import autograd_ops as A
    torch.manual_seed(0)
    W1, b1 = (torch.randn(64, 48) * 0.2).cuda(), torch.zeros(64).cuda()
    W2, b2 = (torch.randn(64, 64) * 0.2).cuda(), torch.zeros(64).cuda()
    x = torch.randn(40, 48)
    x[5, 7] = float("nan")
    x[9, 3] = float("inf")
    for prec in ("bf16", "bf16x6"):
        h = A.linear(x.cuda(), W1, b1, None, A.ACT_RELU, 0.0, prec)
        y = A.linear(h, W2, b2, None, A.ACT_NONE, 0.0, prec, out_f32=True).cpu()
        ref = torch.relu(x @ W1.cpu().t()) @ W2.cpu().t()
        assert torch.isnan(ref[5]).all() and torch.isnan(y[5]).all(), prec           # NaN row stays NaN
        assert not torch.isfinite(y[9]).any() and not torch.isfinite(ref[9]).any(), prec   # inf row: inf / NaN, never finite
        ok = torch.ones(40, dtype=torch.bool); ok[5] = ok[9] = False
        assert torch.isfinite(y[ok]).all()
    # Adam: a NaN gradient element poisons that element only
    p, g = torch.ones(1024).cuda(), torch.zeros(1024).cuda()
    g[17] = float("nan")
    m, v = torch.zeros(1024).cuda(), torch.zeros(1024).cuda()
    ops.adam_step(p, g, m, v, 1)
    assert torch.isnan(p[17]).item() and torch.isfinite(torch.cat((p[:17], p[18:]))).all()


def test_nan_in_the_fused_step_forward(ops):
    """sweep 1 of the explicit critic step (forward-with-save, one fused launch; its translation unit applies ReLU as
    max(v, v * 0) in fp32 -- NaN-propagating -- where the inference unit uses an integer max on the packed bf16 pair that turns
    the matrix pipe's -NaN into 0): a NaN input row never touches another row and reaches ITS logit in both critics,
    train_Fk_discriminator's D_cost reports it, and diverged (NaN) weights give NaN logits everywhere -- as the reference's
    ATen ops would."""
    import argparse
    from dhaug_amd import fused
    from dhaug_amd.models_Fk_GAN import Fk_discriminator as dis, model_fk_gan_train as train
    from test_gpu_models import make_args
    B, D = 300, 256
    args = make_args(batch_size=B, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
    torch.manual_seed(3)
    D3, D2 = dis.Fk_3D_Discriminator("cuda", args).cuda(), dis.Fk_2D_Discriminator(args, 16).cuda()
    x3 = GU.synth_pose16(B, seed=5); x3 = (x3 - x3[:, :1]).reshape(B, 48).cuda()
    x2 = ((torch.rand(B, 32) - 0.5) * 1.6).cuda()
    clean3 = fused.critic3d_forward_save(D3, x3, ops.kcs_forward(x3, True, f32=True, bf16_ld=32)[1])["logits"].clone()
    x3[7, 20] = float("nan"); x2[11, 3] = float("nan")
    kf, kb = ops.kcs_forward(x3, True, f32=True, bf16_ld=32)
    l3 = fused.critic3d_forward_save(D3, x3, kb)["logits"]
    l2 = fused.critic2d_forward_save(D2, x2)["logits"]
    keep = torch.arange(B, device="cuda") != 7
    assert torch.equal(l3[keep], clean3[keep]) and torch.isnan(l3[7]).item()       # its own logit, no other row
    bad2 = torch.isnan(l2[:, 0])
    assert bad2[11].item() and bad2.sum().item() == 1
    opt = train.FusedAdam(D2.parameters(), lr=1e-4, betas=(0.5, 0.9))
    W, C = train.train_Fk_discriminator(D2, x2.clone(), (x2 + 0.01).clone(), argparse.Namespace(train_iter_num=0), None, "d2d", opt, args)
    assert torch.isnan(C).item() and torch.isnan(W).item()
    # a diverged 3D critic (one NaN weight in the first pose layer): every row's logit is NaN
    with torch.no_grad():
        D3.previous[0].weight[5, 9] = float("nan")
    from dhaug_amd import autograd_ops as A
    A.bump_weight_epoch()
    x3c = x3.clone(); x3c[7, 20] = 0.1
    l3d = fused.critic3d_forward_save(D3, x3c, ops.kcs_forward(x3c, True, f32=True, bf16_ld=32)[1])["logits"]
    assert torch.isnan(l3d).all()


def test_nonfinite_in_the_fused_inference_programs(ops):
    """What the fused INFERENCE programs (fused_mlp_kernel<false>: bf16; fused_mlp_x3_kernel: f16x3 -- the sampling pass, the G
    step's flipped evaluations, score_fake_pair) do with non-finite values, in both settings of dhaug_set_nan_propagation:
    * default: their ReLU is an integer max on the bit pattern, which turns the matrix pipe's -NaN into 0 -- a NaN / +-inf input
      row of the 3D critic or the generator trunk gives a FINITE logit / head (documented deviation, include/dhaug.h,
      INTEGRATION.md), and a NaN weight of a ReLU network does not show in the output; the LeakyReLU 2D critic (mul + max in
      fp32) propagates: NaN in that row's logit, NaN weights give NaN logits everywhere;
    * nan_propagation(True): every network does what the reference's ATen ops do (R/models_Fk_GAN/Fk_discriminator.py:180-201,
      253-266): the bad row's own logit / head row is NaN, NaN weights give NaN everywhere;
    * in either setting no other row changes by a bit, and finite inputs give the same bits in both settings."""
    from dhaug_amd import fused, autograd_ops as A
    from dhaug_amd.models_Fk_GAN import Fk_discriminator as dis, Fk_generator as gen
    from test_gpu_models import make_args
    B, D = 512, 256
    args = make_args(batch_size=B, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D, Gen_DenseDim=D)
    torch.manual_seed(3)
    D3, D2 = dis.Fk_3D_Discriminator("cuda", args).cuda(), dis.Fk_2D_Discriminator(args, 16).cuda()
    G = gen.Fk_Generator(None, args, "cuda").cuda()
    gsel = torch.Generator().manual_seed(8)
    x3 = (torch.randn(B, 48, generator=gsel) * 0.3).cuda()
    x2 = (torch.rand(B, 32, generator=gsel) - 0.5).cuda()
    z = torch.randn(B, 128, generator=gsel).cuda()

    def run(mode, x3, x2, z):
        kf, kb = ops.kcs_forward(x3, True, f32=True, bf16_ld=32)
        with torch.no_grad():
            l3 = fused.critic3d(D3, x3 if mode == "f16x3" else x3.bfloat16(), kcs=kf if mode == "f16x3" else kb, mode=mode).float().reshape(-1).clone()
            l2 = fused.critic2d(D2, x2 if mode == "f16x3" else x2.bfloat16(), mode=mode).float().reshape(-1).clone()
            h = fused.generator_head(G, z, mode).float().clone()
        return l3, l2, h

    rows = torch.arange(B, device="cuda")
    for mode in ("bf16", "f16x3"):
        c3, c2, ch = run(mode, x3, x2, z)
        assert torch.isfinite(c3).all() and torch.isfinite(c2).all() and torch.isfinite(ch).all()
        with ops.nan_propagation(True):
            p3, p2, ph = run(mode, x3, x2, z)
        assert torch.equal(p3, c3) and torch.equal(p2, c2) and torch.equal(ph, ch), mode        # finite values: the same bits
        for val in (float("nan"), float("inf"), float("-inf")):
            a3, a2, az = x3.clone(), x2.clone(), z.clone()
            a3[7, 20] = val; a2[11, 3] = val; az[5, 9] = val
            for prop in (False, True):
                with ops.nan_propagation(prop):
                    l3, l2, h = run(mode, a3, a2, az)
                assert torch.equal(l3[rows != 7], c3[rows != 7]) and torch.equal(l2[rows != 11], c2[rows != 11]), (mode, val, prop)
                assert torch.equal(h[rows != 5], ch[rows != 5]), (mode, val, prop)
                assert torch.isnan(l2[11]).item(), (mode, val, prop)                            # LeakyReLU: always propagates
                if prop:
                    assert torch.isnan(l3[7]).item() and torch.isnan(h[5]).all(), (mode, val)
                else:
                    assert torch.isfinite(l3[7]).item() and torch.isfinite(h[5]).all(), (mode, val)   # the documented deviation
        # one NaN weight in each network's first layer
        keep = [(m, m.weight[5, 9].item()) for m in (D3.previous[0], D2.pose_layer_1, G.preprocess[0])]
        with torch.no_grad():
            for m, _ in keep:
                m.weight[5, 9] = float("nan")
        A.bump_weight_epoch()
        for prop in (False, True):
            with ops.nan_propagation(prop):
                l3, l2, h = run(mode, x3, x2, z)
            assert torch.isnan(l2).all(), (mode, prop)
            if prop:
                assert torch.isnan(l3).all() and torch.isnan(h).all(), mode
            else:
                assert torch.isfinite(l3).all() and torch.isfinite(h).all(), mode                # the documented deviation
        with torch.no_grad():
            for m, w in keep:
                m.weight[5, 9] = w
        A.bump_weight_epoch()


def test_fk_angles_far_outside_the_joint_range(ops):
    g = torch.Generator().manual_seed(9)
    N = 4096
    a = (torch.rand(N, 37, generator=g) * 2 - 1) * 1.0e4
    bl = torch.rand(N, 15, generator=g) * 0.4 + 0.1
    rt = torch.randn(N, 3, generator=g)
    out = ops.fk_forward(a.cuda(), bl.cuda(), rt.cuda()).cpu()
    ref = O.fk_forward16(a, bl, rt)
    assert (out - ref).abs().max().item() <= 1e-5
    # and at the edge of the reduction's fast path (131072 rad ~ 7.5e6 deg) the library path takes over seamlessly
    a2 = a * 1.0e3
    out2 = ops.fk_forward(a2.cuda(), bl.cuda(), rt.cuda()).cpu()
    assert (out2 - O.fk_forward16(a2, bl, rt)).abs().max().item() <= 1e-5


def test_fk_nan_pose_does_not_leak_into_other_poses(ops):
    a, bl, rt = GU.synth_fk_inputs(256, seed=12)
    clean = ops.fk_forward(a.cuda(), bl.cuda(), rt.cuda()).cpu()
    a2 = a.clone()
    a2[100, 12] = float("nan")
    a2[101, 3] = float("inf")
    out = ops.fk_forward(a2.cuda(), bl.cuda(), rt.cuda()).cpu()
    keep = torch.ones(256, dtype=torch.bool); keep[100] = keep[101] = False
    assert torch.equal(out[keep], clean[keep])
    # zero bone lengths: joints collapse onto their parents exactly as in the reference
    bl0 = bl.clone(); bl0[:, [0, 12]] = 0.0
    o0 = ops.fk_forward(a.cuda(), bl0.cuda(), rt.cuda()).cpu()
    assert (o0 - O.fk_forward16(a, bl0, rt)).abs().max().item() <= 1e-5
    assert (o0[:, 6] - o0[:, 5]).abs().max().item() <= 1e-6


def test_empty_batches(ops):
    """N = 0 through the entry points of the path: nothing is launched, shapes are kept, accumulating outputs are untouched"""
    z = lambda *s: torch.zeros(*s, device="cuda")
    assert ops.fk_forward(z(0, 37), z(0, 15), z(0, 3)).shape == (0, 16, 3)
    fake, ang = ops.gen_tail_forward(z(0, 35), z(0, 15), None, True)
    assert fake.shape[0] == 0
    out = ops.gen_tail_forward_critics(z(0, 35), z(0, 15), None, True, ([1.0, 0, 0, 0], [0.0, 0, 0], [1.0, 1, 0, 0, 0, 0, 0, 0, 0]))
    assert all(t.shape[0] == 0 for t in out)
    kf, kb = ops.kcs_forward(z(0, 48), True, f32=True, bf16_ld=32)
    assert kf.shape == (0, 30) and kb.shape == (0, 32)
    assert ops.bone_length(z(0, 16, 3)).shape[0] == 0
    A = torch.zeros(0, 64, device="cuda", dtype=torch.bfloat16)
    W = torch.zeros(32, 64, device="cuda", dtype=torch.bfloat16)
    cb, _ = ops.gemm_nt(A, W, 32, 64, out_bf16=True)
    assert cb.shape[0] == 0
    acc = torch.full((32, 64), 3.0, device="cuda")
    got = ops.gemm_tn(torch.zeros(0, 32, device="cuda", dtype=torch.bfloat16), A, 32, 64, out=acc, accumulate=True)
    assert (got == 3.0).all()                                    # dW += 0 rows
    fresh = ops.gemm_tn(torch.zeros(0, 32, device="cuda", dtype=torch.bfloat16), A, 32, 64)
    assert fresh.shape == (32, 64) and (fresh == 0).all()
    torch.cuda.synchronize()
