"""GPU: the explicit generator step (dhaug_amd/gen_step.py) against the autograd composite it replaces -- same weights, noise,
jitter, camera -- in the fp32-grade arithmetic (tight) and in bf16 (direction), for the single-frame loop's step (D3 + D2, L/R
flip copies) and the video loop's (D3, D2 + both motion critics, flip + frame-reversed copies on the reference's (-1, R, 32)
view).  (The goldens captured from the reference's own G step are checked through gan_iteration / video_gan_iteration in
tests/test_gpu_loops.py -- those entry points take this path.)"""
import argparse

import pytest
import torch

import golden_util as GU
import loop_util as LU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dhaug_amd
    dhaug_amd._lib.lib()
    from dhaug_amd import gen_step
    from dhaug_amd.common.camera import camera_params9
    from dhaug_amd.common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params
    from dhaug_amd.models_Fk_GAN import forward_kinematics_DH_model as fkm, model_fk_gan_train as train
    ext = h36m_cameras_extrinsic_params["S1"][1]
    cam = ([float(v) for v in ext["orientation"]], [float(v) / 1000.0 for v in ext["translation"]],
           camera_params9(h36m_cameras_intrinsic_params[1]))
    return argparse.Namespace(fkm=fkm, train=train, gs=gen_step, cam=cam)


def _models(M, B, D, R, prec):
    from test_gpu_models import make_args
    video = R > 1
    args = make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D, video_Dis_DenseDim_3D=D,
                     video_Dis_DenseDim_2D=D, **(dict(single_or_multi_train_mode="multi", architecture="3,3",
                                                      GAN_3d_motion_loss_weight=0.7, GAN_2d_motion_loss_weight=0.4) if video else {}))
    fk = M.fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
    d = M.train.video_mode_my_get_poseFk_model(args, None, fk, R) if video else M.train.my_get_poseFk_model(args, None, fk)
    sds = dict(model_G=GU.seeded_state_dict(GU.shapes_generator(D, frames=R) if video else GU.shapes_generator(D), 21),
               model_d3d=GU.seeded_state_dict(GU.shapes_d3(D), 22), model_d2d=GU.seeded_state_dict(GU.shapes_d2(D), 23))
    if video:
        s3, s2 = LU.motion_shapes(D, R)
        sds.update(model_motion_d3d=GU.seeded_state_dict(s3, 24), model_motion_d2d=GU.seeded_state_dict(s2, 25))
    for k, sd in sds.items():
        with torch.no_grad():
            for n, p in d[k].named_parameters():
                p.copy_(sd[n].cuda())
        d[k].precision = prec
    from dhaug_amd import autograd_ops as A
    A.bump_weight_epoch()
    d["model_G"].GAN_generator_get_bone_length(GU.synth_pose16(B * R, seed=4).cuda())
    return args, d


def _step(M, B, D, R, prec, explicit, flip=True, playback=True):
    args, d = _models(M, B, D, R, prec)
    video = R > 1
    critics = (d["model_d3d"], d["model_d2d"]) + ((d["model_motion_d3d"], d["model_motion_d2d"]) if video else ())
    weights = (1.0, 0.2) + ((0.7, 0.4) if video else ())
    noise = torch.randn(B, 128, generator=torch.Generator().manual_seed(9)).cuda()
    scaler = (torch.randint(-200, 200, (B, 8), generator=torch.Generator().manual_seed(10)) / 1000.0).cuda()
    before = {k: p.detach().clone() for k, p in d["model_G"].named_parameters()}
    old = M.train.EXPLICIT_G_STEP
    M.train.EXPLICIT_G_STEP = explicit
    try:
        cost = M.train.generator_step(args, d["model_G"], d["optimizer_G"], critics, weights, M.cam, flip, noise, scaler, frames=R,
                                      playback=playback and video)
    finally:
        M.train.EXPLICIT_G_STEP = old
    G = d["model_G"]
    return (cost.item(), {k: p.grad.detach().float().clone() for k, p in G.named_parameters()},
            {k: (p.detach() - before[k]) for k, p in G.named_parameters()})


@pytest.mark.parametrize("B,D,R", [(72, 64, 1), (2048, 256, 1), (16, 32, 9), (256, 64, 9)])
def test_explicit_generator_step_equals_autograd(M, B, D, R):
    assert M.train.EXPLICIT_G_STEP
    ce, ge, de = _step(M, B, D, R, "bf16x6", True)
    ca, ga, da = _step(M, B, D, R, "bf16x6", False)
    assert abs(ce - ca) <= 2e-5 * max(1.0, abs(ca)), (ce, ca)
    for k in ga:
        scale = ga[k].abs().max().item()
        assert (ge[k] - ga[k]).abs().max().item() <= 1e-9 + 5e-5 * scale, (k, (ge[k] - ga[k]).abs().max().item(), scale)
        # one Adam step: both moved by lr * sign(g) where |g| is above the rounding of the two summation orders
        big = ga[k].abs() > 1e-3 * scale
        assert (de[k] - da[k])[big].abs().max().item() <= 2e-6 if big.any() else True, k
    # bf16 (the throughput arithmetic): same step up to bf16 rounding of the chain
    cb, gb, _ = _step(M, B, D, R, "bf16", True)
    cd, gd, _ = _step(M, B, D, R, "bf16", False)
    assert abs(cb - ce) <= 5e-2 * max(1.0, abs(ce)) and abs(cb - cd) <= 2e-2 * max(1.0, abs(cd))
    cs = torch.nn.functional.cosine_similarity
    cos = [cs(gb[k].reshape(-1).double(), ge[k].reshape(-1).double(), dim=0).item() for k in ge if ge[k].abs().max() > 0]
    cos2 = [cs(gb[k].reshape(-1).double(), gd[k].reshape(-1).double(), dim=0).item() for k in ge if ge[k].abs().max() > 0]
    # (the two bf16 paths run the same arithmetic; against the fp32-grade step only the direction is comparable, and only
    # where the batch averages the bf16 rounding out: 16 clips through 32-wide bf16 layers do not)
    assert min(cos2) > 0.99, cos2
    if R == 1:
        assert min(cos) > 0.9, cos


def test_explicit_generator_step_variants(M):
    """no flip copies / no playback: the weights of the terms change (1 instead of 1/2), nothing else"""
    for flip, playback, R in ((False, False, 1), (False, True, 9), (True, False, 9)):
        B, D = (72, 32) if R == 1 else (16, 32)
        ce, ge, _ = _step(M, B, D, R, "bf16x6", True, flip, playback)
        ca, ga, _ = _step(M, B, D, R, "bf16x6", False, flip, playback)
        assert abs(ce - ca) <= 2e-5 * max(1.0, abs(ca)), (flip, playback, ce, ca)
        for k in ga:
            scale = ga[k].abs().max().item()
            assert (ge[k] - ga[k]).abs().max().item() <= 1e-9 + 5e-5 * scale, (flip, playback, k)


def test_no_autograd_node_in_the_explicit_step(M):
    """the step runs under no_grad and leaves no graph: every parameter gradient lives in the optimizer's flat bucket"""
    args, d = _models(M, 72, 32, 1, "bf16")
    G, oG = d["model_G"], d["optimizer_G"]
    noise = torch.randn(72, 128, device="cuda")
    cost = M.train.generator_step(args, G, oG, (d["model_d3d"], d["model_d2d"]), (1.0, 0.2), M.cam, True, noise, None)
    assert cost.grad_fn is None and not cost.requires_grad
    lo, hi = oG.flat_grad.data_ptr(), oG.flat_grad.data_ptr() + oG.flat_grad.numel() * 4
    for p in G.parameters():
        assert p.grad is not None and p.grad.grad_fn is None and lo <= p.grad.data_ptr() < hi
    assert int(oG.step_dev.item()) == 1
