"""GPU: the hipGraph path of the training iteration (dhaug_amd/graphs.py) against the eager path it replaces in bench.py.

N iterations eager and N iterations as captured graphs, from identical weights, identical inputs and constant draws
(ConstDraws: a graph bakes its draws in): same weights, same Adam state, same step counts, same scalars.  What differs
between the two paths and is therefore under test: the in-capture weight re-pack (optimizer prologue + FusedNet in-place
re-pack), the device-side Adam step count, the static input buffers, and that BUILDING a graph (two warm-up calls + the
capture) advances nothing."""
import argparse

import pytest
import torch

import golden_util as GU

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dhaug_amd
    dhaug_amd._lib.lib()
    from dhaug_amd import graphs
    from dhaug_amd.common.camera import camera_params9
    from dhaug_amd.common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params
    from dhaug_amd.models_Fk_GAN import forward_kinematics_DH_model as fkm, model_fk_gan_train as train
    ext = h36m_cameras_extrinsic_params["S1"][0]
    cam = ([float(v) for v in ext["orientation"]], [float(v) / 1000.0 for v in ext["translation"]],
           camera_params9(h36m_cameras_intrinsic_params[0]))
    return argparse.Namespace(graphs=graphs, fkm=fkm, train=train, cam=cam)


def _build(M, args, D):
    fk = M.fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
    d = M.train.my_get_poseFk_model(args, None, fk)
    for key, shapes, seed in (("model_G", GU.shapes_generator(D), 11), ("model_d3d", GU.shapes_d3(D), 12), ("model_d2d", GU.shapes_d2(D), 13)):
        sd = GU.seeded_state_dict(shapes, seed)
        with torch.no_grad():
            for k, p in d[key].named_parameters():
                p.copy_(sd[k].cuda())                       # in place: the parameters stay views of the optimizer's flat buffer
    from dhaug_amd import autograd_ops as A
    A.bump_weight_epoch()
    return d


@pytest.mark.parametrize("B,D,split", [(96, 64, False), (2048, 256, False), (2048, 256, True)])
def test_graphed_iterations_equal_eager(M, B, D, split, monkeypatch):
    """split: sweep 4 in two parts (critic_step.TN_SPLIT) in both forms -- the capture then keeps the critics' steps on its own
    stream and forks there (run_critic_steps at 'long' batches; the threshold is lowered to this test's batch), the eager
    iteration forks inside its concurrent critic streams.  Without it both forms sum in one part."""
    from test_gpu_models import make_args
    from dhaug_amd import critic_step as CS
    monkeypatch.setattr(CS, "TN_SPLIT", split)
    monkeypatch.setattr(M.train, "LONG_ROWS", 1024 if split else 1 << 30)
    args = make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
    gen = torch.Generator().manual_seed(5)
    x3 = GU.synth_pose16(B, seed=3).cuda()
    x3 = x3 + torch.tensor([0.0, 0.0, 4.5], device="cuda")                       # camera space: in front of the camera
    x2 = ((torch.rand(B, 16, 2, generator=gen) - 0.5) * 1.2).cuda()
    cp = torch.zeros(B, 16, device="cuda")
    cp[:, 9:13] = torch.tensor(M.cam[0], device="cuda")
    cp[:, 13:16] = torch.tensor(M.cam[1], device="cuda")
    mk = lambda: M.train.ConstDraws(noise=[torch.randn(B, 128, generator=torch.Generator().manual_seed(1)).cuda()],
                                    scaler=[(torch.randint(-200, 200, (B, 8), generator=torch.Generator().manual_seed(2)) / 1000.0).cuda()],
                                    alpha=[torch.rand(B, 1, generator=torch.Generator().manual_seed(3)).cuda()])
    # five iterations, the G step at the end of the fifth: the critics' ten steps each are deterministic kernels (graph and
    # eager must agree to the last bit or two); the G step's short / ragged weight-gradient shapes still add with fp32 atomics,
    # and Adam's g / (|g| + eps) turns a last-bit difference of a near-zero gradient into a visible one -- so nothing that
    # DEPENDS on the stepped generator is compared (a sixth iteration would be)
    N = 5
    # ---- eager
    de = _build(M, args, D)
    dr = mk()
    eager = []
    for i in range(N):
        r = M.train.gan_iteration(args, de, x3, cp, x2, ["S1"], None, None, do_g_step=(i % 5 == 4), camera=M.cam, draws=dr)
        eager.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in r.items()})
        if i == 0:
            first_e = {ok: de[ok].flat_param.clone() for ok in ("optimizer_d3d", "optimizer_d2d")}
    # ---- graphs (two graphs: with / without the G step); building one must not advance anything
    dg = _build(M, args, D)
    dr2 = mk()
    G = M.graphs.GraphedGanIteration(M.train.gan_iteration, args, dg, ["S1"], None)
    for i in range(N):
        r = G(x3, cp, x2, i % 5 == 4, M.cam, draws=dr2)
        e = eager[i]
        for k in ("Wasserstein_D_3D", "D_cost_3D", "Wasserstein_D_2D", "D_cost_2D", "G_cost"):
            if e[k] is None:
                assert r[k] is None
                continue
            # the first iterations to 1e-4; later ones follow the drift ONE last-bit difference of a gradient grows into
            # (Adam's g / (|g| + eps) turns it into a +-lr step of that weight: measured with the two summation orders of
            # sweep 4, tools/determinism.py, 4e-5 relative after three iterations, 9e-5 after four)
            tol = 1e-4 if i < 2 else 1e-3
            assert abs(r[k].item() - e[k].item()) <= tol * max(1.0, abs(e[k].item())), (i, k, r[k].item(), e[k].item())
        assert (r["pos_3d_cam"] - e["pos_3d_cam"]).abs().max().item() <= 1e-5
        if i == 0 and B >= 2048:
            # after the first replay (two steps of each critic, every kernel of them summing in a fixed order): the same bits
            for ok, pe in first_e.items():
                assert (dg[ok].flat_param - pe).abs().max().item() <= 1e-7 * pe.abs().max().item(), ok
    assert len(G.graphs) == 2
    steps = {"optimizer_G": 1, "optimizer_d3d": 2 * N, "optimizer_d2d": 2 * N}
    for ok, n in steps.items():
        assert int(dg[ok].step_dev.item()) == n == int(de[ok].step_dev.item()), (ok, dg[ok].step_dev.item(), de[ok].step_dev.item())
        assert dg[ok].state_dict()["dhaug_flat"]["step_count"] == n
        for name in ("flat_param", "exp_avg", "exp_avg_sq"):
            a, b = getattr(dg[ok], name), getattr(de[ok], name)
            scale = b.abs().max().item()
            # after N iterations: within what one flipped last bit of a near-zero gradient grows into (+-lr per step in that
            # weight); the bit-for-bit comparison is the one after the first iteration, above
            tol = 1e-2 * scale + 1e-12 if name != "flat_param" else 1.1e-4 * n
            assert (a - b).abs().max().item() <= tol, (ok, name, (a - b).abs().max().item(), scale)
    for mk_, mo in (("model_G", "optimizer_G"), ("model_d3d", "optimizer_d3d"), ("model_d2d", "optimizer_d2d")):
        sg, se = dg[mk_].state_dict(), de[mk_].state_dict()
        assert sg.keys() == se.keys()
        for k in sg:
            assert (sg[k] - se[k]).abs().max().item() <= 1.1e-4 * steps[mo], (mk_, k)


def test_graphed_video_iterations_equal_eager(M):
    """the video loop's iteration (four critics, playback copies, explicit G step) as hipGraphs against eager, like the
    single-frame test above"""
    from dhaug_amd.models_Fk_GAN import video_GAN_fun as V
    import loop_util as LU
    from test_gpu_models import make_args
    B, R, D, N = 32, 9, 32, 5
    args = make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D, video_Dis_DenseDim_3D=D, video_Dis_DenseDim_2D=D,
                     single_or_multi_train_mode="multi", architecture="3,3", single_dis_warmup_epoch=0, GAN_video_playback_input=True,
                     GAN_3d_motion_loss_weight=1.0, GAN_2d_motion_loss_weight=1.0)
    x3 = (GU.synth_pose16(B * R, seed=3) + torch.tensor([0.0, 0.0, 4.5])).reshape(B, R, 16, 3).cuda()
    x2 = ((torch.rand(B, R, 16, 2, generator=torch.Generator().manual_seed(5)) - 0.5) * 1.2).cuda()
    cp = torch.zeros(B, 16, device="cuda")
    cp[:, 9:13] = torch.tensor(M.cam[0], device="cuda"); cp[:, 13:16] = torch.tensor(M.cam[1], device="cuda")
    mk = lambda: M.train.ConstDraws(noise=[torch.randn(B, 128, generator=torch.Generator().manual_seed(1)).cuda()],
                                    scaler=[(torch.randint(-200, 200, (B, 8), generator=torch.Generator().manual_seed(2)) / 1000.0).cuda()],
                                    alpha=[torch.rand(B * R, 1, generator=torch.Generator().manual_seed(3)).cuda()])

    def build():
        fk = M.fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
        d = M.train.video_mode_my_get_poseFk_model(args, None, fk, R)
        s3, s2 = LU.motion_shapes(D, R)
        for key, shapes, seed in (("model_G", GU.shapes_generator(D, frames=R), 11), ("model_d3d", GU.shapes_d3(D), 12),
                                  ("model_d2d", GU.shapes_d2(D), 13), ("model_motion_d3d", s3, 14), ("model_motion_d2d", s2, 15)):
            sd = GU.seeded_state_dict(shapes, seed)
            with torch.no_grad():
                for k, p in d[key].named_parameters():
                    p.copy_(sd[k].cuda())
        from dhaug_amd import autograd_ops as A
        A.bump_weight_epoch()
        return d

    summ = argparse.Namespace(epoch=10, train_iter_num=0)
    de, dr = build(), mk()
    eager = []
    for i in range(N):
        r = V.video_gan_iteration(args, de, x3, cp, x2, ["S1"], summ, None, do_g_step=(i % 5 == 4), camera=M.cam, draws=dr)
        eager.append(r["pos_3d_cam"].clone())
    dg, dr2 = build(), mk()
    G = M.graphs.GraphedGanIteration(V.video_gan_iteration, args, dg, ["S1"], summ)
    for i in range(N):
        r = G(x3, cp, x2, i % 5 == 4, M.cam, draws=dr2)
        assert (r["pos_3d_cam"] - eager[i]).abs().max().item() <= 1e-5
    for ok in ("optimizer_d3d", "optimizer_d2d", "optimizer_motion_d3d", "optimizer_motion_d2d", "optimizer_G"):
        assert int(dg[ok].step_dev.item()) == int(de[ok].step_dev.item()) > 0, ok
        a, b = dg[ok].flat_param, de[ok].flat_param
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-12, (ok, (a - b).abs().max().item())


def test_a_diverged_generator_shows_in_the_iteration_eager_and_graphed(M):
    """gan_iteration runs the fused inference programs (the sampling pass G(z), the G step's value-only evaluations) in their
    NaN-propagating form (ops.nan_propagation: read when a kernel is launched, so a captured iteration keeps it on replay): NaN
    weights in the generator reach the critic steps' costs -- D_cost, Wasserstein_D -- as they would through the reference's ATen
    arithmetic (R/models_Fk_GAN/model_fk_gan_train.py:305-341), in the eager iteration and in its hipGraph replay alike."""
    from test_gpu_models import make_args
    from dhaug_amd import autograd_ops as A
    B, D = 256, 256
    args = make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
    x3 = GU.synth_pose16(B, seed=3).cuda() + torch.tensor([0.0, 0.0, 4.5], device="cuda")
    x2 = ((torch.rand(B, 16, 2, generator=torch.Generator().manual_seed(5)) - 0.5) * 1.2).cuda()
    cp = torch.zeros(B, 16, device="cuda")
    cp[:, 9:13] = torch.tensor(M.cam[0], device="cuda")
    cp[:, 13:16] = torch.tensor(M.cam[1], device="cuda")
    for graphed in (False, True):
        d = _build(M, args, D)
        run = (M.graphs.GraphedGanIteration(M.train.gan_iteration, args, d, ["S1"], argparse.Namespace(epoch=10, train_iter_num=0))
               if graphed else None)
        call = (lambda: run(x3, cp, x2, False, M.cam)) if graphed else (
            lambda: M.train.gan_iteration(args, d, x3, cp, x2, ["S1"], None, None, do_g_step=False, camera=M.cam))
        if not graphed:
            r = call()
            assert torch.isfinite(r["D_cost_3D"]).item() and torch.isfinite(r["D_cost_2D"]).item()
        with torch.no_grad():
            d["model_G"].block2.fc1.weight[3, 7] = float("nan")          # a diverged generator: one NaN weight in its trunk
        A.bump_weight_epoch()
        for _ in range(2):                                               # (graphed: the capturing call, then a pure replay)
            r = call()
            assert torch.isnan(r["D_cost_3D"]).item() and torch.isnan(r["D_cost_2D"]).item(), ("graphed" if graphed else "eager")
            assert torch.isnan(r["pos_3d_cam"]).any().item()


@pytest.mark.parametrize("D,B,graphed", [(256, 2048, False), (256, 2048, True), (1000, 160, False)])
def test_whole_iterations_read_no_unwritten_memory(M, D, B, graphed, monkeypatch):
    """Five whole single-frame iterations -- sampling pass, four critic steps each, the explicit G step on the fifth (gen_step.py: trunk
    with saved activations, FK tail, camera, both critics' chains, the flipped value-only evaluations through the fused inference
    programs, the grouped weight gradients) -- with EVERY torch.empty / empty_like buffer pre-filled with NaN, then with 3e38: the same
    critics (bit for bit after the first iteration; the later ones and the G step within what its atomically summed short contractions
    allow) as with fresh memory, eager and as hipGraphs, at DenseDim 256 (fused programs) and at DenseDim 1000 (layer GEMMs incl. the
    ping-pong kernel, whose 1000-wide column blocks are read 1008 wide: the guard rows of critic_step / gen_step).  An operand row,
    pad column or guard row that nobody wrote would show as NaN / as a huge value.  (tests/test_gpu_critic_step.py holds single critic
    steps to the same probe; tools/poison.py is the development form.)"""
    from test_gpu_models import make_args
    if D == 1000:
        monkeypatch.setenv("DHAUG_GEMM_WIDE_MIN_TILES", "1")
    args = make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
    x3 = GU.synth_pose16(B, seed=3).cuda() + torch.tensor([0.0, 0.0, 4.5], device="cuda")
    x2 = ((torch.rand(B, 16, 2, generator=torch.Generator().manual_seed(5)) - 0.5) * 1.2).cuda()
    cp = torch.zeros(B, 16, device="cuda")
    cp[:, 9:13] = torch.tensor(M.cam[0], device="cuda")
    cp[:, 13:16] = torch.tensor(M.cam[1], device="cuda")
    mk = lambda: M.train.ConstDraws(noise=[torch.randn(B, 128, generator=torch.Generator().manual_seed(1)).cuda()],
                                    scaler=[(torch.randint(-200, 200, (B, 8), generator=torch.Generator().manual_seed(2)) / 1000.0).cuda()],
                                    alpha=[torch.rand(B, 1, generator=torch.Generator().manual_seed(3)).cuda()])

    def run():
        d, dr = _build(M, args, D), mk()
        it = (M.graphs.GraphedGanIteration(M.train.gan_iteration, args, d, ["S1"], None) if graphed else None)
        first, res = None, None
        for i in range(5):
            if graphed:
                res = it(x3, cp, x2, i == 4, M.cam, draws=dr)
            else:
                res = M.train.gan_iteration(args, d, x3, cp, x2, ["S1"], None, None, do_g_step=(i == 4), camera=M.cam, draws=dr)
            if i == 0:
                first = {ok: d[ok].flat_param.clone() for ok in ("optimizer_d3d", "optimizer_d2d")}
        torch.cuda.synchronize()
        return first, {ok: d[ok].flat_param.clone() for ok in ("optimizer_d3d", "optimizer_d2d", "optimizer_G")}, \
            {k: (v.item() if torch.is_tensor(v) and v.numel() == 1 else None) for k, v in res.items()}
    ref_first, ref_last, ref_s = run()
    real_empty, real_empty_like = torch.empty, torch.empty_like
    for poison in (float("nan"), 3.0e38):
        def fill(t):
            if t.is_cuda and t.numel():
                if t.dtype.is_floating_point:
                    t.fill_(poison)
                else:
                    t.view(torch.uint8).fill_(255)
            return t
        monkeypatch.setattr(torch, "empty", lambda *a, **k: fill(real_empty(*a, **k)))
        monkeypatch.setattr(torch, "empty_like", lambda *a, **k: fill(real_empty_like(*a, **k)))
        try:
            first, last, s = run()
        finally:
            monkeypatch.setattr(torch, "empty", real_empty)
            monkeypatch.setattr(torch, "empty_like", real_empty_like)
        for ok in ref_first:                                     # two steps of each critic: deterministic kernels, the same bits
            assert torch.equal(first[ok], ref_first[ok]), (ok, poison, (first[ok] - ref_first[ok]).abs().max().item())
        for ok, steps in (("optimizer_d3d", 10), ("optimizer_d2d", 10), ("optimizer_G", 1)):
            assert torch.isfinite(last[ok]).all(), (ok, poison)
            # (later iterations: one last-bit difference of an atomically summed gradient becomes a +-lr step of that weight)
            assert (last[ok] - ref_last[ok]).abs().max().item() <= 2.05e-4 * steps, (ok, poison, (last[ok] - ref_last[ok]).abs().max().item())
        for k, v in ref_s.items():
            if v is not None:
                assert s[k] is not None and abs(s[k] - v) <= 1e-3 * max(1.0, abs(v)), (k, poison, s[k], v)
