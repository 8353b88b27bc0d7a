"""Drop-in for R/models_Fk_GAN/model_fk_gan_train.py: model/optimizer factory (:97-173), the WGAN-GP critic step
train_Fk_discriminator (:177-230) and the single-frame GAN epoch GAN_solutions_FK_generator (:236-511).

Differences that are deliberate (documented in DESIGN.md):
  * device work only: no per-iteration .cpu().numpy() of the generated pairs (reference :487-489); the epoch's
    fake pairs stay on the device and are handed over as one tensor-backed loader ("next" row N2).
  * RNG draws (noise, GP alpha, bone jitter, camera choice) happen on the device / are injectable.
  * the three backward calls of the critic step are one backward of -D(real).mean() + D(fake).mean() + GP
    (same gradients; fewer passes over the weights)."""
import os

import numpy as np
import torch

from .. import autograd_ops as A
from .. import critic_step
from .. import gen_step
from .. import ops
from ..common import camera as cam
from ..common.h36m_dataset import h36m_cameras_extrinsic_params, h36m_cameras_intrinsic_params
from ..optim import FusedAdam
from .Fk_discriminator import (Fk_2D_Discriminator, Fk_3D_Discriminator, Video_motion_Fk_2D_Discriminator,
                               Video_motion_Fk_3D_Discriminator, calc_gradient_penalty)
from .Fk_generator import Fk_Generator, Video_Fk_Generator
from .video_mode_operate import frames_from_args


def set_grad(nets, requires_grad=False):
    """R/utils/utils.py:123-127."""
    for net in nets:
        if net is not None:
            for p in net.parameters():
                p.requires_grad = requires_grad


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("the DH-AUG hot path needs a GPU (no CPU fallback exists)")
    return torch.device("cuda")


def my_get_poseFk_model(args, dataset, FK_DH_Class):
    device = _device()
    num_joints = dataset.skeleton().num_joints() if dataset is not None else 16
    model_G = Fk_Generator(FK_DH_Class, args, device, INPUT_VEC_DIM=128).to(device)
    model_d3d = Fk_3D_Discriminator(device, args).to(device)
    model_d2d = Fk_2D_Discriminator(args, num_joints).to(device)
    lr = 1e-4                                    # lr_g / lr_d are parsed but ignored by the reference (:112)
    return {
        'model_G': model_G, 'model_d3d': model_d3d, 'model_d2d': model_d2d,
        'optimizer_G': FusedAdam(model_G.parameters(), lr=lr, betas=(0.5, 0.9)),
        'optimizer_d3d': FusedAdam(model_d3d.parameters(), lr=lr, betas=(0.5, 0.9)),
        'optimizer_d2d': FusedAdam(model_d2d.parameters(), lr=lr, betas=(0.5, 0.9)),
    }


def video_mode_my_get_poseFk_model(args, dataset, FK_DH_Class, video_frame_num):
    device = _device()
    model_G = Video_Fk_Generator(video_frame_num, FK_DH_Class, args, device, INPUT_VEC_DIM=128).to(device)
    model_d3d = Fk_3D_Discriminator(device, args).to(device)
    model_d2d = Fk_2D_Discriminator(args).to(device)
    model_motion_d3d = Video_motion_Fk_3D_Discriminator(device, args, video_frame_num).to(device)
    model_motion_d2d = Video_motion_Fk_2D_Discriminator(device, args, video_frame_num).to(device)
    mk = lambda m: FusedAdam(m.parameters(), lr=1e-4, betas=(0.5, 0.9))
    return {
        'model_G': model_G, 'model_d3d': model_d3d, 'model_d2d': model_d2d,
        'model_motion_d3d': model_motion_d3d, 'model_motion_d2d': model_motion_d2d,
        'optimizer_G': mk(model_G), 'optimizer_d3d': mk(model_d3d), 'optimizer_d2d': mk(model_d2d),
        'optimizer_motion_d3d': mk(model_motion_d3d), 'optimizer_motion_d2d': mk(model_motion_d2d),
    }


def traditional_solutions_FK_generator(args, FK_DH_Class, data_dict, train_subjects):
    """'normal' (non-GAN) augmentation, R/models_Fk_GAN/model_fk_gan_train.py:37-93: sample
    args.generator_whole_number poses with the FK model's handler_but_generater and view them through the 4 cameras
    of every training subject; the pairs become data_dict['train_fake2d3d_loader'] (device-resident)."""
    device = _device()
    pos32, _, _, _, _ = FK_DH_Class.handler_but_generater()
    from ..common.h36m_dataset import H36M_32_To_16_Table
    world16 = torch.as_tensor(pos32[:, H36M_32_To_16_Table, :], device=device)
    buf = FakePairBuffer(args.batch_size)
    for subject in train_subjects:
        for cam_id in range(4):
            ext = h36m_cameras_extrinsic_params[subject][cam_id]
            quat = [float(v) for v in ext['orientation']]
            trans = [float(v) / 1000.0 for v in ext['translation']]
            cam9 = cam.camera_params9(h36m_cameras_intrinsic_params[cam_id])
            c3, p2 = ops.world_to_camera_project(world16, quat, trans, cam9)
            buf.append(c3, p2, [0.0] * 9)            # the reference stores a dummy camera column here (:80-84)
    data_dict['train_fake2d3d_loader'] = buf
    return


class MeanFn(torch.autograd.Function):
    """mean of the (M,1) logits through the column-sum kernel."""

    @staticmethod
    def forward(ctx, x):
        ctx.shape = x.shape
        return (ops.colsum(x.reshape(-1, 1).contiguous(), N=1) / x.numel()).reshape(())

    @staticmethod
    def backward(ctx, g):
        n = 1
        for s in ctx.shape:
            n *= s
        return (g / n).expand(ctx.shape).contiguous()


ANALYTIC_CRITIC_STEP = os.environ.get("DHAUG_NO_ANALYTIC_STEP") is None
EXPLICIT_G_STEP = os.environ.get("DHAUG_NO_EXPLICIT_G_STEP") is None


def train_Fk_discriminator(model_dis, data_real, data_fake, summary, writer, writer_name, optimizerD, args,
                           one=None, mone=None, dis_mode='single', alpha=None):
    """One WGAN-GP critic step; returns (Wasserstein_D, D_cost) as 0-dim device tensors.

    The four critics take the explicit four-sweep schedule of critic_step.py (forward, backward chain, tangent sweep, weight
    gradients over one 3B-row batch; no autograd graph).  Anything else (other modules, unusual shapes) runs the same step
    through autograd: -D(real).mean() + D(fake).mean() + GP in one backward."""
    device = _device()
    data_real = data_real.to(device)
    data_fake = data_fake.to(device)
    real_used_num = frames_from_args(args) if dis_mode != 'motion' else 1
    rows = args.batch_size * real_used_num                       # BATCH_SIZE of calc_gradient_penalty (:210)
    if ANALYTIC_CRITIC_STEP and critic_step.supported(model_dis, optimizerD, data_real, data_fake, rows):
        if alpha is None:
            alpha = torch.rand(rows, 1, device=device)
        sc = critic_step.critic_step(model_dis, optimizerD, data_real.reshape(rows, -1), data_fake.reshape(rows, -1),
                                     alpha.to(device).reshape(-1, 1)[:rows], args.GAN_LAMBDA)     # (an injected draw may be longer)
        D_real, D_fake, Wasserstein_D, D_cost = sc[0], sc[1], sc[3], sc[4]
    else:
        model_dis.zero_grad()
        optimizerD.zero_grad()
        # real and fake rows go through the critic as ONE batch (rows are independent; the two means are taken over its
        # halves): every layer GEMM, weight-gradient GEMM and activation pass of the two passes of the reference
        # (R/models_Fk_GAN/model_fk_gan_train.py:251-262) runs once over 2B rows instead of twice over B
        # (the split is on the LOGIT count: a motion critic folds R input rows into one logit)
        if data_fake.shape == data_real.shape:
            logits = model_dis(torch.cat((data_real, data_fake), 0))
            h = logits.shape[0] // 2
            assert logits.shape[0] == 2 * h and h > 0, "critic returned %d logits for a real+fake batch" % logits.shape[0]
            D_real, D_fake = MeanFn.apply(logits[:h]), MeanFn.apply(logits[h:])
        else:
            D_real = MeanFn.apply(model_dis(data_real))
            D_fake = MeanFn.apply(model_dis(data_fake))
        gradient_penalty = calc_gradient_penalty(model_dis, data_real.detach(), data_fake.detach(), rows, args.GAN_LAMBDA,
                                                 device, alpha=alpha)
        (D_fake - D_real + gradient_penalty).backward()      # == backward(mone) + backward(one) + GP.backward()
        D_cost = (D_fake - D_real + gradient_penalty).detach()
        Wasserstein_D = (D_real - D_fake).detach()
        optimizerD.step()
    if writer is not None:
        it = getattr(summary, "train_iter_num", 0)
        writer.add_scalar('train_G_iter_PoseFk/{}_D_real'.format(writer_name), D_real.detach(), it)
        writer.add_scalar('train_G_iter_PoseFk/{}_D_fake'.format(writer_name), D_fake.detach(), it)
        writer.add_scalar('train_G_iter_PoseFk/{}_Wasserstein_D'.format(writer_name), Wasserstein_D, it)
    return Wasserstein_D, D_cost


def pick_camera(train_subjects, rng=np.random):
    """random subject + camera (R/models_Fk_GAN/model_fk_gan_train.py:344-364) -> (quat4, trans3 [m], cam9)."""
    subject = train_subjects[rng.randint(0, len(train_subjects))]
    cam_id = rng.randint(0, 4)
    ext = h36m_cameras_extrinsic_params[subject][cam_id]
    quat = [float(v) for v in ext['orientation']]
    trans = [float(v) / 1000.0 for v in ext['translation']]
    return quat, trans, cam.camera_params9(h36m_cameras_intrinsic_params[cam_id])


class FakePairBuffer:
    """The product of the augmentation epoch ("next" row N2): (pos_3d_cam (M,16,3), 2D (M,16,2), cam (M,9)) kept on
    the device; iterating yields shuffled batches like the reference's DataLoader(PoseDataSet(...)) (:504-510)."""

    def __init__(self, batch_size):
        self.batch_size = batch_size
        self.p3, self.p2, self.cam = [], [], []

    def append(self, pos_3d_cam, pos_2d, cam9):
        self.p3.append(pos_3d_cam.detach())
        self.p2.append(pos_2d.detach())
        self.cam.append(torch.as_tensor(cam9, dtype=torch.float32, device=pos_2d.device).reshape(1, 9).expand(pos_2d.shape[0], 9))

    def tensors(self):
        return torch.cat(self.p3), torch.cat(self.p2), torch.cat(self.cam)

    def __len__(self):
        n = sum(t.shape[0] for t in self.p3)
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        p3, p2, c = self.tensors()
        perm = torch.randperm(p3.shape[0], device=p3.device)
        for i in range(0, p3.shape[0], self.batch_size):
            j = perm[i:i + self.batch_size]
            yield p3[j], p2[j], ['none'] * j.shape[0], c[j]


CONCURRENT_CRITICS = os.environ.get("DHAUG_NO_CONCURRENT_CRITICS") is None
LONG_ROWS = 16384           # from this batch on one critic's kernels fill the card (run_critic_steps)
# critics side by side on disjoint CU sets (run_critic_steps).  Measured (B = 65 536): one critic's step alone on 256 / 192 / 128 / 64
# CUs takes 2.47 / 2.64 / 3.16 / 5.79 ms (3D) and 0.84 / 0.91 / 1.07 / 1.83 ms (2D) -- but the iteration with the 3D critic held to 192
# and the 2D critic to 64 takes 6.97 ms against 6.90 with each launch on the whole card (176/80: 7.43, 208/48: 7.37): the chains
# are HBM-bound together.  An option, off.
PARTITION = os.environ.get("DHAUG_PARTITION") is not None
CU_SHARES = {("d3", "d2"): tuple(int(v) for v in os.environ.get("DHAUG_CU_SHARES", "192,64").split(","))}
_SIDE = {}


def _side_streams(n):
    dev = torch.cuda.current_device()
    if len(_SIDE.get(dev, ())) < n:
        _SIDE[dev] = [torch.cuda.Stream() for _ in range(n)]
    return _SIDE[dev]


def run_critic_steps(steps, optimizers, interleave, long_rows=False):
    """steps: [(key, fn)] in the reference's order, fn() -> (Wasserstein_D, D_cost).  Returns {index: result}.
    interleave (multi-rank runs): the steps of different networks are independent given the fakes, so they are issued
    round-robin over the networks -- per-network order kept -- with the optimizers in overlap mode: while one network's
    gradient bucket is all-reduced, the next network's step computes.  Single-rank runs keep the reference's order."""
    res = {}
    if (not interleave and long_rows and critic_step.TN_SPLIT and torch.cuda.is_available()
            and torch.cuda.is_current_stream_capturing() and critic_step.capture_root()):
        # a hipGraph capture allows ONE fork level (critic_step.can_split).  At long batches the kernels of two critics
        # hardly overlap (each fills the card: 72 % of an eager iteration has one kernel resident, tools/timeline_analyze.py),
        # what pays is sweep 4's first part beside the penalty and the tangent sweep: the steps stay on the capture's own
        # stream and fork there
        for i, (_, fn) in enumerate(steps):
            res[i] = fn()
        return res
    from .. import graphs
    rec = graphs.RECORDER
    if (not interleave and rec is not None and hasattr(rec, "fork") and CONCURRENT_CRITICS and len({k for k, _ in steps}) > 1
            and torch.cuda.is_current_stream_capturing()):
        # inside a forked capture (graphs.ForkedCall): every network's steps become a graph of their own, replayed side by side
        keys = []
        for k, _ in steps:
            if k not in keys:
                keys.append(k)
        streams = _side_streams(len(keys))
        chain = lambda k: (lambda: {i: fn() for i, (kk, fn) in enumerate(steps) if kk == k})
        for part in rec.fork([(st, chain(k)) for k, st in zip(keys, streams)]):
            res.update(part)
        return res
    if not interleave and CONCURRENT_CRITICS and len({k for k, _ in steps}) > 1 and torch.cuda.is_available():
        # single rank: the steps of DIFFERENT networks run on side streams, one per network (per-network order kept), and
        # join before anything reads their results -- the launch-bound kernels of one critic's step (narrow layers,
        # elementwise passes, the tails of the persistent kernels) fill what the other's leaves idle
        main = torch.cuda.current_stream()
        streams = _side_streams(len({k for k, _ in steps}))
        keys = []
        for k, _ in steps:
            if k not in keys:
                keys.append(k)
        shares = CU_SHARES.get(tuple(keys)) if (long_rows and PARTITION and not torch.cuda.is_current_stream_capturing()) else None
        if shares is not None:
            # long batches: every persistent launch of a critic's step fills the card, so two critics' steps on two streams
            # merely take turns.  Here each network's launches are held to ITS share of the CUs (dhaug_set_workgroup_cap) and the
            # steps are issued round robin over the networks: the chains then really run side by side, and what one leaves of
            # HBM in its matrix-bound phases (the forward-with-save launch) the other uses.
            for st in streams[:len(keys)]:
                st.wait_stream(main)
            queues = {k: [(i, fn) for i, (kk, fn) in enumerate(steps) if kk == k] for k in keys}
            old_split = critic_step.TN_SPLIT
            critic_step.TN_SPLIT = False                      # (a third concurrent launch would oversubscribe the CUs)
            try:
                while any(queues.values()):
                    for k, st, share in zip(keys, streams, shares):
                        if queues[k]:
                            i, fn = queues[k].pop(0)
                            with torch.cuda.stream(st), ops.workgroup_cap(share):
                                res[i] = fn()
            finally:
                critic_step.TN_SPLIT = old_split
            for st in streams[:len(keys)]:
                main.wait_stream(st)
            return res
        for k, st in zip(keys, streams):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                for i, (kk, fn) in enumerate(steps):
                    if kk == k:
                        res[i] = fn()
        for st in streams[:len(keys)]:
            main.wait_stream(st)
        return res
    if not interleave:
        for i, (_, fn) in enumerate(steps):
            res[i] = fn()
        return res
    for o in optimizers:
        o.overlap = True
    queues = {}
    for i, (key, fn) in enumerate(steps):
        queues.setdefault(key, []).append((i, fn))
    while any(queues.values()):
        for key in list(queues):
            if queues[key]:
                i, fn = queues[key].pop(0)
                res[i] = fn()
    for o in optimizers:
        o.flush()
        o.overlap = False
    return res


def _multi_rank():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class Draws:
    """Replayable random draws of one iteration (parity tests): noise / scaler / alpha are lists consumed in the order the
    reference draws them (R/models_Fk_GAN/model_fk_gan_train.py:303, R/models_Fk_GAN/Fk_generator.py:197,
    R/models_Fk_GAN/Fk_discriminator.py:210).  An exhausted or absent list means "draw on the device"."""

    def __init__(self, noise=(), scaler=(), alpha=()):
        self.q = dict(noise=list(noise), scaler=list(scaler), alpha=list(alpha))

    def take(self, kind, device):
        q = self.q[kind]
        return q.pop(0).to(device) if q else None


class ConstDraws(Draws):
    """the same device tensors at every take: a captured hipGraph bakes its draws in, so an eager run that is to be compared
    with graph replays must see constant draws too (tests/test_gpu_graphs.py)"""

    def take(self, kind, device):
        q = self.q[kind]
        return q[0].to(device) if q else None


def generator_step(args, G, oG, critics, weights, camera, flip, noise=None, scaler=None, frames=1, playback=False):
    """The G step of both epoch loops (R/models_Fk_GAN/model_fk_gan_train.py:415-484, R/models_Fk_GAN/video_GAN_fun.py:421-566).
    critics = (D3, D2) or (D3, D2, M3, M2), weights the matching loss weights.  Flipped copies contribute their value
    but no gradient (.detach().clone() in the reference).  Returns G_cost = -gen_loss (0-dim device tensor)."""
    device = _device()
    quat, trans, cam9 = camera
    if noise is None:
        noise = torch.randn(args.batch_size, 128, device=device)
    if EXPLICIT_G_STEP and gen_step.supported(G, oG, critics):
        # explicit schedule (gen_step.py): forward with kept activations, one input-gradient chain per critic, the FK tail's
        # reverse mode, the trunk's backward chain + grouped weight-gradient launch -- no autograd graph
        with torch.no_grad():
            return gen_step.generator_step(args, G, oG, critics, weights, camera, flip, noise.to(device), scaler, frames, playback)
    set_grad(critics, False)
    set_grad([G], True)
    G.zero_grad()
    oG.zero_grad()
    fw = G(noise, bone_len_scaler=scaler).reshape(-1, 16, 3)
    _, f2d = A.W2CProjectFn.apply(fw, tuple(quat), tuple(trans), tuple(cam9))
    fc = A.center_flip(fw, True, False)
    mean = lambda net, x: MeanFn.apply(net(x))
    R = frames

    def terms(fc, f2d):
        t = [mean(critics[0], fc), mean(critics[1], f2d)]
        if len(critics) == 4:
            m3, m2 = mean(critics[2], fc.reshape(-1, 48)), mean(critics[3], f2d.reshape(-1, 32))
            if playback:
                # reference quirk: the 3D clip is viewed as (-1, R, 32) before the frame flip (:467,:521)
                m3 = (m3 + mean(critics[2], torch.flip(fc.reshape(-1, R, 32), dims=[1]).reshape(-1, 48))) / 2
                m2 = (m2 + mean(critics[3], torch.flip(f2d.reshape(-1, R, 32), dims=[1]).reshape(-1, 32))) / 2
            t += [m3, m2]
        return t

    t = terms(fc, f2d)
    if flip:
        with torch.no_grad():
            tf = terms(ops.center_flip(fc.detach(), False, True), ops.center_flip(f2d.detach(), False, True))
        t = [(a + b) / 2 for a, b in zip(t, tf)]
    gen_loss = sum(a * w for a, w in zip(t, weights))
    (-gen_loss).backward()                                               # gen_loss.backward(mone)
    oG.step()
    set_grad(critics, True)
    return (-gen_loss).detach()


def gan_iteration(*a, **k):
    """_gan_iteration with the fused inference programs it launches (the sampling pass, the G step's value-only evaluations) in
    their NaN-propagating form: a diverged generator or critic shows as NaN in the iteration's costs, as in the reference's ATen
    arithmetic, instead of being clipped to finite values by the integer-max ReLU of the inference kernels (+5 % on launches that
    are ~2 % of an iteration).  The flag is read when a kernel is LAUNCHED, so an iteration captured into a hipGraph keeps it on
    replay: set here, inside the captured region's function, eager and graphed iterations behave alike."""
    with ops.nan_propagation(True):
        return _gan_iteration(*a, **k)


def _gan_iteration(args, poseFk_dict, inputs_3d, cam_param, target_d2d, train_subjects, summary=None, writer=None,
                   do_g_step=False, camera=None, rng=np.random, draws=None):
    """One pass of R/models_Fk_GAN/model_fk_gan_train.py:281-489 on one real batch.
    inputs_3d (B,16,3) camera-space real poses, cam_param (B,>=16) with quaternion at [9:13] and translation at
    [13:16], target_d2d (B,16,2).  draws: a Draws object replaying recorded noise / jitter / GP coefficients.
    Returns dict(pos_3d_cam, pos_2d, cam9, Wasserstein_D_3D, ..., G_cost)."""
    device = _device()
    G, D3, D2 = poseFk_dict['model_G'], poseFk_dict['model_d3d'], poseFk_dict['model_d2d']
    oG, o3, o2 = poseFk_dict['optimizer_G'], poseFk_dict['optimizer_d3d'], poseFk_dict['optimizer_d2d']
    B = args.batch_size
    inputs_3d, cam_param, target_d2d = inputs_3d.to(device), cam_param.to(device), target_d2d.to(device)
    G.GAN_generator_get_bone_length(inputs_3d)
    real_world = cam.GAN_torch_camera_to_world_batch(inputs_3d.reshape(-1, 16, 3), cam_param[:, 9:13].contiguous(),
                                                     cam_param[:, 13:16].contiguous())
    real_c = ops.center_flip(real_world, True, False)                       # :295
    set_grad([D3, D2], True)
    set_grad([G], False)
    draws = draws or Draws()
    with torch.no_grad():
        noise = draws.take("noise", device)
        if noise is None:
            noise = torch.randn(B, 128, device=device)
        fake_world = G(noise, bone_len_scaler=draws.take("scaler", device)).reshape(-1, 16, 3)   # :305-310 (.data: no graph)
    fake_c = ops.center_flip(fake_world, True, False)                        # :312
    out = {}
    flip = bool(args.flip_GAN_model_input)
    quat, trans, cam9 = camera if camera is not None else pick_camera(train_subjects, rng)
    pos_3d_cam, pos_2d = ops.world_to_camera_project(fake_world, quat, trans, cam9)      # :374-376
    # the critic steps in the reference's order (D3, D3 flipped :319-341, D2, D2 flipped :387-409); their interpolation
    # coefficients are taken in that order whatever order the steps are issued in
    alphas = [draws.take("alpha", device) for _ in range(4 if flip else 2)]
    mk = lambda net, r, f, name, opt, a: (lambda: train_Fk_discriminator(net, r, f, summary, writer, name, opt, args, alpha=a))
    steps = [("d3", mk(D3, real_c, fake_c, 'Fk_d3d', o3, alphas[0]))]
    if flip:
        steps.append(("d3", mk(D3, ops.center_flip(real_c, False, True), ops.center_flip(fake_c, False, True), 'Fk_d3d', o3,
                               alphas[1])))
    steps.append(("d2", mk(D2, target_d2d, pos_2d, 'd2d', o2, alphas[2 if flip else 1])))
    if flip:
        steps.append(("d2", mk(D2, ops.center_flip(target_d2d, False, True), ops.center_flip(pos_2d, False, True), 'd2d', o2,
                               alphas[3])))
    res = run_critic_steps(steps, (o3, o2), _multi_rank(), long_rows=B >= LONG_ROWS)
    if flip:
        W3, C3 = (res[0][0] + res[1][0]) / 2, (res[0][1] + res[1][1]) / 2
        W2, C2 = (res[2][0] + res[3][0]) / 2, (res[2][1] + res[3][1]) / 2
    else:
        (W3, C3), (W2, C2) = res[0], res[1]
    G_cost = None
    if do_g_step:                                                            # :415-484
        G_cost = generator_step(args, G, oG, (D3, D2), (args.GAN_3d_loss_weight, args.GAN_2d_loss_weight),
                                (quat, trans, cam9), flip, draws.take("noise", device), draws.take("scaler", device))
    out.update(pos_3d_cam=pos_3d_cam, pos_2d=pos_2d, cam9=cam9, Wasserstein_D_3D=W3, D_cost_3D=C3,
               Wasserstein_D_2D=W2, D_cost_2D=C2, G_cost=G_cost)
    return out


def GAN_solutions_FK_generator(args, poseFk_dict, data_dict, model_pos, summary, writer, train_subjects):
    """The per-epoch loop; side effect data_dict['train_fake2d3d_loader'] (a FakePairBuffer)."""
    for k in ('model_G', 'model_d3d', 'model_d2d'):
        poseFk_dict[k].train()
    if model_pos is not None:
        model_pos.train()
        set_grad([model_pos], False)
    buf = FakePairBuffer(args.batch_size)
    for (inputs_3d, _, _, cam_param), target_d2d, _target_d3d in zip(
            data_dict['train_gt2d3d_loader'], data_dict['target_2d_loader'], data_dict['target_3d_loader']):
        if inputs_3d.shape[0] < args.batch_size:        # ragged last batch is dropped, as in the reference (:276)
            continue
        r = gan_iteration(args, poseFk_dict, inputs_3d, cam_param, target_d2d, train_subjects, summary, writer,
                          do_g_step=(summary.train_iter_num % 5 == 4))
        if hasattr(summary, "summary_train_discrim_update"):
            summary.summary_train_discrim_update()
        if r['G_cost'] is not None and hasattr(summary, "summary_train_fakepose_iter_num_update"):
            summary.summary_train_fakepose_iter_num_update()
        buf.append(r['pos_3d_cam'], r['pos_2d'], r['cam9'])
        if hasattr(summary, "summary_train_iter_num_update"):
            summary.summary_train_iter_num_update()
        else:
            summary.train_iter_num += 1
    data_dict['train_fake2d3d_loader'] = buf
    return
