#!/usr/bin/env python3
"""Golden vectors for the LOOPS of the hot path, captured by RUNNING THE REFERENCE's own epoch functions on CPU:

  gan_loop_D32.npz         GAN_solutions_FK_generator, 5 iterations (G step on the 5th), flip on
                           (R/models_Fk_GAN/model_fk_gan_train.py:236-511) -- SURVEY.md section 8(c) item 8
  gan_loop_D256.npz        the same loop at DenseDim 256 (the width of the reference's README command and of the benchmark;
                           `python make_golden_loops.py gan_loop_D256`): weights / gradients as compact records
  video_loop_D32.npz       video_mode_GAN_solutions_FK_generator, R = 9, 5 iterations, motion critics on, playback and
                           flip on (R/models_Fk_GAN/video_GAN_fun.py:79-601) incl. the (-1, R, 32) view of quirk q6
  motion_step_m{3,2}_D32   one train_Fk_discriminator call on each motion critic in the mode the video loop uses for it
                           (M3: dis_mode='motion', GP over B clips; M2: default mode, GP over B*R frames -- :219-232,:341-346)
  motion_step_m{3,2}_D1000 the same at DenseDim 1000 (the reference's default width, README video command), B = 128 clips:
                           gradients and weight changes as compact records (`python make_golden_loops.py motion_step_D1000`)
  gan_loop_D1000.npz       the single-frame loop at DenseDim 1000 (R/function_aug/config.py:101-109: the default of every
                           *DenseDim*), B = 128: the frame critics' twenty steps and the G step at the video path's width
                           (`python make_golden_loops.py gan_loop_D1000`); compact records

Build-container only (imports /root/reference through _ref_import.py).  The reference draws its random numbers from the
global torch / numpy generators inside the loop; the draws are RECORDED here (torch.randn / rand / randint and the FK
model's RandomState are wrapped for the duration of the call) and stored with the outputs, so that the build can replay
exactly the same noise / jitter / interpolation coefficients.  `torch.device("cuda")` inside the reference functions is
answered with the CPU device; the plotting call at the end of the video epoch is skipped (no arithmetic).  Fixtures are
data (inputs + outputs); no reference source is stored.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import _ref_import as RI                      # noqa: E402
from golden_util import seeded_state_dict, synth_pose16   # noqa: E402
from make_golden import save                   # noqa: E402

torch.set_num_threads(1)


class Recorder:
    """records every torch.randn / rand / randint result (in call order) while active"""

    def __init__(self):
        self.log = []

    def __enter__(self):
        self.orig = {k: getattr(torch, k) for k in ("randn", "rand", "randint")}
        for k, f in self.orig.items():
            setattr(torch, k, self._wrap(k, f))
        return self

    def _wrap(self, kind, f):
        def g(*a, **kw):
            out = f(*a, **kw)
            self.log.append((kind, out.detach().clone()))
            return out
        return g

    def __exit__(self, *exc):
        for k, f in self.orig.items():
            setattr(torch, k, f)

    def of(self, kind):
        return [t for k, t in self.log if k == kind]


class RecordingRandomState:
    """numpy RandomState proxy that logs randint draws (Video_Fk_Generator's jitter source, R/...Fk_generator.py:383)"""

    def __init__(self, rs):
        self.rs, self.log = rs, []

    def randint(self, *a, **kw):
        out = self.rs.randint(*a, **kw)
        self.log.append(np.array(out))
        return out

    def __getattr__(self, k):
        return getattr(self.rs, k)


def cpu_torch_proxy():
    cpu = torch.device("cpu")
    proxy = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    proxy.device = lambda *a, **k: cpu
    # the recorder patches the real module; route the three RNG entry points through it at call time
    for k in ("randn", "rand", "randint"):
        setattr(proxy, k, (lambda kk: (lambda *a, **kw: getattr(torch, kk)(*a, **kw)))(k))
    return proxy


def hook_step(opt, fn):
    orig = opt.step

    def step(*a, **kw):
        fn()
        return orig(*a, **kw)
    opt.step = step


def sd_arrays(prefix, sd):
    return {prefix + k: v.detach().clone() for k, v in sd.items()}


def adam(m):
    return torch.optim.Adam(m.parameters(), lr=1e-4, betas=(0.5, 0.9))


def cam_param_rows(h36m, B, subject="S1", cam_id=1):
    ext = h36m.h36m_cameras_extrinsic_params[subject][cam_id]
    cp = np.zeros((B, 16), np.float32)
    cp[:, 9:13] = np.array(ext["orientation"], np.float32)
    cp[:, 13:16] = np.array(ext["translation"], np.float32) / 1000.0
    return cp


def chosen_cameras(h36m, subjects, cams):
    """quaternion / translation [m] of the (subject, camera) pairs the loop drew"""
    q = np.array([h36m.h36m_cameras_extrinsic_params[subjects[s]][c]["orientation"] for s, c in cams], np.float32)
    t = np.array([h36m.h36m_cameras_extrinsic_params[subjects[s]][c]["translation"] for s, c in cams], np.float64) / 1000.0
    return dict(cam_quat=q, cam_trans=t.astype(np.float32))


def scalars(writer):
    names = sorted({n for n, _, _ in writer.scalars})
    out = {}
    for n in names:
        out["scalar__" + n.replace("/", "|")] = np.array([v for m, v, _ in writer.scalars if m == n], np.float64)
    return out


def single_frame_loop(M, D=32, B=64):
    """D = 32: every tensor whole (gan_loop_D32).  D = 256 -- the width of the reference's README command and of the benchmark
    (gan_loop_D256): the networks have 0.44 / 0.88 / 0.27 M parameters, so weights and gradients are kept as compact records of
    their CHANGE from the seeded initial weights (golden_util.compact: strided samples + seeded +-1 projections; biases and
    narrow layers whole)."""
    train, gen, dis, fkm, h36m = M["train"], M["gen"], M["dis"], M["fkm"], M["h36m"]
    import utils.utils as ru
    import golden_util as GU
    ITERS = 5
    small = D == 32
    args = RI.make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D, flip_GAN_model_input=True)
    train.torch = cpu_torch_proxy()
    fk = fkm.Forward_Kinematics_DH_Model(args, ["S1", "S5"], None)
    G = gen.Fk_Generator(fk, args, "cpu")
    D3 = dis.Fk_3D_Discriminator("cpu", args)
    D2 = dis.Fk_2D_Discriminator(args, 16)
    seeds = dict(G=1100, D3=1200, D2=1300) if small else (dict(G=1150, D3=1250, D2=1350) if D == 256 else dict(G=1170, D3=1270, D2=1370))
    init = {}
    for tag, net, s in (("G", G, seeds["G"]), ("d3", D3, seeds["D3"]), ("d2", D2, seeds["D2"])):
        net.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=s))
        init[tag] = {k: v.detach().clone() for k, v in net.state_dict().items()}

    def records(prefix, tag, tensors, delta):
        """whole tensors (D = 32) or compact records of (tensor - initial weight) / of the tensor (gradients)"""
        if small:
            return {prefix + k: v.detach().clone() for k, v in tensors.items()}
        out = {}
        for i, (k, v) in enumerate(tensors.items()):
            t = (v.detach() - init[tag][k]) if delta else v.detach()
            for part, r in GU.compact(t, 100 + i).items():
                out["%s%s__%s" % (prefix, part, k)] = r
        return out
    d = dict(model_G=G, model_d3d=D3, model_d2d=D2, optimizer_G=adam(G), optimizer_d3d=adam(D3), optimizer_d2d=adam(D2))
    cp = torch.tensor(cam_param_rows(h36m, B))
    real3d = [synth_pose16(B, seed=800 + i) + torch.tensor([0.1, -0.2, 4.5]) for i in range(ITERS)]
    real2d = [(torch.rand(B, 16, 2, generator=torch.Generator().manual_seed(850 + i)) - 0.5) * 1.6 for i in range(ITERS)]
    # a ragged last batch: the reference skips it (:276)
    data = dict(train_gt2d3d_loader=[(x, None, None, cp) for x in real3d] + [(real3d[0][:5], None, None, cp[:5])],
                target_2d_loader=real2d + [real2d[0][:5]], target_3d_loader=[None] * (ITERS + 1))
    summary = ru.Summary("/tmp/dhaug_ref_summary")
    writer = M["Writer"]()
    gstep = {}

    raw = {}

    def at_g_step():                                  # (raw copies: compact() draws its signs with torch.randint, which is being recorded)
        raw["d3"] = {k: v.detach().clone() for k, v in D3.state_dict().items()}
        raw["d2"] = {k: v.detach().clone() for k, v in D2.state_dict().items()}
        raw["grad"] = {k: p.grad.detach().clone() for k, p in G.named_parameters()}
    hook_step(d["optimizer_G"], at_g_step)
    np.random.seed(4242)
    torch.manual_seed(777)
    with Recorder() as rec:
        train.GAN_solutions_FK_generator(args, d, data, torch.nn.Linear(1, 1), summary, writer, ["S1", "S5"])
    gstep.update(records("gstep_d3__", "d3", raw["d3"], True))
    gstep.update(records("gstep_d2__", "d2", raw["d2"], True))
    gstep.update(records("gstep_grad__", "G", raw["grad"], False))
    np.random.seed(4242)
    cams = np.array([[np.random.randint(0, 2), np.random.randint(0, 4)] for _ in range(ITERS)])
    ds = data["train_fake2d3d_loader"].dataset
    out = dict(real3d=torch.stack(real3d), real2d=torch.stack(real2d), cam_param=cp, cams=cams,
               noise=torch.stack(rec.of("randn")), alpha=torch.stack(rec.of("rand")),
               scaler=torch.stack(rec.of("randint")).float() / 1000.0,
               buf_p3=ds._poses_3d, buf_p2=ds._poses_2d, buf_cam=ds._cams,
               seeds=np.array([seeds["G"], seeds["D3"], seeds["D2"]]), iters=np.array(summary.train_iter_num))
    assert out["noise"].shape[0] == ITERS + 1 and out["alpha"].shape[0] == 4 * ITERS and out["scaler"].shape[0] == ITERS + 1
    out.update(chosen_cameras(h36m, ["S1", "S5"], cams))
    out.update(records("final_G__", "G", G.state_dict(), True))
    out.update(records("final_d3__", "d3", D3.state_dict(), True))
    out.update(records("final_d2__", "d2", D2.state_dict(), True))
    out.update(gstep)
    out.update(scalars(writer))
    save("gan_loop_D%d" % D, **out)


def motion_shapes(net):
    return {k: tuple(v.shape) for k, v in net.state_dict().items()}


def video_loop(M):
    train, gen, dis, fkm, h36m = M["train"], M["gen"], M["dis"], M["fkm"], M["h36m"]
    import importlib
    import utils.utils as ru
    cwd = os.getcwd()
    os.chdir(RI.REF_ROOT)
    try:
        video = importlib.import_module("models_Fk_GAN.video_GAN_fun")
    finally:
        os.chdir(cwd)
    video.torch = cpu_torch_proxy()
    video.my_visual_GAN_video = lambda *a, **k: None          # plotting at the end of the epoch: no arithmetic
    B, R, D, ITERS = 8, 9, 32, 5
    ckpt = tempfile.mkdtemp(prefix="dhaug_ref_ckpt_")
    os.makedirs(os.path.join(ckpt, "tmp"))
    args = RI.make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D, video_Dis_DenseDim_3D=D,
                        video_Dis_DenseDim_2D=D, single_or_multi_train_mode="multi", architecture="3,3",
                        single_dis_warmup_epoch=0, GAN_video_playback_input=True, flip_GAN_model_input=True,
                        checkpoint=ckpt, random_seed=3)
    fk = fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
    fk.random = RecordingRandomState(fk.random)
    G = gen.Video_Fk_Generator(R, fk, args, "cpu")
    nets = dict(G=G, d3=dis.Fk_3D_Discriminator("cpu", args), d2=dis.Fk_2D_Discriminator(args, 16),
                m3=dis.Video_motion_Fk_3D_Discriminator("cpu", args, R), m2=dis.Video_motion_Fk_2D_Discriminator("cpu", args, R))
    seeds = dict(G=2100, d3=2200, d2=2300, m3=2400, m2=2500)
    for k, net in nets.items():
        net.load_state_dict(seeded_state_dict(motion_shapes(net), seed=seeds[k]))
    d = dict(model_G=G, model_d3d=nets["d3"], model_d2d=nets["d2"], model_motion_d3d=nets["m3"], model_motion_d2d=nets["m2"],
             optimizer_G=adam(G), optimizer_d3d=adam(nets["d3"]), optimizer_d2d=adam(nets["d2"]),
             optimizer_motion_d3d=adam(nets["m3"]), optimizer_motion_d2d=adam(nets["m2"]))
    cp = cam_param_rows(h36m, B)
    real3d = [(synth_pose16(B * R, seed=900 + i) + torch.tensor([0.1, -0.2, 4.5])).reshape(B, R, 16, 3).numpy() for i in range(ITERS)]
    real2d = [((torch.rand(B * R, 16, 2, generator=torch.Generator().manual_seed(950 + i)) - 0.5) * 1.6).reshape(B, R, 16, 2).numpy()
              for i in range(ITERS)]

    class Loader:
        num_batches = ITERS

        def next_epoch(self):
            for x3, x2 in zip(real3d, real2d):
                yield cp, x3, x2

    data = dict(target_GAN_loader=Loader())
    summary = ru.Summary("/tmp/dhaug_ref_summary")
    summary.epoch = 1
    writer = M["Writer"]()
    gstep = {}

    def at_g_step():
        for k in ("d3", "d2", "m3", "m2"):
            gstep.update(sd_arrays("gstep_%s__" % k, nets[k].state_dict()))
        gstep.update({"gstep_grad__" + k: p.grad.detach().clone() for k, p in G.named_parameters()})
    hook_step(d["optimizer_G"], at_g_step)
    np.random.seed(5151)
    torch.manual_seed(888)
    with Recorder() as rec:
        video.video_mode_GAN_solutions_FK_generator(args, d, data, torch.nn.Linear(1, 1), summary, writer, ["S1"])
    np.random.seed(5151)
    cams = np.array([[np.random.randint(0, 1), np.random.randint(0, 4)] for _ in range(ITERS)])
    ds = data["train_fake2d3d_loader"].dataset
    out = dict(real3d=np.stack(real3d), real2d=np.stack(real2d), cam_param=cp, cams=cams,
               noise=torch.stack(rec.of("randn")), alpha_list_len=np.array(len(rec.of("rand"))),
               scaler=np.stack(fk.random.log).astype(np.float32) / 1000.0,
               buf_p3=ds._poses_3d, buf_p2=ds._poses_2d, buf_cam=ds._cams,
               seeds=np.array([seeds[k] for k in ("G", "d3", "d2", "m3", "m2")]), iters=np.array(summary.train_iter_num))
    # GP interpolation coefficients come in two shapes ((B*R,1) for the single-frame critics and the 2D motion critic,
    # (B,1) for the 3D motion critic): stored in call order, one array each
    for i, a in enumerate(rec.of("rand")):
        out["alpha_%03d" % i] = a
    assert out["noise"].shape[0] == ITERS + 1 and out["scaler"].shape[0] == ITERS + 1 and len(rec.of("randint")) == 0
    out.update(chosen_cameras(h36m, ["S1"], cams))
    for k, net in nets.items():
        out.update(sd_arrays("final_%s__" % k, net.state_dict()))
    out.update(gstep)
    out.update(scalars(writer))
    save("video_loop_D32", **out)

    # ---- one isolated step of each motion critic, in the mode the loop uses for it -------------------------------
    for tag, mode in (("m3", "motion"), ("m2", "single")):
        net = (dis.Video_motion_Fk_3D_Discriminator if tag == "m3" else dis.Video_motion_Fk_2D_Discriminator)("cpu", args, R)
        sd = seeded_state_dict(motion_shapes(net), seed=2600 + len(mode))
        net.load_state_dict(sd)
        if tag == "m3":
            xr = synth_pose16(B * R, seed=61); xr = (xr - xr[:, :1]).reshape(B * R, 48)
            xf = synth_pose16(B * R, seed=62); xf = (xf - xf[:, :1]).reshape(B * R, 48)
        else:
            xr = ((torch.rand(B * R, 16, 2, generator=torch.Generator().manual_seed(63)) - 0.5) * 1.6)
            xf = ((torch.rand(B * R, 16, 2, generator=torch.Generator().manual_seed(64)) - 0.5) * 1.6)
        opt = adam(net)
        one = torch.tensor(1, dtype=torch.float32)
        grads = {}
        hook_step(opt, lambda: grads.update({"grad__" + k: p.grad.detach().clone() for k, p in net.named_parameters()}))
        torch.manual_seed(999)
        with Recorder() as rec:
            kw = dict(dis_mode="motion") if mode == "motion" else {}
            W, C = train.train_Fk_discriminator(net, xr.clone(), xf.clone(), summary, M["Writer"](), "motion_" + tag, opt,
                                                args, one, one * -1, **kw)
        newp = {"new__" + k: p.detach().clone() for k, p in net.named_parameters()}
        save("motion_step_%s_D32" % tag, real=xr, fake=xf, alpha=rec.of("rand")[0], Wasserstein_D=W.detach(),
             D_cost=C.detach(), weight_seed=np.array(2600 + len(mode)), **grads, **newp)


def video_D1000(M):
    """BASELINE configs[4] widths (DenseDim 1000 everywhere, R = 9): forward outputs of the video generator and of the four
    critics on a small batch.  Weights come from seeded_state_dict (seeds stored), so only inputs and outputs are kept."""
    gen, dis, fkm = M["gen"], M["dis"], M["fkm"]
    B, R, D = 4, 9, 1000
    args = RI.make_args(batch_size=B, Gen_DenseDim=D, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D, video_Dis_DenseDim_3D=D,
                        video_Dis_DenseDim_2D=D, single_or_multi_train_mode="multi", architecture="3,3", random_seed=5)
    fk = fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
    fk.random = RecordingRandomState(fk.random)
    nets = dict(G=gen.Video_Fk_Generator(R, fk, args, "cpu"), d3=dis.Fk_3D_Discriminator("cpu", args),
                d2=dis.Fk_2D_Discriminator(args, 16), m3=dis.Video_motion_Fk_3D_Discriminator("cpu", args, R),
                m2=dis.Video_motion_Fk_2D_Discriminator("cpu", args, R))
    seeds = dict(G=3100, d3=3200, d2=3300, m3=3400, m2=3500)
    for k, net in nets.items():
        net.load_state_dict(seeded_state_dict(motion_shapes(net), seed=seeds[k]))
    real = synth_pose16(B * R, seed=71).view(B, R, 16, 3)
    nets["G"].GAN_generator_get_bone_length(real)
    z = torch.randn(B, 128, generator=torch.Generator().manual_seed(72))
    with torch.no_grad():
        fake = nets["G"](z)
        x3 = synth_pose16(B * R, seed=73); x3 = x3 - x3[:, :1]
        x2 = (torch.rand(B * R, 16, 2, generator=torch.Generator().manual_seed(74)) - 0.5) * 1.6
        out = dict(z=z, real16=real, bone_len=nets["G"].boneLength, scaler=np.stack(fk.random.log)[0].astype(np.float32) / 1000.0,
                   fake=fake, x3=x3, x2=x2, logit_d3=nets["d3"](x3), logit_d2=nets["d2"](x2), logit_m3=nets["m3"](x3),
                   logit_m2=nets["m2"](x2), seeds=np.array([seeds[k] for k in ("G", "d3", "d2", "m3", "m2")]))
    save("video_D1000", **out)


def motion_steps_D1000(M):
    """One train_Fk_discriminator call on each MOTION critic at the width of the reference's README video command (DenseDim 1000:
    R/function_aug/config.py:101-109, 25.5 M / 12.5 M parameters), R = 9, B = 128 clips (round 6; 16 before: too few rows for an element-wise bf16 comparison), in the mode the video loop uses for it
    (M3: dis_mode='motion', penalty over B clips; M2: default mode, penalty over B*R frames): the training kernels of the video
    path (1000-wide NT layers, wide grouped contractions, adam_nt) are pinned to the reference above DenseDim 32.  Gradients and
    the weights' CHANGE are kept as compact records (golden_util.compact)."""
    import golden_util as GU
    train, dis = M["train"], M["dis"]
    import utils.utils as ru
    train.torch = cpu_torch_proxy()
    B, R, D = 128, 9, 1000
    args = RI.make_args(batch_size=B, video_Dis_DenseDim_3D=D, video_Dis_DenseDim_2D=D, single_or_multi_train_mode="multi",
                        architecture="3,3", random_seed=7)
    summary = ru.Summary("/tmp/dhaug_ref_summary")
    for tag, mode, seed in (("m3", "motion", 4600), ("m2", "single", 4700)):
        net = (dis.Video_motion_Fk_3D_Discriminator if tag == "m3" else dis.Video_motion_Fk_2D_Discriminator)("cpu", args, R)
        sd = seeded_state_dict(motion_shapes(net), seed=seed)
        net.load_state_dict(sd)
        if tag == "m3":
            xr = synth_pose16(B * R, seed=161); xr = (xr - xr[:, :1]).reshape(B * R, 48)
            xf = synth_pose16(B * R, seed=162); xf = (xf - xf[:, :1]).reshape(B * R, 48)
        else:
            xr = ((torch.rand(B * R, 16, 2, generator=torch.Generator().manual_seed(163)) - 0.5) * 1.6)
            xf = ((torch.rand(B * R, 16, 2, generator=torch.Generator().manual_seed(164)) - 0.5) * 1.6)
        opt = adam(net)
        one = torch.tensor(1, dtype=torch.float32)
        grads = {}
        hook_step(opt, lambda: grads.update({k: p.grad.detach().clone() for k, p in net.named_parameters()}))
        torch.manual_seed(1999)
        with Recorder() as rec:
            kw = dict(dis_mode="motion") if mode == "motion" else {}
            W, C = train.train_Fk_discriminator(net, xr.clone(), xf.clone(), summary, M["Writer"](), "motion_" + tag, opt,
                                                args, one, one * -1, **kw)
        out = dict(real=xr, fake=xf, alpha=rec.of("rand")[0], Wasserstein_D=W.detach(), D_cost=C.detach(), weight_seed=np.array(seed))
        for i, (k, p) in enumerate(net.named_parameters()):
            for kind, t in (("grad", grads[k]), ("delta", p.detach() - sd[k])):
                for part, v in GU.compact(t, 100 + i).items():
                    out["%s__%s__%s" % (kind, part, k)] = v
        save("motion_step_%s_D1000" % tag, **out)


def main():
    M = RI.load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "motion_step_D1000":
        torch.set_num_threads(8)
        motion_steps_D1000(M)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "video_D1000":
        video_D1000(M)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "gan_loop_D256":
        torch.set_num_threads(8)
        single_frame_loop(M, 256)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "gan_loop_D1000":
        torch.set_num_threads(8)
        single_frame_loop(M, 1000, B=128)
        return
    single_frame_loop(M)
    video_loop(M)
    video_D1000(M)


if __name__ == "__main__":
    main()
