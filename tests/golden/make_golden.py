#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ by RUNNING THE REFERENCE.

Build-container only: imports the reference Python from /root/reference (read-only) through
tests/golden/_ref_import.py, feeds it seeded inputs and stores inputs + outputs as small .npz
fixtures.  The fixtures are data; no reference source is copied.  Re-run with
    python tests/golden/make_golden.py
The GPU box never runs this (no /root/reference there); it only reads the .npz files.

Each fixture name maps to a SURVEY.md section-8(c) capture item.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, ".."))
import _ref_import as RI                      # noqa: E402
from golden_util import seeded_state_dict, synth_fk_inputs, synth_pose16   # noqa: E402

torch.set_num_threads(1)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-34s %7.1f KB  %d arrays" % (name + ".npz", os.path.getsize(path) / 1024, len(out)))


def split_angles(a):
    """generator_angle layout -> reference kwargs (R/models_Fk_GAN/Fk_generator.py:179-186)."""
    return dict(right_leg_joints_angle=a[:, 0:5], left_leg_joints_angle=a[:, 5:10],
                body_joints_angle=a[:, 10:23], right_hand_joints_angle=a[:, 23:28],
                left_hand_joints_angle=a[:, 28:33], generator_global_rot_3d_pos_angle=a[:, 34:37])


BONE_KW = ["left_small_leg_len", "right_small_leg_len", "left_big_leg_len", "right_big_leg_len", "left_hip_len",
           "right_hip_len", "waist_len", "thorax_len", "left_shoulder_len", "right_shoulder_len",
           "left_big_arm_len", "right_big_arm_len", "left_small_arm_len", "right_small_arm_len", "neck_len"]


def ref_fk(M, args, angles, bone_len, root):
    fk = M["fkm"].Forward_Kinematics_DH_Model(args, ["S1"], None)
    kw = split_angles(angles)
    kw.update({n: bone_len[:, i] for i, n in enumerate(BONE_KW)})
    return fk.change_3d_joint_angle(root_3d_pos=root, **kw)


def main():
    M = RI.load_reference()
    fkm, gen, dis, sop, train = M["fkm"], M["gen"], M["dis"], M["sop"], M["train"]

    # ---- 1. FK, torch branch (a1-a4, a6) --------------------------------------------------------
    for N in (1, 8, 1024):
        a, bl, rt = synth_fk_inputs(N, seed=N)
        out32 = ref_fk(M, RI.make_args(batch_size=N), a, bl, rt)
        save("fk_N%d" % N, angles=a, bone_len=bl, root=rt, out32=out32)

    # single-DOF sweeps: pose i has only angle slot i non-zero (localises axis / sign errors)
    a = torch.zeros(37, 37)
    a[torch.arange(37), torch.arange(37)] = 37.0
    bl = torch.tensor([[0.45, 0.44, 0.46, 0.47, 0.13, 0.14, 0.24, 0.26, 0.15, 0.16, 0.28, 0.29, 0.25, 0.26, 0.12]]).repeat(37, 1)
    rt = torch.zeros(37, 3)
    save("fk_single_dof", angles=a, bone_len=bl, root=rt, out32=ref_fk(M, RI.make_args(batch_size=37), a, bl, rt))

    # video mode: N = B*R poses, root given as (B,R,3)
    B, R = 16, 9
    a, bl, rt = synth_fk_inputs(B * R, seed=99)
    args_v = RI.make_args(batch_size=B, single_or_multi_train_mode="multi", architecture="3,3")
    save("fk_video_B16_R9", angles=a, bone_len=bl, root=rt.view(B, R, 3),
         out32=ref_fk(M, args_v, a, bl, rt.view(B, R, 3)))

    # T-pose known answer + numpy (float64 scalar) branch on 4 random poses (a5)
    fk = fkm.Forward_Kinematics_DH_Model(RI.make_args(batch_size=1), ["S1"], None)
    tpose = fk.init_Fk_DH_angle()
    a, bl, rt = synth_fk_inputs(4, seed=5)
    outs = []
    for i in range(4):
        kw = {k: v[i].tolist() for k, v in split_angles(a.numpy()).items()}
        kw.update({n: float(bl[i, j]) for j, n in enumerate(BONE_KW)})
        outs.append(fk.change_3d_joint_angle(root_3d_pos=rt[i].numpy(), **kw).copy())
    save("fk_numpy_branch", tpose32=tpose, angles=a, bone_len=bl, root=rt, out32=np.stack(outs))

    # ---- 2. dh_matrix / rotationMatrix (a1, a2) -------------------------------------------------
    g = torch.Generator().manual_seed(2)
    al = torch.tensor([0.0, 90.0, -90.0, 0.0] * 4)
    aa = torch.rand(16, generator=g) - 0.5
    dd = torch.rand(16, generator=g) - 0.5
    th = (torch.rand(16, generator=g) - 0.5) * 720
    args16 = RI.make_args(batch_size=16)
    ang3 = (torch.rand(16, 3, generator=g) - 0.5) * 360
    save("dh_rot_16", alpha=al, a=aa, d=dd, theta=th, dh=fkm.dh_matrix(al, aa, dd, th, args16),
         ang3=ang3, rot=fkm.rotationMatrix(ang3[:, 0], ang3[:, 1], ang3[:, 2], args16))

    # ---- 4. bone vectors / KCS (a8, a11) --------------------------------------------------------
    pose = synth_pose16(256, seed=4)
    save("kcs_256", pose16=pose, bonevec=sop.Fk_get_boneVecByPose3d(pose),
         kcs30=dis.special_KCS_Input_transform(pose.clone(), "cpu"),
         kcs15=dis.video_mode_special_KCS_Input_transform(pose.clone(), "cpu"))

    # ---- 3. generators (a9, a10) ----------------------------------------------------------------
    for D, B in ((32, 64), (256, 256)):
        args = RI.make_args(batch_size=B, Gen_DenseDim=D)
        fk = fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
        G = gen.Fk_Generator(fk, args, "cpu")
        sd = seeded_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=100 + D)
        G.load_state_dict(sd)
        real = synth_pose16(B, seed=7)
        G.GAN_generator_get_bone_length(real)
        z = torch.randn(B, 128, generator=torch.Generator().manual_seed(8))
        heads = []
        hk = G.deconv_out.register_forward_hook(lambda m, i, o: heads.append(o.detach().clone()))
        torch.manual_seed(1234)
        scaler = torch.randint(-200, 200, size=(B, 8)) / 1000.0      # what forward will draw first
        torch.manual_seed(1234)
        fake = G(z)
        hk.remove()
        extra = dict(weight_seed=np.array(100 + D))
        save("gen_D%d" % D, z=z, real16=real, bone_len=G.boneLength, scaler=scaler, head=heads[0],
             angle37=G.distribute_angle[-1], fake=fake, **extra)
        # no-preAngle variant (angle = 180 * tanh), D=32 only
        if D == 32:
            args2 = RI.make_args(batch_size=B, Gen_DenseDim=D, GAN_whether_use_preAngle=False)
            G2 = gen.Fk_Generator(fkm.Forward_Kinematics_DH_Model(args2, ["S1"], None), args2, "cpu")
            G2.load_state_dict(sd)
            G2.GAN_generator_get_bone_length(real)
            torch.manual_seed(1234)
            save("gen_D32_nopre", fake=G2(z), angle37=G2.distribute_angle[-1])

    # video generator, R=9, D=32
    B, R, D = 8, 9, 32
    args = RI.make_args(batch_size=B, Gen_DenseDim=D, single_or_multi_train_mode="multi", architecture="3,3")
    fk = fkm.Forward_Kinematics_DH_Model(args, ["S1"], None)
    G = gen.Video_Fk_Generator(R, fk, args, "cpu")
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in G.state_dict().items()}, seed=77)
    G.load_state_dict(sd)
    real = synth_pose16(B * R, seed=9).view(B, R, 16, 3)
    G.GAN_generator_get_bone_length(real)
    z = torch.randn(B, 128, generator=torch.Generator().manual_seed(10))
    scaler = np.random.RandomState(args.random_seed).randint(-200, 200, size=(B, 8)) / 1000.0
    save("gen_video_D32", z=z, real16=real, bone_len=G.boneLength, scaler=scaler.astype(np.float32),
         fake=G(z), angle37=G.distribute_angle[-1], weight_seed=np.array(77))

    # ---- 5. critics (a12, a13, a17) -------------------------------------------------------------
    for D, B in ((32, 64), (256, 256)):
        args = RI.make_args(batch_size=B, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
        D3 = dis.Fk_3D_Discriminator("cpu", args)
        D2 = dis.Fk_2D_Discriminator(args, 16)
        sd3 = seeded_state_dict({k: tuple(v.shape) for k, v in D3.state_dict().items()}, seed=200 + D)
        sd2 = seeded_state_dict({k: tuple(v.shape) for k, v in D2.state_dict().items()}, seed=300 + D)
        D3.load_state_dict(sd3); D2.load_state_dict(sd2)
        x3 = synth_pose16(B, seed=11); x3 = x3 - x3[:, :1]
        x2 = (torch.rand(B, 16, 2, generator=torch.Generator().manual_seed(12)) - 0.5) * 1.6
        save("critics_D%d" % D, x3=x3, x2=x2, logit3=D3(x3), logit2=D2(x2),
             weight_seed3=np.array(200 + D), weight_seed2=np.array(300 + D))

    B, R, D = 8, 9, 32
    args = RI.make_args(batch_size=B, video_Dis_DenseDim_3D=D, video_Dis_DenseDim_2D=D,
                        single_or_multi_train_mode="multi", architecture="3,3")
    M3 = dis.Video_motion_Fk_3D_Discriminator("cpu", args, R)
    M2 = dis.Video_motion_Fk_2D_Discriminator("cpu", args, R)
    sdm3 = seeded_state_dict({k: tuple(v.shape) for k, v in M3.state_dict().items()}, seed=400)
    sdm2 = seeded_state_dict({k: tuple(v.shape) for k, v in M2.state_dict().items()}, seed=500)
    M3.load_state_dict(sdm3); M2.load_state_dict(sdm2)
    x3 = synth_pose16(B * R, seed=13); x3 = x3 - x3[:, :1]
    x2 = (torch.rand(B * R, 16, 2, generator=torch.Generator().manual_seed(14)) - 0.5) * 1.6
    save("motion_critics_D32", x3=x3, x2=x2, logit3=M3(x3), logit2=M2(x2),
         weight_seed3=np.array(400), weight_seed2=np.array(500))

    # ---- 6 + 7. gradient penalty and one critic step (a14, a15) ---------------------------------
    B, D = 64, 32
    args = RI.make_args(batch_size=B, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)
    cpu = torch.device("cpu")
    proxy = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
    proxy.device = lambda *a, **k: cpu                       # train_Fk_discriminator hard-codes "cuda"
    train.torch = proxy
    for tag, make, xr, xf in (
            ("d3", lambda: dis.Fk_3D_Discriminator("cpu", args),
             synth_pose16(B, seed=21), synth_pose16(B, seed=22)),
            ("d2", lambda: dis.Fk_2D_Discriminator(args, 16),
             (torch.rand(B, 16, 2, generator=torch.Generator().manual_seed(23)) - 0.5) * 1.6,
             (torch.rand(B, 16, 2, generator=torch.Generator().manual_seed(24)) - 0.5) * 1.6)):
        if tag == "d3":
            xr = xr - xr[:, :1]; xf = xf - xf[:, :1]
        net = make()
        sd = seeded_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()}, seed=600 + len(tag))
        net.load_state_dict(sd)
        torch.manual_seed(4321)
        alpha = torch.rand(B, 1)
        # (6) GP value and d GP / d theta
        torch.manual_seed(4321)
        net.zero_grad()
        gp = dis.calc_gradient_penalty(net, xr, xf, B, args.GAN_LAMBDA, "cpu")
        gp.backward()
        gp_grads = {("gpgrad__" + k): (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                    for k, p in net.named_parameters()}
        # (7) one full critic step
        net.load_state_dict(sd)
        opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.5, 0.9))
        summary = types.SimpleNamespace(train_discrim_iter_num=1, train_iter_num=1)
        writer = M["Writer"]()
        one = torch.tensor(1, dtype=torch.float32)
        torch.manual_seed(4321)
        W, C = train.train_Fk_discriminator(net, xr.clone(), xf.clone(), summary, writer, "Fk_" + tag, opt, args,
                                            one, one * -1)
        grads = {("grad__" + k): p.grad.clone() for k, p in net.named_parameters()}
        newp = {("new__" + k): p.detach().clone() for k, p in net.named_parameters()}
        save("critic_step_%s_D32" % tag, real=xr, fake=xf, alpha=alpha, gp=gp.detach(), Wasserstein_D=W.detach(),
             D_cost=C.detach(), weight_seed=np.array(600 + len(tag)), **gp_grads, **grads, **newp)

    # ---- N4: 'normal'-mode sampler handler_but_generater on a synthetic two-subject data set -----------------
    rs = np.random.RandomState(3)
    ds_pos = {s: {a: {c: (rs.standard_normal((20 + 5 * i, 16, 3)) * 0.3 + np.array([0.1 * i, 0.2, 0.9])).astype(np.float32)
                      for c in range(2)} for i, a in enumerate(("Walk", "Sit"))} for s in ("S1", "S5")}
    args_n = RI.make_args(batch_size=4, generator_whole_number=48, generator_choose_BoneLen=True,
                          generator_choose_root_pos=True, generator_global_rot=True, random_seed=11)
    fkn = fkm.Forward_Kinematics_DH_Model(args_n, ["S1", "S5"], None)
    fkn.dataSet_world_3d_pos = ds_pos
    fkn.dataSet_2d_pos = {s: {a: {c: np.zeros((v.shape[0], 16, 2), np.float32) for c, v in cams.items()}
                              for a, cams in acts.items()} for s, acts in ds_pos.items()}
    pos, ang, grot, blen, roots = fkn.handler_but_generater()
    flat = {"%s|%s|%d" % (s, a, c): v for s, acts in ds_pos.items() for a, cams in acts.items() for c, v in cams.items()}
    save("normal_sampler_48", pos=pos, angles=np.asarray(ang), global_rot=np.asarray(grot), bone_len=np.asarray(blen),
         root=np.asarray(roots), **{("ds__" + k): v for k, v in flat.items()})

    # ---- N3: random_bl_aug (bone-length swap) + per-sample projection ------------------------------------------
    import importlib
    os.chdir(RI.REF_ROOT)                      # random_bl_aug loads ./data_extra/... relative to the reference root
    du = importlib.import_module("function_aug.dataloader_update")
    xs = synth_pose16(96, seed=41) + torch.tensor([0.2, -0.1, 4.0])
    np.random.seed(77)
    idx = np.random.choice(5, 96)
    np.random.seed(77)
    swapped = du.random_bl_aug(xs.clone())
    os.chdir(HERE)
    camp = torch.tensor(np.random.RandomState(5).uniform(-0.1, 0.1, (96, 9)), dtype=torch.float32)
    camp[:, :2] += 2.2
    save("bl_aug_96", x=xs, idx=idx, out=swapped, cam=camp, proj=M["camera"].project_to_2d(swapped, camp))

    # ---- N1: camera / projection / flip ("next" row, pinned with the same recipe) ----------------
    cam = M["camera"]
    h36m = M["h36m"]
    ext = h36m.h36m_cameras_extrinsic_params["S1"][0]
    intr = h36m.h36m_cameras_intrinsic_params[0]
    Rq = torch.tensor(np.array(ext["orientation"]).reshape(1, 4), dtype=torch.float32)
    t = torch.tensor(np.array(ext["translation"]).reshape(1, 3) / 1000.0, dtype=torch.float32)
    res_w, res_h = float(intr["res_w"]), float(intr["res_h"])
    f = np.array(intr["focal_length"]) / res_w * 2.0
    c = cam.normalize_screen_coordinates(np.array(intr["center"]), w=res_w, h=res_h).astype("float32")
    camp = np.zeros((128, 9)); camp[:, :2] = f; camp[:, 2:4] = c
    camp[:, 4:7] = np.array(intr["radial_distortion"]); camp[:, 7:] = np.array(intr["tangential_distortion"])
    camp = torch.tensor(camp, dtype=torch.float32)
    X = synth_pose16(128, seed=31) + torch.tensor([0.3, -0.2, 0.9])
    Xc = cam.GAN_torch_world_to_camera(X, R=Rq.clone(), t=t.clone())
    x2d = cam.project_to_2d(Xc, camp)
    Xw = cam.GAN_torch_camera_to_world_batch(Xc, R=Rq.repeat(128, 1), t=t.repeat(128, 1))
    flip = X.detach().clone(); flip[:, :, 0] *= -1
    L, Rr = [4, 5, 6, 10, 11, 12], [1, 2, 3, 13, 14, 15]
    flip[:, L + Rr, :] = flip[:, Rr + L, :]
    save("camera_128", X=X, R=Rq, t=t, cam=camp, Xc=Xc, x2d=x2d, Xw=Xw, flip=flip)


if __name__ == "__main__":
    main()
