"""Drop-in for R/models_Fk_GAN/video_GAN_fun.py:79-601 (multi-frame GAN epoch, R = prod(filter widths) frames per
sample folded into the batch axis for FK / single-frame critics and into the feature axis for the motion critics).

Per iteration (reference lines in brackets): sample fakes [:193-200]; D3 step [:208]; motion-D3 step on the clip and on
its time reversal once epoch >= single_dis_warmup_epoch [:213-232]; the same on L/R-flipped copies [:237-289]; random
camera, world->camera, projection [:291-330]; D2 / motion-D2 steps, reversed, flipped [:335-418]; every 5th iteration
the G step with up to four adversarial terms [:421-566]; generated pairs appended to the device-resident buffer.
Time reversal is a frame permutation (torch.flip, data movement only)."""
import numpy as np
import torch

from .. import ops
from ..common import camera as cam
from .model_fk_gan_train import (Draws, FakePairBuffer, generator_step, pick_camera, run_critic_steps, set_grad,
                                  train_Fk_discriminator, _device, _multi_rank)
from .video_mode_operate import frames_from_args


def _rev(x, R, width):
    """time reversal of (B*R, ...) clips -> same layout"""
    return torch.flip(x.reshape(-1, R, width), dims=[1]).contiguous()


def video_gan_iteration(*a, **k):
    """_video_gan_iteration with the fused inference programs in their NaN-propagating form (see gan_iteration)"""
    with ops.nan_propagation(True):
        return _video_gan_iteration(*a, **k)


def _video_gan_iteration(args, poseFk_dict, inputs_3d, cam_param, inputs_2d, train_subjects, summary, writer=None,
                         do_g_step=False, camera=None, rng=np.random, draws=None):
    """inputs_3d (B,R,16,3) camera-space real clips, cam_param (B,>=16), inputs_2d (B,R,16,2).

    Critic-step conventions of the reference, reproduced: the 3D motion critic steps run with dis_mode='motion' (gradient
    penalty over B clips of R*48 values, :219-232), the 2D motion critic steps with the default mode (penalty over B*R
    frames of 32 values with one interpolation coefficient per FRAME, :341-346,:353-358)."""
    device = _device()
    R, B = frames_from_args(args), args.batch_size
    G, D3, D2 = poseFk_dict['model_G'], poseFk_dict['model_d3d'], poseFk_dict['model_d2d']
    M3, M2 = poseFk_dict['model_motion_d3d'], poseFk_dict['model_motion_d2d']
    oG, o3, o2 = poseFk_dict['optimizer_G'], poseFk_dict['optimizer_d3d'], poseFk_dict['optimizer_d2d']
    om3, om2 = poseFk_dict['optimizer_motion_d3d'], poseFk_dict['optimizer_motion_d2d']
    motion_on = summary.epoch >= args.single_dis_warmup_epoch
    playback, flip = bool(args.GAN_video_playback_input), bool(args.flip_GAN_model_input)
    inputs_3d, cam_param, inputs_2d = inputs_3d.to(device), cam_param.to(device), inputs_2d.to(device)
    G.GAN_generator_get_bone_length(inputs_3d)
    camR = cam_param[:, 9:13].unsqueeze(1).repeat(1, R, 1).reshape(-1, 4).contiguous()
    camT = cam_param[:, 13:16].unsqueeze(1).repeat(1, R, 1).reshape(-1, 3).contiguous()
    real = ops.center_flip(cam.GAN_torch_camera_to_world_batch(inputs_3d.reshape(-1, 16, 3), camR, camT), True, False)
    set_grad([D3, D2, M3, M2], True)
    set_grad([G], False)
    draws = draws or Draws()
    with torch.no_grad():
        noise = draws.take("noise", device)
        if noise is None:
            noise = torch.randn(B, 128, device=device)
        fake_world = G(noise, bone_len_scaler=draws.take("scaler", device)).reshape(-1, 16, 3)
    fake = ops.center_flip(fake_world, True, False)
    quat, trans, cam9 = camera if camera is not None else pick_camera(train_subjects, rng)
    pos_3d_cam, pos_2d = ops.world_to_camera_project(fake_world, quat, trans, cam9)
    real2d = inputs_2d.reshape(-1, 16, 2)

    # every critic step of the iteration, in the reference's order (the recorded interpolation coefficients follow it):
    # 3D: D3, M3, M3 reversed [:208-232], the same on L/R-flipped copies [:237-289]; 2D likewise [:335-418]
    steps, slots = [], {}

    def add(key, net, opt, name, r, f, mode):
        a = draws.take("alpha", device)
        steps.append((key, lambda: train_Fk_discriminator(net, r, f, summary, writer, name, opt, args, dis_mode=mode, alpha=a)))
        return len(steps) - 1

    def group(tag, key_s, net_s, opt_s, name_s, key_m, net_m, opt_m, names_m, r, f, width, mode_m):
        """one (single-frame critic, motion critic, motion critic on the reversed clip) triple"""
        idx = dict(s=add(key_s, net_s, opt_s, name_s, r, f, 'single'))
        if motion_on:
            rr, ff = r.reshape(-1, width), f.reshape(-1, width)
            idx['m'] = add(key_m, net_m, opt_m, names_m[0], rr, ff, mode_m)
            if playback:
                idx['b'] = add(key_m, net_m, opt_m, names_m[1], _rev(rr, R, width), _rev(ff, R, width), mode_m)
        slots[tag] = idx

    group('3', 'd3', D3, o3, 'Fk_d3d', 'm3', M3, om3, ('motion_Fk_d3d', 'back_motion_Fk_d3d'), real, fake, 48, 'motion')
    if flip:
        group('3f', 'd3', D3, o3, 'Fk_d3d', 'm3', M3, om3, ('motion_Fk_d3d', 'back_flip_motion_Fk_d3d'),
              ops.center_flip(real, False, True), ops.center_flip(fake, False, True), 48, 'motion')
    group('2', 'd2', D2, o2, 'd2d', 'm2', M2, om2, ('motion_d2d', 'back_motion_d2d'), real2d, pos_2d, 32, 'single')
    if flip:            # (the reference logs the flipped motion-2D step under 'd2d', :398-401)
        group('2f', 'd2', D2, o2, 'd2d', 'm2', M2, om2, ('d2d', 'back_flip_motion_d2d'),
              ops.center_flip(real2d, False, True), ops.center_flip(pos_2d, False, True), 32, 'single')
    res = run_critic_steps(steps, (o3, o2, om3, om2), _multi_rank())
    avg = lambda a, b: tuple((x + y) / 2 for x, y in zip(a, b))

    def motion(tag):
        i = slots[tag]
        return avg(res[i['m']], res[i['b']]) if 'b' in i else res[i['m']]

    out = {}
    for dim in ('3', '2'):
        out['d' + dim] = res[slots[dim]['s']]
        if motion_on:
            out['m' + dim] = motion(dim)
        if flip:
            out['d' + dim] = avg(out['d' + dim], res[slots[dim + 'f']['s']])
            if motion_on:
                out['m' + dim] = avg(out['m' + dim], motion(dim + 'f'))
    out['G_cost'] = None
    if do_g_step:
        if motion_on:
            critics = (D3, D2, M3, M2)
            weights = (args.GAN_3d_loss_weight, args.GAN_2d_loss_weight, args.GAN_3d_motion_loss_weight,
                       args.GAN_2d_motion_loss_weight)
        else:
            critics, weights = (D3, D2), (args.GAN_3d_loss_weight, args.GAN_2d_loss_weight)
        out['G_cost'] = generator_step(args, G, oG, critics, weights, (quat, trans, cam9), flip,
                                       draws.take("noise", device), draws.take("scaler", device), frames=R,
                                       playback=playback)
        set_grad([D3, D2, M3, M2], True)
    out.update(pos_3d_cam=pos_3d_cam.reshape(B, R, 16, 3), pos_2d=pos_2d.reshape(B, R, 16, 2), cam9=cam9)
    return out


def video_mode_GAN_solutions_FK_generator(args, poseFk_dict, data_dict, model_pos, summary, writer, train_subjects):
    """epoch loop over data_dict['target_GAN_loader'].next_epoch() -> (cam_param, inputs_3d, inputs_2d) numpy batches;
    side effect data_dict['train_fake2d3d_loader'] (device-resident FakePairBuffer of clips)."""
    for k in ('model_G', 'model_d3d', 'model_d2d', 'model_motion_d3d', 'model_motion_d2d'):
        poseFk_dict[k].train()
    if model_pos is not None:
        model_pos.train()
        set_grad([model_pos], False)
    buf = FakePairBuffer(args.batch_size)
    for cam_param, inputs_3d, inputs_2d in data_dict['target_GAN_loader'].next_epoch():
        if inputs_3d.shape[0] < args.batch_size:
            continue
        t = lambda x: x if torch.is_tensor(x) else torch.from_numpy(np.asarray(x).astype('float32'))
        r = video_gan_iteration(args, poseFk_dict, t(inputs_3d), t(cam_param), t(inputs_2d), train_subjects, summary,
                                writer, do_g_step=(summary.train_iter_num % 5 == 4))
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
