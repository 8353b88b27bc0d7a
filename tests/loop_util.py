"""Replay helpers for the loop goldens (tests/golden/{gan,video}_loop_D32.npz): unpack the recorded draws in the order the
reference consumed them and drive the oracle's loop restatement with them.  Test infrastructure (imports the oracle)."""
import torch

import golden_util as GU
from oracle import dhaug_oracle as O

# writer names the reference logs a critic step under (R/models_Fk_GAN/model_fk_gan_train.py:226-229)
SCALAR_PREFIX = "scalar__train_G_iter_PoseFk|"


def keys(g, prefix):
    return [k[len(prefix):] for k in g if k.startswith(prefix)]


def scalar_series(g):
    """name -> float64 tensor of the values logged under it, in call order"""
    return {k[len(SCALAR_PREFIX):]: g[k].double() for k in g if k.startswith(SCALAR_PREFIX)}


def motion_shapes(D, R):
    s3 = {}
    for name, width in (("special_KCS", R * 15), ("diff_special_KCS", (R - 1) * 15), ("pos_3d", R * 48),
                        ("diff_pos_3d", (R - 1) * 48)):
        s3[name + "_previous.0.weight"] = (D, width); s3[name + "_previous.0.bias"] = (D,)
        for i in (1, 2, 3):
            GU._res(s3, "%s_block%d" % (name, i), D)
    s3["kcs_merge_previous.0.weight"] = (100, 4 * D); s3["kcs_merge_previous.0.bias"] = (100,)
    GU._res(s3, "kcs_merge_block1", 100)
    s3["kcs_output.weight"] = (1, 100); s3["kcs_output.bias"] = (1,)
    s2 = {}
    for name, width in (("pos_2d", R * 32), ("root_diff_2d", (R - 1) * 2)):
        s2[name + "_previous.0.weight"] = (D, width); s2[name + "_previous.0.bias"] = (D,)
        for i in (1, 2, 3):
            GU._res(s2, "%s_block%d" % (name, i), D)
    s2["merge_previous.0.weight"] = (100, 2 * D); s2["merge_previous.0.bias"] = (100,)
    GU._res(s2, "merge_block1", 100)
    s2["merge_output.weight"] = (1, 100); s2["merge_output.bias"] = (1,)
    return s3, s2


class logged_steps:
    """while active, every O.critic_step_net call also records D_real / D_fake / Wasserstein_D under the next name of
    `names` -- the series the reference's writer holds (one entry per critic step, in call order)"""

    def __init__(self, names, scal):
        self.names, self.scal = iter(names), scal

    def __enter__(self):
        self.orig = O.critic_step_net

        def logged(net, r, f, a, rows, lam=10.0):
            with torch.no_grad():
                dr, df = net(r).mean().item(), net(f).mean().item()
            res = self.orig(net, r, f, a, rows, lam)
            name = next(self.names)
            for k, v in (("D_real", dr), ("D_fake", df), ("Wasserstein_D", dr - df)):
                self.scal.setdefault("%s_%s" % (name, k), []).append(v)
            return res
        O.critic_step_net = logged
        return self

    def __exit__(self, *exc):
        O.critic_step_net = self.orig


def single_state_dicts(g, D=32):
    sG, s3, s2 = (int(v) for v in g["seeds"])
    return (GU.seeded_state_dict(GU.shapes_generator(D), sG), GU.seeded_state_dict(GU.shapes_d3(D), s3),
            GU.seeded_state_dict(GU.shapes_d2(D), s2))


def video_state_dicts(g, D=32, R=9):
    sG, s3, s2, sm3, sm2 = (int(v) for v in g["seeds"])
    m3, m2 = motion_shapes(D, R)
    return dict(G=GU.seeded_state_dict(GU.shapes_generator(D, frames=R), sG), d3=GU.seeded_state_dict(GU.shapes_d3(D), s3),
                d2=GU.seeded_state_dict(GU.shapes_d2(D), s2), m3=GU.seeded_state_dict(m3, sm3),
                m2=GU.seeded_state_dict(m2, sm2))


def compact_record(g, prefix, k):
    """the compact record (golden_util.compact) stored for tensor k under `prefix` (gan_loop_D256: weights as their change
    from the seeded initial values, gradients as they are)"""
    return {part: g["%s%s__%s" % (prefix, part, k)] for part in ("full", "sample", "proj") if "%s%s__%s" % (prefix, part, k) in g}


def compact_names(g, prefix):
    """parameter names, in state-dict order, that have a compact record under `prefix`"""
    out = []
    for key in g:
        for part in ("full__", "sample__"):
            if key.startswith(prefix + part):
                out.append(key[len(prefix + part):])
    return out


def replay_single_oracle(g, D=32):
    sdG, sd3, sd2 = single_state_dicts(g, D)
    G = O.Net(sdG, lambda z, p, bl, sc: O.generator_forward(z, p, bl, sc)[0])
    D3, D2 = O.Net(sd3, O.d3_forward), O.Net(sd2, O.d2_forward)
    iters = g["real3d"].shape[0]
    B = g["real3d"].shape[1]
    p3, p2 = [], []
    res, scal = {}, {}
    for i in range(iters):
        cam = (g["cam_quat"][i:i + 1], g["cam_trans"][i:i + 1], g["buf_cam"][i * B:(i + 1) * B].float())
        gs = dict(noise=g["noise"][iters], scaler=g["scaler"][iters]) if i == iters - 1 else None
        with logged_steps(["Fk_d3d", "Fk_d3d", "d2d", "d2d"], scal):
            r = O.gan_iteration(G, D3, D2, g["real3d"][i], g["cam_param"], g["real2d"][i], cam, g["noise"][i], g["scaler"][i],
                                [g["alpha"][4 * i + j] for j in range(4)], flip=True, g_step=gs)
        p3.append(r["pos_3d_cam"].detach()); p2.append(r["pos_2d"].detach())
        res[i] = r
    last = res[iters - 1]
    return dict(buf_p3=torch.cat(p3), buf_p2=torch.cat(p2), G=G, D3=D3, D2=D2, g_grads=last["g_grads"],
                gstep_d=last["d_states"], scalars=scal)


def video_alphas(g):
    n = int(g["alpha_list_len"])
    return [g["alpha_%03d" % i] for i in range(n)]


def replay_video_oracle(g, R=9):
    sds = video_state_dicts(g, R=R)
    nets = dict(G=O.Net(sds["G"], lambda z, p, bl, sc: O.generator_forward(z, p, bl, sc, frames=R)[0]),
                d3=O.Net(sds["d3"], O.d3_forward), d2=O.Net(sds["d2"], O.d2_forward),
                m3=O.Net(sds["m3"], lambda x, p: O.motion_d3_forward(x, p, R)),
                m2=O.Net(sds["m2"], lambda x, p: O.motion_d2_forward(x, p, R)))
    iters, B = g["real3d"].shape[0], g["real3d"].shape[1]
    al = video_alphas(g)
    per = len(al) // iters
    assert per * iters == len(al) and per == 12
    p3, p2 = [], []
    scal = {}
    names3 = ["Fk_d3d", "motion_Fk_d3d", "back_motion_Fk_d3d", "Fk_d3d", "motion_Fk_d3d", "back_flip_motion_Fk_d3d"]
    names2 = ["d2d", "motion_d2d", "back_motion_d2d", "d2d", "d2d", "back_flip_motion_d2d"]   # sic: the flipped motion-2D step logs as 'd2d'
    last = None
    for i in range(iters):
        cam9 = torch.as_tensor(g["buf_cam"][i * B:(i + 1) * B]).float().reshape(B * R, 9)
        cam = (g["cam_quat"][i:i + 1], g["cam_trans"][i:i + 1], cam9)
        gs = dict(noise=g["noise"][iters], scaler=torch.as_tensor(g["scaler"][iters])) if i == iters - 1 else None
        with logged_steps(names3 + names2, scal):
            r = O.video_gan_iteration(nets["G"], nets["d3"], nets["d2"], nets["m3"], nets["m2"], R,
                                      torch.as_tensor(g["real3d"][i]), torch.as_tensor(g["cam_param"]),
                                      torch.as_tensor(g["real2d"][i]), cam, g["noise"][i], torch.as_tensor(g["scaler"][i]),
                                      al[per * i:per * (i + 1)], g_step=gs)
        p3.append(r["pos_3d_cam"].detach()); p2.append(r["pos_2d"].detach())
        last = r
    return dict(buf_p3=torch.cat(p3), buf_p2=torch.cat(p2), nets=nets, g_grads=last["g_grads"], scalars=scal)
