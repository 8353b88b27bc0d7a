"""Drop-in for R/models_Fk_GAN/Fk_discriminator.py: KCS transforms, Fk_3D_Discriminator (:149-201),
calc_gradient_penalty (:205-231), Fk_2D_Discriminator (:236-266) and the two video motion critics (:381-587).

All Linear layers are bf16 MFMA GEMMs with fused bias / ReLU|LeakyReLU / residual; KCS features are one HIP kernel
(value, VJP and JVP) -- the critics are differentiable to second order so that calc_gradient_penalty works exactly
as in the reference (autograd.grad(..., create_graph=True))."""
import torch
import torch.autograd as autograd
import torch.nn as nn

from .. import autograd_ops as A
from .. import fused
from .Fk_generator import default_precision, forward_precision, graph_precision
from .special_operate import myResNet


def special_KCS_Input_transform(pos_16_3d, device=None):
    """(N,16,3)|(N,48) -> (N,30): 15 adjacent-bone cosines + 15 bone lengths."""
    return A.KcsFn.apply(pos_16_3d.reshape(-1, 48), True)


def video_mode_special_KCS_Input_transform(pos_16_3d, device=None):
    """(N,16,3)|(N,48) -> (N,15): cosines only."""
    return A.KcsFn.apply(pos_16_3d.reshape(-1, 48), False)


def _no_graph(module, x):
    return not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters())))


def _branch(x, first, blocks, prec, out=None):
    """Linear + ReLU and the myResNet blocks of one input branch.  out: the branch's column block of the concatenation buffer (_cat_buffer):
    the last layer writes its result there"""
    h = A.linear(x, first.weight, first.bias, None, A.ACT_RELU, 0.0, prec)
    for b in (blocks if out is None else blocks[:-1]):
        h = b(h, prec)
    if out is not None:
        b = blocks[-1]
        h1 = A.linear(h, b.fc1.weight, b.fc1.bias, None, A.ACT_RELU, 0.0, prec)
        h = A.linear(h1, b.fc2.weight, b.fc2.bias, h, A.ACT_RELU, 0.0, prec, out=out)
    return h


def _cat_buffer(n, Dw, x, prec):
    """Passes without a graph in bf16 or as f16x3 layer GEMMs (the layer-by-layer forward at widths the fused programs do not cover: the
    reference's default DenseDim 1000): the concatenation of n branch results as ONE buffer the branches' last layers write their column blocks into -- as the
    training steps do -- instead of a torch.cat copy (262 MB and 224 us of a 4 ms forward at B = 65 536).  Returns the (M, ceil16(n Dw))
    buffer and the n (M, Dw) views, or None where the blocks would not be 16-byte aligned / a graph is being built."""
    if torch.is_grad_enabled() or Dw % 8 != 0 or not x.is_cuda:
        return None
    M, total = x.shape[0], n * Dw
    if prec == A.F16X3_LAYER and A.F16X3_PLANES and A.ops.gemm_f16x3_ok(Dw, 3 * A.ceil16(Dw)):
        buf = torch.empty((M, total), dtype=torch.float32, device=x.device)      # (fp32 activations: the layers' c_f32 takes a row pitch)
    elif prec == "bf16":
        buf = torch.empty((M, A.ceil16(total)), dtype=torch.bfloat16, device=x.device)
        if buf.shape[1] > total:
            buf[:, total:].zero_()
    else:
        return None
    return buf, [buf[:, i * Dw:(i + 1) * Dw] for i in range(n)]


def _cat(outs, widths, prec):
    """concatenation of branch outputs along the feature axis.  bf16 activations are zero-padded to a multiple of 16 columns
    (1000 -> 1008): the pads are cut out, and the result is padded once (the consumer reads ceil16(sum) columns)."""
    if prec != "bf16":
        return torch.cat(outs, dim=-1)
    total = sum(widths)
    if all(w % 16 == 0 for w in widths):
        return torch.cat(outs, dim=-1)
    parts = [o[:, :w] for o, w in zip(outs, widths)]
    pad = A.ceil16(total) - total
    if pad:
        parts.append(torch.zeros((outs[0].shape[0], pad), dtype=outs[0].dtype, device=outs[0].device))
    return torch.cat(parts, dim=-1)


def _frame_diff(x, frames, width):
    x = x.reshape(-1, frames, width)
    return (x[:, 1:] - x[:, :-1]).reshape(-1, (frames - 1) * width)


class Fk_3D_Discriminator(nn.Module):
    def __init__(self, device, args):
        super().__init__()
        self.device, self.args = device, args
        self.precision = default_precision()
        D = args.Dis_DenseDim_3D
        self.previous = nn.Sequential(nn.Linear(16 * 3, D), nn.ReLU(True))
        self.block1, self.block2, self.block3 = myResNet(D), myResNet(D), myResNet(D)
        self.special_KCS_previous = nn.Sequential(nn.Linear(30, D), nn.ReLU(True))
        self.special_KCS_block1, self.special_KCS_block2, self.special_KCS_block3 = myResNet(D), myResNet(D), myResNet(D)
        self.merge_previous = nn.Sequential(nn.Linear(D + D, 100), nn.ReLU(True))
        self.merge_block1 = myResNet(100)
        self.output = nn.Linear(100, 1)

    def forward(self, input, center=False, kcs=None):
        """center=True scores `input - input[:, :1]` (what every caller of the reference feeds the critic) without a
        separate centring pass where the fused path applies"""
        p = self.precision
        x = input.reshape(-1, 48)
        if p in fused.MODES and x.is_cuda and _no_graph(self, x) and fused.supported(self.args.Dis_DenseDim_3D):
            return fused.critic3d(self, x.float(), center, kcs, p)  # one launch, activations stay in LDS
        p = forward_precision(p)
        if center:
            x = A.center_flip(x.reshape(-1, 16, 3), True, False).reshape(-1, 48)
        Dw = self.args.Dis_DenseDim_3D
        cb = _cat_buffer(2, Dw, x, p)
        k = _branch(A.KcsFn.apply(x, True), self.special_KCS_previous[0],
                    (self.special_KCS_block1, self.special_KCS_block2, self.special_KCS_block3), p, out=None if cb is None else cb[1][0])
        q = _branch(x, self.previous[0], (self.block1, self.block2, self.block3), p, out=None if cb is None else cb[1][1])
        m = _cat((k, q), (Dw, Dw), p) if cb is None else cb[0]
        m = A.linear(m, self.merge_previous[0].weight, self.merge_previous[0].bias, None, A.ACT_RELU, 0.0, p)
        m = self.merge_block1(m, p)
        return A.linear(m, self.output.weight, self.output.bias, None, A.ACT_NONE, 0.0, p, out_f32=True)


def calc_gradient_penalty(netD, real_data, fake_data, BATCH_SIZE, LAMBDA, device, alpha=None):
    """LAMBDA * mean((||dD/dx_hat||_2 - 1)^2).  `alpha` (B,1) may be injected (parity tests); otherwise it is drawn
    on the device (the reference draws it from the global CPU generator, :210)."""
    real_data = real_data.reshape(BATCH_SIZE, -1)
    fake_data = fake_data.reshape(BATCH_SIZE, -1)
    if alpha is None:
        alpha = torch.rand(BATCH_SIZE, 1, device=real_data.device)
    alpha = alpha.to(real_data.device).reshape(BATCH_SIZE, 1)
    interpolates = (alpha * real_data + ((1 - alpha) * fake_data)).detach().requires_grad_(True)
    disc_interpolates = netD(interpolates)
    gradients = autograd.grad(outputs=disc_interpolates, inputs=interpolates,
                              grad_outputs=torch.ones_like(disc_interpolates),
                              create_graph=True, retain_graph=True, only_inputs=True)[0]
    return ((gradients.norm(2, dim=1) - 1) ** 2).mean() * LAMBDA


class Fk_2D_Discriminator(nn.Module):
    def __init__(self, args, num_joints=16):
        super().__init__()
        self.args = args
        self.precision = default_precision()
        D = args.Dis_DenseDim_2D
        self.pose_layer_1 = nn.Linear(num_joints * 2, D)
        self.pose_layer_2 = nn.Linear(D, D)
        self.pose_layer_3 = nn.Linear(D, D)
        self.pose_layer_4 = nn.Linear(D, D)
        self.layer_last = nn.Linear(D, D)
        self.layer_pred = nn.Linear(D, 1)
        self.relu = nn.LeakyReLU()
        self.slope = 0.01

    def forward(self, x):
        p, s, L = self.precision, self.slope, A.ACT_LRELU
        x = x.reshape(-1, 32)
        if p in fused.MODES and x.is_cuda and _no_graph(self, x) and fused.supported(self.args.Dis_DenseDim_2D):
            return fused.critic2d(self, x.float(), p)
        p = forward_precision(p)
        d1 = A.linear(x, self.pose_layer_1.weight, self.pose_layer_1.bias, None, L, s, p)
        d2 = A.linear(d1, self.pose_layer_2.weight, self.pose_layer_2.bias, None, L, s, p)
        d3 = A.linear(d2, self.pose_layer_3.weight, self.pose_layer_3.bias, d1, L, s, p)
        d4 = A.linear(d3, self.pose_layer_4.weight, self.pose_layer_4.bias, None, A.ACT_NONE, 0.0, p)
        dl = A.linear(d4, self.layer_last.weight, self.layer_last.bias, None, L, s, p)
        return A.linear(dl, self.layer_pred.weight, self.layer_pred.bias, None, A.ACT_NONE, 0.0, p, out_f32=True)


class Video_motion_Fk_3D_Discriminator(nn.Module):
    def __init__(self, device, args, video_frame_num):
        super().__init__()
        self.video_frame_num, self.device, self.args = video_frame_num, device, args
        self.precision = default_precision()
        D, R = args.video_Dis_DenseDim_3D, video_frame_num
        for name, width in (("special_KCS", R * 15), ("diff_special_KCS", (R - 1) * 15), ("pos_3d", R * 48),
                            ("diff_pos_3d", (R - 1) * 48)):
            setattr(self, name + "_previous", nn.Sequential(nn.Linear(width, D), nn.ReLU(True)))
            for i in (1, 2, 3):
                setattr(self, "%s_block%d" % (name, i), myResNet(D))
        self.use_pos = bool(args.motion_Dis_whether_use_3dPos_branch)
        self.use_diff = bool(args.motion_Dis_whether_use_3dDiff_branch)
        self.branch_num = 2 + int(self.use_pos) + int(self.use_diff)
        self.kcs_merge_previous = nn.Sequential(nn.Linear(D * self.branch_num, 100), nn.ReLU(True))
        self.kcs_merge_block1 = myResNet(100)
        self.kcs_output = nn.Linear(100, 1)

    def _b(self, x, name, out=None):
        return _branch(x.contiguous(), getattr(self, name + "_previous")[0],
                       [getattr(self, "%s_block%d" % (name, i)) for i in (1, 2, 3)], forward_precision(self.precision), out=out)

    def forward(self, input):
        R, p = self.video_frame_num, forward_precision(self.precision)
        x = input.reshape(-1, 48)
        kc = A.KcsFn.apply(x, False).reshape(-1, R * 15)
        Dw = self.args.video_Dis_DenseDim_3D
        cb = _cat_buffer(self.branch_num, Dw, kc, p)               # (passes without a graph in bf16: no torch.cat copy)
        view = (lambda i: None) if cb is None else (lambda i: cb[1][i])
        outs = [self._b(kc, "special_KCS", view(0)), self._b(_frame_diff(kc, R, 15), "diff_special_KCS", view(1))]
        if self.use_pos:
            outs.append(self._b(x.reshape(-1, R * 48), "pos_3d", view(len(outs))))
        if self.use_diff:
            outs.append(self._b(_frame_diff(x, R, 48), "diff_pos_3d", view(len(outs))))
        m = _cat(outs, [Dw] * len(outs), p) if cb is None else cb[0]
        m = A.linear(m, self.kcs_merge_previous[0].weight, self.kcs_merge_previous[0].bias, None, A.ACT_RELU, 0.0, p)
        m = self.kcs_merge_block1(m, p)
        return A.linear(m, self.kcs_output.weight, self.kcs_output.bias, None, A.ACT_NONE, 0.0, p, out_f32=True)


class Video_motion_Fk_2D_Discriminator(nn.Module):
    def __init__(self, device, args, video_frame_num):
        super().__init__()
        self.video_frame_num, self.device, self.args = video_frame_num, device, args
        self.precision = default_precision()
        D, R = args.video_Dis_DenseDim_2D, video_frame_num
        for name, width in (("pos_2d", R * 32), ("root_diff_2d", (R - 1) * 2)):
            setattr(self, name + "_previous", nn.Sequential(nn.Linear(width, D), nn.ReLU(True)))
            for i in (1, 2, 3):
                setattr(self, "%s_block%d" % (name, i), myResNet(D))
        self.merge_previous = nn.Sequential(nn.Linear(D + D, 100), nn.ReLU(True))
        self.merge_block1 = myResNet(100)
        self.merge_output = nn.Linear(100, 1)

    def _b(self, x, name, out=None):
        return _branch(x.contiguous(), getattr(self, name + "_previous")[0],
                       [getattr(self, "%s_block%d" % (name, i)) for i in (1, 2, 3)], forward_precision(self.precision), out=out)

    def forward(self, input):
        R, p = self.video_frame_num, forward_precision(self.precision)
        x = input.reshape(-1, 32)
        Dw = self.args.video_Dis_DenseDim_2D
        xa = x.reshape(-1, R * 32)
        cb = _cat_buffer(2, Dw, xa, p)                             # (passes without a graph in bf16: no torch.cat copy)
        a = self._b(xa, "pos_2d", None if cb is None else cb[1][0])
        b = self._b(_frame_diff(x.reshape(-1, 16, 2)[:, 0, :], R, 2), "root_diff_2d", None if cb is None else cb[1][1])
        m = _cat((a, b), (Dw, Dw), p) if cb is None else cb[0]
        m = A.linear(m, self.merge_previous[0].weight, self.merge_previous[0].bias, None, A.ACT_RELU, 0.0, p)
        m = self.merge_block1(m, p)
        return A.linear(m, self.merge_output.weight, self.merge_output.bias, None, A.ACT_NONE, 0.0, p, out_f32=True)


def score_fake_pair(d3d, d2d, pose_centered, kcs, proj2d):
    """Both critics on one batch of generated poses (the evaluation / sampling passes: no graph is built): one launch
    where the fused kernel applies (dhaug_mlp_forward over the 3D critic's program followed by the 2D critic's),
    otherwise the two forward() calls of the reference (R/models_Fk_GAN/model_fk_gan_train.py:463-468).
    pose_centered (N,16,3)|(N,48) root-relative, kcs (N,32) bf16 or None, proj2d (N,16,2) -> (logit3d, logit2d)"""
    x3 = pose_centered.reshape(-1, 48)
    both = (d3d.precision in fused.MODES and d2d.precision == d3d.precision and x3.is_cuda and kcs is not None
            and _no_graph(d3d, x3) and _no_graph(d2d, proj2d)
            and fused.supported(d3d.args.Dis_DenseDim_3D, d2d.args.Dis_DenseDim_2D))
    if both and d3d.precision == "f16x3":                                  # fp32-grade: fp32 inputs, fp32 KCS features
        from .. import ops
        x3 = x3.float()
        # (fp32 features handed in are used; the bf16 operand of the bf16 path is not: the launch computes them from x3)
        return fused.critics(d3d, d2d, x3, kcs if kcs.dtype == torch.float32 else None, proj2d.float(), "f16x3")
    if both:
        keep = lambda t: t if t.dtype == torch.bfloat16 else t.float()      # bf16 inputs are loaded as they are
        return fused.critics(d3d, d2d, keep(x3), kcs, keep(proj2d))
    return d3d(pose_centered.float(), kcs=kcs), d2d(proj2d.float())

