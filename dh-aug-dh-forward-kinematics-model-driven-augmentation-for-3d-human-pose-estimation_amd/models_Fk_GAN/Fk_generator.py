"""Drop-in for R/models_Fk_GAN/Fk_generator.py: Fk_Generator (:79-261), Video_Fk_Generator (:264-458).

Trunk (Linear+ReLU, 3 x myResNet, Linear head) = bf16 MFMA GEMMs with fused epilogues; everything after the
head (tanh, 31->37 slot scatter, joint-limit map, bone-length jitter, FK, 32->16 gather) is ONE HIP kernel
(dhaug_gen_tail_forward) instead of ~7 100 ATen ops.  state_dict keys are the reference's
(preprocess.0, block{1,2,3}.fc{1,2}, deconv_out).

RNG: the reference draws bone_len_scaler from the global CPU generator inside forward (:197) / numpy (:383).  Here
it is drawn on the device, or injected through `bone_len_scaler=` (parity tests)."""
import os

import torch
import torch.nn as nn

from .. import autograd_ops as A
from .. import ops
from .special_operate import myResNet

GAN_global_rotation_table = {k: {"range": (-180, 180), "changeRate": (-5, 5)} for k in ("angle_x", "angle_y", "angle_z")}
_LO = [-110, -110, -110, -180, 0, -65, -65, -110, -180, 0] + [-180] * 12 + [0, 0] + [-155, -155, -100, 0, 0, -65, -65, -100, 0, 0]
_HI = [65, 65, 180, 0, 0, 110, 110, 180, 0, 0] + [180] * 12 + [0, 0] + [65, 65, 180, 180, 0, 155, 155, 180, 180, 0]
GAN_angle_range_table = {"joint%d" % (i + 1): {"range": (_LO[i], _HI[i])} for i in range(34)}


def default_precision():
    """'bf16' (one bf16 MFMA pass: the throughput arithmetic), 'f16x3' (fp16 hi+lo operands, three MFMA terms: fp32-grade
    fused forward; passes that build a graph run 'bf16x6'), 'bf16x3' / 'bf16x6' (layer-by-layer split-bf16 GEMMs)"""
    return os.environ.get("DHAUG_PRECISION", "bf16")


def graph_precision(p):
    """arithmetic of the layer-by-layer (autograd) path for a module precision"""
    return "bf16x6" if p == "f16x3" else p


def forward_precision(p):
    """arithmetic of a layer-by-layer FORWARD pass for a module precision: 'f16x3' without a graph -> the same arithmetic as layer GEMMs
    (autograd_ops.F16X3_LAYER, dhaug_gemm_f16x3: any width, the reference's default DenseDim 1000 included); with a graph -> 'bf16x6'"""
    if p == "f16x3" and not torch.is_grad_enabled():
        return A.F16X3_LAYER
    return graph_precision(p)


class _GeneratorBase(nn.Module):
    def __init__(self, frames, FK_DH_Class, args, device, INPUT_VEC_DIM):
        super().__init__()
        self.video_frame_num = frames
        self.OUTPUT_DIM = args.GAN_OUTPUT_DIM
        assert self.OUTPUT_DIM == 35, "the FK tail consumes 32 + 3 head columns (R/function_aug/config.py:85)"
        self.BATCH_SIZE = args.batch_size
        self.FK_DH_Class = FK_DH_Class
        self.train_num = 0
        self.args = args
        self.INPUT_VEC_DIM = INPUT_VEC_DIM
        self.device = device
        self.precision = default_precision()
        self.boneLength = torch.zeros((self.BATCH_SIZE, 15), dtype=torch.float32)
        self.distribute_angle = []            # last generator_angle only (the reference leaks one per call, :170)
        self.record_angles = False
        D = args.Gen_DenseDim
        self.preprocess = nn.Sequential(nn.Linear(INPUT_VEC_DIM, D), nn.ReLU(True))
        self.block1 = myResNet(D)
        self.block2 = myResNet(D)
        self.block3 = myResNet(D)
        self.deconv_out = nn.Linear(D, frames * self.OUTPUT_DIM)

    def GAN_generator_get_bone_length(self, input):
        """bone lengths of the real batch -> self.boneLength (N,15)   (:107-111 / :294-300)."""
        self.boneLength = ops.bone_length(input.reshape(-1, 16, 3))

    def _use_fused(self, x):
        """one-launch fused forward (csrc/dhaug_mlp.hip) for passes that need no autograd graph"""
        from .. import fused
        return (self.precision in fused.MODES and x.is_cuda and not (torch.is_grad_enabled() and (
            x.requires_grad or any(p.requires_grad for p in self.parameters())))
            and fused.supported(self.args.Gen_DenseDim) and self.INPUT_VEC_DIM % 64 == 0
            and self.deconv_out.weight.shape[0] <= 64)

    def trunk(self, z):
        if self._use_fused(z):
            from .. import fused
            return fused.generator_head(self, z.float(), self.precision)
        p = forward_precision(self.precision)
        lin = self.preprocess[0]
        x = A.linear(z, lin.weight, lin.bias, None, A.ACT_RELU, 0.0, p)
        x = self.block3(self.block2(self.block1(x, p), p), p)
        return A.linear(x, self.deconv_out.weight, self.deconv_out.bias, None, A.ACT_NONE, 0.0, p, out_f32=True)

    def _scaler(self, B, bone_len_scaler):
        mode = self.args.bone_len_scaler
        if bone_len_scaler is not None:
            s = bone_len_scaler.to(device=self.boneLength.device, dtype=torch.float32)
        elif mode == "different":
            # integers drawn straight into fp32 (same values as .float() of the int64 draw, one launch fewer)
            s = torch.randint(-200, 200, (B, 8), device=self.boneLength.device, dtype=torch.float32).div_(1000.0)
        elif mode == "same":            # crashes in the reference; implemented as documented
            s = torch.randint(-200, 200, (B, 1), device=self.boneLength.device, dtype=torch.float32).div_(1000.0).repeat(1, 8)
        elif mode == "":
            return None
        else:
            raise ValueError("args.bone_len_scaler")
        if self.video_frame_num > 1:    # one draw per sample, shared by its frames (:389-390)
            s = s.reshape(B, 1, 8).repeat(1, self.video_frame_num, 1).reshape(-1, 8)
        return s.contiguous()

    @staticmethod
    def _jitter_stream(head):
        """(seed, offset) of the device generator for the in-kernel Philox jitter; the offset advances by the 8 draws a
        pose takes, so consecutive calls continue the stream and torch.manual_seed reproduces it"""
        g = torch.cuda.default_generators[head.device.index if head.device.index is not None else torch.cuda.current_device()]
        off = g.get_offset()
        g.set_offset(off + 8)
        return g.initial_seed(), off

    def forward(self, input, bone_len_scaler=None):
        B, R = input.shape[0], self.video_frame_num
        head = self.trunk(input).reshape(B * R, 35)
        use_rt = getattr(self.args, "whether_use_RT", True)
        if not use_rt:                   # global rotation off: tanh^-1(0) = 0 on the three rotation columns
            head = head.clone()
            head[:, 28:31] = 0.0
        bl = self.boneLength
        if bl.shape[0] != B * R:
            raise RuntimeError("boneLength has %d rows, the batch needs %d (call GAN_generator_get_bone_length)"
                               % (bl.shape[0], B * R))
        if (bone_len_scaler is None and self.args.bone_len_scaler == "different" and R == 1 and head.is_cuda
                and not self.record_angles and not (torch.is_grad_enabled() and head.requires_grad)
                and not torch.cuda.is_current_stream_capturing()):   # (a captured graph would replay one (seed, offset))
            # sampling pass (no graph): the jitter is drawn inside the tail kernel (see sample_for_critics) -- same
            # distribution, two RNG launches fewer than torch.randint + div
            seed, off = self._jitter_stream(head)
            fake = ops.gen_tail_forward_critics(head.contiguous(), bl, None, bool(self.args.GAN_whether_use_preAngle), None,
                                                rng=(seed, off), want_critic_inputs=False)[0]
            self.train_num += 1
            return fake.reshape(B, 48)
        scaler = self._scaler(B, bone_len_scaler)
        fake = A.GenTailFn.apply(head.contiguous(), bl, scaler, bool(self.args.GAN_whether_use_preAngle))
        self.train_num += 1
        if self.record_angles:
            with torch.no_grad():
                self.distribute_angle = [ops.gen_tail_forward(head.detach(), bl, scaler,
                                                              bool(self.args.GAN_whether_use_preAngle), True)[1]]
        fake = fake.reshape(B * R, 48)
        return fake.reshape(B, R, 48) if R > 1 else fake


class Fk_Generator(_GeneratorBase):
    def sample_for_critics(self, input, camera=None, bone_len_scaler=None, inputs_bf16=False):
        """Inference-only forward (no graph) that also returns the critics' inputs from the same launch as the FK tail:
        (fake (B,16,3) world, centered (B,48), kcs bf16 (B,32), proj2d (B,16,2) | None) -- what
        R/models_Fk_GAN/model_fk_gan_train.py:305-312,374-376 computes in three more passes over the fake batch.
        inputs_bf16: centered / proj2d as bf16, the precision the bf16 critics read them in anyway (score_fake_pair gives
        bit-identical logits either way; half the bytes between the two launches)."""
        with torch.no_grad():
            B = input.shape[0]
            head = self.trunk(input).reshape(B, 35)
            if not getattr(self.args, "whether_use_RT", True):
                head = head.clone()
                head[:, 28:31] = 0.0
            bl = self.boneLength
            if bl.shape[0] != B:
                raise RuntimeError("boneLength has %d rows, the batch needs %d (call GAN_generator_get_bone_length)"
                                   % (bl.shape[0], B))
            self.train_num += 1
            pre = bool(self.args.GAN_whether_use_preAngle)
            if (bone_len_scaler is None and self.args.bone_len_scaler == "different"
                    and not torch.cuda.is_current_stream_capturing()):
                # the jitter is drawn inside the tail kernel from the device generator's (seed, offset) stream: same
                # distribution as torch.randint(-200, 200) / 1000, reproducible under torch.manual_seed
                seed, off = self._jitter_stream(head)
                return ops.gen_tail_forward_critics(head.contiguous(), bl, None, pre, camera, rng=(seed, off),
                                                    inputs_bf16=inputs_bf16)
            scaler = self._scaler(B, bone_len_scaler)
            return ops.gen_tail_forward_critics(head.contiguous(), bl, scaler, pre, camera, inputs_bf16=inputs_bf16)

    def __init__(self, FK_DH_Class, args, device, INPUT_VEC_DIM=128):
        super().__init__(1, FK_DH_Class, args, device, INPUT_VEC_DIM)


class Video_Fk_Generator(_GeneratorBase):
    def __init__(self, video_frame_num, FK_DH_Class, args, device, INPUT_VEC_DIM=128):
        super().__init__(video_frame_num, FK_DH_Class, args, device, INPUT_VEC_DIM)
