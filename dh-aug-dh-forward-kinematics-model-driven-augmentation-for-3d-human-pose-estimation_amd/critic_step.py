"""One WGAN-GP critic step (train_Fk_discriminator, R/models_Fk_GAN/model_fk_gan_train.py:177-230, with
calc_gradient_penalty, R/models_Fk_GAN/Fk_discriminator.py:205-231) for the two single-frame critics as an explicit
schedule of kernels -- no autograd graph, no double backward.

The reference runs D(real), D(fake) and D(x_hat) as three passes, back-propagates each and differentiates the penalty
through a second-order graph.  The critics are piecewise linear in their hidden layers (ReLU / LeakyReLU MLPs; the only
smooth non-linearity, the KCS transform, sits at the input), so the whole step is four sweeps over ONE batch of 3B rows
[real; fake; x_hat = a real + (1 - a) fake]:

  1. forward            y_l = act(W_l y_{l-1} + b_l [+ skip])                      rows [0, 3B)
  2. backward chain     gz_l = (gz_{l+1} W_{l+1} [+ skip]) * act'(y_l)             rows [0, 3B), seeded with the logit
                        cotangents (-1/B, +1/B, 1): on the x_hat rows it ends in g_b = dD/dx_hat_b
  3. tangent forward    u_l = (W_l u_{l-1} [+ skip]) * act'(y_l)                   rows [2B, 3B), seeded with the penalty's
                        cotangent v_b = (2 lambda / B) (||g_b|| - 1) / ||g_b|| * g_b
  4. weight gradients   dW_l = gz_l[real,fake]^T y_{l-1}[real,fake]  +  gz_l[x_hat]^T u_{l-1}      (one TN GEMM each)
                        db_l = column sums of gz_l over the real / fake rows (the penalty has no bias gradient)

(4) is d/dW of  -mean D(real) + mean D(fake) + lambda mean((||g|| - 1)^2): the penalty depends on W_l only through g, and
<v, g> = <u_{l-1}, W_l^T gz_l> layer by layer.  The masks act'(y) are constants almost everywhere, exactly as in the
reference's autograd (relu'' = 0).  Same gradients as the composite path (tests/test_gpu_critic_step.py: the goldens
captured from the reference in the fp32-grade arithmetic, and the autograd path in bf16); per 256-wide layer 5 launches
instead of ~13, nothing but HIP kernels of libdhaug.so on the path.

`prec`: 'bf16' (the throughput arithmetic) or 'bf16x3' / 'bf16x6' (split operands, fp32 activations: parity tests)."""
import os

import torch

from . import autograd_ops as A
from . import ops

# sweep 1 (forward) of the single-frame critics as ONE fused launch that also saves every layer's output
# (fused.critic3d_forward_save / critic2d_forward_save) instead of one GEMM launch per layer; bf16 arithmetic only
FUSED_STEP_FORWARD = os.environ.get("DHAUG_NO_FUSED_STEP_FORWARD") is None
# sweep 4: the weight / bias gradients of all layers up to 256 wide in ONE grouped launch (ops.gemm_tn_group) instead of one
# contraction launch per layer
TN_GROUP = os.environ.get("DHAUG_NO_TN_GROUP") is None
# the forward-with-save launch writes the block layers' images for the real / fake rows only (the interpolated rows are read
# through their sign bits alone, and receive the tangents): a third fewer stores in the launch that pays most for them
SKIP_XHAT_SAVES = os.environ.get("DHAUG_SAVE_ALL_ROWS") is None
RANK1 = os.environ.get("DHAUG_NO_RANK1") is None
# sweep 4 in two parts: the real / fake rows' contractions are launched on a side stream right behind the backward chain and
# run (HBM-bound) beside the penalty and the launch-bound tangent sweep; the interpolated rows' part follows the tangents
TN_SPLIT = os.environ.get("DHAUG_NO_TN_SPLIT") is None
_TN_SIDE = {}
# workgroups of sweep 4's side-stream part: it runs beside the penalty and the tangent sweep.  Round 4 measured an optimum near 100
# (iteration at B = 65 536: 256 -> 7.53 ms, 192 -> 7.44, 128 -> 7.31, 96 -> 7.28, 80 -> 7.49, 64 -> 7.86: the tangent sweep's
# launch-bound kernels crawled while a one-per-CU launch held every CU); since the tangent sweep runs as block / top-of-the-critic
# launches (round 5) the side part is the longer of the two and wants the whole card -- round 6 (bench.py --workload gan_step), one box:
# 96 -> 7.10 ms, 112 -> 6.91, 128 -> 6.64, 144 -> 6.67, 160 -> 6.60; another: 128 -> 6.49 (three runs), 160 -> 6.46, 192 -> 6.46,
# 224 -> 6.39, 256 -> 6.40 (four runs, 6.35 - 6.44); the video iteration does not care (13.25 either way)
TN_SIDE_WGS = int(os.environ.get("DHAUG_TN_SIDE_WGS", "256"))
# the second part's contractions started beside what is left of the first (only the sums wait): measured 6.87 against 6.94 ms
# per iteration -- the contractions are HBM-bound either way -- so it is an option, off
TN_PHASED = os.environ.get("DHAUG_TN_PHASED") is not None
TN_MAIN_WGS = int(os.environ.get("DHAUG_TN_MAIN_WGS", "0"))  # workgroups of the second part (0: one per CU)

# the layers of independent branches at the same depth as ONE launch (_Math.mm_group, dhaug_gemm_bf16_group): a motion critic's four /
# two branch layers.  On by default since round 5 (DHAUG_NO_NT_GROUP=1: one launch per layer): the grouped launch runs on 128 x 128
# tiles -- four 1 536 x 1000 x 1000 layers are 384 workgroups, ONE round of the card's slots, half the staged bytes of 64 x 64 tiles.
# (Round 4's form, the same launch on 64 x 64 tiles = 1 536 workgroups: 35 us alone against 4 x 12.7, but 17.2 ms per video iteration
# against 15.4 -- it left no slots to the other three critics' streams; DHAUG_NT_GROUP_TILE=64 still selects it.)
NT_GROUP = os.environ.get("DHAUG_NO_NT_GROUP") is None
# split-operand arithmetic: one activation-side split per tensor and step (_Math.split0); DHAUG_NO_SPLIT_CACHE=1: one per use
SPLIT_CACHE = os.environ.get("DHAUG_NO_SPLIT_CACHE") is None
# split-operand arithmetic ("bf16x6"): an operand whose padded width is 64 * 2^j is split into its three distinct pieces ONCE, as planes
# [hi | mid | lo] (6 bytes per value instead of 12, and one split where the forward / backward chains and sweep 4 wanted two layouts);
# the layer products (ops.gemm_nt_planes) and the grouped weight-gradient launch read the pieces in the order of the six-segment operand --
# bit-identical results.  DHAUG_X6_PLANES=0: six-segment operands everywhere.
PLANES = os.environ.get("DHAUG_X6_PLANES", "1") != "0"
# ... and a layer product whose RESULT is such an operand writes the result's planes itself (dhaug_gemm_bf16x6_planes, c_planes): bit for
# bit the split of the fp32 result, without the split launch (DHAUG_X6_PLANES_OUT=0: split launches)
PLANES_OUT = os.environ.get("DHAUG_X6_PLANES_OUT", "1") != "0"

# bf16 operands that a kernel of the step writes beside its fp32 result anyway (the assembled real / fake rows, the penalty's cotangent,
# the KCS operand) are registered as the casts of those tensors instead of being cast again (DHAUG_NO_SEED_CASTS=1: cast launches)
SEED_CASTS = os.environ.get("DHAUG_NO_SEED_CASTS") is None

# the 3D critic's penalty step (KCS pull-back, norm, penalty cotangent, KCS tangent, bf16 operands) as one launch: ops.d3_penalty
D3_PENALTY_FUSED = os.environ.get("DHAUG_NO_D3_PENALTY_FUSED") is None

BF16 = torch.bfloat16
NONE, RELU, LRELU = A.ACT_NONE, A.ACT_RELU, A.ACT_LRELU
ceil16 = A.ceil16


CAPTURE_ROOT = None        # handle of the stream the running hipGraph capture was begun on (graphs.capture_streams): the one
                           # fork level a capture survives belongs to code running directly on it


def tn_side_stream(cur):
    """the stream a step running on `cur` launches the first part of its sweep 4 on"""
    key = (cur.device.index, cur.cuda_stream)
    if key not in _TN_SIDE:
        _TN_SIDE[key] = torch.cuda.Stream()
    return _TN_SIDE[key]


def capture_root():
    """inside a hipGraph capture: is the current stream the one the capture was begun on (and its helper streams exist)?"""
    cur = torch.cuda.current_stream()
    return cur.cuda_stream == CAPTURE_ROOT and (cur.device.index, cur.cuda_stream) in _TN_SIDE


class _Math:
    """the three products of a layer in one arithmetic; activations are bf16 (M, ceil16 n) in 'bf16', fp32 (M, n) otherwise"""

    def __init__(self, prec):
        self.prec, self.bf16 = prec, prec == "bf16"
        self.T = 1 if self.bf16 else A.TERMS[prec]
        self.tn = []                                 # weight-gradient contractions waiting for the grouped launch (flush)
        self.tn2, self._tn_slots = [], set()         # split-operand arithmetic: second contributions to a slot already in self.tn
        self._splits, self._split_src = [], []       # split-operand arithmetic: (address, rows, cols, mode, split) of this step
        self._casts = {}                             # bf16: the casts of fp32 inputs made in this step

    def width(self, n):
        return ceil16(n) if self.bf16 else n

    def empty(self, M, n, dev):
        return torch.empty((M, self.width(n)), dtype=BF16 if self.bf16 else torch.float32, device=dev)

    def empty_blocks(self, M, nb, Dw, dev):
        """(M, nb * Dw) buffer whose Dw-wide column blocks become A operands of GEMMs with K = ceil16(Dw) (the cotangent of a
        concatenation, handed to the branches block by block).  Where Dw is not a multiple of 16 (DenseDim 1000) a block's rows are
        read 16 - Dw % 16 columns beyond the block -- against zero columns of the weights' operand copy, so the values only have to be
        FINITE: inside the buffer they are the next block's / the next row's cotangents, but the last row of the last block would be
        read beyond the allocation (found in round 5 with torch.empty poisoned: NaN x 0).  Such a buffer gets one more row, its head zeroed."""
        if not self.bf16 or Dw % 16 == 0:
            return None                              # (no over-read: the product allocates its output as usual)
        buf = torch.empty((M + 1, self.width(nb * Dw)), dtype=BF16, device=dev)
        buf[M, :16].zero_()
        return buf[:M]

    def _a(self, a, k):
        """activation-side operand of a (fp32 (M,k) network input / bf16 hidden / fp32 hidden)"""
        if self.bf16:
            return a if a.dtype == BF16 else self.cast(a, ceil16(k))
        return self.split0(a, k)

    def cast(self, a, width):
        """bf16 copy of an fp32 network input / tangent seed, made once per tensor and step (the tangent seeds of a branch are an
        operand of the tangent sweep's first GEMM and of sweep 4)"""
        key = (a.data_ptr(), tuple(a.shape), tuple(a.stride()), width)
        hit = self._casts.get(key)
        if hit is None:
            hit = (ops.cast_pad_bf16(a, width), a)       # (the source stays referenced: no address is reused inside the step)
            self._casts[key] = hit
        return hit[0]

    def seed_cast(self, a, width, b):
        """b IS the bf16 copy cast(a, width) would make -- its producer wrote it beside a (ops.gp_assemble / gp_penalty, the KCS
        operand of ops.kcs_forward): registered, no launch (the casts were 8 launches and ~90 us of a single-frame iteration)"""
        if self.bf16 and b is not None and b.dtype == BF16 and tuple(b.shape) == (a.shape[0], width):
            self._casts[(a.data_ptr(), tuple(a.shape), tuple(a.stride()), width)] = (b, a)

    def split0(self, a, k, mode=0):
        """the activation-side split of fp32 a (rows, k) -- made ONCE per tensor and step: a layer's input, cotangent and tangent
        are each an operand of one sweep AND of sweep 4 (there as row ranges of the same tensor), and a split moves 3.5 x the
        bytes of the tensor it splits (the splits were 29 of the step's 95 ms).  The sources are kept referenced until flush()
        (no address is reused inside a step); nothing in the schedule writes a tensor after it was an operand."""
        if not a.is_contiguous():                    # (a column block: split where it lies, one per use)
            return ops.split_bf16(a if (a.dim() == 2 and a.stride(1) == 1) else a.contiguous(), mode, self.T, ceil16(k))
        hit = self.cached_split(a, k, mode)
        if hit is not None:
            return hit
        p, rows, cols = a.data_ptr(), a.shape[0], a.shape[1]
        sp = ops.split_bf16(a, mode, self.T, ceil16(k))
        if SPLIT_CACHE and cols == k:
            self._splits.append((p, rows, cols, mode, sp))
            self._split_src.append(a)                # (referenced until flush: no address is reused inside a step)
        return sp

    def planes_ok(self, k):
        """an fp32 operand k wide may travel as three planes (see PLANES)"""
        kp = ceil16(k)
        return PLANES and not self.bf16 and self.T == 6 and kp >= 64 and (kp & (kp - 1)) == 0

    def cached_split(self, a, k, mode):
        """the split of (a row range of) this tensor made earlier in the step, or None"""
        if not (SPLIT_CACHE and a.is_contiguous() and a.shape[1] == k):
            return None
        p, rows, cols = a.data_ptr(), a.shape[0], a.shape[1]
        for bp, brows, bcols, bmode, sp in self._splits:
            if bmode == mode and bcols == cols and p >= bp and (p - bp) % (4 * cols) == 0:
                r0 = (p - bp) // (4 * cols)
                if r0 + rows <= brows:
                    return sp if (r0 == 0 and rows == brows) else sp[r0:r0 + rows]
        return None

    def mm(self, a, W, orient, bias=None, res=None, act=NONE, slope=0.0, mask=None, mask_act=NONE, out=None, out_f32=False):
        """(a @ W^T if orient == 'nt' else a @ W) + bias + res, then act(.) or, with `mask`, * mask_act'(mask)."""
        N, K = W.shape
        n, k = (N, K) if orient == "nt" else (K, N)
        kp = ceil16(k)
        if (self.planes_ok(k) and a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1
                and self.cached_split(a, k, 1 if (orient == "nn" and SPLIT_CACHE) else 0) is None):   # (a six-segment split made for a product the ping-pong kernel does not take serves this one too)
            masked = mask is not None and mask_act != NONE
            resc = res if (res is None or res.is_contiguous()) else res.contiguous()
            if ((resc is None or resc.dtype == torch.float32) and (not masked or (mask.dtype == torch.float32 and mask.stride(1) == 1))
                    and (out is None or (out.dtype == torch.float32 and out.stride(1) == 1))
                    and ops.gemm_planes_ok(n, kp, bias, resc, mask if masked else None, out)):
                swap = orient == "nn" and SPLIT_CACHE        # (the weights' operand copies as below)
                Bop = A._w_nt(W, kp, self.prec) if orient == "nt" else A._w_nn(W, self.prec, 0 if swap else 1)
                # a result that is itself an operand of that width (the next layer's input, sweep 4's) leaves the GEMM with its planes
                # beside it -- registered as this step's split of the tensor: no split launch for it
                emit = PLANES_OUT and SPLIT_CACHE and self.planes_ok(n) and ceil16(n) == n and (out is None or out.is_contiguous())
                r = ops.gemm_nt_planes(self.split0(a, k, 2), Bop, n, kp, bias=bias, res_f32=resc, act=act, slope=slope,
                                       dmask_f32=mask if masked else None, dmask_act=mask_act if masked else NONE, dmask_slope=slope,
                                       out=out, x_order=1 if swap else 0, planes_out=emit)
                if emit:
                    y, yp = r
                    self._splits.append((y.data_ptr(), y.shape[0], n, 2, yp))
                    self._split_src.append(y)
                    return y
                return r
        if (PLANES_OUT and SPLIT_CACHE and orient == "nt" and not self.bf16 and self.T == 6 and not self.planes_ok(k) and self.planes_ok(n)
                and ceil16(n) == n and a.dtype == torch.float32 and a.dim() == 2 and a.stride(1) == 1 and (out is None or out.is_contiguous())):
            # a narrow input layer (30 / 48 -> DenseDim) whose result is an operand of plane width: the six-segment product on the same
            # kernel, which writes the result's planes beside it
            masked = mask is not None and mask_act != NONE
            resc = res if (res is None or res.is_contiguous()) else res.contiguous()
            if ((resc is None or resc.dtype == torch.float32) and (not masked or (mask.dtype == torch.float32 and mask.stride(1) == 1))
                    and (out is None or (out.dtype == torch.float32 and out.stride(1) == 1))
                    and ops.gemm_planes_ok(n, kp, bias, resc, mask if masked else None, out, six=True)):
                y, yp = ops.gemm_nt_planes(self.split0(a, k, 0), A._w_nt(W, kp, self.prec), n, kp, bias=bias, res_f32=resc, act=act, slope=slope,
                                           dmask_f32=mask if masked else None, dmask_act=mask_act if masked else NONE, dmask_slope=slope,
                                           out=out, x_order=2, planes_out=True)
                self._splits.append((y.data_ptr(), y.shape[0], n, 2, yp))
                self._split_src.append(y)
                return y
        # split-operand arithmetic, backward chain (orient "nn"): the cotangent is split ONCE, in the weight-side layout -- the
        # layout sweep 4 contracts it in (autograd_ops._raw_outer) -- and meets the weights in the activation-side layout
        swap = (not self.bf16) and orient == "nn" and SPLIT_CACHE
        Bop = A._w_nt(W, kp, self.prec) if orient == "nt" else A._w_nn(W, self.prec, 0 if swap else 1)
        a_op = self.split0(a, k, 1) if swap else self._a(a, k)
        if (self.bf16 and RANK1 and orient == "nn" and N == 1 and a.dtype == BF16 and res is None and bias is None
                and mask is not None and mask_act != NONE and not out_f32 and mask.stride(0) % 8 == 0 and mask.shape[1] >= ceil16(K)
                and ceil16(K) <= 1024):
            # the logit layer's input cotangent: seed (rows,1) x weight row (1,K), masked -- a streaming kernel, not a K = 1 GEMM
            return ops.rank1_mask(a, A._w_nn(W, self.prec)[:, 0], mask, K, mask_act, slope, out=out)
        if self.bf16:
            rb = res if (res is not None and res.dtype == BF16) else None
            rf = res if (res is not None and res.dtype != BF16) else None
            if mask is not None and mask_act != NONE and not out_f32 and rf is None and bias is None:
                return ops.gemm_nt_dmask(a_op, Bop, n, kp, mask, mask_act, slope, res_bf16=rb, out=out)
            if mask is not None and mask_act != NONE and not out_f32:
                y, _ = ops.gemm_nt(a_op, Bop, n, kp, bias=bias, res_bf16=rb, res_f32=rf, out_bf16=True, n_pad=ceil16(n), c_bf16=out)
                w = min(y.shape[1], mask.shape[1])           # (a column block of a wider buffer carries no pad columns)
                ya, ma = (y, mask) if (y.shape[1] == w and mask.shape[1] == w) else (y[:, :w], mask[:, :w])
                ops.act_backward(ya, ma, mask_act, slope, out=ya)
                return y
            cb, cf = ops.gemm_nt(a_op, Bop, n, kp, bias=bias, res_bf16=rb, res_f32=rf, act=act, slope=slope,
                                 out_bf16=not out_f32, n_pad=ceil16(n), out_f32=out_f32, c_bf16=None if out_f32 else out,
                                 c_f32=out if out_f32 else None)
            return cf if out_f32 else cb
        # split-operand arithmetic (parity tests): fp32 activations; strided views are staged through contiguous copies
        if res is not None and not res.is_contiguous():
            res = res.contiguous()
        masked = mask is not None and mask_act != NONE
        if masked and bias is None and act == NONE and mask.dtype == torch.float32 and mask.stride(1) == 1 and (out is None or out.stride(1) == 1):
            # the mask rides the GEMM's epilogue (fp32 mask from the producing layer's activation): one launch, not two
            return ops.gemm_nt_dmask_f32(a_op, Bop, n, self.T * kp, mask, mask_act, slope, res_f32=res, out=out)
        direct = out is not None and not (masked and not out.is_contiguous())
        _, cf = ops.gemm_nt(a_op, Bop, n, self.T * kp, bias=bias, res_f32=res, act=act, slope=slope, out_f32=True,
                            c_f32=out if direct else None)
        if masked:
            ops.act_backward(cf, mask if mask.is_contiguous() else mask.contiguous(), mask_act, slope, out=cf)
        if out is not None and not direct:
            out.copy_(cf)
            return out
        return cf

    def mm_group(self, calls):
        """The products of the SAME layer position in several independent branches: `calls` are keyword dicts for mm().  In bf16,
        where every call is a plain or mask-fused generic GEMM of one shape, they are ONE launch (ops.gemm_nt_group: a motion
        critic's four / two branch layers); otherwise they run one by one."""
        if not (self.bf16 and NT_GROUP and 2 <= len(calls) <= ops.NT_GROUP_MAX):
            return [self.mm(**kw) for kw in calls]
        members, shape = [], None
        for kw in calls:
            a, W, orient = kw["a"], kw["W"], kw["orient"]
            N, K = W.shape
            n, k = (N, K) if orient == "nt" else (K, N)
            kp = ceil16(k)
            res, mask, mask_act = kw.get("res"), kw.get("mask"), kw.get("mask_act", NONE)
            masked = mask is not None and mask_act != NONE
            out = kw.get("out")
            ok = (a.dtype == BF16 and (a.shape[1] == kp or (a.shape[1] == k and a.stride(0) >= kp)) and a.stride(1) == 1
                  and not kw.get("out_f32", False) and (res is None or res.dtype == BF16)
                  and n > 256 and kp > 256 and (out is None or (out.dtype == BF16 and out.stride(1) == 1))
                  and (not masked or (kw.get("bias") is None and kw.get("act", NONE) == NONE and mask.dtype == BF16
                                      and getattr(mask, "_dhaug_bits", None) is None and getattr(mask, "_dhaug_bits_cols", None) is None)))
            sh = (a.shape[0], n, kp, masked)
            if not ok or (shape is not None and sh != shape):
                return [self.mm(**kw) for kw in calls]
            shape = sh
            Bop = A._w_nt(W, kp, self.prec) if orient == "nt" else A._w_nn(W, self.prec)
            members.append(dict(A=a, B=Bop, N=n, K=kp, bias=kw.get("bias"), res_bf16=res, act=kw.get("act", NONE), slope=kw.get("slope", 0.0),
                                dmask=mask if masked else None, dmask_act=mask_act if masked else NONE, dmask_slope=kw.get("slope", 0.0),
                                out=out, n_pad=ceil16(n)))
        return ops.gemm_nt_group(members)

    def flush(self):
        """launch the collected weight-gradient contractions (before the optimizer step reads the gradient bucket)"""
        if (self.tn and TN_PHASED and getattr(self, "_side", None) is not None and len(self.tn) <= ops._lib.TN_GROUP_MAX
                and all(it[2] <= 256 and it[3] <= 256 for it in self.tn) and not self.tn2
                and len({it[4].data_ptr() for it in self.tn}) == len(self.tn)):
            # (layers of at most 256 x 256 with distinct gradient slots only: a wide layer is several blocks of the launch, and
            # the phased form takes one chunk)
            # both parts add into the same gradient slots -- but only their SUMS touch the slots: this part's contractions
            # start now, beside what is left of the side part, and only the sums wait for it
            ws = ops._tn_group_workspace(self.tn[0][0].device)
            ops.gemm_tn_group(self.tn, phase=1, workspace=ws, max_workgroups=TN_MAIN_WGS)
            self.join()
            ops.gemm_tn_group(self.tn, phase=2, workspace=ws, max_workgroups=TN_MAIN_WGS)
            self.tn = []
            self.tn2, self._tn_slots = [], set()
            self._splits, self._split_src, self._casts = [], [], {}
            return
        self.join()                          # (both parts accumulate into the same gradient slots: never concurrently)
        if self.tn:
            ops.gemm_tn_group(self.tn)
            self.tn = []
        if self.tn2:
            ops.gemm_tn_group(self.tn2)
        self.tn2, self._tn_slots = [], set()
        self._splits, self._split_src, self._casts = [], [], {}

    def flush_side(self):
        """the same on a side stream of the current one (join() / flush() makes the current stream wait for it): the
        operands stay referenced until then"""
        if not self.tn:
            return
        cur = torch.cuda.current_stream()
        st = tn_side_stream(cur)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            ops.gemm_tn_group(self.tn, max_workgroups=TN_SIDE_WGS)
        self._side = (st, self.tn)
        self.tn = []

    def join(self):
        side = getattr(self, "_side", None)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side[0])
            self._side = None

    def can_split(self, B):
        """sweep 4 in two parts (TN_SPLIT): bf16, whole 32-row stages in both parts, batches long enough for the grouped launch"""
        # (inside a hipGraph capture only on the stream the capture was begun on: hipStreamEndCapture of this HIP release
        # crashes on a fork inside a fork, or on an edge between sibling branches -- a step that runs on a forked stream
        # there, concurrent critics, keeps sweep 4 in one part)
        return (self.bf16 and TN_SPLIT and TN_GROUP and B % 32 == 0 and ops.tn_group_ok(B, 1, 1, 0)
                and (not torch.cuda.is_current_stream_capturing() or capture_root()))

    def fusable(self, n, k, rows):
        """the mask (and skip) ride the GEMM epilogue of every bf16 kernel (an element's mask value is read by the thread
        that then writes that element), so the output may overwrite the mask"""
        return self.bf16

    def outer(self, g, x, N, K, wslot, bslot=None, colsum_rows=None):
        """wslot (N,K) += g^T x;  bslot (N) += column sums of g over rows [0, colsum_rows) (all rows by default; through the
        pairing column-sum kernel when N < 16)"""
        if self.bf16:
            gb = g if g.dtype == BF16 else self.cast(g, ceil16(N))
            xb = x if x.dtype == BF16 else self.cast(x, ceil16(K))
            narrow = N < 16
            M = gb.shape[0]
            cr = M if colsum_rows is None else colsum_rows
            if TN_GROUP and ops.tn_group_ok(M, min(N, 256), min(K, 256), cr):
                # joins the step's grouped launch (a DenseDim-1000 layer is 4 x 4 blocks of 256 x 256: ops.gemm_tn_group hands it
                # over whole where the group has blocks enough to fill the card, block by block otherwise; the bias sums ride
                # with the first column block of every row block)
                cs = bslot if (bslot is not None and not narrow) else None
                self.tn.append((gb, xb, N, K, wslot, cs, cr if cs is not None else 0, True, M, None, None))
            else:
                ops.gemm_tn(gb, xb, N, K, colsum=bslot if (bslot is not None and not narrow) else None, out=wslot, accumulate=True,
                            colsum_rows=colsum_rows)
            if bslot is not None and narrow:
                ops.colsum(gb if colsum_rows is None else gb[:colsum_rows], N=N, out=bslot, accumulate=True)
            return
        rowm = lambda t: t if (t.dim() == 2 and t.stride(1) == 1) else t.contiguous()      # (row-major, any row pitch)
        gc, xc = rowm(g), rowm(x)
        TM, Np, Kp = self.T * gc.shape[0], ceil16(N), ceil16(K)
        grouped = TN_GROUP and ops.tn_group_ok(TM, min(N, 256), min(K, 256), 0) and (wslot.data_ptr(), 2) not in self._tn_slots
        # (planes: the grouped launch only, layers of one 256 x 256 block; it reads piece (0 1 0 1 2 0)[t] of g and (0 0 1 1 0 2)[t] of x)
        pa = 2 if (grouped and N <= 256 and K <= 256 and self.planes_ok(N) and self.cached_split(gc, N, 1) is None) else 0
        pb = 1 if (grouped and N <= 256 and K <= 256 and self.planes_ok(K) and self.cached_split(xc, K, 0) is None) else 0
        g1, x3 = self.split0(gc, N, 2 if pa else 1), self.split0(xc, K, 2 if pb else 0)
        if grouped:
            # the split-operand contraction (ONE contraction over T * M rows: autograd_ops._raw_outer) joins the step's grouped launch
            # like a bf16 one.  A second contribution to the same gradient (the interpolated rows' part) waits for a second launch:
            # the items of one launch are summed into their slots concurrently.
            item = (g1.view(-1, Np), x3.view(-1, Kp), N, K, wslot, None, 0, True, TM, None, None, pa, pb)
            key = wslot.data_ptr()
            if key in self._tn_slots:
                self.tn2.append(item)
                self._tn_slots.add((key, 2))         # (a third contribution would run on its own, right away)
            else:
                self.tn.append(item)
                self._tn_slots.add(key)
        else:
            A._raw_outer(gc, xc, N, K, self.prec, out=wslot, split=(g1, x3))
        if bslot is not None:
            ops.colsum(g if g.is_contiguous() else g.contiguous(), N=N, out=bslot, accumulate=True)


def _slot(p):
    s = getattr(p, "_dhaug_grad_slot", None)
    if s is None or p.grad is not s:
        raise RuntimeError("critic_step needs the critic's parameters under a FusedAdam flat gradient bucket")
    return s


class _Lin:
    """one nn.Linear inside the schedule: forward, backward-chain, tangent and weight-gradient products"""

    def __init__(self, lin, act, slope=0.0):
        self.W, self.b, self.act, self.slope = lin.weight, lin.bias, act, slope
        self.N, self.K = lin.weight.shape

    def fwd(self, m, x, res=None, out=None, out_f32=False):
        return m.mm(x, self.W, "nt", bias=self.b, res=res, act=self.act, slope=self.slope, out=out, out_f32=out_f32)

    def _bslot(self):
        return None if self.b is None else _slot(self.b)

    def bwd(self, m, gz, mask, mask_act, slope, skip=None, out=None, out_f32=False):
        """(gz W + skip) * mask_act'(mask): cotangent at the producer's pre-activation"""
        return m.mm(gz, self.W, "nn", res=skip, mask=mask, mask_act=mask_act, slope=slope, out=out, out_f32=out_f32)

    def tan(self, m, u, y, skip=None, out=None, inplace=False):
        """(u W^T + skip) * act'(y): tangent through this layer.  inplace: where the mask rides the GEMM epilogue (or there is
        no mask) the result overwrites y -- the interpolated rows of an activation buffer then hold its tangent, and the
        layer's weight gradient is ONE contraction over all 3B rows (see grads)."""
        if inplace and out is None and (self.act == NONE or m.fusable(self.N, ceil16(self.K), u.shape[0])) and m.bf16:
            out = y
        return m.mm(u, self.W, "nt", res=skip, mask=y, mask_act=self.act, slope=self.slope, out=out)

    def grads(self, m, gz, x, B2, u, bias_is_zero=False):
        """dW += gz[:2B]^T x[:2B] + gz[2B:]^T u,  db += colsum(gz[:2B]).  bias_is_zero: the logit layer of a critic step --
        its cotangent is -1/B on the B real rows and +1/B on the B fake rows, so the bias gradient is EXACTLY zero (the
        reference's two backward passes cancel to the last bit as well; the pairing column-sum kernel reproduced that 0 in
        26 us per step): the slot, zeroed by zero_grad, is left alone"""
        if bias_is_zero:
            bz = self.b
            self.b = None
            try:
                return self._grads(m, gz, x, B2, u)
            finally:
                self.b = bz
        return self._grads(m, gz, x, B2, u)

    def grads_part(self, m, g, x, with_bias):
        """dW += g^T x over the given rows (one part of sweep 4), db += colsum(g) if with_bias"""
        m.outer(g, x, self.N, self.K, _slot(self.W), self._bslot() if with_bias else None)

    def _grads(self, m, gz, x, B2, u):
        if (m.bf16 and x.dtype == BF16 and u.dtype == BF16 and B2 % 128 == 0 and u.data_ptr() == x[B2:].data_ptr()
                and u.stride(0) == x.stride(0)):
            # the tangent was written over the interpolated rows of x: one launch contracts all 3B rows
            m.outer(gz, x, self.N, self.K, _slot(self.W), self._bslot(), colsum_rows=B2)
            return
        if m.bf16 and x.dtype != BF16 and u.dtype != BF16 and B2 % 128 == 0 and x.shape[0] == B2 + u.shape[0]:
            # a network input layer: [x(real, fake); tangent seed] cast into one bf16 operand, one contraction
            xb = torch.empty((x.shape[0], ceil16(self.K)), dtype=BF16, device=x.device)
            ops.cast_pad_bf16(x[:B2], ceil16(self.K), out=xb[:B2])
            ops.cast_pad_bf16(u, ceil16(self.K), out=xb[B2:])
            m.outer(gz, xb, self.N, self.K, _slot(self.W), self._bslot(), colsum_rows=B2)
            return
        m.outer(gz[:B2], x[:B2], self.N, self.K, _slot(self.W), self._bslot())
        m.outer(gz[B2:], u, self.N, self.K, _slot(self.W))


class _Block:
    """myResNet: y = relu(fc2(relu(fc1(x))) + x)"""

    def __init__(self, blk):
        self.fc1, self.fc2 = _Lin(blk.fc1, RELU), _Lin(blk.fc2, RELU)

    def fwd(self, m, x):
        h = self.fc1.fwd(m, x)
        return h, self.fc2.fwd(m, h, res=x)

    def bwd(self, m, gz2, h, x, x_act=RELU, out=None):
        """gz2: cotangent at fc2's pre-activation.  Returns (gz1, cotangent at the pre-activation of x's producer)."""
        if x_act == RELU and _pair_ok(m, self.fc2, self.fc1, gz2, h, x, out):
            # both layers in one launch: gz1 stays in LDS between them, gz2 is read once (dhaug_gemm_block2_bf16)
            return ops.gemm_block2(gz2, A._w_nn(self.fc2.W, m.prec), A._w_nn(self.fc1.W, m.prec), h, x, RELU, 0.0, out2=out)
        gz1 = self.fc2.bwd(m, gz2, h, RELU, 0.0)
        return gz1, self.fc1.bwd(m, gz1, x, x_act, 0.0, skip=gz2, out=out)

    def tan(self, m, u, h, y):
        if _pair_ok(m, self.fc1, self.fc2, u, h, y, y) and h.data_ptr() != u.data_ptr() and y.data_ptr() != u.data_ptr():
            # (in place: the tangents overwrite the interpolated rows of h and y -- the masks are their sign bits)
            return ops.gemm_block2(u, A._w_nt(self.fc1.W, 256, m.prec), A._w_nt(self.fc2.W, 256, m.prec), h, y, RELU, 0.0, out1=h, out2=y)
        uh = self.fc1.tan(m, u, h, inplace=True)
        return uh, self.fc2.tan(m, uh, y, skip=u, inplace=True)


BLOCK2_STACK = os.environ.get("DHAUG_NO_BLOCK2_STACK") is None


def stack_bwd(m, blocks, top, hs, ys):
    """the backward chain through a branch's blocks (last to first): top = cotangent at the last block's fc2 pre-activation;
    hs[i], ys[i]: block i's hidden activation and INPUT.  Returns (a1, a2) with a1[i] = cotangent at block i's fc1
    pre-activation... of fc2's input, a2[i] = cotangent at the pre-activation producing ys[i]; a2[n] = top."""
    n = len(blocks)
    a2, a1 = [None] * n + [top], [None] * n
    if (BLOCK2_STACK and 2 <= n <= ops._lib.BLOCK2_MAX and all(_pair_ok(m, b.fc2, b.fc1, top, hs[i], ys[i], None) for i, b in enumerate(blocks))):
        # one launch for the whole chain (dhaug_gemm_block2_stack_bf16): block i takes block i + 1's result
        desc = [(A._w_nn(blocks[i].fc2.W, m.prec), A._w_nn(blocks[i].fc1.W, m.prec), hs[i], ys[i], None, None) for i in range(n - 1, -1, -1)]
        outs = ops.gemm_block2_stack(top, desc, RELU, 0.0)
        for j, i in enumerate(range(n - 1, -1, -1)):
            a1[i], a2[i] = outs[j]
        return a1, a2
    for i in range(n - 1, -1, -1):
        a1[i], a2[i] = blocks[i].bwd(m, a2[i + 1], hs[i], ys[i])
    return a1, a2


def stack_tan(m, blocks, u0, hs, ys):
    """the tangent chain through a branch's blocks (first to last), in place over hs[i] / ys[i] (x_hat rows of the saved hidden
    activations / block OUTPUTS, sign bits attached).  Returns (uh, u): uh[i] = tangent of block i's hidden activation,
    u[i] = tangent of its output."""
    n = len(blocks)
    ok = BLOCK2_STACK and 2 <= n <= ops._lib.BLOCK2_MAX
    x = u0
    for i, b in enumerate(blocks):
        ok = ok and _pair_ok(m, b.fc1, b.fc2, u0, hs[i], ys[i], ys[i]) and hs[i].data_ptr() != x.data_ptr() and ys[i].data_ptr() != x.data_ptr()
        x = ys[i]
    if ok:
        desc = [(A._w_nt(b.fc1.W, 256, m.prec), A._w_nt(b.fc2.W, 256, m.prec), hs[i], ys[i], hs[i], ys[i]) for i, b in enumerate(blocks)]
        outs = ops.gemm_block2_stack(u0, desc, RELU, 0.0)
        return [o[0] for o in outs], [o[1] for o in outs]
    uh, u, x = [], [], u0
    for i, b in enumerate(blocks):
        hh, x = b.tan(m, x, hs[i], ys[i])
        uh.append(hh); u.append(x)
    return uh, u


def _pair_ok(m, la, lb, x, mask1, mask2, out):
    """two 256 -> 256 layers behind each other, bf16, both masks with sign bits: one launch (ops.gemm_block2)"""
    return (m.bf16 and la.N == 256 and la.K == 256 and lb.N == 256 and lb.K == 256 and ops.block2_ok(x, mask1, mask2, x.shape[0])
            and (out is None or (out.dtype == BF16 and out.stride(0) % 8 == 0 and out.shape[1] >= 256)))


def _seeds(B, m, dev):
    """logit cotangents of the 3B rows: -1/B (real), +1/B (fake), 1 (x_hat: grad_outputs = ones)"""
    s = torch.empty((3 * B, 1), dtype=torch.float32, device=dev)
    s[:B] = -1.0 / B
    s[B:2 * B] = 1.0 / B
    s[2 * B:] = 1.0
    return ops.cast_pad_bf16(s, 16) if m.bf16 else s


_SEED_CACHE = {}


def seeds(B, m, dev):
    key = (B, m.bf16, str(dev))
    if key not in _SEED_CACHE:
        if torch.cuda.is_current_stream_capturing():         # (a cached tensor must not live in one graph's memory pool)
            return _seeds(B, m, dev)
        _SEED_CACHE[key] = _seeds(B, m, dev)
    return _SEED_CACHE[key]


def _finish(optimizerD, logits, pen, pen_rows, lam, rows=None):
    """scalars (logit means over `rows` real / fake rows, penalty mean over the pen_rows = len(pen) penalty terms) and the
    optimizer step"""
    sc = ops.critic_scalars(logits, pen, pen_rows if rows is None else rows, lam)
    optimizerD.step()
    return sc


def step_d2(D, optimizerD, real, fake, alpha, lam, prec=None):
    """Fk_2D_Discriminator (R/models_Fk_GAN/Fk_discriminator.py:236-266).  real, fake (B,32); alpha (B,1).
    Returns the (5,) device tensor D_real, D_fake, GP, Wasserstein_D, D_cost; gradients are left in the optimizer's bucket
    and the Adam step is taken."""
    m = _Math(prec or D.precision)
    L = [_Lin(D.pose_layer_1, LRELU, D.slope), _Lin(D.pose_layer_2, LRELU, D.slope), _Lin(D.pose_layer_3, LRELU, D.slope),
         _Lin(D.pose_layer_4, NONE), _Lin(D.layer_last, LRELU, D.slope), _Lin(D.layer_pred, NONE)]
    s = D.slope
    optimizerD.zero_grad()
    X = ops.gp_assemble(real, fake, alpha, bf16_rows=m.bf16 and SEED_CASTS)
    B = X.shape[0] // 3
    B2 = 2 * B
    m.seed_cast(X[:B2], ceil16(X.shape[1]), getattr(X, "_dhaug_bf16_rows", None))
    from . import fused
    if m.bf16 and FUSED_STEP_FORWARD and fused.step_forward_supported(D):
        r = fused.critic2d_forward_save(D, X, save_rows=B2 if (SKIP_XHAT_SAVES and fused.partial_save_ok(B)) else 0)
        (d1, d2, d3, d4, dl), logits = r["d"], r["logits"]
    else:
        d1 = L[0].fwd(m, X)
        d2 = L[1].fwd(m, d1)
        d3 = L[2].fwd(m, d2, res=d1)
        d4 = L[3].fwd(m, d3)
        dl = L[4].fwd(m, d4)
        logits = L[5].fwd(m, dl, out_f32=True)
    gzp = seeds(B, m, X.device)
    gzl = L[5].bwd(m, gzp, dl, LRELU, s)
    gz4 = L[4].bwd(m, gzl, d4, NONE, 0.0)
    gz3 = L[3].bwd(m, gz4, d3, LRELU, s)
    if _pair_ok(m, L[2], L[1], gz3, d2, d1, None):            # d3 = lrelu(L3 d2 + d1): the same two-layer pattern as a myResNet block
        gz2, gz1 = ops.gemm_block2(gz3, A._w_nn(L[2].W, m.prec), A._w_nn(L[1].W, m.prec), d2, d1, LRELU, s)
    else:
        gz2 = L[2].bwd(m, gz3, d2, LRELU, s)
        gz1 = L[1].bwd(m, gz2, d1, LRELU, s, skip=gz3)
    split = m.can_split(B)
    if split:                                                 # sweep 4, real / fake rows: beside the penalty and the tangent sweep
        for lay, gz, x in ((L[0], gz1, X), (L[1], gz2, d1), (L[2], gz3, d2), (L[3], gz4, d3), (L[4], gzl, d4), (L[5], gzp, dl)):
            lay.grads_part(m, gz[:B2], x[:B2], lay is not L[5])
        m.flush_side()
    g = L[0].bwd(m, gz1[B2:], None, NONE, 0.0, out_f32=True)                 # (B,32) fp32: dD/dx_hat
    v, pen = ops.gp_penalty(g, 2.0 * lam / B, bf16=m.bf16 and SEED_CASTS)
    m.seed_cast(v, ceil16(v.shape[1]), getattr(v, "_dhaug_bf16", None))
    tail = ops.tail_rows                                      # (x_hat rows of a saved activation, its sign bits attached)
    u1 = L[0].tan(m, v, tail(d1, B2), inplace=True)
    t2, t3 = tail(d2, B2), tail(d3, B2)
    if _pair_ok(m, L[1], L[2], u1, t2, t3, t3) and t2.data_ptr() != u1.data_ptr():
        u2, u3 = ops.gemm_block2(u1, A._w_nt(L[1].W, 256, m.prec), A._w_nt(L[2].W, 256, m.prec), t2, t3, LRELU, s, out1=t2, out2=t3)
    else:
        u2 = L[1].tan(m, u1, t2, inplace=True)
        u3 = L[2].tan(m, u2, t3, skip=u1, inplace=True)
    u4 = L[3].tan(m, u3, tail(d4, B2), inplace=True)
    ul = L[4].tan(m, u4, tail(dl, B2), inplace=True)
    layers = ((L[0], gz1, X, v), (L[1], gz2, d1, u1), (L[2], gz3, d2, u2), (L[3], gz4, d3, u3), (L[4], gzl, d4, u4), (L[5], gzp, dl, ul))
    if split:
        for lay, gz, x, u in layers:
            lay.grads_part(m, gz[B2:], u, False)
    else:
        for lay, gz, x, u in layers:
            lay.grads(m, gz, x, B2, u, bias_is_zero=(lay is L[5] and m.bf16))
    m.flush()
    return _finish(optimizerD, logits, pen, B, lam)


class _Branch:
    """Linear + ReLU, three myResNet blocks (one input branch of the 3D critic / the motion critics)"""

    def __init__(self, first, blocks):
        self.first, self.blocks = _Lin(first, RELU), [_Block(b) for b in blocks]


def _layer_major(m, branches):
    """wide branches of equal depth in bf16: their layers at the same depth run as one launch each (_Math.mm_group)"""
    return (m.bf16 and NT_GROUP and len(branches) > 1 and branches[0].first.N > 256
            and all(len(br.blocks) == len(branches[0].blocks) and br.first.N == branches[0].first.N for br in branches))


def _fwd_layer_major(m, branches, F, cat, Dw):
    """sweep 1 of the branches, depth by depth (same products as _Block.fwd, same buffers)"""
    nb, nblk = len(branches), len(branches[0].blocks)
    y = [[br.first.fwd(m, F[bi])] for bi, br in enumerate(branches)]
    h = [[] for _ in range(nb)]
    for i in range(nblk):
        hh = m.mm_group([dict(a=y[bi][-1], W=br.blocks[i].fc1.W, orient="nt", bias=br.blocks[i].fc1.b, act=RELU) for bi, br in enumerate(branches)])
        last = i == nblk - 1
        yy = m.mm_group([dict(a=hh[bi], W=br.blocks[i].fc2.W, orient="nt", bias=br.blocks[i].fc2.b, res=y[bi][-1], act=RELU,
                              out=cat[:, bi * Dw:(bi + 1) * Dw] if last else None) for bi, br in enumerate(branches)])
        for bi in range(nb):
            h[bi].append(hh[bi]); y[bi].append(yy[bi])
    return y, h


def _bwd_layer_major(m, branches, gcat, h, y, Dw):
    """sweep 2 through the branches' blocks, depth by depth (same products as _Block.bwd's two-launch form)"""
    nb, nblk = len(branches), len(branches[0].blocks)
    a2 = [[None] * nblk + [gcat[:, bi * Dw:(bi + 1) * Dw]] for bi in range(nb)]
    a1 = [[None] * nblk for _ in range(nb)]
    for i in range(nblk - 1, -1, -1):
        g1 = m.mm_group([dict(a=a2[bi][i + 1], W=br.blocks[i].fc2.W, orient="nn", mask=h[bi][i], mask_act=RELU) for bi, br in enumerate(branches)])
        g2 = m.mm_group([dict(a=g1[bi], W=br.blocks[i].fc1.W, orient="nn", res=a2[bi][i + 1], mask=y[bi][i], mask_act=RELU)
                         for bi, br in enumerate(branches)])
        for bi in range(nb):
            a1[bi][i], a2[bi][i] = g1[bi], g2[bi]
    return a1, a2


def _tan_layer_major(m, branches, u_first, hs, ys):
    """sweep 3 through the branches' blocks, depth by depth, in place over the interpolated rows of hs / ys (as _Block.tan)"""
    nb, nblk = len(branches), len(branches[0].blocks)
    u = [[u_first[bi]] for bi in range(nb)]
    uh = [[] for _ in range(nb)]
    for i in range(nblk):
        t1 = m.mm_group([dict(a=u[bi][-1], W=br.blocks[i].fc1.W, orient="nt", mask=hs[bi][i], mask_act=RELU, out=hs[bi][i])
                         for bi, br in enumerate(branches)])
        t2 = m.mm_group([dict(a=t1[bi], W=br.blocks[i].fc2.W, orient="nt", res=u[bi][-1], mask=ys[bi][i], mask_act=RELU, out=ys[bi][i])
                         for bi, br in enumerate(branches)])
        for bi in range(nb):
            uh[bi].append(t1[bi]); u[bi].append(t2[bi])
    return u, uh


def _top_shapes(m, Lm, Mb, Lo, nb, Dw):
    """a concatenation of two 256-wide branches -> Linear(n0 <= 112) + ReLU -> myResNet(n0) -> Linear(1): what ops.critic_top_* cover"""
    n0 = Lm.N
    return (m.bf16 and nb == 2 and Dw == 256 and Lm.K == 512 and Lo.N == 1 and Lo.K == n0 and Mb.fc1.N == n0 and Mb.fc1.K == n0
            and Mb.fc2.N == n0 and Mb.fc2.K == n0 and Lm.act == RELU and Mb.fc1.act == RELU and Mb.fc2.act == RELU)


def _top_fusable(m, Lm, Mb, Lo, nb, Dw, M, masks, cat):
    """the top of the critic (merge layer -> merge block -> logit layer) as one launch per sweep: ops.critic_top_backward"""
    return _top_shapes(m, Lm, Mb, Lo, nb, Dw) and ops.top_backward_ok(M, Lm.N, 512, masks, getattr(cat, "_dhaug_bits_cols", None))


def step_branchnet(m, optimizerD, branches, Lm, Mb, Lo, X, rows, lam, feats, input_grad, tangents, pen_view=None, fwd=None, penalty=None):
    """The four sweeps for a critic of the form  cat_b(branch_b(feat_b(x))) -> Linear(100)+ReLU -> myResNet(100) -> Linear(1).
    X (3*rows, W) fp32 = [real; fake; x_hat] (ops.gp_assemble); feats(X) -> one fp32 input per branch (3*rows each);
    input_grad([g_b]) -> dD/dx_hat (rows, W) fp32 from the branches' input cotangents (x_hat rows); tangents(v) -> one fp32
    tangent input per branch (rows each).  pen_view: shape the penalty sees g in (default (rows, W); the 2D motion critic is
    stepped with one interpolation coefficient per FRAME)."""
    dev = X.device
    B = rows
    B2, M3 = 2 * B, 3 * B
    nb, Dw = len(branches), branches[0].first.N
    F = feats(X)
    # ---- 1. forward (the branch outputs land side by side: the concatenation is a buffer, not a copy)
    if fwd is not None:                                      # one fused launch that saves every layer's output (same buffers)
        r = fwd()
        cat, y, h, m0, mh, m1, logits = r["cat"], r["y"], r["h"], r["m0"], r["mh"], r["m1"], r["logits"]
    elif _layer_major(m, branches):
        cat = m.empty(M3, nb * Dw, dev)
        y, h = _fwd_layer_major(m, branches, F, cat, Dw)
        m0 = Lm.fwd(m, cat)
        mh, m1 = Mb.fwd(m, m0)
        logits = Lo.fwd(m, m1, out_f32=True)
    else:
        cat = m.empty(M3, nb * Dw, dev)
        y, h = [], []
        for bi, br in enumerate(branches):
            ys, hs = [br.first.fwd(m, F[bi])], []
            for i, blk in enumerate(br.blocks):
                hh = blk.fc1.fwd(m, ys[-1])
                hs.append(hh)
                ys.append(blk.fc2.fwd(m, hh, res=ys[-1], out=cat[:, bi * Dw:(bi + 1) * Dw] if i == len(br.blocks) - 1 else None))
            y.append(ys); h.append(hs)
        m0 = Lm.fwd(m, cat)
        mh, m1 = Mb.fwd(m, m0)
        logits = Lo.fwd(m, m1, out_f32=True)
    # ---- 2. backward chain
    gzo = seeds(B, m, dev)
    if Dw == 256 and all(getattr(y[bi][-1], "_dhaug_bits", None) is not None for bi in range(nb)):
        cat._dhaug_bits_cols = [y[bi][-1]._dhaug_bits for bi in range(nb)]      # (the mask of column block bi: its branch's sign bits)
    if _top_fusable(m, Lm, Mb, Lo, nb, Dw, M3, (m1, mh, m0), cat):
        # merge layer, merge block and logit layer in ONE launch: the 100-wide cotangents stay in LDS between the layers
        gz_m2, gz_m1, gz_m0, gcat = ops.critic_top_backward(
            gzo, A._w_nn(Lo.W, m.prec)[:, 0], m1, mh, m0, A._w_nn(Mb.fc2.W, m.prec), A._w_nn(Mb.fc1.W, m.prec), A._w_nn(Lm.W, m.prec),
            cat._dhaug_bits_cols, Lm.N, RELU, 0.0)
    else:
        gz_m2 = Lo.bwd(m, gzo, m1, RELU, 0.0)
        gz_m1, gz_m0 = Mb.bwd(m, gz_m2, mh, m0)
        gcat = Lm.bwd(m, gz_m0, cat, RELU, 0.0, out=m.empty_blocks(M3, nb, Dw, dev))   # (3B, nb*D): cotangents at every branch's last fc2
    g1, g2, gin = [], [], []
    if _layer_major(m, branches):
        g1, g2 = _bwd_layer_major(m, branches, gcat, h, y, Dw)
        gin = [br.first.bwd(m, g2[bi][0][B2:], None, NONE, 0.0, out_f32=True) for bi, br in enumerate(branches)]
    for bi, br in enumerate(branches if not gin else []):
        # a2[i]: cotangent at the pre-activation producing y[i]
        a1, a2 = stack_bwd(m, br.blocks, gcat[:, bi * Dw:(bi + 1) * Dw], h[bi], y[bi])
        g1.append(a1); g2.append(a2)
        gin.append(br.first.bwd(m, a2[0][B2:], None, NONE, 0.0, out_f32=True))   # (B, w_b) fp32, x_hat rows only
    split = m.can_split(B)
    if split:                                                # sweep 4, real / fake rows: beside the penalty and the tangent sweep
        for bi, br in enumerate(branches):
            br.first.grads_part(m, g2[bi][0][:B2], F[bi][:B2], True)
            for i, blk in enumerate(br.blocks):
                blk.fc1.grads_part(m, g1[bi][i][:B2], y[bi][i][:B2], True)
                blk.fc2.grads_part(m, g2[bi][i + 1][:B2], h[bi][i][:B2], True)
        Lm.grads_part(m, gz_m0[:B2], cat[:B2], True)
        Mb.fc1.grads_part(m, gz_m1[:B2], m0[:B2], True); Mb.fc2.grads_part(m, gz_m2[:B2], mh[:B2], True)
        Lo.grads_part(m, gzo[:B2], m1[:B2], False)           # (its bias gradient is exactly zero: see _Lin.grads)
        m.flush_side()
    # ---- 3. penalty and tangent sweep (x_hat rows)
    if penalty is not None and m.bf16:
        # dD/dx_hat, its norm, the penalty's cotangent and the branches' tangent inputs (bf16 operands) in ONE launch
        T, pen = penalty(gin, 2.0 * lam / B)
        n_pen = B
    else:
        g = input_grad(gin)                                  # dD/dx_hat
        gv = g if pen_view is None else g.reshape(pen_view)
        v, pen = ops.gp_penalty(gv, 2.0 * lam / gv.shape[0])
        T = tangents(v.reshape(g.shape))
        n_pen = gv.shape[0]
    u, uh = [], []
    if _layer_major(m, branches):
        u_first = [br.first.tan(m, T[bi], ops.tail_rows(y[bi][0], B2), inplace=True) for bi, br in enumerate(branches)]
        u, uh = _tan_layer_major(m, branches, u_first, [[ops.tail_rows(t, B2) for t in h[bi]] for bi in range(nb)],
                                 [[ops.tail_rows(t, B2) for t in y[bi][1:]] for bi in range(nb)])
    for bi, br in enumerate(branches if not u else []):
        u_first = br.first.tan(m, T[bi], ops.tail_rows(y[bi][0], B2), inplace=True)
        uhs, uo = stack_tan(m, br.blocks, u_first, [ops.tail_rows(t, B2) for t in h[bi]], [ops.tail_rows(t, B2) for t in y[bi][1:]])
        u.append([u_first] + uo); uh.append(uhs)
    if all(u[bi][-1].data_ptr() == cat[B2:, bi * Dw:].data_ptr() for bi in range(nb)):
        ucat = cat[B2:]                                      # every branch tangent was written in place
    else:
        ucat = m.empty(B, nb * Dw, dev)
        for bi in range(nb):
            ucat[:, bi * Dw:(bi + 1) * Dw].copy_(u[bi][-1][:, :Dw])
    if (_top_shapes(m, Lm, Mb, Lo, nb, Dw) and ops.top_tangent_ok(B, Lm.N, 512, ucat, (m0[B2:], mh[B2:], m1[B2:]))
            and ucat.data_ptr() != m0[B2:].data_ptr()):
        # merge layer and merge block of the tangent sweep in ONE launch, in place over the interpolated rows of m0, mh, m1
        # (it runs beside sweep 4's side-stream part; holding it to the CUs that part leaves free -- 96 / 128 / 160 workgroups -- was
        # measured: no difference, 6.02-6.04 ms per iteration)
        um0, umh, um1 = ops.critic_top_tangent(ucat, m0[B2:], mh[B2:], m1[B2:], A._w_nt(Lm.W, 512, m.prec), A._w_nt(Mb.fc1.W, 112, m.prec),
                                               A._w_nt(Mb.fc2.W, 112, m.prec), Lm.N, RELU, 0.0)
    else:
        um0 = Lm.tan(m, ucat, m0[B2:], inplace=True)
        umh, um1 = Mb.tan(m, um0, mh[B2:], m1[B2:])
    # ---- 4. weight / bias gradients (the interpolated rows' part where the real / fake rows' part is already under way)
    if split:
        for bi, br in enumerate(branches):
            br.first.grads_part(m, g2[bi][0][B2:], T[bi], False)
            for i, blk in enumerate(br.blocks):
                blk.fc1.grads_part(m, g1[bi][i][B2:], u[bi][i], False)
                blk.fc2.grads_part(m, g2[bi][i + 1][B2:], uh[bi][i], False)
        Lm.grads_part(m, gz_m0[B2:], ucat, False)
        Mb.fc1.grads_part(m, gz_m1[B2:], um0, False); Mb.fc2.grads_part(m, gz_m2[B2:], umh, False)
        Lo.grads_part(m, gzo[B2:], um1, False)
    else:
        for bi, br in enumerate(branches):
            br.first.grads(m, g2[bi][0], F[bi], B2, T[bi])
            for i, blk in enumerate(br.blocks):
                blk.fc1.grads(m, g1[bi][i], y[bi][i], B2, u[bi][i])
                blk.fc2.grads(m, g2[bi][i + 1], h[bi][i], B2, uh[bi][i])
        Lm.grads(m, gz_m0, cat, B2, ucat)
        Mb.fc1.grads(m, gz_m1, m0, B2, um0); Mb.fc2.grads(m, gz_m2, mh, B2, umh)
        Lo.grads(m, gzo, m1, B2, um1, bias_is_zero=m.bf16)
    m.flush()
    return _finish(optimizerD, logits, pen, n_pen, lam, rows=B)


def step_d3(D, optimizerD, real, fake, alpha, lam, prec=None):
    """Fk_3D_Discriminator (R/models_Fk_GAN/Fk_discriminator.py:149-201).  real, fake (B,16,3)|(B,48) root-relative."""
    m = _Math(prec or D.precision)
    br = [_Branch(D.special_KCS_previous[0], (D.special_KCS_block1, D.special_KCS_block2, D.special_KCS_block3)),
          _Branch(D.previous[0], (D.block1, D.block2, D.block3))]
    optimizerD.zero_grad()
    X = ops.gp_assemble(real, fake, alpha, bf16_rows=m.bf16 and SEED_CASTS)  # (3B,48)
    B = X.shape[0] // 3
    xh = X[2 * B:]
    from . import fused
    use = m.bf16 and FUSED_STEP_FORWARD and fused.step_forward_supported(D)
    kf, kb = ops.kcs_forward(X, True, f32=True, bf16_ld=32 if use else 0)    # fp32 features (first layer's weight gradient) [+ bf16 operand]
    if SEED_CASTS:                                                           # sweep 4's bf16 operands of the two input layers exist already
        m.seed_cast(X[:2 * B], ceil16(X.shape[1]), getattr(X, "_dhaug_bf16_rows", None))
        if use and kb is not None:
            m.seed_cast(kf[:2 * B], 32, kb[:2 * B])
    sr = 2 * B if (use and SKIP_XHAT_SAVES and fused.partial_save_ok(B)) else 0
    return step_branchnet(
        m, optimizerD, br, _Lin(D.merge_previous[0], RELU), _Block(D.merge_block1), _Lin(D.output, NONE), X, B, lam,
        feats=lambda X: [kf, X],
        input_grad=lambda gs: ops.add_f32(ops.kcs_backward(xh, gs[0], True), gs[1]),      # KCS^T path + pose path
        tangents=lambda v: [ops.kcs_jvp(xh, v, True), v],
        fwd=(lambda: fused.critic3d_forward_save(D, X, kb, save_rows=sr)) if use else None,
        penalty=(lambda gs, coef: (lambda r: ([r[0], r[1]], r[2]))(ops.d3_penalty(xh, gs[0], gs[1], coef))) if D3_PENALTY_FUSED else None)


def step_m3(D, optimizerD, real, fake, alpha, lam, prec=None):
    """Video_motion_Fk_3D_Discriminator (R/models_Fk_GAN/Fk_discriminator.py:381-512), stepped with dis_mode='motion':
    real, fake (B, R*48) clips, alpha (B,1), penalty over the B clips.  Branches: per-frame KCS cosines (R*15), their frame
    differences ((R-1)*15), the poses (R*48), their frame differences ((R-1)*48)."""
    m = _Math(prec or D.precision)
    R = D.video_frame_num
    names = ["special_KCS", "diff_special_KCS"] + (["pos_3d"] if D.use_pos else []) + (["diff_pos_3d"] if D.use_diff else [])
    br = [_Branch(getattr(D, n + "_previous")[0], [getattr(D, "%s_block%d" % (n, i)) for i in (1, 2, 3)]) for n in names]
    optimizerD.zero_grad()
    X = ops.gp_assemble(real, fake, alpha)                                   # (3B, R*48)
    B = X.shape[0] // 3
    xh = X[2 * B:].reshape(B * R, 48)

    def feats(X):
        kc = ops.kcs_forward(X.reshape(-1, 48), False, f32=True)[0].reshape(-1, R * 15)
        out = [kc, ops.frame_diff(kc, R, 15)]
        if D.use_pos:
            out.append(X)
        if D.use_diff:
            out.append(ops.frame_diff(X, R, 48))
        return out

    def input_grad(gs):
        gk = ops.add_f32(gs[0], ops.frame_diff(gs[1], R, 15, adjoint=True))   # cotangent of the per-frame cosines
        g = ops.kcs_backward(xh, gk.reshape(B * R, 15), False).reshape(B, R * 48)
        i = 2
        if D.use_pos:
            g = ops.add_f32(g, gs[i]); i += 1
        if D.use_diff:
            g = ops.add_f32(g, ops.frame_diff(gs[i], R, 48, adjoint=True))
        return g

    def tangents(v):
        tk = ops.kcs_jvp(xh, v.reshape(B * R, 48), False).reshape(B, R * 15)
        out = [tk, ops.frame_diff(tk, R, 15)]
        if D.use_pos:
            out.append(v)
        if D.use_diff:
            out.append(ops.frame_diff(v, R, 48))
        return out

    return step_branchnet(m, optimizerD, br, _Lin(D.kcs_merge_previous[0], RELU), _Block(D.kcs_merge_block1),
                          _Lin(D.kcs_output, NONE), X, B, lam, feats, input_grad, tangents)


def step_m2(D, optimizerD, real, fake, alpha, lam, prec=None):
    """Video_motion_Fk_2D_Discriminator (R/models_Fk_GAN/Fk_discriminator.py:516-587) as the video loop steps it: the DEFAULT
    mode of train_Fk_discriminator, i.e. real, fake (B*R, 32) frames, alpha (B*R, 1) -- one interpolation coefficient per
    frame -- and the penalty over the B*R per-frame gradient norms (R/models_Fk_GAN/video_GAN_fun.py:341-346).  Branches: the
    clip's 2D poses (R*32) and the frame differences of its root joint ((R-1)*2)."""
    m = _Math(prec or D.precision)
    R = D.video_frame_num
    br = [_Branch(getattr(D, n + "_previous")[0], [getattr(D, "%s_block%d" % (n, i)) for i in (1, 2, 3)])
          for n in ("pos_2d", "root_diff_2d")]
    optimizerD.zero_grad()
    Xf = ops.gp_assemble(real, fake, alpha)                                  # (3*B*R, 32): interpolated per frame
    B = Xf.shape[0] // (3 * R)
    X = Xf.reshape(3 * B, R * 32)                                            # clips (real, fake and interpolated frames stay together)
    return step_branchnet(
        m, optimizerD, br, _Lin(D.merge_previous[0], RELU), _Block(D.merge_block1), _Lin(D.merge_output, NONE), X, B, lam,
        feats=lambda X: [X, ops.frame_diff(X, R, 32, 2)],
        input_grad=lambda gs: ops.add_f32(gs[0], ops.frame_diff(gs[1], R, 32, 2, adjoint=True)),
        tangents=lambda v: [v, ops.frame_diff(v, R, 32, 2)],
        pen_view=(B * R, 32))


def supported(model_dis, optimizerD, real, fake, rows):
    """The explicit schedule covers the four critics under a FusedAdam bucket on one real + fake batch, with `rows` =
    BATCH_SIZE of calc_gradient_penalty: (rows,48) / (rows,32) for the single-frame critics, (rows, R*48) clips for the 3D
    motion critic (dis_mode='motion'), (rows,32) FRAMES with rows a multiple of R for the 2D motion critic (default mode)."""
    from .models_Fk_GAN.Fk_discriminator import (Fk_2D_Discriminator, Fk_3D_Discriminator, Video_motion_Fk_2D_Discriminator,
                                                 Video_motion_Fk_3D_Discriminator)
    from .optim import FusedAdam
    if not isinstance(optimizerD, FusedAdam) or real.shape != fake.shape or not real.is_cuda or rows < 1:
        return False
    if real.numel() % rows:
        return False
    w = real.numel() // rows
    t = type(model_dis)
    if t is Fk_3D_Discriminator:
        ok = w == 48
    elif t is Fk_2D_Discriminator:
        ok = w == 32
    elif t is Video_motion_Fk_3D_Discriminator:
        ok = w == 48 * model_dis.video_frame_num
    elif t is Video_motion_Fk_2D_Discriminator:
        ok = w == 32 and rows % model_dis.video_frame_num == 0
    else:
        ok = False
    return ok and model_dis.precision in ("bf16", "bf16x3", "bf16x6", "f16x3") and all(p.requires_grad for p in model_dis.parameters())


def critic_step(model_dis, optimizerD, real, fake, alpha, lam):
    """real, fake: (rows, W) with rows = the BATCH_SIZE calc_gradient_penalty is called with (one row per penalty term)"""
    from .models_Fk_GAN.Fk_discriminator import (Fk_3D_Discriminator, Video_motion_Fk_2D_Discriminator,
                                                 Video_motion_Fk_3D_Discriminator)
    from .models_Fk_GAN.Fk_generator import graph_precision
    prec = graph_precision(model_dis.precision)
    if isinstance(model_dis, Video_motion_Fk_3D_Discriminator):
        return step_m3(model_dis, optimizerD, real, fake, alpha, lam, prec)
    if isinstance(model_dis, Video_motion_Fk_2D_Discriminator):
        return step_m2(model_dis, optimizerD, real, fake, alpha, lam, prec)
    if isinstance(model_dis, Fk_3D_Discriminator):
        return step_d3(model_dis, optimizerD, real.reshape(-1, 48), fake.reshape(-1, 48), alpha, lam, prec)
    return step_d2(model_dis, optimizerD, real.reshape(-1, 32), fake.reshape(-1, 32), alpha, lam, prec)
