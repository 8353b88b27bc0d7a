"""torch.autograd plumbing over the HIP kernels.

Every Function's backward is itself written with Functions of this file, so gradients of any order exist -- the
WGAN-GP penalty (R/models_Fk_GAN/Fk_discriminator.py:205-231) differentiates the critic's input-gradient with
create_graph=True, and the drop-in critics support exactly that through:

    Linear(x, W)  = x W^T (+ bias + residual, activation)        d/dx -> LinearT,  d/dW -> Outer
    LinearT(g, W) = g W                                          d/dg -> Linear,   d/dW -> Outer
    Outer(g, x)   = g^T x                                        d/dg -> Linear,   d/dx -> LinearT
    ActBwd(g, y)  = g * act'(y)      (ReLU / LeakyReLU: piecewise linear, act'' = 0 a.e.)
    KCS, KCS-VJP, KCS-JVP close the same way (VJP and JVP are each other's adjoint in the cotangent argument).

Precision ("prec"):
    'bf16'   : operands rounded to bf16, fp32 accumulate on MFMA, hidden activations stored as bf16 (zero-padded to a
               multiple of 16 columns so they are directly the next GEMM's operand).  The build's default.
    'bf16x3' : x = hi + lo on the same bf16 MFMA kernel: A' = [hi|hi|lo], B' = [hi|lo|hi], K' = 3K (product terms
               down to 2^-16 relative); activations stay fp32.
    'bf16x6' : x = hi + mid + lo, six product terms, K' = 6K: fp32-grade (2^-24).  Used for the 1e-4 logit parity
               tests against the fp32 reference.
"""
import os

import torch

from . import ops

BF16 = torch.bfloat16
ACT_NONE, ACT_RELU, ACT_LRELU = 0, 1, 2
TERMS = {"bf16x3": 3, "bf16x6": 6}
F16X3_LAYER = "f16x3l"     # forward-only layer arithmetic: IEEE-half pairs on dhaug_gemm_f16x3 (see _raw_linear)
F16X3_PLANES = os.environ.get("DHAUG_F16X3_PLANES", "1") != "0"     # ... with the operand split written by the producing layer's epilogue


def _pow2_width(k):
    """piece width of an operand carried as planes: the power of two >= max(k, 64)"""
    w = 64
    while w < k:
        w *= 2
    return w
# (activation-side segment, weight-side segment) holding (hi, mid|lo ...) for the TN products
_SEG = {3: dict(hi=0, lo=2), 6: dict(hi=0, mid=2, lo=5)}
_PAIRS = {3: (("hi", "hi"), ("hi", "lo"), ("lo", "hi")),
          6: (("hi", "hi"), ("hi", "mid"), ("mid", "hi"), ("mid", "mid"), ("hi", "lo"), ("lo", "hi"))}


def ceil16(v):
    return (v + 15) // 16 * 16


# ---------------------------------------------------------------------------------------------------
# packed-weight cache: bf16 copies in the two orientations the GEMMs need, rebuilt when the fp32 master changes
# ---------------------------------------------------------------------------------------------------
class _Packed:
    __slots__ = ("key", "nt", "nn", "nt3", "nn3")

    def __init__(self):
        self.key = None
        self.nt = {}
        self.nn = None
        self.nt3 = {}
        self.nn3 = None


# Bumped by every optimizer step (the fused Adam kernel writes the masters through raw pointers, which torch's
# version counter does not see); every packed copy made under an older epoch is stale.
DIRECT_WGRAD = True       # LinearFn.backward may accumulate weight/bias gradients straight into FusedAdam's flat bucket
WEIGHT_EPOCH = 0
# Non-zero while a hipGraph is being captured (graphs.GraphedCall): part of every cache key, so that a capture never
# re-uses a packed copy made outside it.  A captured kernel holds the ADDRESS of its operands; a copy cached before the
# capture would neither be refreshed by the replay (its pack kernel is not in the graph) nor stay alive (the cache drops it
# the next time the weights change) -- the second case is a GPU memory fault on replay.
CAPTURE_ID = 0


def bump_weight_epoch():
    """every packed copy of every weight is stale (raw writes through .data: broadcasts, loads)"""
    global WEIGHT_EPOCH
    WEIGHT_EPOCH += 1


def pack_key(W):
    return (W.data_ptr(), W._version, tuple(W.shape), WEIGHT_EPOCH, getattr(W, "_dhaug_epoch", 0), CAPTURE_ID)


def install_packed(W, nt, nn):
    """the optimizer re-packed W itself (dhaug_repack_weights): register the fresh bf16 copies under the current key"""
    ent = _Packed()
    ent.key = pack_key(W)
    ent.nt[(nt.shape[1], "bf16")] = nt
    ent.nn = nn
    W._dhaug_pack = ent


def _pack(W):
    """cache entry for a weight-like fp32 (N,K) tensor.  The entry lives ON the parameter object (an id()-keyed dict
    would alias a freed parameter whose id / address get reused), and is valid for one (storage, version, epoch)."""
    key = pack_key(W)
    if isinstance(W, torch.nn.Parameter):
        ent = getattr(W, "_dhaug_pack", None)
        if ent is not None and ent.key == key:
            return ent
        ent = _Packed()
        ent.key = key
        W._dhaug_pack = ent
        return ent
    ent = _Packed()
    ent.key = key
    return ent


def _w_nt(W, Kp, prec):
    """B operand of x W^T: (N, Kp) bf16, or (N, 3Kp) for bf16x3."""
    ent = _pack(W)
    d = ent.nt if prec == "bf16" else ent.nt3
    key = (Kp, prec)
    if key not in d:
        Wd = W.detach()
        if prec == F16X3_LAYER:
            d[key] = ops.split_f16(Wd, 1, Kp)
        else:
            d[key] = ops.cast_pad_bf16(Wd, Kp) if prec == "bf16" else ops.split_bf16(Wd, 1, TERMS[prec], Kp)
    return d[key]


def _w_nn(W, prec, mode=1):
    """B operand of g W (= g (W^T)^T): (K, Np) bf16, or (K, 3Np).  mode 0: the split in the ACTIVATION-side layout, for a g that
    comes in the weight-side layout (the product pairs are symmetric in the two layouts: critic_step._Math.mm)."""
    ent = _pack(W)
    if prec == "bf16":
        if ent.nn is None:
            ent.nn = ops.cast_transpose_bf16(W.detach())
        return ent.nn
    if ent.nn3 is None:
        ent.nn3 = {}
    if (prec, mode) not in ent.nn3:
        ent.nn3[(prec, mode)] = ops.split_bf16(W.detach().t().contiguous(), mode, TERMS[prec])
    return ent.nn3[(prec, mode)]


def clear_weight_cache():
    bump_weight_epoch()


# ---------------------------------------------------------------------------------------------------
# raw (non-autograd) products
# ---------------------------------------------------------------------------------------------------
def _operand(x, width):
    """activation-side bf16 operand (M, width) of an fp32 (M, <=width) or bf16 (M, width) tensor"""
    if x.dtype == BF16:
        assert x.shape[1] == width, (x.shape, width)
        return x
    return ops.cast_pad_bf16(x, width)


def _raw_linear(x, W, bias, res, act, slope, prec, out_f32, out=None):
    N, K = W.shape
    Kp = ceil16(K)
    if prec == "bf16":
        xb = _operand(x, Kp)
        rb = res if (res is not None and res.dtype == BF16) else None
        rf = res if (res is not None and res.dtype != BF16) else None
        # out: a bf16 column block of a wider buffer (a branch's result written into its place in the concatenation; no column beyond
        # the view's own is touched)
        cb, cf = ops.gemm_nt(xb, _w_nt(W, Kp, prec), N, Kp, bias=bias, res_bf16=rb, res_f32=rf, act=act, slope=slope,
                             out_bf16=not out_f32, n_pad=ceil16(N), out_f32=out_f32, c_bf16=None if out_f32 else out)
        return cf if out_f32 else cb
    assert out is None or (prec == F16X3_LAYER and F16X3_PLANES), "an output view is taken by the bf16 and the f16x3 layer paths only"
    if prec == F16X3_LAYER:
        # the fused parity programs' arithmetic (IEEE-half pairs, three product terms) as a layer GEMM on the ping-pong tiles: forward
        # passes without a graph at widths the fused programs do not cover (DenseDim 1000); a layer whose shape the kernel does not take
        # (the 100-wide merge block, the 1-wide logit layer, K < 48) runs in "bf16x6" -- fp32-grade either way
        xf = x.float() if x.dtype == BF16 else x
        rf = res if (res is None or (res.dtype == torch.float32 and res.stride(-1) == 1)) else res.float().contiguous()
        if Kp < 48:                                  # (a 30- / 32-wide input layer: zero columns up to the kernel's shortest contraction, 3 x 48 --
            Kp = 48                                  # the same products, and the layer writes its result's pieces like the others)
        if ops.gemm_f16x3_ok(N, 3 * Kp, bias, rf) and xf.dim() == 2 and xf.stride(1) == 1:
            if F16X3_PLANES:
                # the split left out of the chain of layers: a layer whose result is wide enough to be the next layer's operand writes the
                # result's two pieces [hi | lo] from its epilogue (piece width: the next power of two -- 1 024 at DenseDim 1000) and hangs
                # them on the tensor it returns; the next layer finds them there (a tensor made any other way has none and is split as before)
                kp2 = _pow2_width(K)
                have = getattr(x, "_dhaug_f16_planes", None)
                if have is not None and (have.shape[1] != 2 * kp2 or have.shape[0] != xf.shape[0]):
                    have = None
                emit = _pow2_width(N) if (N >= 64 and out is None) else 0       # (out: a block of a concatenation -- its consumer splits the whole)
                if have is not None:
                    r = ops.gemm_nt_f16x3_planes(have, _w_nt(W, kp2, prec), N, kp2, True, bias=bias, res_f32=rf, act=act, slope=slope, planes_kp=emit,
                                                 out=out)
                else:
                    r = ops.gemm_nt_f16x3_planes(ops.split_f16(xf, 0, Kp), _w_nt(W, Kp, prec), N, Kp, False, bias=bias, res_f32=rf, act=act,
                                                 slope=slope, planes_kp=emit, out=out)
                if emit:
                    r[0]._dhaug_f16_planes = r[1]
                    return r[0]
                return r
            return ops.gemm_nt_f16x3(ops.split_f16(xf, 0, Kp), _w_nt(W, Kp, prec), N, 3 * Kp, bias=bias, res_f32=rf, act=act, slope=slope)
        if out is not None:                          # (a layer the kernel does not take, e.g. an unaligned bias: computed as usual, then placed)
            out.copy_(_raw_linear(x, W, bias, res, act, slope, "bf16x6", True))
            return out
        prec = "bf16x6"
    T = TERMS[prec]
    x3 = ops.split_bf16(x.float() if x.dtype == BF16 else x, 0, T, Kp)
    _, cf = ops.gemm_nt(x3, _w_nt(W, Kp, prec), N, T * Kp, bias=bias, res_f32=res, act=act, slope=slope, out_f32=True)
    return cf


def _raw_linear_t(g, W, prec, out_f32):
    N, K = W.shape
    Np = ceil16(N)
    if prec == "bf16":
        gb = _operand(g, Np)
        cb, cf = ops.gemm_nt(gb, _w_nn(W, prec), K, Np, out_bf16=not out_f32, n_pad=ceil16(K), out_f32=out_f32)
        return cf if out_f32 else cb
    T = TERMS[prec]
    g3 = ops.split_bf16(g, 0, T, Np)
    _, cf = ops.gemm_nt(g3, _w_nn(W, prec), K, T * Np, out_f32=True)
    return cf


def _raw_outer(g, x, N, K, prec, colsum=None, out=None, split=None):
    if prec == "bf16":
        gb = _operand(g, ceil16(N)) if g.dtype != BF16 else g
        xb = _operand(x, ceil16(K)) if x.dtype != BF16 else x
        return ops.gemm_tn(gb, xb, N, K, colsum=colsum, out=out, accumulate=out is not None)
    Np, Kp, T = ceil16(N), ceil16(K), TERMS[prec]
    M = g.shape[0]
    # ONE contraction over T * M rows: a split is (M, T * pad) with the terms side by side, i.e. -- the same memory -- a
    # (T * M, pad) matrix whose row T m + t holds term t of row m.  With g in the weight-side layout and x in the activation-side
    # layout, row T m + t of the two carries exactly the t-th product pair (dhaug_split_bf16: [hi|mid|hi|mid|lo|hi] against
    # [hi|hi|mid|mid|hi|lo]), so sum_{m,t} g''[Tm+t]^T x''[Tm+t] is the six-term (three-term) product.  (Before: one contraction
    # per pair on column blocks of two activation-side splits -- six launches, six partial sums.)
    if split is not None:                      # (the caller holds the operands' splits already: critic_step._Math)
        g1, x3 = split
    else:
        g1 = ops.split_bf16(g, 1, T, Np)
        x3 = ops.split_bf16(x, 0, T, Kp)
    return ops.gemm_tn(g1.view(T * M, Np), x3.view(T * M, Kp), N, K, out=out, accumulate=out is not None)


# ---------------------------------------------------------------------------------------------------
# dense-layer Functions
# ---------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + bias + res).  x: fp32 (M,K) network input or bf16 (M,ceil16 K) hidden activation."""

    @staticmethod
    def forward(ctx, x, W, bias, res, act, slope, prec, out_f32, out=None):
        y = _raw_linear(x, W, bias, res, act, slope, prec, out_f32, out)
        ctx.save_for_backward(x, W, y if act != ACT_NONE else None)
        ctx.cfg = (act, slope, prec, bias is not None, None if res is None else (res.dtype, res.shape))
        # gradient slots of a FusedAdam flat bucket (set by the optimizer), see backward
        ctx.slots = (getattr(W, "_dhaug_grad_slot", None), getattr(bias, "_dhaug_grad_slot", None) if bias is not None else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, W, y = ctx.saved_tensors
        act, slope, prec, has_bias, res_info = ctx.cfg
        N, K = W.shape
        gz = ActBwdFn.apply(gy.contiguous(), y, act, slope) if act != ACT_NONE else gy.contiguous()
        gx = gW = gb = gres = None
        if ctx.needs_input_grad[0]:
            gx = LinearTFn.apply(gz, W, prec, x.dtype != BF16)
        want_b = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            wslot, bslot = ctx.slots
            if (DIRECT_WGRAD and prec == "bf16" and not torch.is_grad_enabled() and wslot is not None and W.grad is wslot
                    and (not want_b or bslot is not None)):
                # first-order pass under a FusedAdam bucket: the weight-gradient GEMM accumulates straight into the
                # parameter's slot of the flat gradient (and the bias gradient, A^T * ones on the MFMA pipe, into the
                # bias slot): no zero-fill of a temporary, no AccumulateGrad add.  autograd sees "no gradient" for W/b.
                # (a logit layer's bias gradient goes through the column-sum kernel: it pairs the real and the fake half of
                # the batch, whose cotangents cancel exactly -- see dhaug_colsum_*)
                narrow = N < 16
                _raw_outer(gz, x, N, K, prec, colsum=bslot if (want_b and not narrow) else None, out=wslot)
                if want_b and narrow:
                    ops.colsum(gz, N=N, out=bslot, accumulate=True)
                want_b = False
            elif want_b and prec == "bf16" and not torch.is_grad_enabled() and N >= 16:
                # first-order pass: the weight-gradient GEMM also emits the bias gradient (A^T * ones on the MFMA pipe)
                gb = torch.empty((N,), dtype=torch.float32, device=gz.device)
                gW = _raw_outer(gz, x, N, K, prec, colsum=gb)
                want_b = False
            else:
                gW = OuterFn.apply(gz, x, N, K, prec)
        if want_b:
            gb = ColSumFn.apply(gz, N)
        if res_info is not None and ctx.needs_input_grad[3]:
            gres = gz if (gz.dtype == res_info[0] and gz.shape == res_info[1]) else ReshapeGradFn.apply(gz, res_info)
        return gx, gW, gb, gres, None, None, None, None, None


class LinearTFn(torch.autograd.Function):
    """out = g W  (input gradient of Linear).  g: bf16 (M, ceil16 N) or fp32 (M,N)."""

    @staticmethod
    def forward(ctx, g, W, prec, out_f32):
        ctx.save_for_backward(g, W)
        ctx.cfg = (prec, g.dtype != BF16)
        return _raw_linear_t(g, W, prec, out_f32 or prec != "bf16")

    @staticmethod
    def backward(ctx, go):
        g, W = ctx.saved_tensors
        prec, g_f32 = ctx.cfg
        N, K = W.shape
        gg = gW = None
        if ctx.needs_input_grad[0]:
            gg = LinearFn.apply(go.contiguous(), W, None, None, ACT_NONE, 0.0, prec, g_f32 or prec != "bf16")
        if ctx.needs_input_grad[1]:
            gW = OuterFn.apply(g, go.contiguous(), N, K, prec)
        return gg, gW, None, None


class OuterFn(torch.autograd.Function):
    """out (N,K) fp32 = g^T x  (weight gradient of Linear)."""

    @staticmethod
    def forward(ctx, g, x, N, K, prec):
        ctx.save_for_backward(g, x)
        ctx.cfg = (N, K, prec)
        return _raw_outer(g, x, N, K, prec)

    @staticmethod
    def backward(ctx, G):
        g, x = ctx.saved_tensors
        N, K, prec = ctx.cfg
        gg = gx = None
        G = G.contiguous()
        if ctx.needs_input_grad[0]:
            gg = LinearFn.apply(x, G, None, None, ACT_NONE, 0.0, prec, g.dtype != BF16 or prec != "bf16")
        if ctx.needs_input_grad[1]:
            gx = LinearTFn.apply(g, G, prec, x.dtype != BF16)
        return gg, gx, None, None, None


class ActBwdFn(torch.autograd.Function):
    """g * act'(y); linear in g, y enters only through its sign."""

    @staticmethod
    def forward(ctx, g, y, act, slope):
        if g.dtype != y.dtype or g.shape != y.shape:         # fp32 upstream gradient of a bf16 activation
            g = ops.cast_pad_bf16(g, y.shape[1]) if y.dtype == BF16 else g.float()
        ctx.save_for_backward(y)
        ctx.cfg = (act, slope)
        return ops.act_backward(g, y, act, slope)

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        return ActBwdFn.apply(gg.contiguous(), y, ctx.cfg[0], ctx.cfg[1]), None, None, None


class ColSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, N):
        ctx.shape = (g.shape, g.dtype)
        return ops.colsum(g, N=N)

    @staticmethod
    def backward(ctx, gb):
        shape, dtype = ctx.shape
        out = torch.zeros(shape, dtype=dtype, device=gb.device)
        out[:, :gb.shape[0]] = gb.to(dtype)
        return out, None


class ReshapeGradFn(torch.autograd.Function):
    """dtype / padding adapter between a gradient and the tensor it belongs to (fp32 (M,N) <-> bf16 (M,ceil16 N))."""

    @staticmethod
    def forward(ctx, g, info):
        dtype, shape = info
        ctx.info = (g.dtype, g.shape)
        if dtype == BF16:
            return ops.cast_pad_bf16(g.float()[:, :shape[1]].contiguous(), shape[1]) if g.dtype != BF16 else g[:, :shape[1]].contiguous()
        return g.float()[:, :shape[1]].contiguous()

    @staticmethod
    def backward(ctx, gg):
        return ReshapeGradFn.apply(gg, ctx.info), None


class LinearTMaskFn(torch.autograd.Function):
    """out = (g W + res) * act'(ymask) in ONE launch (dhaug_gemm_bf16_dmask), differentiable: the pieces of a residual
    block's input-gradient chain under create_graph (WGAN-GP) without separate activation-backward / add kernels.
    g (M,D) bf16, W (D,D), ymask (M,D) bf16 | None (no mask), res (M,D) bf16 | None.  The backward is composed of the
    differentiable Functions above (t = go * act'(ymask) is what all three gradients start from)."""

    @staticmethod
    def forward(ctx, g, W, ymask, res, act, slope, prec):
        D = W.shape[0]
        ctx.save_for_backward(g, W, ymask)
        ctx.cfg = (act if ymask is not None else ACT_NONE, slope, prec, res is not None)
        return ops.gemm_nt_dmask(g, _w_nn(W, prec), D, D, ymask if ymask is not None else g,
                                 act if ymask is not None else ACT_NONE, slope, res_bf16=res)

    @staticmethod
    def backward(ctx, go):
        g, W, ymask = ctx.saved_tensors
        act, slope, prec, has_res = ctx.cfg
        D = W.shape[0]
        t = ActBwdFn.apply(go.contiguous(), ymask, act, slope) if act != ACT_NONE else go.contiguous()
        gg = gW = gres = None
        if ctx.needs_input_grad[0]:
            gg = LinearFn.apply(t, W, None, None, ACT_NONE, 0.0, prec, False)
        if ctx.needs_input_grad[1]:
            gW = OuterFn.apply(g, t, D, D, prec)
        if has_res and ctx.needs_input_grad[3]:
            gres = t
        return gg, gW, None, gres, None, None, None


FUSED_GP_CHAIN = os.environ.get("DHAUG_NO_FUSED_GP_CHAIN") is None


class ResBlockFn(torch.autograd.Function):
    """y = act(fc2(act(fc1(x))) + x) for bf16 activations of width D (myResNet, R/models_Fk_GAN/special_operate.py:490-510).
    First-order backward in 2 + 2 GEMM launches per block: the hidden layer's activation backward rides the epilogue of
    the GEMM that produces its input gradient, the skip connection is that of the block's input gradient GEMM, and
    weight / bias gradients go where LinearFn puts them.  Under create_graph (WGAN-GP) the backward is the composite
    of the differentiable pieces, exactly what two LinearFn nodes would record."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, act, slope, prec):
        h = _raw_linear(x, W1, b1, None, act, slope, prec, False)
        y = _raw_linear(h, W2, b2, x, act, slope, prec, False)
        ctx.save_for_backward(x, h, y, W1, W2)
        ctx.cfg = (act, slope, prec)
        ctx.slots = tuple(getattr(t, "_dhaug_grad_slot", None) for t in (W1, b1, W2, b2))
        return y

    @staticmethod
    def backward(ctx, gy):
        x, h, y, W1, W2 = ctx.saved_tensors
        act, slope, prec = ctx.cfg
        D = W1.shape[0]
        if torch.is_grad_enabled():                             # differentiable composite (double backward)
            gz2 = ActBwdFn.apply(gy.contiguous(), y, act, slope)
            if FUSED_GP_CHAIN and gz2.dtype == BF16:
                gz1 = LinearTMaskFn.apply(gz2, W2, h, None, act, slope, prec)         # (gz2 W2) * act'(h)
                gx = LinearTMaskFn.apply(gz1, W1, None, gz2, act, slope, prec)        # gz1 W1 + gz2
            else:
                gh = LinearTFn.apply(gz2, W2, prec, False)
                gz1 = ActBwdFn.apply(gh, h, act, slope)
                gx = LinearTFn.apply(gz1, W1, prec, False) + gz2
            gW2 = OuterFn.apply(gz2, h, D, D, prec) if ctx.needs_input_grad[3] else None
            gW1 = OuterFn.apply(gz1, x, D, D, prec) if ctx.needs_input_grad[1] else None
            gb2 = ColSumFn.apply(gz2, D) if ctx.needs_input_grad[4] else None
            gb1 = ColSumFn.apply(gz1, D) if ctx.needs_input_grad[2] else None
            return gx, gW1, gb1, gW2, gb2, None, None, None
        gy = gy.contiguous()
        if gy.dtype != BF16:
            gy = ops.cast_pad_bf16(gy, D)
        gz2 = ops.act_backward(gy, y, act, slope)
        gz1 = ops.gemm_nt_dmask(gz2, _w_nn(W2, prec), D, D, h, act, slope)              # (gz2 W2) * act'(h)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = ops.gemm_nt_dmask(gz1, _w_nn(W1, prec), D, D, y, ACT_NONE, 0.0, res_bf16=gz2)   # gz1 W1 + gz2
        grads = [None, None, None, None]                        # W1, b1, W2, b2
        for k, (gz, inp, W) in enumerate(((gz1, x, W1), (gz2, h, W2))):
            wslot, bslot = ctx.slots[2 * k], ctx.slots[2 * k + 1]
            need_w, need_b = ctx.needs_input_grad[1 + 2 * k], ctx.needs_input_grad[2 + 2 * k]
            if not need_w:
                if need_b:
                    grads[2 * k + 1] = ops.colsum(gz, N=D)
                continue
            if DIRECT_WGRAD and wslot is not None and W.grad is wslot and (not need_b or bslot is not None):
                _raw_outer(gz, inp, D, D, prec, colsum=bslot if need_b else None, out=wslot)
            else:
                gb = torch.empty((D,), dtype=torch.float32, device=gz.device) if need_b else None
                grads[2 * k] = _raw_outer(gz, inp, D, D, prec, colsum=gb)
                grads[2 * k + 1] = gb
        return gx, grads[0], grads[1], grads[2], grads[3], None, None, None


def res_block(x, W1, b1, W2, b2, act=ACT_RELU, slope=0.0, prec="bf16"):
    """myResNet forward; one autograd node when the activations are bf16 hidden states of a 16-aligned width"""
    D = W1.shape[0]
    if (prec == "bf16" and x.dtype == BF16 and x.is_cuda and W1.shape == (D, D) and W2.shape == (D, D) and D % 16 == 0
            and x.shape[1] == D and b1 is not None and b2 is not None):
        return ResBlockFn.apply(x, W1, b1, W2, b2, act, slope, prec)
    h = linear(x, W1, b1, None, act, slope, prec)
    return linear(h, W2, b2, x, act, slope, prec)


def linear(x, W, bias=None, res=None, act=ACT_NONE, slope=0.0, prec="bf16", out_f32=False, out=None):
    """out (bf16 arithmetic, passes without a graph): write the result into this (M, N) column block of a wider bf16 buffer"""
    if out is not None:
        assert prec in ("bf16", F16X3_LAYER) and not torch.is_grad_enabled()
        return LinearFn.apply(x, W, bias, res, act, slope, prec, prec != "bf16", out)
    return LinearFn.apply(x, W, bias, res, act, slope, prec, out_f32 or prec != "bf16")


# ---------------------------------------------------------------------------------------------------
# pose features / FK / camera
# ---------------------------------------------------------------------------------------------------
class KcsFn(torch.autograd.Function):
    """special_KCS_Input_transform: (M,48) fp32 -> (M,30|15) fp32."""

    @staticmethod
    def forward(ctx, x, with_lengths):
        x = x.reshape(-1, 48)
        ctx.save_for_backward(x)
        ctx.wl = with_lengths
        return ops.kcs_forward(x, with_lengths, f32=True)[0]

    @staticmethod
    def backward(ctx, gf):
        (x,) = ctx.saved_tensors
        return KcsVjpFn.apply(x, gf.contiguous(), ctx.wl), None


class KcsVjpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gf, with_lengths):
        ctx.save_for_backward(x)
        ctx.wl = with_lengths
        return ops.kcs_backward(x, gf, with_lengths)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        # adjoint in the cotangent argument; the second derivative w.r.t. the pose itself is not provided
        # (no caller differentiates the penalty w.r.t. its interpolation points)
        return None, KcsJvpFn.apply(x, g.contiguous(), ctx.wl), None


class KcsJvpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, with_lengths):
        ctx.save_for_backward(x)
        ctx.wl = with_lengths
        return ops.kcs_jvp(x, t, with_lengths)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return None, KcsVjpFn.apply(x, g.contiguous(), ctx.wl), None


class FkFn(torch.autograd.Function):
    """change_3d_joint_angle + 32->16 gather: angles (N,37), bone_len (N,15), root (N,3) -> (N,16,3)."""

    @staticmethod
    def forward(ctx, angles, bone_len, root):
        ctx.save_for_backward(angles, bone_len)
        return ops.fk_forward(angles, bone_len, root, 16)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        angles, bone_len = ctx.saved_tensors
        ga, gb, gr = ops.fk_backward(angles, bone_len, g.contiguous())
        return ga, gb, gr


class GenTailFn(torch.autograd.Function):
    """Fk_Generator tail: head (N,35), bone_len (N,15), scaler (N,8)|None -> fake (N,16,3)."""

    @staticmethod
    def forward(ctx, head, bone_len, scaler, use_preangle):
        ctx.save_for_backward(head, bone_len, scaler)
        ctx.pre = use_preangle
        return ops.gen_tail_forward(head, bone_len, scaler, use_preangle)[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        head, bone_len, scaler = ctx.saved_tensors
        return ops.gen_tail_backward(head, bone_len, scaler, g.contiguous(), ctx.pre), None, None, None


class CenterFlipFn(torch.autograd.Function):
    """x - x[:, :1] and/or L/R flip of (N,16,C); linear, so its backward is the transposed kernel."""

    @staticmethod
    def forward(ctx, x, center, flip, adjoint):
        ctx.cfg = (center, flip, adjoint)
        return ops.center_flip(x, center, flip, adjoint)

    @staticmethod
    def backward(ctx, g):
        center, flip, adjoint = ctx.cfg
        return CenterFlipFn.apply(g.contiguous(), center, flip, not adjoint), None, None, None


class W2CProjectFn(torch.autograd.Function):
    """GAN_torch_world_to_camera + project_to_2d with one shared camera -> (cam3d (N,16,3), proj2d (N,16,2))."""

    @staticmethod
    def forward(ctx, x, quat, trans, cam9):
        ctx.save_for_backward(x)
        ctx.cam = (quat, trans, cam9)
        c3, p2 = ops.world_to_camera_project(x, quat, trans, cam9)
        return c3, p2

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g3, g2):
        (x,) = ctx.saved_tensors
        quat, trans, cam9 = ctx.cam
        gx = ops.world_to_camera_project_backward(x, quat, trans, cam9, None if g3 is None else g3.contiguous(),
                                                  None if g2 is None else g2.contiguous())
        return gx.reshape(x.shape), None, None, None


def center_flip(x, center=False, flip=False):
    return CenterFlipFn.apply(x, center, flip, False)
