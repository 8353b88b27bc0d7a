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
import torch

from . import autograd_ops as A
from . import ops

BF16 = torch.bfloat16
NONE, RELU, LRELU = A.ACT_NONE, A.ACT_RELU, A.ACT_LRELU
ceil16 = A.ceil16


class _Math:
    """the three products of a layer in one arithmetic; activations are bf16 (M, ceil16 n) in 'bf16', fp32 (M, n) otherwise"""

    def __init__(self, prec):
        self.prec, self.bf16 = prec, prec == "bf16"
        self.T = 1 if self.bf16 else A.TERMS[prec]

    def width(self, n):
        return ceil16(n) if self.bf16 else n

    def empty(self, M, n, dev):
        return torch.empty((M, self.width(n)), dtype=BF16 if self.bf16 else torch.float32, device=dev)

    def _a(self, a, k):
        """activation-side operand of a (fp32 (M,k) network input / bf16 hidden / fp32 hidden)"""
        if self.bf16:
            return a if a.dtype == BF16 else ops.cast_pad_bf16(a, ceil16(k))
        return ops.split_bf16(a if a.is_contiguous() else a.contiguous(), 0, self.T, ceil16(k))

    def mm(self, a, W, orient, bias=None, res=None, act=NONE, slope=0.0, mask=None, mask_act=NONE, out=None, out_f32=False):
        """(a @ W^T if orient == 'nt' else a @ W) + bias + res, then act(.) or, with `mask`, * mask_act'(mask)."""
        N, K = W.shape
        n, k = (N, K) if orient == "nt" else (K, N)
        kp = ceil16(k)
        Bop = A._w_nt(W, kp, self.prec) if orient == "nt" else A._w_nn(W, self.prec)
        a_op = self._a(a, k)
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
        direct = out is not None and not (masked and not out.is_contiguous())
        _, cf = ops.gemm_nt(a_op, Bop, n, self.T * kp, bias=bias, res_f32=res, act=act, slope=slope, out_f32=True,
                            c_f32=out if direct else None)
        if masked:
            ops.act_backward(cf, mask if mask.is_contiguous() else mask.contiguous(), mask_act, slope, out=cf)
        if out is not None and not direct:
            out.copy_(cf)
            return out
        return cf

    def fusable(self, n, k, rows):
        """shapes whose mask (and skip) ride the GEMM epilogue (gemm_nt256s_kernel): the output may then overwrite the mask"""
        return self.bf16 and n == 256 and k in (128, 256) and rows % 64 == 0

    def outer(self, g, x, N, K, wslot, bslot=None, colsum_rows=None):
        """wslot (N,K) += g^T x;  bslot (N) += column sums of g over rows [0, colsum_rows) (all rows by default; through the
        pairing column-sum kernel when N < 16)"""
        if self.bf16:
            gb = g if g.dtype == BF16 else ops.cast_pad_bf16(g, ceil16(N))
            xb = x if x.dtype == BF16 else ops.cast_pad_bf16(x, ceil16(K))
            narrow = N < 16
            ops.gemm_tn(gb, xb, N, K, colsum=bslot if (bslot is not None and not narrow) else None, out=wslot, accumulate=True,
                        colsum_rows=colsum_rows)
            if bslot is not None and narrow:
                ops.colsum(gb if colsum_rows is None else gb[:colsum_rows], N=N, out=bslot, accumulate=True)
            return
        A._raw_outer(g if g.is_contiguous() else g.contiguous(), x if x.is_contiguous() else x.contiguous(), N, K, self.prec,
                     out=wslot)
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

    def grads(self, m, gz, x, B2, u):
        """dW += gz[:2B]^T x[:2B] + gz[2B:]^T u,  db += colsum(gz[:2B])"""
        if (m.bf16 and x.dtype == BF16 and u.dtype == BF16 and B2 % 128 == 0 and u.data_ptr() == x[B2:].data_ptr()
                and u.stride(0) == x.stride(0)):
            # the tangent was written over the interpolated rows of x: one launch contracts all 3B rows
            m.outer(gz, x, self.N, self.K, _slot(self.W), _slot(self.b), colsum_rows=B2)
            return
        m.outer(gz[:B2], x[:B2], self.N, self.K, _slot(self.W), _slot(self.b))
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
        gz1 = self.fc2.bwd(m, gz2, h, RELU, 0.0)
        return gz1, self.fc1.bwd(m, gz1, x, x_act, 0.0, skip=gz2, out=out)

    def tan(self, m, u, h, y):
        uh = self.fc1.tan(m, u, h, inplace=True)
        return uh, self.fc2.tan(m, uh, y, skip=u, inplace=True)


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
        _SEED_CACHE[key] = _seeds(B, m, dev)
    return _SEED_CACHE[key]


def _finish(optimizerD, logits, pen, B, lam):
    sc = ops.critic_scalars(logits, pen, B, lam)
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
    X = ops.gp_assemble(real, fake, alpha)
    B = X.shape[0] // 3
    B2 = 2 * B
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
    gz2 = L[2].bwd(m, gz3, d2, LRELU, s)
    gz1 = L[1].bwd(m, gz2, d1, LRELU, s, skip=gz3)
    g = L[0].bwd(m, gz1[B2:], None, NONE, 0.0, out_f32=True)                 # (B,32) fp32: dD/dx_hat
    v, pen = ops.gp_penalty(g, 2.0 * lam / B)
    u1 = L[0].tan(m, v, d1[B2:])
    u2 = L[1].tan(m, u1, d2[B2:], inplace=True)
    u3 = L[2].tan(m, u2, d3[B2:], skip=u1, inplace=True)
    u4 = L[3].tan(m, u3, d4[B2:], inplace=True)
    ul = L[4].tan(m, u4, dl[B2:], inplace=True)
    for lay, gz, x, u in ((L[0], gz1, X, v), (L[1], gz2, d1, u1), (L[2], gz3, d2, u2), (L[3], gz4, d3, u3),
                          (L[4], gzl, d4, u4), (L[5], gzp, dl, ul)):
        lay.grads(m, gz, x, B2, u)
    return _finish(optimizerD, logits, pen, B, lam)


def step_d3(D, optimizerD, real, fake, alpha, lam, prec=None):
    """Fk_3D_Discriminator (R/models_Fk_GAN/Fk_discriminator.py:149-201).  real, fake (B,16,3)|(B,48) root-relative."""
    m = _Math(prec or D.precision)
    dev = real.device
    Lk, Lp = _Lin(D.special_KCS_previous[0], RELU), _Lin(D.previous[0], RELU)
    Kb = [_Block(b) for b in (D.special_KCS_block1, D.special_KCS_block2, D.special_KCS_block3)]
    Pb = [_Block(b) for b in (D.block1, D.block2, D.block3)]
    Lm, Mb, Lo = _Lin(D.merge_previous[0], RELU), _Block(D.merge_block1), _Lin(D.output, NONE)
    Dw = Lk.N
    optimizerD.zero_grad()
    X = ops.gp_assemble(real, fake, alpha)                                   # (3B,48)
    B = X.shape[0] // 3
    B2, M3 = 2 * B, 3 * B
    Kf = ops.kcs_forward(X, True, f32=True)[0]                                # (3B,30) fp32
    # ---- 1. forward (the two branch outputs land side by side: the concatenation is a buffer, not a copy)
    cat = m.empty(M3, 2 * Dw, dev)
    k = [Lk.fwd(m, Kf)]
    kh = []
    for i, b in enumerate(Kb):
        h = b.fc1.fwd(m, k[-1])
        kh.append(h)
        k.append(b.fc2.fwd(m, h, res=k[-1], out=cat[:, :Dw] if i == 2 else None))
    p = [Lp.fwd(m, X)]
    ph = []
    for i, b in enumerate(Pb):
        h = b.fc1.fwd(m, p[-1])
        ph.append(h)
        p.append(b.fc2.fwd(m, h, res=p[-1], out=cat[:, Dw:] if i == 2 else None))
    m0 = Lm.fwd(m, cat)
    mh, m1 = Mb.fwd(m, m0)
    logits = Lo.fwd(m, m1, out_f32=True)
    # ---- 2. backward chain
    gzo = seeds(B, m, dev)
    gz_m2 = Lo.bwd(m, gzo, m1, RELU, 0.0)
    gz_m1, gz_m0 = Mb.bwd(m, gz_m2, mh, m0)
    gcat = Lm.bwd(m, gz_m0, cat, RELU, 0.0)                                  # (3B, 2D): cotangents at both branches' last fc2
    gk2, gp2 = [None] * 3 + [gcat[:, :Dw]], [None] * 3 + [gcat[:, Dw:]]       # index i: cotangent at the pre-activation producing k[i]
    gk1, gp1 = [None] * 3, [None] * 3
    for i in (2, 1, 0):
        gk1[i], gk2[i] = Kb[i].bwd(m, gk2[i + 1], kh[i], k[i])
        gp1[i], gp2[i] = Pb[i].bwd(m, gp2[i + 1], ph[i], p[i])
    xh = X[B2:]
    g_feat = Lk.bwd(m, gk2[0][B2:], None, NONE, 0.0, out_f32=True)           # (B,30)
    g_kcs = ops.kcs_backward(xh, g_feat, True)                               # (B,48): the KCS^T path
    g = Lp.bwd(m, gp2[0][B2:], None, NONE, 0.0, skip=g_kcs, out_f32=True)    # dD/dx_hat = pose path + KCS^T path
    # ---- 3. penalty and tangent sweep (x_hat rows)
    v, pen = ops.gp_penalty(g, 2.0 * lam / B)
    tk = ops.kcs_jvp(xh, v, True)                                            # (B,30)
    uk, up = [Lk.tan(m, tk, k[0][B2:])], [Lp.tan(m, v, p[0][B2:])]
    ukh, uph = [], []
    for i in range(3):
        h, y = Kb[i].tan(m, uk[-1], kh[i][B2:], k[i + 1][B2:])
        ukh.append(h); uk.append(y)
        h, y = Pb[i].tan(m, up[-1], ph[i][B2:], p[i + 1][B2:])
        uph.append(h); up.append(y)
    if uk[3].data_ptr() == cat[B2:].data_ptr() and up[3].data_ptr() == cat[B2:, Dw:].data_ptr():
        ucat = cat[B2:]                                                      # both branch tangents were written in place
    else:
        ucat = m.empty(B, 2 * Dw, dev)
        ucat[:, :Dw].copy_(uk[3][:, :Dw]); ucat[:, Dw:2 * Dw].copy_(up[3][:, :Dw])
    um0 = Lm.tan(m, ucat, m0[B2:])
    umh, um1 = Mb.tan(m, um0, mh[B2:], m1[B2:])
    # ---- 4. weight / bias gradients
    Lk.grads(m, gk2[0], Kf, B2, tk)
    Lp.grads(m, gp2[0], X, B2, v)
    for i in range(3):
        Kb[i].fc1.grads(m, gk1[i], k[i], B2, uk[i]); Kb[i].fc2.grads(m, gk2[i + 1], kh[i], B2, ukh[i])
        Pb[i].fc1.grads(m, gp1[i], p[i], B2, up[i]); Pb[i].fc2.grads(m, gp2[i + 1], ph[i], B2, uph[i])
    Lm.grads(m, gz_m0, cat, B2, ucat)
    Mb.fc1.grads(m, gz_m1, m0, B2, um0); Mb.fc2.grads(m, gz_m2, mh, B2, umh)
    Lo.grads(m, gzo, m1, B2, um1)
    return _finish(optimizerD, logits, pen, B, lam)


def supported(model_dis, optimizerD, real, fake):
    """the explicit schedule covers the two single-frame critics under a FusedAdam bucket on one real + fake batch"""
    from .models_Fk_GAN.Fk_discriminator import Fk_2D_Discriminator, Fk_3D_Discriminator
    from .optim import FusedAdam
    if not isinstance(optimizerD, FusedAdam) or real.shape != fake.shape or not real.is_cuda:
        return False
    if type(model_dis) not in (Fk_2D_Discriminator, Fk_3D_Discriminator):
        return False
    return model_dis.precision in ("bf16", "bf16x3", "bf16x6", "f16x3") and all(p.requires_grad for p in model_dis.parameters())


def critic_step(model_dis, optimizerD, real, fake, alpha, lam):
    from .models_Fk_GAN.Fk_discriminator import Fk_3D_Discriminator
    from .models_Fk_GAN.Fk_generator import graph_precision
    prec = graph_precision(model_dis.precision)
    if isinstance(model_dis, Fk_3D_Discriminator):
        return step_d3(model_dis, optimizerD, real.reshape(-1, 48), fake.reshape(-1, 48), alpha, lam, prec)
    return step_d2(model_dis, optimizerD, real.reshape(-1, 32), fake.reshape(-1, 32), alpha, lam, prec)
