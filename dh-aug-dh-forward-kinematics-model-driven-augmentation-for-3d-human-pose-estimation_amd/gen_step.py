"""The generator step of both epoch loops (R/models_Fk_GAN/model_fk_gan_train.py:415-484, R/models_Fk_GAN/video_GAN_fun.py:421-566)
as an explicit schedule of kernels -- no autograd graph.

    gen_loss = sum_c  w_c * mean D_c(fake)          (flipped / reversed copies averaged in as the reference does)
    (-gen_loss).backward();  optimizer_G.step()

with the critics frozen.  The critics are piecewise linear behind their input features, so d(-gen_loss)/d(fake) is ONE
backward chain per critic seeded with the constant logit cotangent -w_c / rows (critic_step.py's sweep 2 on the fake rows
only, input layers included), then the transposed feature maps (KCS VJP, frame differences, centring, camera + projection
VJP), the FK tail's reverse mode (dhaug_gen_tail_backward) and the generator trunk's backward chain; the trunk's weight
gradients are one grouped contraction launch.  What the reference's copies contribute:
  * L/R-flipped copies (.detach().clone(), :455,:459): value only -- scored by the fused no-grad programs;
  * frame-reversed copies of the motion critics (video loop, `playback`): value AND gradient (torch.flip is differentiable),
    on the view the reference takes them on ((-1, R, 32) for both the 3D and the 2D clip)."""
import torch

from . import autograd_ops as A
from . import critic_step as CS
from . import ops

NONE, RELU, LRELU = A.ACT_NONE, A.ACT_RELU, A.ACT_LRELU
BF16 = torch.bfloat16


def _const_seed(rows, value, m, dev):
    s = torch.full((rows, 1), float(value), dtype=torch.float32, device=dev)
    return ops.cast_pad_bf16(s, 16) if m.bf16 else s


class _BranchNet:
    """cat_b(branch_b(feat_b(x))) -> Linear(100)+ReLU -> myResNet(100) -> Linear(1): forward keeping every activation, and
    the input-gradient chain of a constant logit cotangent (weights frozen: no weight gradients)."""

    def __init__(self, branches, Lm, Mb, Lo):
        self.br, self.Lm, self.Mb, self.Lo = branches, Lm, Mb, Lo

    def forward(self, m, F, fwd=None):
        if fwd is not None:
            return fwd()
        rows, dev = F[0].shape[0], F[0].device
        nb, Dw = len(self.br), self.br[0].first.N
        cat = m.empty(rows, nb * Dw, dev)
        y, h = [], []
        for bi, br in enumerate(self.br):
            ys, hs = [br.first.fwd(m, F[bi])], []
            for i, blk in enumerate(br.blocks):
                hh = blk.fc1.fwd(m, ys[-1])
                hs.append(hh)
                ys.append(blk.fc2.fwd(m, hh, res=ys[-1], out=cat[:, bi * Dw:(bi + 1) * Dw] if i == len(br.blocks) - 1 else None))
            y.append(ys); h.append(hs)
        m0 = self.Lm.fwd(m, cat)
        mh, m1 = self.Mb.fwd(m, m0)
        logits = self.Lo.fwd(m, m1, out_f32=True)
        return dict(cat=cat, y=y, h=h, m0=m0, mh=mh, m1=m1, logits=logits)

    def input_grads(self, m, s, seed):
        """s: forward(); seed (rows,1) logit cotangent -> one fp32 input cotangent per branch"""
        Dw = self.br[0].first.N
        cat = s["cat"]
        if Dw == 256 and all(getattr(s["y"][bi][-1], "_dhaug_bits", None) is not None for bi in range(len(self.br))):
            cat._dhaug_bits_cols = [s["y"][bi][-1]._dhaug_bits for bi in range(len(self.br))]
        if CS._top_fusable(m, self.Lm, self.Mb, self.Lo, len(self.br), Dw, cat.shape[0], (s["m1"], s["mh"], s["m0"]), cat):
            # (merge layer, merge block and logit layer in one launch, as in the critic step: ops.critic_top_backward)
            _, _, _, gcat = ops.critic_top_backward(
                seed, CS.A._w_nn(self.Lo.W, m.prec)[:, 0], s["m1"], s["mh"], s["m0"], CS.A._w_nn(self.Mb.fc2.W, m.prec),
                CS.A._w_nn(self.Mb.fc1.W, m.prec), CS.A._w_nn(self.Lm.W, m.prec), cat._dhaug_bits_cols, self.Lm.N, RELU, 0.0)
        else:
            gz_m2 = self.Lo.bwd(m, seed, s["m1"], RELU, 0.0)
            _, gz_m0 = self.Mb.bwd(m, gz_m2, s["mh"], s["m0"])
            gcat = self.Lm.bwd(m, gz_m0, cat, RELU, 0.0, out=m.empty_blocks(cat.shape[0], len(self.br), Dw, cat.device))
        gin = []
        for bi, br in enumerate(self.br):
            _, a2 = CS.stack_bwd(m, br.blocks, gcat[:, bi * Dw:(bi + 1) * Dw], s["h"][bi], s["y"][bi])
            gin.append(br.first.bwd(m, a2[0], None, NONE, 0.0, out_f32=True))
        return gin


def _d3_net(D):
    return _BranchNet([CS._Branch(D.special_KCS_previous[0], (D.special_KCS_block1, D.special_KCS_block2, D.special_KCS_block3)),
                       CS._Branch(D.previous[0], (D.block1, D.block2, D.block3))],
                      CS._Lin(D.merge_previous[0], RELU), CS._Block(D.merge_block1), CS._Lin(D.output, NONE))


def _d3_value_grad(m, D, fc, coef):
    """fc (N,48) root-relative fakes -> (logits (N,1), coef * d mean D3 / d fc  (N,48) fp32)"""
    from . import fused
    net = _d3_net(D)
    N = fc.shape[0]
    use = m.bf16 and CS.FUSED_STEP_FORWARD and fused.step_forward_supported(D)
    kf, kb = ops.kcs_forward(fc, True, f32=True, bf16_ld=32 if use else 0)
    sr = -1 if (CS.SKIP_XHAT_SAVES and fused.partial_save_ok(N)) else 0          # (only masks are read back: the block layers leave their sign bits)
    s = net.forward(m, [kf, fc], fwd=(lambda: fused.critic3d_forward_save(D, fc, kb, save_rows=sr)) if use else None)
    gk, gp = net.input_grads(m, s, _const_seed(N, coef / N, m, fc.device))
    return s["logits"], ops.add_f32(ops.kcs_backward(fc, gk, True), gp)


def _d2_value_grad(m, D, x, coef):
    """x (N,32) projections -> (logits, coef * d mean D2 / d x (N,32) fp32).  R/models_Fk_GAN/Fk_discriminator.py:253-266"""
    from . import fused
    sl = D.slope
    L = [CS._Lin(D.pose_layer_1, LRELU, sl), CS._Lin(D.pose_layer_2, LRELU, sl), CS._Lin(D.pose_layer_3, LRELU, sl),
         CS._Lin(D.pose_layer_4, NONE), CS._Lin(D.layer_last, LRELU, sl), CS._Lin(D.layer_pred, NONE)]
    N = x.shape[0]
    if m.bf16 and CS.FUSED_STEP_FORWARD and fused.step_forward_supported(D):
        r = fused.critic2d_forward_save(D, x, save_rows=-1 if (CS.SKIP_XHAT_SAVES and fused.partial_save_ok(N)) else 0)
        (d1, d2, d3, d4, dl), logits = r["d"], r["logits"]
    else:
        d1 = L[0].fwd(m, x)
        d2 = L[1].fwd(m, d1)
        d3 = L[2].fwd(m, d2, res=d1)
        d4 = L[3].fwd(m, d3)
        dl = L[4].fwd(m, d4)
        logits = L[5].fwd(m, dl, out_f32=True)
    gzl = L[5].bwd(m, _const_seed(N, coef / N, m, x.device), dl, LRELU, sl)
    gz4 = L[4].bwd(m, gzl, d4, NONE, 0.0)
    gz3 = L[3].bwd(m, gz4, d3, LRELU, sl)
    gz2 = L[2].bwd(m, gz3, d2, LRELU, sl)
    gz1 = L[1].bwd(m, gz2, d1, LRELU, sl, skip=gz3)
    return logits, L[0].bwd(m, gz1, None, NONE, 0.0, out_f32=True)


def _m3_value_grad(m, D, fc, coef):
    """3D motion critic on clips: fc (B*R,48) -> (logits (B,1), coef * d mean / d fc (B*R,48)).  Branch features as in
    critic_step.step_m3 (R/models_Fk_GAN/Fk_discriminator.py:381-512)."""
    R = D.video_frame_num
    names = ["special_KCS", "diff_special_KCS"] + (["pos_3d"] if D.use_pos else []) + (["diff_pos_3d"] if D.use_diff else [])
    net = _BranchNet([CS._Branch(getattr(D, n + "_previous")[0], [getattr(D, "%s_block%d" % (n, i)) for i in (1, 2, 3)]) for n in names],
                     CS._Lin(D.kcs_merge_previous[0], RELU), CS._Block(D.kcs_merge_block1), CS._Lin(D.kcs_output, NONE))
    X = fc.reshape(-1, R * 48)
    B = X.shape[0]
    kc = ops.kcs_forward(fc.reshape(-1, 48), False, f32=True)[0].reshape(B, R * 15)
    F = [kc, ops.frame_diff(kc, R, 15)]
    if D.use_pos:
        F.append(X)
    if D.use_diff:
        F.append(ops.frame_diff(X, R, 48))
    s = net.forward(m, F)
    gs = net.input_grads(m, s, _const_seed(B, coef / B, m, fc.device))
    gk = ops.add_f32(gs[0], ops.frame_diff(gs[1], R, 15, adjoint=True))
    g = ops.kcs_backward(fc.reshape(-1, 48), gk.reshape(B * R, 15), False).reshape(B, R * 48)
    i = 2
    if D.use_pos:
        g = ops.add_f32(g, gs[i]); i += 1
    if D.use_diff:
        g = ops.add_f32(g, ops.frame_diff(gs[i], R, 48, adjoint=True))
    return s["logits"], g.reshape(B * R, 48)


def _m2_value_grad(m, D, x, coef):
    """2D motion critic: x (B*R,32) -> (logits (B,1), coef * d mean / d x (B*R,32)).  R/models_Fk_GAN/Fk_discriminator.py:516-587"""
    R = D.video_frame_num
    net = _BranchNet([CS._Branch(getattr(D, n + "_previous")[0], [getattr(D, "%s_block%d" % (n, i)) for i in (1, 2, 3)])
                      for n in ("pos_2d", "root_diff_2d")],
                     CS._Lin(D.merge_previous[0], RELU), CS._Block(D.merge_block1), CS._Lin(D.merge_output, NONE))
    X = x.reshape(-1, R * 32)
    B = X.shape[0]
    s = net.forward(m, [X, ops.frame_diff(X, R, 32, 2)])
    gs = net.input_grads(m, s, _const_seed(B, coef / B, m, x.device))
    g = ops.add_f32(gs[0], ops.frame_diff(gs[1], R, 32, 2, adjoint=True))
    return s["logits"], g.reshape(B * R, 32)


CONCURRENT = __import__("os").environ.get("DHAUG_NO_CONCURRENT_CRITICS") is None
_STREAMS = {}


def side_streams(cur, n):
    """the n side streams of `cur` (created on first use; graphs.prepare_streams makes them before a capture)"""
    pool = _STREAMS.setdefault((cur.device.index, cur.cuda_stream), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream())
    return pool


def _parallel(fns):
    """[fn() for fn in fns], every fn on its own side stream (forked from the current one, joined before returning): the
    critics' value + input-gradient chains are independent of each other and mostly launch-bound at the G step's B rows.
    Tensors that come back were allocated on a side stream and are read on the current one: record_stream tells the
    allocator."""
    if not CONCURRENT or len(fns) < 2:
        return [fn() for fn in fns]
    cur = torch.cuda.current_stream()
    if torch.cuda.is_current_stream_capturing():
        # no stream creation inside a capture, and forks only from the stream the capture was begun on (one level: see
        # graphs.prepare_streams); whatever has no stream runs in line
        from . import critic_step
        pool = _STREAMS.get((cur.device.index, cur.cuda_stream), []) if cur.cuda_stream == critic_step.CAPTURE_ROOT else []
    else:
        pool = side_streams(cur, len(fns))
    out = []
    for i, fn in enumerate(fns):
        if i >= len(pool):
            out.append(fn())
            continue
        st = pool[i]
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            r = fn()
        _record_stream(r, cur)
        out.append(r)
    for st in pool[:len(fns)]:
        cur.wait_stream(st)
    return out


def _record_stream(r, stream):
    """every tensor of a (nested) result: it was allocated on a side stream and will be read on `stream`"""
    if torch.is_tensor(r):
        r.record_stream(stream)
    elif isinstance(r, (tuple, list)):
        for t in r:
            _record_stream(t, stream)
    elif isinstance(r, dict):
        for t in r.values():
            _record_stream(t, stream)


def supported(G, oG, critics):
    """the explicit schedule covers the reference's generators and critics under a FusedAdam bucket, in every arithmetic of
    the layer path"""
    from .models_Fk_GAN.Fk_discriminator import (Fk_2D_Discriminator, Fk_3D_Discriminator, Video_motion_Fk_2D_Discriminator,
                                                 Video_motion_Fk_3D_Discriminator)
    from .models_Fk_GAN.Fk_generator import _GeneratorBase
    from .optim import FusedAdam
    if not isinstance(oG, FusedAdam) or not isinstance(G, _GeneratorBase) or len(critics) not in (2, 4):
        return False
    kinds = (Fk_3D_Discriminator, Fk_2D_Discriminator, Video_motion_Fk_3D_Discriminator, Video_motion_Fk_2D_Discriminator)
    if any(type(c) is not k for c, k in zip(critics, kinds)):
        return False
    ok = ("bf16", "bf16x3", "bf16x6", "f16x3")
    return G.precision in ok and all(c.precision in ok for c in critics) and getattr(G.args, "whether_use_RT", True)


def generator_step(args, G, oG, critics, weights, camera, flip, noise, scaler, frames=1, playback=False):
    """One G step; returns G_cost = -gen_loss (0-dim device tensor).  noise (B,128); scaler: the (B,8) bone-length jitter
    draw or None (drawn on the device like Fk_Generator.forward does)."""
    from .models_Fk_GAN.Fk_generator import graph_precision
    dev = noise.device
    quat, trans, cam9 = camera
    R = frames
    mG = CS._Math(graph_precision(G.precision))
    oG.zero_grad()
    # ---- forward: trunk (every activation kept), FK tail, camera
    Lp = CS._Lin(G.preprocess[0], RELU)
    blocks = [CS._Block(b) for b in (G.block1, G.block2, G.block3)]
    Lh = CS._Lin(G.deconv_out, NONE)
    z = noise.contiguous().float()
    B = z.shape[0]
    N = B * R
    ys, hs = [Lp.fwd(mG, z)], []
    for blk in blocks:
        h, y = blk.fwd(mG, ys[-1])
        hs.append(h); ys.append(y)
    head = Lh.fwd(mG, ys[-1], out_f32=True)                       # (B, 35 R) fp32
    head2 = head.reshape(N, 35)
    bl = G.boneLength
    if bl.shape[0] != N:
        raise RuntimeError("boneLength has %d rows, the batch needs %d (call GAN_generator_get_bone_length)" % (bl.shape[0], N))
    pre = bool(args.GAN_whether_use_preAngle)
    sc = G._scaler(B, scaler)
    fw = ops.gen_tail_forward(head2, bl, sc, pre)[0]             # (N,16,3) world
    G.train_num += 1
    _, f2d = ops.world_to_camera_project(fw, quat, trans, cam9)
    fc = ops.center_flip(fw, True, False).reshape(N, 48)
    x2 = f2d.reshape(N, 32)
    # ---- critics: value + input gradient of -w * mean D(.) (the L/R-flipped copies halve the weight of the plain ones)
    half = 0.5 if flip else 1.0
    m3d, m2d = CS._Math(graph_precision(critics[0].precision)), CS._Math(graph_precision(critics[1].precision))
    jobs = [lambda: _d3_value_grad(m3d, critics[0], fc, -weights[0] * half),
            lambda: _d2_value_grad(m2d, critics[1], x2, -weights[1] * half)]
    wts = [weights[0] * half, weights[1] * half]
    kinds = ["fc", "x2"]
    rev = lambda t: ops.frame_reverse(t.reshape(-1, R * 32), R, 32)
    if len(critics) == 4:
        mm3, mm2 = CS._Math(graph_precision(critics[2].precision)), CS._Math(graph_precision(critics[3].precision))
        ph = 0.5 if playback else 1.0
        jobs += [lambda: _m3_value_grad(mm3, critics[2], fc, -weights[2] * half * ph),
                 lambda: _m2_value_grad(mm2, critics[3], x2, -weights[3] * half * ph)]
        wts += [weights[2] * half * ph, weights[3] * half * ph]
        kinds += ["fc", "x2"]
        if playback:
            # the reference reverses the frames of the clip VIEWED as (-1, R, 32) -- also for the 3D clip: a
            # permutation of the clip's values that is its own transpose
            def rev3_job():
                l, g = _m3_value_grad(CS._Math(graph_precision(critics[2].precision)), critics[2], rev(fc).reshape(N, 48),
                                      -weights[2] * half * ph)
                return l, rev(g).reshape(N, 48)

            def rev2_job():
                l, g = _m2_value_grad(CS._Math(graph_precision(critics[3].precision)), critics[3], rev(x2).reshape(N, 32),
                                      -weights[3] * half * ph)
                return l, rev(g).reshape(N, 32)
            jobs += [rev3_job, rev2_job]
            wts += [weights[2] * half * ph, weights[3] * half * ph]
            kinds += ["fc", "x2"]
    flip_terms = []
    if flip:                                                      # value only (R/...:455-468): fused no-grad programs
        def flip_job():
            with torch.no_grad():
                fcf, x2f = ops.center_flip(fc.reshape(N, 16, 3), False, True), ops.center_flip(x2.reshape(N, 16, 2), False, True)
                t = [(critics[0](fcf), weights[0] * 0.5), (critics[1](x2f), weights[1] * 0.5)]
                if len(critics) == 4:
                    ph2 = 0.5 if playback else 1.0
                    t += [(critics[2](fcf.reshape(-1, 48)), weights[2] * 0.5 * ph2), (critics[3](x2f.reshape(-1, 32)), weights[3] * 0.5 * ph2)]
                    if playback:
                        t += [(critics[2](rev(fcf).reshape(-1, 48)), weights[2] * 0.5 * ph2),
                              (critics[3](rev(x2f).reshape(-1, 32)), weights[3] * 0.5 * ph2)]
            return tuple(l for l, _ in t), [w for _, w in t]
        jobs.append(flip_job)
    res = _parallel(jobs)
    terms = []                                                    # (logits, weight) of every mean in gen_loss
    g_fc = g_x2 = None
    for (l, g), w, kind in zip(res[:len(wts)], wts, kinds):
        terms.append((l, w))
        if kind == "fc":
            g_fc = g if g_fc is None else ops.add_f32(g_fc, g)
        else:
            g_x2 = g if g_x2 is None else ops.add_f32(g_x2, g)
    if flip:
        ls, ws = res[-1]
        terms += list(zip(ls, ws))
    G_cost = ops.weighted_means([t for t, _ in terms], [-w for _, w in terms])          # -gen_loss
    # ---- back through centring, camera + projection, the FK tail
    g_fw = ops.add_f32(ops.center_flip(g_fc.reshape(N, 16, 3), True, False, adjoint=True).reshape(N, 48),
                       ops.world_to_camera_project_backward(fw, quat, trans, cam9, None, g_x2).reshape(N, 48))
    g_head = ops.gen_tail_backward(head2, bl, sc, g_fw, pre).reshape(B, 35 * R)
    # ---- trunk: backward chain + weight gradients (one grouped contraction launch where the shapes allow)
    slot = CS._slot
    gz = Lh.bwd(mG, g_head, ys[-1], RELU, 0.0)                    # cotangent at block3.fc2's pre-activation
    mG.outer(g_head, ys[-1], Lh.N, Lh.K, slot(Lh.W), slot(Lh.b))
    for i in range(len(blocks) - 1, -1, -1):
        gz1, gz_in = blocks[i].bwd(mG, gz, hs[i], ys[i])
        mG.outer(gz, hs[i], blocks[i].fc2.N, blocks[i].fc2.K, slot(blocks[i].fc2.W), slot(blocks[i].fc2.b))
        mG.outer(gz1, ys[i], blocks[i].fc1.N, blocks[i].fc1.K, slot(blocks[i].fc1.W), slot(blocks[i].fc1.b))
        gz = gz_in
    mG.outer(gz, z, Lp.N, Lp.K, slot(Lp.W), slot(Lp.b))
    mG.flush()
    oG.step()
    return G_cost
