"""GPU: the explicit four-sweep WGAN-GP critic step (dhaug_amd/critic_step.py) against (a) the autograd composite it
replaces, same weights / batch / interpolation coefficients, in the fp32-grade arithmetic (tight) and in bf16 (direction),
(b) the oracle's fp64 critic step.  (The goldens captured from the reference's own train_Fk_discriminator are checked
through train_Fk_discriminator in tests/test_gpu_models.py and tests/test_gpu_loops.py -- that entry point takes this path.)"""
import argparse

import pytest
import torch

import golden_util as GU
from oracle import dhaug_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dhaug_amd
    dhaug_amd._lib.lib()
    from dhaug_amd import critic_step
    from dhaug_amd.models_Fk_GAN import Fk_discriminator, model_fk_gan_train
    return argparse.Namespace(dis=Fk_discriminator, train=model_fk_gan_train, cs=critic_step)


def _args(B, D):
    from test_gpu_models import make_args
    return make_args(batch_size=B, Dis_DenseDim_3D=D, Dis_DenseDim_2D=D)


def _data(tag, B, seed):
    g = torch.Generator().manual_seed(seed)
    if tag == "d3":
        r = GU.synth_pose16(B, seed=seed); r = r - r[:, :1]
        f = r + 0.08 * torch.randn(B, 16, 3, generator=g); f = f - f[:, :1]
    else:
        r = (torch.rand(B, 16, 2, generator=g) - 0.5) * 1.6
        f = r + 0.1 * torch.randn(B, 16, 2, generator=g)
    return r, f, torch.rand(B, 1, generator=g)


def _net(M, tag, args, sd, prec):
    net = M.dis.Fk_3D_Discriminator("cuda", args) if tag == "d3" else M.dis.Fk_2D_Discriminator(args, 16)
    net.load_state_dict(sd)
    net.precision = prec
    net = net.cuda()
    return net, M.train.FusedAdam(net.parameters(), lr=1e-4, betas=(0.5, 0.9))


def _run(M, tag, args, sd, prec, data, analytic):
    net, opt = _net(M, tag, args, sd, prec)
    old = M.train.ANALYTIC_CRITIC_STEP
    M.train.ANALYTIC_CRITIC_STEP = analytic
    try:
        r, f, a = data
        W, C = M.train.train_Fk_discriminator(net, r.cuda(), f.cuda(), argparse.Namespace(train_iter_num=0), None, tag, opt,
                                              args, alpha=a.cuda())
    finally:
        M.train.ANALYTIC_CRITIC_STEP = old
    return W.item(), C.item(), {k: p.grad.detach().float().cpu().clone() for k, p in net.named_parameters()}, \
        {k: p.detach().cpu().clone() for k, p in net.named_parameters()}


@pytest.mark.parametrize("tag,D,B", [("d3", 64, 72), ("d2", 64, 72), ("d3", 256, 300), ("d2", 256, 300)])
def test_explicit_step_equals_autograd_and_oracle(M, tag, D, B):
    args = _args(B, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, 900 + D)
    data = _data(tag, B, 31 + D)
    # fp32-grade arithmetic: explicit schedule == autograd composite == fp64 oracle
    Wa, Ca, ga, pa = _run(M, tag, args, sd, "bf16x6", data, True)
    Wc, Cc, gc, pc = _run(M, tag, args, sd, "bf16x6", data, False)
    fwd = O.d3_forward if tag == "d3" else O.d2_forward
    ref = O.critic_step(fwd, {k: v.double() for k, v in sd.items()}, data[0].double(), data[1].double(), data[2].double())
    assert abs(Wa - ref["Wasserstein_D"].item()) <= 2e-6 and abs(Ca - ref["D_cost"].item()) <= 1e-5 * max(1.0, abs(Ca))
    assert abs(Wa - Wc) <= 2e-6 and abs(Ca - Cc) <= 1e-5 * max(1.0, abs(Cc))
    for k, r in ref["grads"].items():
        scale = r.abs().max().item()
        # fp64 vs fp32-grade arithmetic: a hidden unit within rounding of 0 flips its ReLU mask on ONE row, which moves the
        # gradients it feeds by that row's O(1/B) term -- so: all but a few elements tight, every element within a row's worth
        e = (ga[k].double() - r).abs().reshape(-1)
        kth = max(1, int(e.numel() * 0.995))
        assert e.kthvalue(kth).values.item() <= 2e-6 + 2e-4 * scale, (k, e.kthvalue(kth).values.item(), scale)
        assert e.max().item() <= 2e-4 + 1e-3 * scale, (k, e.max().item(), scale)
        assert (ga[k] - gc[k]).abs().max().item() <= 1e-8 + 2e-5 * scale, (k, (ga[k] - gc[k]).abs().max().item(), scale)
    # the logit layer's bias sees -1/B on B rows and +1/B on B rows: exactly zero, so Adam leaves it alone (B is not a
    # power of two here: an unpaired summation order leaves ~1e-7 and costs a full lr step)
    last = "output.bias" if tag == "d3" else "layer_pred.bias"
    assert ga[last].abs().max().item() == 0.0 and (pa[last] - sd[last]).abs().max().item() == 0.0
    assert gc[last].abs().max().item() == 0.0
    # bf16 (the throughput arithmetic): same step up to bf16 rounding
    Wb, Cb, gb, _ = _run(M, tag, args, sd, "bf16", data, True)
    Wd, Cd, gd, _ = _run(M, tag, args, sd, "bf16", data, False)
    assert abs(Wb - Wa) <= 3e-2 * max(1.0, abs(Wa)) and abs(Cb - Ca) <= 5e-2 * max(1.0, abs(Ca))
    cos_ref, cos_path = [], []
    for k, r in ref["grads"].items():
        if r.abs().max() > 0:
            cs = torch.nn.functional.cosine_similarity
            cos_ref.append(cs(gb[k].reshape(-1).double(), r.reshape(-1), dim=0).item())
            cos_path.append(cs(gb[k].reshape(-1).double(), gd[k].reshape(-1).double(), dim=0).item())
    # (a 72-row batch through bf16 layers: the weakest parameter's direction agrees to ~0.97 with fp64; the two bf16
    # paths agree with each other at least as well)
    assert min(cos_ref) > 0.95, cos_ref
    assert min(cos_path) > 0.95, cos_path
    assert gb[last].abs().max().item() == 0.0


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_explicit_step_vs_reference_at_dense_dim_256(M, golden, tag):
    """ONE critic step at the benchmark's width against the reference's own train_Fk_discriminator (fixtures
    critic_step_{d3,d2}_D256 captured by tests/golden/make_golden_d256.py: scalars, every gradient, the Adam update) -- the
    explicit schedule in the fp32-grade arithmetic; then the throughput arithmetic (bf16) for direction."""
    g = golden("critic_step_%s_D256" % tag)
    B, D = g["real"].shape[0], 256
    args = _args(B, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, int(g["weight_seed"]))
    rec = lambda kind, k: {part: g["%s__%s__%s" % (kind, part, k)] for part in ("full", "sample", "proj")
                           if "%s__%s__%s" % (kind, part, k) in g}
    data = (g["real"], g["fake"], g["alpha"])
    W, C, grads, params = _run(M, tag, args, sd, "bf16x6", data, True)
    assert abs(W - g["Wasserstein_D"].item()) <= 1e-5 and abs(C - g["D_cost"].item()) <= 1e-4 * max(1.0, abs(g["D_cost"].item()))
    for i, k in enumerate(sd):
        GU.compact_close(grads[k], rec("grad", k), 100 + i, 2e-6, 3e-4, k)
        ref = rec("delta", k)
        got = GU.compact(params[k] - sd[k], 100 + i)
        key = "full" if "full" in ref else "sample"
        gref = rec("grad", k)[key]
        # Adam's first step is lr * g / (|g| + eps): compared where |g| is not within rounding of zero
        well = gref.abs() > max(1e-3 * gref.abs().max().item(), 1e-7)
        if well.any():
            assert (got[key].double() - ref[key].double())[well].abs().max().item() <= 2e-6, k
        assert (got[key].double() - ref[key].double()).abs().max().item() <= 2.01e-4, k
    Wb, Cb, gb, _ = _run(M, tag, args, sd, "bf16", data, True)
    assert abs(Wb - W) <= 3e-2 * max(1.0, abs(W))
    cos = []
    for i, k in enumerate(sd):
        r = rec("grad", k)
        key = "full" if "full" in r else "sample"
        a, b = GU.compact(gb[k], 100 + i)[key].double(), r[key].double()
        if b.abs().max() > 0:
            cos.append(torch.nn.functional.cosine_similarity(a, b, dim=0).item())
    assert min(cos) > 0.95, cos


@pytest.mark.parametrize("tag,D,B", [("d3", 256, 304), ("d2", 256, 304)])
def test_bf16_explicit_step_vs_bf16_emulated_oracle(M, tag, D, B):
    """The TIMED arithmetic (bf16 operands, fp32 accumulate, bf16 activations and cotangents) against the oracle's emulation of
    it element by element: the oracle's forward rounds where the kernels round (operands and stored activations to bf16:
    dhaug_oracle._linear / _store) and autograd through that dtype round trip rounds the cotangents and tangents to bf16 as well
    (the gradient of a bf16 tensor is bf16) -- where the explicit step stores them; what is left is summation order.  Every gradient
    element within 2e-2 of its tensor's scale (a dropped skip term or a wrong mask in one of the 17 layers moves whole
    tensors by O(1)); cosine > 0.95 was all the earlier test asked of this path."""
    args = _args(B, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, 950 + D)
    data = _data(tag, B, 77 + D)
    Wb, Cb, gb, _ = _run(M, tag, args, sd, "bf16", data, True)
    f32 = O.d3_forward if tag == "d3" else O.d2_forward
    # ('bf16_fused': at DenseDim 256 the step's forward sweep is the one-launch program, whose 3D critic parks half of its merge
    # layer as bf16 -- one rounding point more than the layer path)
    ref = O.critic_step(lambda x, p: f32(x, p, precision="bf16_fused"), sd, data[0], data[1], data[2])
    assert abs(Wb - ref["Wasserstein_D"].item()) <= 2e-3 * max(1.0, abs(Wb)) and abs(Cb - ref["D_cost"].item()) <= 1e-2 * max(1.0, abs(Cb))
    worst = {1: 0.0, 2: 0.0}
    for k, r in ref["grads"].items():
        scale = r.abs().max().item()
        if scale == 0.0:
            assert gb[k].abs().max().item() == 0.0, k
            continue
        e = (gb[k].double() - r.double()).abs().max().item() / scale
        worst[r.dim()] = max(worst[r.dim()], e)
        # weight gradients: 2e-2 of the tensor's scale.  A bias gradient is the DIFFERENCE of two column sums of equal size (the
        # real rows' cotangents carry -1/B, the fake rows' +1/B): its scale is a residue of that cancellation, so the same
        # absolute noise is a larger share of it -- 4e-2
        assert e <= (2e-2 if r.dim() == 2 else 4e-2), (k, e, scale)
    print("bf16 explicit %s step vs bf16-emulated oracle: worst element error %.2e (weights) / %.2e (biases) of the tensor's scale"
          % (tag, worst[2], worst[1]))


def test_gradient_penalty_dead_rows(M):
    """A row whose input gradient is exactly zero (every ReLU unit behind the merge layer dead): the reference's
    gradients.norm(2, dim=1) back-propagates 0 there (torch's norm subgradient at 0), the penalty term is (0 - 1)^2.  The
    explicit step must give the autograd path's finite gradients, not -inf * 0 = NaN (R/models_Fk_GAN/Fk_discriminator.py:229)."""
    from dhaug_amd import ops
    g = torch.randn(64, 48, device="cuda")
    g[5] = 0.0
    g[63] = 0.0
    v, pen = ops.gp_penalty(g, 0.25)
    n = g.norm(dim=1, keepdim=True)
    ref = torch.where(n > 0, 0.25 * (n - 1) / n.clamp_min(1e-30) * g, torch.zeros_like(g))
    assert torch.isfinite(v).all() and (v[5] == 0).all() and (v[63] == 0).all()
    assert (v - ref).abs().max().item() <= 1e-5 and ((pen - (n[:, 0] - 1) ** 2).abs() / (1 + pen)).max().item() <= 1e-5
    assert pen[5].item() == 1.0
    # through the whole step: the 3D critic with its merge layer switched off by a large negative bias
    B, D = 72, 64
    args = _args(B, D)
    sd = GU.seeded_state_dict(GU.shapes_d3(D), 123)
    sd["merge_previous.0.bias"] = torch.full_like(sd["merge_previous.0.bias"], -1e3)
    data = _data("d3", B, 8)
    for prec in ("bf16x6", "bf16"):
        Wa, Ca, ga, pa = _run(M, "d3", args, sd, prec, data, True)
        Wc, Cc, gc, pc = _run(M, "d3", args, sd, prec, data, False)
        assert all(torch.isfinite(t).all() for t in ga.values()) and all(torch.isfinite(t).all() for t in pa.values())
        assert abs(Ca - Cc) <= 1e-5 * max(1.0, abs(Cc)) and abs(Ca - args.GAN_LAMBDA) <= 1e-4     # GP = lambda * mean(1)
        for k in ga:
            assert (ga[k] - gc[k]).abs().max().item() <= 1e-7, k


def test_explicit_step_full_batch_properties(M):
    """BASELINE configs[2] size (B = 65 536, D = 256, bf16): finite, every parameter moves by at most ~lr, the step of a
    2x replicated batch equals the step of the batch (means are batch-size independent), scalars match a bf16x6 subsample."""
    B, D = 65536, 256
    args = _args(B, D)
    for tag in ("d3", "d2"):
        shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
        sd = GU.seeded_state_dict(shapes, 1900)
        r, f, a = _data(tag, 4096, 77)
        rep = lambda t: t.repeat(B // 4096, *([1] * (t.dim() - 1)))
        W, C, g, p = _run(M, tag, args, sd, "bf16", (rep(r), rep(f), rep(a)), True)
        args_s = _args(4096, D)
        Ws, Cs, gs, ps = _run(M, tag, args_s, sd, "bf16", (r, f, a), True)
        assert all(torch.isfinite(v).all() for v in g.values())
        assert abs(W - Ws) <= 1e-3 * max(1.0, abs(Ws)) and abs(C - Cs) <= 2e-3 * max(1.0, abs(Cs))
        for k in g:
            scale = gs[k].abs().max().item()
            assert (g[k] - gs[k]).abs().max().item() <= 2e-2 * scale + 1e-9, (tag, k)
            assert (p[k] - sd[k]).abs().max().item() <= 1.01e-4, (tag, k)


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_forward_with_save_layer_by_layer(M, tag):
    """The explicit step's sweep 1 as one fused launch (fused.critic3d_forward_save / critic2d_forward_save): every SAVED
    activation must be the layer function of the saved activations in front of it -- act(W x + b [+ skip]) evaluated by
    torch in fp32 on the bf16 values the kernel consumed -- to bf16 rounding (one ulp of the value + the partial-sum
    rounding of the merge layer's two halves), on a ragged row count (900 = 7 tiles of 128 + 4 rows)."""
    from dhaug_amd import fused, ops
    D, rows = 256, 900
    args = _args(300, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, 4242)
    net = (M.dis.Fk_3D_Discriminator("cuda", args) if tag == "d3" else M.dis.Fk_2D_Discriminator(args, 16)).cuda()
    net.load_state_dict(sd)
    net.precision = "bf16"
    gen = torch.Generator().manual_seed(5)
    W = lambda k: sd[k + ".weight"].cuda().bfloat16().float()
    b = lambda k: sd[k + ".bias"].cuda()
    relu = torch.relu

    def close(got, ref, n, extra=0.0):
        got, ref = got[:, :n].float(), ref
        tol = 2.0 ** -8 * ref.abs() + 2.0 ** -8 * extra + 1e-6              # bf16: half an ulp of the value (+ slack where a partial sum was rounded)
        bad = ((got - ref).abs() > tol)
        assert bad.float().mean().item() <= 2e-4, bad.float().mean().item()  # (fp32 summation order: a value within rounding of a bf16 tie)
        assert (got - ref).abs().max().item() <= 2.0 ** -6 * max(1.0, ref.abs().max().item())

    if tag == "d3":
        x = (GU.synth_pose16(rows, seed=9) - GU.synth_pose16(rows, seed=9)[:, :1]).reshape(rows, 48).cuda()
        kf, kb = ops.kcs_forward(x, True, f32=True, bf16_ld=32)
        r = fused.critic3d_forward_save(net, x, kb)
        lin = lambda k, v: v @ W(k).t() + b(k)
        for bi, (first, blocks, inp) in enumerate((("special_KCS_previous.0", ("special_KCS_block1", "special_KCS_block2", "special_KCS_block3"), kb[:, :30].float()),
                                                   ("previous.0", ("block1", "block2", "block3"), x.bfloat16().float()))):
            close(r["y"][bi][0], relu(lin(first, inp)), D)
            for i, blk in enumerate(blocks):
                yin = r["y"][bi][i][:, :D].float()
                close(r["h"][bi][i], relu(lin(blk + ".fc1", yin)), D)
                close(r["y"][bi][i + 1], relu(lin(blk + ".fc2", r["h"][bi][i][:, :D].float()) + yin), D)
        cat = r["cat"].float()
        pre = cat @ W("merge_previous.0").t() + b("merge_previous.0")
        half = (cat[:, :D] @ W("merge_previous.0")[:, :D].t()).abs()           # the KCS half waits in LDS as bf16
        close(r["m0"], relu(pre), 100, extra=half.max().item())
        m0 = r["m0"][:, :100].float()
        close(r["mh"], relu(lin("merge_block1.fc1", m0)), 100)
        close(r["m1"], relu(lin("merge_block1.fc2", r["mh"][:, :100].float()) + m0), 100)
        assert (r["m0"][:, 100:] == 0).all() and (r["m1"][:, 100:] == 0).all()
        logit = r["m1"][:, :100].float() @ W("output").t() + b("output")
        assert (r["logits"] - logit).abs().max().item() <= 1e-4 * max(1.0, logit.abs().max().item())
    else:
        x = (torch.rand(rows, 32, generator=gen) - 0.5).cuda()
        r = fused.critic2d_forward_save(net, x)
        s = net.slope
        lr = lambda v: torch.nn.functional.leaky_relu(v, s)
        lin = lambda k, v: v @ W(k).t() + b(k)
        d = [t[:, :D].float() for t in r["d"]]
        close(r["d"][0], lr(lin("pose_layer_1", x.bfloat16().float())), D)
        close(r["d"][1], lr(lin("pose_layer_2", d[0])), D)
        close(r["d"][2], lr(lin("pose_layer_3", d[1]) + d[0]), D)
        close(r["d"][3], lin("pose_layer_4", d[2]), D)
        close(r["d"][4], lr(lin("layer_last", d[3])), D)
        logit = lin("layer_pred", d[4])
        assert (r["logits"] - logit).abs().max().item() <= 1e-4 * max(1.0, logit.abs().max().item())
    # the sign-bit arrays the run layers leave beside their images: bit == (saved value > 0), element for element
    from dhaug_amd import fused as F
    saved = ([t for br in r["y"] for t in br] + [t for br in r["h"] for t in br]) if tag == "d3" else list(r["d"])
    with_bits = [t for t in saved if getattr(t, "_dhaug_bits", None) is not None]
    assert len(with_bits) == (14 if tag == "d3" else 4)
    for t in with_bits:
        assert torch.equal(F.decode_bits(t._dhaug_bits, rows), (t[:, :D].float() > 0).cpu())


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_sweep4_in_two_parts_equals_one_part(M, tag):
    """the weight gradients as two grouped launches (real / fake rows on a side stream beside the tangent sweep, interpolated
    rows after it: critic_step.TN_SPLIT) against one launch over all 3B rows: the same sums in another order"""
    B, D = 2048, 256
    args = _args(B, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, 77)
    data = _data(tag, B, 5)
    assert M.cs.TN_SPLIT
    W1, C1, g1, p1 = _run(M, tag, args, sd, "bf16", data, True)
    M.cs.TN_SPLIT = False
    try:
        W0, C0, g0, p0 = _run(M, tag, args, sd, "bf16", data, True)
    finally:
        M.cs.TN_SPLIT = True
    assert W1 == W0 and C1 == C0
    for k in g0:
        scale = g0[k].abs().max().item()
        # (fp32 sums of the same terms in another grouping; the bias sums cancel to 1e-3 of their terms)
        assert (g1[k] - g0[k]).abs().max().item() <= 1e-4 * scale + 1e-12, (k, (g1[k] - g0[k]).abs().max().item(), scale)


def test_bf16_operands_written_beside_fp32_results_equal_the_cast_launches(M):
    """dhaug_gp_assemble_bf16 / dhaug_gp_penalty_bf16 / the KCS operand of dhaug_kcs_forward: the bf16 tensors these launches write
    beside their fp32 results are, bit for bit, what dhaug_cast_pad_bf16 makes of those results (critic_step registers them as the
    casts of those tensors: _Math.seed_cast)"""
    from dhaug_amd import ops
    g = torch.Generator().manual_seed(3)
    for B, W in ((1000, 32), (2048 + 64, 48), (7, 48)):
        r, f, a = torch.randn(B, W, generator=g).cuda(), torch.randn(B, W, generator=g).cuda() * 3, torch.rand(B, 1, generator=g).cuda()
        X0 = ops.gp_assemble(r, f, a)
        X1 = ops.gp_assemble(r, f, a, bf16_rows=True)
        assert torch.equal(X0, X1) and torch.equal(X1._dhaug_bf16_rows, ops.cast_pad_bf16(X1[:2 * B], W))
        gr = torch.randn(B, W, generator=g).cuda()
        gr[B // 2] = 0.0                                         # (a dead row: the penalty's cotangent is 0 there)
        v0, p0 = ops.gp_penalty(gr, 0.37)
        v1, p1 = ops.gp_penalty(gr, 0.37, bf16=True)
        assert torch.equal(v0, v1) and torch.equal(p0, p1) and torch.equal(v1._dhaug_bf16, ops.cast_pad_bf16(v1, W))
    x = (GU.synth_pose16(3000, seed=4)).reshape(3000, 48).cuda()
    kf, kb = ops.kcs_forward(x, True, f32=True, bf16_ld=32)
    assert torch.equal(kb, ops.cast_pad_bf16(kf, 32))


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_step_with_registered_casts_equals_step_with_cast_launches(M, tag):
    """critic_step.SEED_CASTS: the same step, bit for bit (the operands are the same bits; only the launches that made them differ)"""
    B, D = 2048, 256
    args = _args(B, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, 41)
    data = _data(tag, B, 12)
    assert M.cs.SEED_CASTS
    W1, C1, g1, p1 = _run(M, tag, args, sd, "bf16", data, True)
    M.cs.SEED_CASTS = False
    try:
        W0, C0, g0, p0 = _run(M, tag, args, sd, "bf16", data, True)
    finally:
        M.cs.SEED_CASTS = True
    assert W1 == W0 and C1 == C0
    for k in g0:
        if g0[k].numel() > 1:                                    # (the logit layer's bias slot: see test_step_is_bit_reproducible_under_load)
            assert torch.equal(g1[k], g0[k]), k
            assert torch.equal(p1[k], p0[k]), k


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_step_is_bit_reproducible_under_load(M, tag):
    """every kernel of the explicit step sums in a fixed order (the one exception, the logit layer's bias slot, is excluded): the
    same step run again -- while a second stream keeps the memory system and the matrix pipes busy with unrelated work of
    varying length -- gives the same gradient bucket and scalars, bit for bit.  A stale LDS stage, a missing wait or a missing
    barrier in one of the hand-scheduled kernels shows up here as a rare mismatch (tools/stress_determinism.py runs 400 of them)."""
    B, D = 2048, 256
    args = _args(B, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, 31)
    data = _data(tag, B, 9)
    side = torch.cuda.Stream()
    big = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
    buf = torch.empty(32 << 20, device="cuda", dtype=torch.uint8)
    ref = None
    for rep in range(12):
        if rep:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(rep % 4):
                    buf.add_(1)
                    torch.mm(big, big)
        W, C, g, p = _run(M, tag, args, sd, "bf16", data, True)
        cur = torch.cat([v.reshape(-1).float().cpu() for k, v in sorted(g.items()) if v.numel() > 1] + [torch.tensor([float(W), float(C)])])
        if ref is None:
            ref = cur
        else:
            assert torch.equal(cur, ref), (tag, rep, (cur - ref).abs().max().item())
    torch.cuda.current_stream().wait_stream(side)


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_block_launches_equal_single_layer_launches(M, tag):
    """the residual blocks of sweeps 2 and 3 as two-layer / three-block launches (dhaug_gemm_block2_stack_bf16) and the merge
    layer's cotangent masked by sign bits (dhaug_gemm_bf16_dbits_wide) against the same step with one launch per layer: the
    kernels compute the same products in the same order -- identical gradients"""
    from dhaug_amd import ops
    B, D = 2048, 256
    args = _args(B, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, 41)
    data = _data(tag, B, 11)
    calls = ops._lib.CALLS[0]
    W1, C1, g1, p1 = _run(M, tag, args, sd, "bf16", data, True)
    n_block = ops._lib.CALLS[0] - calls
    ops.BLOCK2 = False
    try:
        calls = ops._lib.CALLS[0]
        W0, C0, g0, p0 = _run(M, tag, args, sd, "bf16", data, True)
        n_single = ops._lib.CALLS[0] - calls
    finally:
        ops.BLOCK2 = True
    assert n_block < n_single                                     # (fewer C-ABI calls: the launches really were merged)
    assert W1 == W0 and C1 == C0
    for k in g0:
        if g0[k].numel() > 1:
            assert torch.equal(g1[k], g0[k]), k


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_step_reads_no_unwritten_memory(M, tag, monkeypatch):
    """the forward-with-save launch writes the block layers' images for the real / fake rows only, the interpolated rows are
    read through their sign bits and then receive the tangents (critic_step.SKIP_XHAT_SAVES): with every torch.empty buffer
    pre-filled with NaN (or a huge value) the step's gradients, scalars and stepped weights are the same bits as without --
    nothing reads a row nobody wrote.  (tools/poison.py runs the same probe over whole iterations, eager and as hipGraphs.)"""
    B, D = 2048, 256
    args = _args(B, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, 51)
    data = _data(tag, B, 13)
    assert M.cs.SKIP_XHAT_SAVES
    ref = _run(M, tag, args, sd, "bf16", data, True)
    real_empty, real_empty_like = torch.empty, torch.empty_like
    for poison in (float("nan"), 3.0e38):
        def fill(t):
            if t.is_cuda and t.numel():
                if t.dtype.is_floating_point:
                    t.fill_(poison)
                else:
                    t.view(torch.uint8).fill_(255)
            return t
        monkeypatch.setattr(torch, "empty", lambda *a, **k: fill(real_empty(*a, **k)))
        monkeypatch.setattr(torch, "empty_like", lambda *a, **k: fill(real_empty_like(*a, **k)))
        got = _run(M, tag, args, sd, "bf16", data, True)
        monkeypatch.setattr(torch, "empty", real_empty)
        monkeypatch.setattr(torch, "empty_like", real_empty_like)
        assert got[0] == ref[0] and got[1] == ref[1], (tag, poison)
        for k in ref[2]:
            if ref[2][k].numel() > 1:
                assert torch.equal(got[2][k], ref[2][k]), (tag, poison, k)
                assert torch.equal(got[3][k], ref[3][k]), (tag, poison, k)


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_ragged_batch_keeps_every_saved_row(M, tag, monkeypatch):
    """B = 48: the two real / fake parts make whole 32-row tiles (96) but the interpolated part does not start a launch over whole
    tiles (3B = 144), so the backward sweep reads mask IMAGES -- the forward-with-save launch must then write every row
    (fused.partial_save_ok): the step equals the one with DHAUG_SAVE_ALL_ROWS semantics, and poisoned buffers change nothing"""
    B, D = 48, 256
    args = _args(B, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, 61)
    data = _data(tag, B, 17)
    ref = _run(M, tag, args, sd, "bf16", data, True)
    monkeypatch.setattr(M.cs, "SKIP_XHAT_SAVES", False)
    allrows = _run(M, tag, args, sd, "bf16", data, True)
    monkeypatch.setattr(M.cs, "SKIP_XHAT_SAVES", True)
    assert ref[0] == allrows[0] and ref[1] == allrows[1]
    real_empty = torch.empty
    monkeypatch.setattr(torch, "empty", lambda *a, **k: (lambda t: t.fill_(float("nan")) if (t.is_cuda and t.dtype.is_floating_point and t.numel()) else t)(real_empty(*a, **k)))
    got = _run(M, tag, args, sd, "bf16", data, True)
    monkeypatch.setattr(torch, "empty", real_empty)
    assert got[0] == ref[0] and got[1] == ref[1]
    for k in ref[2]:
        if ref[2][k].numel() > 1:
            assert torch.equal(got[2][k], ref[2][k]) and torch.equal(allrows[2][k], ref[2][k]), (tag, k)


@pytest.mark.parametrize("tag", ["d3", "d2"])
def test_split_operands_as_planes_equal_six_segment_operands(M, tag):
    """critic_step.PLANES: the fp32-grade ("bf16x6") step with every 256-wide operand split once into three planes -- layer products on
    dhaug_gemm_bf16x6_planes, weight gradients over virtual rows -- is the step on six-segment operands bit for bit (same product terms, same
    order), from fewer split launches moving half the bytes."""
    B, D = 1536, 256
    args = _args(B, D)
    shapes = GU.shapes_d3(D) if tag == "d3" else GU.shapes_d2(D)
    sd = GU.seeded_state_dict(shapes, 43)
    data = _data(tag, B, 14)
    assert M.cs.PLANES
    from dhaug_amd import _lib
    calls = {}
    real = _lib.call
    def spy(name, *a):
        calls[name] = calls.get(name, 0) + 1
        return real(name, *a)
    _lib.call = spy
    try:
        W1, C1, g1, p1 = _run(M, tag, args, sd, "bf16x6", data, True)
        with_planes = dict(calls)
        calls.clear()
        M.cs.PLANES = False
        try:
            W0, C0, g0, p0 = _run(M, tag, args, sd, "bf16x6", data, True)
        finally:
            M.cs.PLANES = True
    finally:
        _lib.call = real
    assert with_planes.get("dhaug_gemm_bf16x6_planes", 0) >= 12 and calls.get("dhaug_gemm_bf16x6_planes", 0) == 0
    # (results that are operands themselves leave the GEMM with their planes: critic_step.PLANES_OUT)
    assert M.cs.PLANES_OUT and with_planes["dhaug_split_bf16"] <= calls["dhaug_split_bf16"] - 8, (with_planes["dhaug_split_bf16"], calls["dhaug_split_bf16"])
    assert W1 == W0 and C1 == C0
    for k in g0:
        if k.endswith("weight"):
            assert torch.equal(g1[k], g0[k]), k
            assert torch.equal(p1[k], p0[k]), k
        else:                                                     # (bias gradients: dhaug_colsum_f32 adds its row slabs with atomics -- the same
            scale = g0[k].abs().max().item() + 1e-12                  # cotangent bits summed in an order that varies from launch to launch)
            assert (g1[k] - g0[k]).abs().max().item() <= 2e-6 * scale, k
