"""GPU parity tests, kernel level: every C-ABI entry point against the oracle on seeded inputs.
Tolerances: FK joints <= 1e-5 abs (north_star); gradients <= 1e-4 relative to the gradient scale; bf16 GEMMs are
compared with an fp32 matmul of the *bf16-rounded* operands (fp32 accumulate both sides): <= 2e-5 relative to the
row scale, plus the bf16 rounding of the stored output where the output is bf16."""
import numpy as np
import pytest
import torch

import golden_util as GU
from oracle import dhaug_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dhaug_amd
    from dhaug_amd import ops as _ops
    dhaug_amd._lib.lib()            # fail loudly if the HIP extension is missing
    return _ops


def dev(t):
    return t.cuda()


def maxabs(a, b):
    return (a.double().cpu() - b.double().cpu()).abs().max().item()


# ------------------------------------------------------------------------------------------------ FK
@pytest.mark.parametrize("name", ["fk_N1", "fk_N8", "fk_N1024", "fk_single_dof", "fk_video_B16_R9"])
def test_fk_forward_golden(ops, golden, name):
    g = golden(name)
    root = g["root"].reshape(-1, 3)
    out32 = ops.fk_forward(dev(g["angles"]), dev(g["bone_len"]), dev(root), out_joints=32)
    assert out32.shape == g["out32"].shape
    assert maxabs(out32, g["out32"]) <= 1e-5                      # reference (N,32,3), incl. the unused rows
    out16 = ops.fk_forward(dev(g["angles"]), dev(g["bone_len"]), dev(root), out_joints=16)
    assert maxabs(out16, g["out32"][:, O.H36M_32_TO_16]) <= 1e-5


@pytest.mark.parametrize("N", [1, 63, 64, 65, 1000, 65536 + 17])
def test_fk_forward_ragged_sizes(ops, N):
    a, bl, rt = GU.synth_fk_inputs(N, seed=100 + N)
    out = ops.fk_forward(dev(a), dev(bl), dev(rt))
    ref = O.fk_forward16(a, bl, rt)
    assert maxabs(out, ref) <= 1e-5
    # fp64 oracle: the GPU result is as close to exact arithmetic as the fp32 reference is
    if N <= 1000:
        ref64 = O.fk_forward16(a.double(), bl.double(), rt.double())
        assert maxabs(out, ref64) <= 4e-6


def test_fk_forward_empty_and_misuse(ops):
    e = torch.empty((0, 37)).cuda()
    out = ops.fk_forward(e, torch.empty((0, 15)).cuda(), torch.empty((0, 3)).cuda())
    assert out.shape == (0, 16, 3)
    with pytest.raises(RuntimeError):
        ops.fk_forward(torch.zeros(4, 37), torch.zeros(4, 15), torch.zeros(4, 3))      # CPU tensors: no fallback


def test_fk_properties_full_size(ops):
    """size-independent properties at the BASELINE batch (65 536): bone-length invariance, dead angle slots,
    root translation equivariance."""
    N = 65536
    a, bl, rt = GU.synth_fk_inputs(N, seed=7)
    out = ops.fk_forward(dev(a), dev(bl), dev(rt))
    assert maxabs(ops.bone_length(out), bl) <= 2e-6
    a2 = a.clone(); a2[:, [4, 9, 22, 27, 32, 33]] += 77.0
    assert maxabs(ops.fk_forward(dev(a2), dev(bl), dev(rt)), out) <= 1e-6
    shift = torch.tensor([0.5, -0.25, 2.0])
    out_s = ops.fk_forward(dev(a), dev(bl), dev(rt + shift))
    assert maxabs(out_s - shift.cuda(), out) <= 2e-6
    assert maxabs(out[:, 0], rt) == 0.0                          # Hip == root exactly


@pytest.mark.parametrize("N", [5, 64, 1000])
def test_fk_backward(ops, N):
    a, bl, rt = GU.synth_fk_inputs(N, seed=200 + N)
    g = torch.randn(N, 16, 3, generator=torch.Generator().manual_seed(N))
    ad, bd, rd = (t.double().requires_grad_(True) for t in (a, bl, rt))
    (O.fk_forward16(ad, bd, rd) * g.double()).sum().backward()
    ga, gb, gr = ops.fk_backward(dev(a), dev(bl), dev(g))
    for got, ref in ((ga, ad.grad), (gb, bd.grad), (gr, rd.grad)):
        assert maxabs(got, ref) <= 1e-4 * ref.abs().max().item()
    assert ga[:, [4, 9, 22, 27, 32, 33]].abs().max().item() == 0.0


@pytest.mark.parametrize("name,pre", [("gen_D32", True), ("gen_D256", True), ("gen_D32_nopre", False)])
def test_gen_tail_forward_golden(ops, golden, name, pre):
    src = golden("gen_D32" if name == "gen_D32_nopre" else name)
    g = golden(name)
    fake, ang = ops.gen_tail_forward(dev(src["head"]), dev(src["bone_len"]), dev(src["scaler"]), use_preangle=pre,
                                     want_angles=True)
    assert maxabs(ang, g["angle37"]) <= 2e-4                      # degrees (tanhf on device vs ATen tanh)
    assert maxabs(fake.reshape(-1, 48), g["fake"]) <= 1e-5


def test_gen_tail_video_golden(ops, golden):
    g = golden("gen_video_D32")
    B, R = 8, 9
    sd = GU.seeded_state_dict(GU.shapes_generator(32, frames=R), int(g["weight_seed"]))
    head = O.gen_trunk(g["z"], sd).reshape(B * R, 35)
    scaler = g["scaler"].reshape(B, 1, 8).repeat(1, R, 1).reshape(B * R, 8)
    fake, _ = ops.gen_tail_forward(dev(head), dev(g["bone_len"]), dev(scaler))
    assert maxabs(fake.reshape(B, R, 48), g["fake"]) <= 1e-5


def test_gen_tail_backward(ops, golden):
    s = golden("gen_D32")
    head, bl, sc = s["head"], s["bone_len"], s["scaler"]
    g = torch.randn(head.shape[0], 16, 3, generator=torch.Generator().manual_seed(3))
    hd = head.double().requires_grad_(True)
    fake, _ = O.gen_tail(hd, bl.double(), sc.double())
    (fake.reshape(-1, 16, 3) * g.double()).sum().backward()
    gh = ops.gen_tail_backward(dev(head), dev(bl), dev(sc), dev(g))
    assert maxabs(gh, hd.grad) <= 1e-4 * hd.grad.abs().max().item()
    assert gh[:, 31].abs().max().item() == 0.0
    # no jitter (scaler = NULL) path
    f0, _ = ops.gen_tail_forward(dev(head), dev(bl), None)
    r0, _ = O.gen_tail(head, bl, torch.zeros(head.shape[0], 8))
    assert maxabs(f0.reshape(-1, 48), r0) <= 1e-5


# -------------------------------------------------------------------------------------- pose features
def test_kcs_golden(ops, golden):
    g = golden("kcs_256")
    x = dev(g["pose16"])
    assert maxabs(ops.bone_length(x), O.bone_lengths(g["pose16"])) <= 1e-6
    f, b = ops.kcs_forward(x, with_lengths=True, f32=True, bf16_ld=32)
    assert maxabs(f, g["kcs30"]) <= 5e-6
    assert maxabs(b[:, :30].float(), g["kcs30"].to(torch.bfloat16).float()) <= 1e-2 and b[:, 30:].abs().max() == 0
    assert maxabs(b[:, :30].float(), f.to(torch.bfloat16).float()) == 0.0
    f15, _ = ops.kcs_forward(x, with_lengths=False)
    assert maxabs(f15, g["kcs15"]) <= 5e-6


@pytest.mark.parametrize("wl", [True, False])
def test_kcs_vjp_jvp(ops, wl):
    N = 300
    x = GU.synth_pose16(N, seed=17)
    W = 30 if wl else 15
    gf = torch.randn(N, W, generator=torch.Generator().manual_seed(1))
    tan = torch.randn(N, 16, 3, generator=torch.Generator().manual_seed(2))
    xd = x.double().requires_grad_(True)
    f = O.kcs_features(xd, with_lengths=wl)
    (f * gf.double()).sum().backward()
    got = ops.kcs_backward(dev(x), dev(gf), with_lengths=wl)
    assert maxabs(got.reshape(N, 16, 3), xd.grad) <= 1e-4 * xd.grad.abs().max().item()
    _, jv = torch.autograd.functional.jvp(lambda t: O.kcs_features(t, with_lengths=wl), x.double(), tan.double())
    gj = ops.kcs_jvp(dev(x), dev(tan), with_lengths=wl)
    assert maxabs(gj, jv) <= 1e-4 * jv.abs().max().item()


def test_camera_golden_and_backward(ops, golden):
    g = golden("camera_128")
    q, t, cam = g["R"][0], g["t"][0], g["cam"][0]
    c3, p2 = ops.world_to_camera_project(dev(g["X"]), q, t, cam)
    assert maxabs(c3, g["Xc"]) <= 2e-6 and maxabs(p2, g["x2d"]) <= 2e-6
    w = ops.camera_to_world(dev(g["Xc"]), dev(g["R"].repeat(128, 1)), dev(g["t"].repeat(128, 1)))
    assert maxabs(w, g["Xw"]) <= 2e-6
    assert maxabs(ops.center_flip(dev(g["X"]), False, True), g["flip"]) == 0.0
    cen = g["X"] - g["X"][:, :1]
    assert maxabs(ops.center_flip(dev(g["X"]), True, False), cen) <= 1e-7
    assert maxabs(ops.center_flip(dev(g["X"]), True, True), O.flip_lr(cen)) <= 1e-7
    # backward of w2c+project and of centre/flip (adjoint) against autograd on the oracle
    g3 = torch.randn(128, 16, 3, generator=torch.Generator().manual_seed(5))
    g2 = torch.randn(128, 16, 2, generator=torch.Generator().manual_seed(6))
    X = g["X"].double().requires_grad_(True)
    Xc = O.world_to_camera(X, g["R"].double(), g["t"].double())
    x2 = O.project_to_2d(Xc, g["cam"].double())
    ((Xc * g3.double()).sum() + (x2 * g2.double()).sum()).backward()
    gx = ops.world_to_camera_project_backward(dev(g["X"]), q, t, cam, dev(g3), dev(g2))
    assert maxabs(gx, X.grad) <= 1e-4 * X.grad.abs().max().item()
    Y = g["X"].double().requires_grad_(True)
    yc = Y - Y[:, :1]
    yf = yc.clone(); yf[:, :, 0] = -yf[:, :, 0]
    yf = yf[:, [0, 4, 5, 6, 1, 2, 3, 7, 8, 9, 13, 14, 15, 10, 11, 12]]
    (yf * g3.double()).sum().backward()
    assert maxabs(ops.center_flip(dev(g3), True, True, adjoint=True), Y.grad) <= 1e-5


# ---------------------------------------------------------------------------------------------- GEMM
def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (1000, 256, 128), (129, 100, 512), (4096, 35, 256), (777, 1, 112),
                                   (64, 256, 48), (300, 256, 32), (2048, 112, 112), (515, 315, 1008),
                                   # long batch x wide layer: the 128 x 256-tile kernel (ragged rows, K tail, ragged columns)
                                   (4608, 1000, 1008), (4096 + 77, 512, 256), (13824, 1000, 1008)])
def test_gemm_nt_plain(ops, M, N, K):
    gen = torch.Generator().manual_seed(M + N + K)
    A = _bf(torch.randn(M, K, generator=gen)).cuda()
    B = _bf(torch.randn(N, K, generator=gen) / K ** 0.5).cuda()
    ref = (A.float().cpu().double() @ B.float().cpu().double().t()).float()
    _, cf = ops.gemm_nt(A, B, N, K, out_f32=True)
    assert maxabs(cf, ref) <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("act,slope", [(0, 0.0), (1, 0.0), (2, 0.01)])
def test_gemm_nt_epilogues(ops, act, slope):
    M, N, K = 1111, 100, 256
    gen = torch.Generator().manual_seed(act)
    A = _bf(torch.randn(M, K, generator=gen)).cuda()
    B = _bf(torch.randn(N, K, generator=gen) / 16).cuda()
    bias = torch.randn(N, generator=gen).cuda()
    res = _bf(torch.randn(M, 104, generator=gen)).cuda()
    z = A.float().cpu().double() @ B.float().cpu().double().t() + bias.cpu().double() + res[:, :N].float().cpu().double()
    ref = z if act == 0 else (torch.relu(z) if act == 1 else torch.nn.functional.leaky_relu(z, slope))
    cb, cf = ops.gemm_nt(A, B, N, K, bias=bias, res_bf16=res, act=act, slope=slope, out_bf16=True, n_pad=112, out_f32=True)
    assert maxabs(cf, ref) <= 3e-5 * ref.abs().max().item()
    assert cb.shape == (M, 112) and cb[:, N:].abs().max().item() == 0.0
    assert maxabs(cb[:, :N].float(), cf.to(torch.bfloat16).float()) == 0.0
    # fp32 residual variant
    resf = torch.randn(M, N, generator=gen).cuda()
    _, cf2 = ops.gemm_nt(A, B, N, K, bias=bias, res_f32=resf, act=0, out_f32=True)
    ref2 = A.float().cpu().double() @ B.float().cpu().double().t() + bias.cpu().double() + resf.cpu().double()
    assert maxabs(cf2, ref2) <= 3e-5 * ref2.abs().max().item()


@pytest.mark.parametrize("M", [4608 + 40, 10240 + 40])
def test_gemm_nt_big_tiles_epilogues(ops, M):
    """the 128 x 256-tile kernel (4 648 rows) and the 256 x 256-tile kernel (10 280 rows: 164 tiles) with everything the DenseDim-1000
    training layers ask of them: bias + bf16 residual + ReLU with a zero-padded bf16 output (forward), and the masked input-gradient
    form (backward / tangent); against fp64 on the bf16 operands and against the 64 x 64-tile kernel they replace for long batches"""
    import os
    gen = torch.Generator().manual_seed(5)
    N, K, Kp = 1000, 1000, 1008
    A = torch.zeros(M, Kp); A[:, :K] = torch.randn(M, K, generator=gen) * 0.5
    W = torch.zeros(N, Kp); W[:, :K] = torch.randn(N, K, generator=gen) / K ** 0.5
    R = torch.zeros(M, Kp); R[:, :N] = torch.randn(M, N, generator=gen)
    Y = torch.zeros(M, Kp); Y[:, :N] = torch.randn(M, N, generator=gen)
    A, W, R, Y = (_bf(t).cuda() for t in (A, W, R, Y))
    bias = torch.randn(N, generator=gen).cuda()
    pre = A.float().cpu().double() @ W.float().cpu().double().t()
    cb, _ = ops.gemm_nt(A, W, N, Kp, bias=bias, res_bf16=R, act=1, out_bf16=True, n_pad=Kp)
    ref = torch.relu(pre + bias.cpu().double() + R[:, :N].float().cpu().double())
    assert cb.shape == (M, Kp) and cb[:, N:].abs().max().item() == 0.0
    assert maxabs(cb[:, :N].float(), ref) <= 2.0 ** -8 * max(1.0, ref.abs().max().item())
    g = ops.gemm_nt_dmask(A, W, N, Kp, Y, 1, 0.0, res_bf16=R)
    refg = (pre + R[:, :N].float().cpu().double()) * (Y[:, :N].float().cpu() > 0).double()
    assert maxabs(g[:, :N].float(), refg) <= 2.0 ** -8 * max(1.0, refg.abs().max().item())
    os.environ["DHAUG_GEMM_NOBIG"] = "1"
    try:
        cb_old, _ = ops.gemm_nt(A, W, N, Kp, bias=bias, res_bf16=R, act=1, out_bf16=True, n_pad=Kp)
    finally:
        del os.environ["DHAUG_GEMM_NOBIG"]
    assert (cb.float() - cb_old.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("N", [1, 64, 1000, 65536])
def test_d3_penalty_equals_the_six_launches(ops, N):
    """dhaug_d3_penalty (KCS pull-back of the KCS branch's input cotangent + the pose branch's, the row norm, the penalty and its
    cotangent, the KCS tangent, both tangent inputs as bf16 operands: one launch) against kcs_backward, add_f32, gp_penalty, kcs_jvp and
    the two casts it replaces, a dead row (zero cotangent) included."""
    gen = torch.Generator().manual_seed(31)
    x = GU.synth_pose16(N, seed=5).reshape(N, 48).cuda()
    x = x - x[:, :3].repeat(1, 16)
    gk = (torch.randn(N, 30, generator=gen) * 0.01).cuda()
    gp = (torch.randn(N, 48, generator=gen) * 0.01).cuda()
    if N > 3:
        gk[3] = 0.0; gp[3] = 0.0                                  # every unit of the critic dead on this row: norm 0
    coef = 2.0 * 10.0 / N
    g = ops.add_f32(ops.kcs_backward(x, gk, True), gp)
    v, pen = ops.gp_penalty(g, coef)
    tk = ops.kcs_jvp(x, v, True)
    r_tk, r_v = ops.cast_pad_bf16(tk, 32), ops.cast_pad_bf16(v, 48)
    f_tk, f_v, f_pen = ops.d3_penalty(x, gk, gp, coef)
    # the same operations in the same order; hipcc contracts a few multiply-adds differently in the two kernels, so the fp32 values
    # may differ in the last bit (and, through it, a bf16 value by one ulp on a rounding boundary)
    assert (f_pen - pen).abs().max().item() <= 4e-7 * max(1.0, pen.abs().max().item())
    for a, b in ((f_v, r_v), (f_tk, r_tk)):
        a, b = a.float(), b.float()
        assert (a - b).abs().max().item() <= 2.0 ** -7 * b.abs().max().item() + 1e-30
        assert (a == b).float().mean().item() >= 0.99
    assert torch.isfinite(f_pen).all() and torch.isfinite(f_v.float()).all() and torch.isfinite(f_tk.float()).all()
    if N > 3:
        assert f_pen[3].item() == 1.0 and f_v[3].abs().max().item() == 0.0


@pytest.mark.parametrize("M", [64, 8192, 64 * 515])
def test_critic_top_backward_equals_the_four_launches(ops, M):
    """dhaug_critic_top_backward_bf16 (merge layer, 100-wide merge block and logit layer of the 3D critic's backward chain in one
    launch) against the launches it replaces -- dhaug_rank1_mask_bf16, two dhaug_gemm_bf16_dmask_pad (the second with the skip), one
    dhaug_gemm_bf16_dbits_wide: bit-identical cotangents, zero pad columns included; one tile, one tile per workgroup, ragged tile counts."""
    gen = torch.Generator().manual_seed(21)
    n0 = 100
    act = lambda: _bf(torch.cat([torch.relu(torch.randn(M, n0, generator=gen)), torch.zeros(M, 12)], 1)).cuda()   # saved activations (M, 112)
    m1, mh, m0 = act(), act(), act()
    seed = _bf(torch.cat([torch.randn(M, 1, generator=gen) * 0.01, torch.zeros(M, 15)], 1)).cuda()
    mk = lambda rows, cols, pad: _bf(torch.cat([torch.randn(rows, cols, generator=gen) / cols ** 0.5, torch.zeros(rows, pad - cols)], 1)).cuda()
    W2nn, W1nn, Wmnn = mk(n0, n0, 112), mk(n0, n0, 112), mk(512, n0, 112)
    wout = mk(n0, 1, 16)                                          # the logit layer's "nn" copy: (100, 16), column 0
    nb = (M + 127) // 128 * 4 * 256
    bits = [torch.randint(-2**31, 2**31 - 1, (nb,), dtype=torch.int32, generator=gen).cuda() for _ in range(2)]
    cat = torch.zeros(M, 512, dtype=torch.bfloat16, device="cuda")
    cat._dhaug_bits_cols = bits
    assert ops.top_backward_ok(M, n0, 512, (m1, mh, m0), bits)
    g2, g1, g0, gcat = ops.critic_top_backward(seed, wout[:, 0], m1, mh, m0, W2nn, W1nn, Wmnn, bits, n0, 1, 0.0)
    r2 = ops.rank1_mask(seed, wout[:, 0], m1, n0, 1, 0.0)
    r1 = ops.gemm_nt_dmask(r2, W2nn, n0, 112, mh, 1, 0.0)
    r0 = ops.gemm_nt_dmask(r1, W1nn, n0, 112, m0, 1, 0.0, res_bf16=r2)
    rcat = ops.gemm_nt_dmask(r0, Wmnn, 512, 112, cat, 1, 0.0)
    for name, a, b in (("g2", g2, r2), ("g1", g1, r1), ("g0", g0, r0), ("gcat", gcat, rcat)):
        assert a.shape == b.shape and torch.equal(a, b), (name, (a.float() - b.float()).abs().max().item())
    # and against plain fp64 arithmetic on the same operands (the rounding points named in the kernel)
    f = lambda t: t.float().cpu().double()
    e2 = (f(seed)[:, :1] * f(wout)[:, 0][None, :]) * (f(m1)[:, :n0] > 0)
    assert (f(g2)[:, :n0] - e2).abs().max().item() <= 2.0 ** -8 * max(1e-6, e2.abs().max().item())
    # ... layer by layer from each layer's own (bf16) input: g1 = (g2 W2) relu'(mh), g0 = (g1 W1 + g2) relu'(m0), gcat = (g0 Wm) relu'(cat)
    near = lambda got, want: (got - want).abs().max().item() <= 2.0 ** -8 * max(1e-6, want.abs().max().item())
    e1 = (f(g2)[:, :n0] @ f(W2nn)[:, :n0].t()) * (f(mh)[:, :n0] > 0)
    e0 = (f(g1)[:, :n0] @ f(W1nn)[:, :n0].t() + f(g2)[:, :n0]) * (f(m0)[:, :n0] > 0)
    assert near(f(g1)[:, :n0], e1) and near(f(g0)[:, :n0], e0)
    ecat = f(g0)[:, :n0] @ f(Wmnn)[:, :n0].t()                       # (M, 512) before the branches' sign-bit masks
    gc = f(gcat)
    kept = gc != 0                                                   # where the mask kept the value it must be the product's
    assert kept.float().mean().item() > 0.2 and (gc - ecat)[kept].abs().max().item() <= 2.0 ** -8 * max(1e-6, ecat.abs().max().item())


@pytest.mark.parametrize("M", [64, 4096, 64 * 515])
def test_critic_top_tangent_equals_the_three_launches(ops, M):
    """dhaug_critic_top_tangent_bf16 (merge layer with K = 512 and the 100-wide merge block of the 3D critic's tangent sweep in one
    launch, in place over the activations it masks with) against the three dhaug_gemm_bf16_dmask_pad launches it replaces: same bits."""
    gen = torch.Generator().manual_seed(23)
    n0 = 100
    act = lambda: _bf(torch.cat([torch.relu(torch.randn(M, n0, generator=gen)), torch.zeros(M, 12)], 1)).cuda()
    m0, mh, m1 = act(), act(), act()
    ucat = _bf(torch.randn(M, 512, generator=gen) * 0.1).cuda()
    mk = lambda rows, cols, pad: _bf(torch.cat([torch.randn(rows, cols, generator=gen) / cols ** 0.5, torch.zeros(rows, pad - cols)], 1)).cuda()
    Wm, W1, W2 = mk(n0, 512, 512), mk(n0, n0, 112), mk(n0, n0, 112)
    a0, ah, a1 = m0.clone(), mh.clone(), m1.clone()
    k0, kh, k1 = [(t.float().cpu()[:, :n0] > 0).double() for t in (m0, mh, m1)]      # (both forms overwrite the activations they mask with)
    r0 = ops.gemm_nt_dmask(ucat, Wm, n0, 512, a0, 1, 0.0, out=a0)
    rh = ops.gemm_nt_dmask(r0, W1, n0, 112, ah, 1, 0.0, out=ah)
    r1 = ops.gemm_nt_dmask(rh, W2, n0, 112, a1, 1, 0.0, res_bf16=r0, out=a1)
    assert ops.top_tangent_ok(M, n0, 512, ucat, (m0, mh, m1))
    u0, uh, u1 = ops.critic_top_tangent(ucat, m0, mh, m1, Wm, W1, W2, n0, 1, 0.0)
    for name, a, b in (("um0", u0, r0), ("umh", uh, rh), ("um1", u1, r1)):
        assert torch.equal(a, b), (name, (a.float() - b.float()).abs().max().item())
    # and against plain fp64 arithmetic on the same operands, layer by layer from the launch's own (bf16) inputs of each layer: a layout
    # error shared with the three launches it replaces would pass the comparison above
    f = lambda t: t.float().cpu().double()
    near = lambda got, want: (f(got)[:, :n0] - want).abs().max().item() <= 2.0 ** -8 * max(1e-6, want.abs().max().item())
    e0 = (f(ucat) @ f(Wm).t())[:, :n0] * k0
    eh = (f(u0)[:, :n0] @ f(W1)[:, :n0].t()) * kh
    e1 = (f(uh)[:, :n0] @ f(W2)[:, :n0].t() + f(u0)[:, :n0]) * k1
    assert near(u0, e0) and near(uh, eh) and near(u1, e1)
    assert u0[:, n0:].abs().max().item() == 0.0 and uh[:, n0:].abs().max().item() == 0.0 and u1[:, n0:].abs().max().item() == 0.0


@pytest.mark.parametrize("M,n", [(1536, 4), (512, 4), (1536 + 72, 2), (200, 3)])
def test_gemm_nt_group_128_tiles_equal_single_launches(ops, M, n):
    """dhaug_gemm_bf16_group on 128 x 128 tiles (a motion critic's branch layers at one depth as ONE launch: DenseDim 1000, K = 1008
    with a short last stage, ragged row and column tiles) is BIT-identical to one launch per member on the 64 x 64-tile kernel: the
    k-steps are summed in that kernel's order.  Forward form (bias + bf16 residual + ReLU, zero-padded output) and the masked
    input-gradient form; members differ in operands."""
    gen = torch.Generator().manual_seed(11)
    N, K, Kp = 1000, 1000, 1008
    mk = lambda rows, cols, sc: _bf(torch.cat([torch.randn(rows, cols, generator=gen) * sc, torch.zeros(rows, Kp - cols)], 1)).cuda()
    As, Ws, Rs, Ys = ([mk(M, K, 0.5) for _ in range(n)], [mk(N, K, K ** -0.5) for _ in range(n)], [mk(M, N, 1.0) for _ in range(n)],
                      [mk(M, N, 1.0) for _ in range(n)])
    bias = [torch.randn(N, generator=gen).cuda() for _ in range(n)]
    outs = ops.gemm_nt_group([dict(A=As[i], B=Ws[i], N=N, K=Kp, bias=bias[i], res_bf16=Rs[i], act=1, n_pad=Kp) for i in range(n)])
    for i in range(n):
        one, _ = ops.gemm_nt(As[i], Ws[i], N, Kp, bias=bias[i], res_bf16=Rs[i], act=1, out_bf16=True, n_pad=Kp)
        assert outs[i].shape == (M, Kp) and torch.equal(outs[i], one), i
    ref = torch.relu(As[0].float().cpu().double() @ Ws[0].float().cpu().double().t() + bias[0].cpu().double() + Rs[0].float().cpu().double()[:, :N])
    assert maxabs(outs[0][:, :N].float(), ref) <= 2.0 ** -8 * max(1.0, ref.abs().max().item())
    gs = ops.gemm_nt_group([dict(A=As[i], B=Ws[i], N=N, K=Kp, res_bf16=Rs[i], dmask=Ys[i], dmask_act=1, n_pad=Kp) for i in range(n)])
    for i in range(n):
        one = ops.gemm_nt_dmask(As[i], Ws[i], N, Kp, Ys[i], 1, 0.0, res_bf16=Rs[i])
        assert torch.equal(gs[i][:, :N], one[:, :N]), i


@pytest.mark.parametrize("M,K,act,slope,use_bias,use_res", [(4096, 256, 1, 0.0, True, True), (65536, 256, 2, 0.01, True, False),
                                                            (192, 128, 0, 0.0, False, True), (64, 256, 1, 0.0, False, False),
                                                            (33280, 128, 1, 0.0, True, True)])
def test_gemm_nt_256wide(ops, M, K, act, slope, use_bias, use_res):
    """The weight-stationary 256-feature kernel of the training path (rows copied global->LDS, residual on the matrix
    pipe): bf16 output bit-identical to rounding the fp64 reference except where fp32 accumulation order moves a value
    across a rounding boundary (<= 1 bf16 ulp), and identical to the generic kernel's within the same bound."""
    import os
    N = 256
    gen = torch.Generator().manual_seed(M + K + act)
    A = _bf(torch.randn(M, K, generator=gen)).cuda()
    B = _bf(torch.randn(N, K, generator=gen) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=gen).cuda() if use_bias else None
    res = _bf(torch.randn(M, N, generator=gen)).cuda() if use_res else None
    z = A.float().cpu().double() @ B.float().cpu().double().t()
    if use_bias:
        z = z + bias.cpu().double()
    if use_res:
        z = z + res.float().cpu().double()
    ref = z if act == 0 else (torch.relu(z) if act == 1 else torch.nn.functional.leaky_relu(z, slope))
    cb, _ = ops.gemm_nt(A, B, N, K, bias=bias, res_bf16=res, act=act, slope=slope, out_bf16=True)
    os.environ["DHAUG_GEMM_NO256"] = "1"
    try:
        cb_gen, _ = ops.gemm_nt(A, B, N, K, bias=bias, res_bf16=res, act=act, slope=slope, out_bf16=True)
    finally:
        os.environ.pop("DHAUG_GEMM_NO256")
    scale = ref.abs().max().item()
    assert cb.shape == (M, N)
    assert maxabs(cb.float(), ref) <= 2.0 ** -8 * scale                     # one bf16 rounding of the result
    assert maxabs(cb.float(), cb_gen.float()) <= 2.0 ** -7 * scale           # at most an ulp apart
    assert (cb.float() != cb_gen.float()).float().mean().item() < 1e-3      # and almost always identical


@pytest.mark.parametrize("terms,tol", [(3, 3e-5), (6, 2e-6)])
def test_gemm_split_terms(ops, terms, tol):
    """x = hi + lo (3 products) / hi + mid + lo (6 products): bf16 MFMA passes that reproduce an fp32 product."""
    M, N, K = 1024, 256, 256
    gen = torch.Generator().manual_seed(9)
    X = torch.randn(M, K, generator=gen)
    W = torch.randn(N, K, generator=gen) / 16
    A3 = ops.split_bf16(X.cuda(), 0, terms)
    B3 = ops.split_bf16(W.cuda(), 1, terms)
    _, cf = ops.gemm_nt(A3, B3, N, terms * K, out_f32=True)
    ref = X.double() @ W.double().t()
    assert maxabs(cf, ref) <= tol * ref.abs().max().item()
    one = ops.gemm_nt(ops.cast_pad_bf16(X.cuda()), ops.cast_pad_bf16(W.cuda()), N, K, out_f32=True)[1]
    assert maxabs(one, ref) > 10 * maxabs(cf, ref)                 # single-pass bf16 is visibly coarser


@pytest.mark.parametrize("M,N1,N2", [(4096, 256, 256), (1000, 100, 512), (777, 32, 256), (65536, 256, 48), (130, 1, 100),
                                     (98304, 256, 256), (65536 + 192, 128, 384),  # (these two also run under DHAUG_TN_128=1 by hand)
                                     # ragged feature counts on whole 128-row stages: the LDS-DMA kernel with zero-sourced chunks
                                     (6144, 100, 100), (12288, 1, 100), (12288, 256, 30), (98304, 100, 512), (6144, 1, 256)])
def test_gemm_tn(ops, M, N1, N2):
    gen = torch.Generator().manual_seed(M + N1)
    p1, p2 = (N1 + 7) // 8 * 8, (N2 + 7) // 8 * 8
    A = torch.zeros(M, p1); A[:, :N1] = torch.randn(M, N1, generator=gen)
    B = torch.zeros(M, p2); B[:, :N2] = torch.randn(M, N2, generator=gen)
    A, B = _bf(A).cuda(), _bf(B).cuda()
    ref = A[:, :N1].float().cpu().double().t() @ B[:, :N2].float().cpu().double()
    C = ops.gemm_tn(A, B, N1, N2)
    tol = 1e-4 * max(1.0, ref.abs().max().item())                 # fp32 atomics: order-dependent rounding
    assert maxabs(C, ref) <= tol
    C2 = ops.gemm_tn(A, B, N1, N2, out=C.clone(), accumulate=True)
    assert maxabs(C2, 2 * ref) <= 2 * tol
    # bias gradient from the same launch: column sums of A
    cs = torch.empty(N1, device="cuda")
    C3 = ops.gemm_tn(A, B, N1, N2, colsum=cs)
    assert maxabs(C3, ref) <= tol
    csref = A[:, :N1].float().cpu().double().sum(0)
    assert maxabs(cs, csref) <= 1e-4 * max(1.0, csref.abs().max().item()) + 1e-3
    # bias sums over a leading block of rows only (the explicit critic step: real / fake rows, not the interpolated ones)
    if M % 3 == 0 and (2 * M // 3) % 128 == 0:
        cs2 = torch.zeros(N1, device="cuda")
        C4 = ops.gemm_tn(A, B, N1, N2, out=torch.zeros(N1, N2, device="cuda"), accumulate=True, colsum=cs2, colsum_rows=2 * M // 3)
        assert maxabs(C4, ref) <= tol
        cs2ref = A[:2 * M // 3, :N1].float().cpu().double().sum(0)
        assert maxabs(cs2, cs2ref) <= 1e-4 * max(1.0, cs2ref.abs().max().item()) + 1e-3


@pytest.mark.parametrize("M,cs_rows", [(2048, 2048), (8224, 4096), (32 * 1000, 32 * 333), (196608, 131072)])
def test_gemm_tn256(ops, M, cs_rows):
    """the whole-output weight-gradient kernel (dhaug_gemm_tn_group_bf16, here a group of one): uneven batch slices, operands that are column blocks
    of a wider buffer (the 3D critic's concatenation), row-limited bias sums, accumulate; against fp64
    and against the 64 x 64-tile kernel it replaces for this shape."""
    gen = torch.Generator().manual_seed(M)
    wide = _bf(torch.randn(M, 512, generator=gen)).cuda()
    A, B = wide[:, 256:], _bf(torch.randn(M, 256, generator=gen) * 0.5).cuda()
    assert ops.TN256
    ref = A.float().cpu().double().t() @ B.float().cpu().double()
    csref = A[:cs_rows].float().cpu().double().sum(0)
    tol = 1e-4 * max(1.0, ref.abs().max().item())
    cs = torch.full((256,), 3.0, device="cuda")
    C = ops.gemm_tn(A, B, 256, 256, colsum=cs, colsum_rows=cs_rows)
    assert maxabs(C, ref) <= tol
    assert maxabs(cs, csref) <= 1e-4 * max(1.0, csref.abs().max().item()) + 1e-3
    C2 = ops.gemm_tn(A, B, 256, 256, out=C.clone(), accumulate=True, colsum=cs, colsum_rows=cs_rows)
    assert maxabs(C2, 2 * ref) <= 2 * tol and maxabs(cs, 2 * csref) <= 2e-4 * max(1.0, csref.abs().max().item()) + 2e-3
    if M % 128 == 0 and cs_rows % 128 == 0:
        ops.TN256 = False
        try:
            cs_old = torch.zeros(256, device="cuda")
            C_old = ops.gemm_tn(A, B, 256, 256, colsum=cs_old, colsum_rows=cs_rows)
        finally:
            ops.TN256 = True
        assert maxabs(C, C_old) <= tol


def test_gemm_tn_group(ops):
    """one launch for the weight gradients of a whole step: layers of different widths and batch lengths (the 3D critic's
    shapes: 256 x 256 blocks, the 100-wide merge block, the 1-wide logit layer, the 30 / 48-column input layers, a column
    block of the concatenation), accumulate and overwrite mixed, bias sums over a leading block of rows."""
    gen = torch.Generator().manual_seed(77)
    M = 12288
    c16 = lambda n: (n + 15) // 16 * 16
    shapes = [(256, 256, M, 8192), (256, 256, M, 8192), (100, 256, M, 8192), (100, 100, M, 8192), (1, 100, M, 8192),
              (256, 30, M, 8192), (256, 48, 4096, 4096), (100, 256, M, 0), (7, 9, 2048, 2048)]
    wide = _bf(torch.randn(M, 512, generator=gen) * 0.5).cuda()
    items, refs = [], []
    for i, (N1, N2, m, cr) in enumerate(shapes):
        A = torch.zeros(m, c16(N1)); A[:, :N1] = torch.randn(m, N1, generator=gen)
        A = _bf(A).cuda()
        if i == 7:
            B = wide[:, 256:]                                       # a column block: ld 512
        else:
            B = torch.zeros(m, c16(N2)); B[:, :N2] = torch.randn(m, N2, generator=gen) * 0.5
            B = _bf(B).cuda()
        acc = i % 2 == 1
        out = torch.full((N1, N2), 2.0, device="cuda")
        cs = torch.full((N1,), -1.0, device="cuda") if cr else None
        items.append((A, B, N1, N2, out, cs, cr, acc, None, None, None))
        ref = A[:, :N1].float().cpu().double().t() @ B[:, :N2].float().cpu().double()
        csr = A[:cr, :N1].float().cpu().double().sum(0)
        refs.append((ref + (2.0 if acc else 0.0), csr + (-1.0 if acc else 0.0)))
    ops.gemm_tn_group(items)
    for (A, B, N1, N2, out, cs, cr, acc, _, _, _), (ref, csr) in zip(items, refs):
        assert maxabs(out, ref) <= 1e-4 * max(1.0, ref.abs().max().item()), (N1, N2)
        if cs is not None:
            assert maxabs(cs, csr) <= 1e-4 * max(1.0, csr.abs().max().item()) + 1e-3, (N1, N2)


def test_gemm_tn_group_same_output_twice(ops):
    """two contributions to ONE gradient slot in one list (short batch: every block is left with one workgroup, which adds into
    the slot itself): ops.gemm_tn_group serialises them into two launches; the C-ABI call refuses two layers with one output"""
    import ctypes
    lib = ops._lib
    gen = torch.Generator().manual_seed(5)
    M = 1024 + 32
    A1, B1 = _bf(torch.randn(M, 256, generator=gen)).cuda(), _bf(torch.randn(M, 256, generator=gen) * 0.5).cuda()
    A2, B2 = _bf(torch.randn(64, 256, generator=gen)).cuda(), _bf(torch.randn(64, 256, generator=gen) * 0.5).cuda()
    out, cs = torch.zeros(256, 256, device="cuda"), torch.zeros(256, device="cuda")
    items = [(A1, B1, 256, 256, out, cs, M, True, None, None, None), (A2, B2, 256, 256, out, cs, 64, True, None, None, None)]
    ops.gemm_tn_group(items)
    ref = A1.float().cpu().double().t() @ B1.float().cpu().double() + A2.float().cpu().double().t() @ B2.float().cpu().double()
    csr = A1.float().cpu().double().sum(0) + A2.float().cpu().double().sum(0)
    assert maxabs(out, ref) <= 1e-4 * ref.abs().max().item()
    assert maxabs(cs, csr) <= 1e-4 * csr.abs().max().item() + 1e-3
    arr = (lib.TnLayer * 2)()
    for d, (A, B, N1, N2, o, c, cr, acc, _, _, _) in zip(arr, items):
        d.A, d.lda, d.B, d.ldb, d.C, d.ldc, d.colsum_a, d.colsum_rows = A.data_ptr(), 256, B.data_ptr(), 256, o.data_ptr(), 256, c.data_ptr(), cr
        d.M, d.N1, d.N2, d.accumulate, d.max_workgroups = A.shape[0], 256, 256, 1, 0
    ws = ops._tn_group_workspace(out.device)
    rc = lib.lib().dhaug_gemm_tn_group_bf16_phase(arr, 2, ctypes.c_void_p(ws.data_ptr()), 0, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0


def test_split_of_a_column_view_reads_it_in_place(ops):
    """ops.split_bf16 on a column block of a wider fp32 buffer (row pitch > width, base not 16-byte aligned) = the split of its
    contiguous copy"""
    gen = torch.Generator().manual_seed(6)
    wide = torch.randn(300, 100, generator=gen).cuda()
    for c0, w in ((0, 48), (3, 30), (52, 48)):
        v = wide[:, c0:c0 + w]
        assert not v.is_contiguous()
        for mode, T in ((0, 3), (1, 6)):
            assert torch.equal(ops.split_bf16(v, mode, T), ops.split_bf16(v.contiguous(), mode, T))


def test_gemm_split_operand_mask_in_the_epilogue(ops):
    """dhaug_gemm_bf16_dmask_f32: (A B^T + res) * relu'(mask) with fp32 result / residual / mask on six-term split operands -- the
    backward / tangent GEMM of the parity-grade training step -- against fp64 and against the GEMM + dhaug_act_backward_f32 pair it
    replaces (long batch: the 256 x 256-tile kernel; short: the 64 x 64 one; a ragged width)"""
    gen = torch.Generator().manual_seed(31)
    for M, N, K in ((40960 + 24, 256, 256), (300, 100, 256), (4096, 1000, 48)):
        a = torch.randn(M, K, generator=gen)
        W = torch.randn(N, K, generator=gen) / K ** 0.5
        res = torch.randn(M, N, generator=gen).cuda()
        mask = torch.randn(M, N, generator=gen).cuda()
        kp = (K + 15) // 16 * 16
        a6, w6 = ops.split_bf16(a.cuda(), 0, 6, kp), ops.split_bf16(W.cuda(), 1, 6, kp)
        for slope, act in ((0.0, 1), (0.01, 2)):
            out = ops.gemm_nt_dmask_f32(a6, w6, N, 6 * kp, mask, act, slope, res_f32=res)
            pre = a.double() @ W.double().t() + res.cpu().double()
            ref = torch.where(mask.cpu() > 0, pre, pre * slope)
            assert maxabs(out, ref) <= 3e-6 * max(1.0, ref.abs().max().item()), (M, N, K, act)
            _, two = ops.gemm_nt(a6, w6, N, 6 * kp, res_f32=res, out_f32=True)
            two = ops.act_backward(two, mask, act, slope)
            assert maxabs(out, two) <= 1e-6 * max(1.0, ref.abs().max().item()), (M, N, K, act)


def test_gemm_tn_group_wide_layers(ops, monkeypatch):
    """layers wider than 256 handed over whole (a grid of 256 x 256 blocks, one workgroup each, adding into the gradient slot
    itself): the DenseDim-1000 shapes of a video step -- 1000 x 1000, the 100 x 4000 merge layer, a 1000 x 135 input layer, a
    classic 256-wide layer in the same group -- against a float64 product and against the block-by-block form"""
    gen = torch.Generator().manual_seed(78)
    M = 1536
    c16 = lambda n: (n + 15) // 16 * 16
    shapes = [(1000, 1000, 1024, True), (1000, 1000, 0, False), (100, 4000, 1024, True), (1000, 135, 1024, False),
              (256, 256, 1024, True), (1000, 1000, 1536, True), (300, 700, 512, False), (1000, 1000, 1024, True)] + [(1000, 1000, 0, True)] * 3

    def build():
        g = torch.Generator().manual_seed(79)
        items, refs = [], []
        for (N1, N2, cr, acc) in shapes:
            A = torch.zeros(M, c16(N1)); A[:, :N1] = torch.randn(M, N1, generator=g)
            B = torch.zeros(M, c16(N2)); B[:, :N2] = torch.randn(M, N2, generator=g) * 0.5
            A, B = _bf(A).cuda(), _bf(B).cuda()
            out = torch.full((N1, N2), 2.0, device="cuda")
            cs = torch.full((N1,), -1.0, device="cuda") if cr else None
            items.append((A, B, N1, N2, out, cs, cr, acc, None, None, None))
            ref = A[:, :N1].float().cpu().double().t() @ B[:, :N2].float().cpu().double()
            csr = A[:cr, :N1].float().cpu().double().sum(0)
            refs.append((ref + (2.0 if acc else 0.0), csr + (-1.0 if acc else 0.0)))
        return items, refs
    items, refs = build()
    monkeypatch.setattr(ops, "TN_WIDE_MIN_BLOCKS", 1)
    ops.gemm_tn_group(items)
    for (A, B, N1, N2, out, cs, cr, acc, _, _, _), (ref, csr) in zip(items, refs):
        assert maxabs(out, ref) <= 1e-4 * max(1.0, ref.abs().max().item()), (N1, N2)
        if cs is not None:
            assert maxabs(cs, csr) <= 1e-4 * max(1.0, csr.abs().max().item()) + 1e-3, (N1, N2)
    blocks, _ = build()
    monkeypatch.setattr(ops, "TN_WIDE_MIN_BLOCKS", 10 ** 9)
    ops.gemm_tn_group(blocks)
    for a, b in zip(items, blocks):                  # (the blocks are split over the batch there: another summation order)
        assert maxabs(a[4], b[4]) <= 2e-6 * a[4].abs().max().item(), a[2:4]
        assert a[5] is None or maxabs(a[5], b[5]) <= 2e-6 * max(1.0, a[5].abs().max().item()), a[2:4]


def test_pack_kernels(ops):
    gen = torch.Generator().manual_seed(4)
    W = torch.randn(100, 30, generator=gen)
    p = ops.cast_pad_bf16(W.cuda())
    assert p.shape == (100, 32) and maxabs(p[:, :30].float(), W.to(torch.bfloat16).float()) == 0 and p[:, 30:].abs().max() == 0
    t = ops.cast_transpose_bf16(W.cuda())
    assert t.shape == (30, 112) and maxabs(t[:, :100].float(), W.t().to(torch.bfloat16).float()) == 0
    assert t[:, 100:].abs().max() == 0
    s = ops.split_bf16(W.cuda(), 1, 3)
    hi, lo = s[:, :30].float(), s[:, 32:62].float()
    assert maxabs(hi + lo, W) <= 2e-5 * W.abs().max().item() and maxabs(s[:, 64:94].float(), hi) == 0
    s6 = ops.split_bf16(W.cuda(), 0, 6)                       # [hi|hi|mid|mid|hi|lo]
    seg = [s6[:, 32 * t:32 * t + 30].float() for t in range(6)]
    assert maxabs(seg[0] + seg[2] + seg[5], W) <= 2e-7 * W.abs().max().item()
    assert maxabs(seg[1], seg[0]) == 0 and maxabs(seg[3], seg[2]) == 0 and maxabs(seg[4], seg[0]) == 0


def test_colsum_actbwd_adam(ops):
    gen = torch.Generator().manual_seed(8)
    X = torch.randn(5000, 100, generator=gen)
    assert maxabs(ops.colsum(X.cuda()), X.double().sum(0)) <= 1e-3
    Xb = _bf(torch.cat([X, torch.zeros(5000, 4)], 1)).cuda()
    assert maxabs(ops.colsum(Xb, N=100), Xb[:, :100].float().cpu().double().sum(0)) <= 1e-3
    g = _bf(torch.randn(333, 256, generator=gen)).cuda()
    y = _bf(torch.randn(333, 256, generator=gen)).cuda()
    for act, neg in ((1, 0.0), (2, 0.01), (0, 1.0)):
        ref = torch.where(y.float() > 0, g.float(), g.float() * neg).to(torch.bfloat16)
        assert maxabs(ops.act_backward(g, y, act, 0.01).float(), ref.float()) == 0.0
        reff = torch.where(y.float() > 0, g.float(), g.float() * neg)
        assert maxabs(ops.act_backward(g.float(), y.float(), act, 0.01), reff) == 0.0
    p = torch.randn(10007, generator=gen)
    pr = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-4, betas=(0.5, 0.9))
    pg = p.clone().cuda()
    m, v = torch.zeros_like(pg), torch.zeros_like(pg)
    for step in (1, 2, 3):
        gr = torch.randn(10007, generator=gen)
        pr.grad = gr.clone()
        opt.step()
        ops.adam_step(pg, gr.cuda(), m, v, step)
    assert maxabs(pg, pr.detach()) <= 2e-7


def test_two_launch_optimizer_step_equals_the_four_launches(ops):
    """dhaug_adam_repack_step (count + Adam + the nt copies of every weight in one streaming launch, nn = nt^T in a second) against
    dhaug_counter_add + dhaug_adam_step_dev + dhaug_repack_weights on a network of ragged shapes: parameters, moments, packed
    copies (pad included) and the device step count bit for bit over several steps, and torch.optim.Adam within rounding"""
    from dhaug_amd import optim

    def build():
        torch.manual_seed(11)
        return torch.nn.Sequential(torch.nn.Linear(225, 100), torch.nn.Linear(100, 257), torch.nn.Linear(257, 64),
                                   torch.nn.Linear(64, 1), torch.nn.Linear(1000, 130)).cuda()
    nets = [build(), build(), build()]
    ref = torch.optim.Adam(nets[2].parameters(), lr=1e-4, betas=(0.5, 0.9))
    opts = []
    for fused, net in ((True, nets[0]), (False, nets[1])):
        old = optim.FUSED_STEP
        optim.FUSED_STEP = fused
        try:
            opts.append(optim.FusedAdam(net.parameters(), lr=1e-4, betas=(0.5, 0.9)))
        finally:
            optim.FUSED_STEP = old
    gen = torch.Generator().manual_seed(12)
    for step in range(1, 5):
        grads = [torch.randn(p.shape, generator=gen).cuda() * 0.1 for p in nets[0].parameters()]
        for fused, opt, net in ((True, opts[0], nets[0]), (False, opts[1], nets[1])):
            opt.zero_grad()
            for p, g in zip(net.parameters(), grads):
                p.grad.copy_(g)
            old = optim.FUSED_STEP
            optim.FUSED_STEP = fused
            try:
                opt.step()
            finally:
                optim.FUSED_STEP = old
        for p, g in zip(nets[2].parameters(), grads):
            p.grad = g.clone()
        ref.step()
        a, b = opts
        assert int(a.step_dev.item()) == step == int(b.step_dev.item())
        for name in ("flat_param", "exp_avg", "exp_avg_sq"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (step, name)
        for (_, nta, nna), (_, ntb, nnb) in zip(a._packs[2], b._packs[2]):   # the copies, their zero pads included
            assert torch.equal(nta, ntb) and torch.equal(nna, nnb), step
        for (pa, nta, nna), pr in zip(a._packs[2], [p for p in nets[2].parameters() if p.dim() == 2]):
            N, K = pa.shape
            assert maxabs(pa, pr.detach()) <= 3e-7
            assert torch.equal(nta[:, :K], pa.detach().to(torch.bfloat16)) and nta[:, K:].abs().sum() == 0
            assert torch.equal(nna[:, :N], pa.detach().t().to(torch.bfloat16)) and nna[:, N:].abs().sum() == 0


def test_random_bl_aug_golden(ops, golden):
    """next row N3: bone-length swap + per-sample projection against the reference's own output"""
    from dhaug_amd.function_aug.dataloader_update import random_bl_aug, BL_TEMPLATES
    g = golden("bl_aug_96")
    out = random_bl_aug(dev(g["x"]), template_idx=g["idx"].numpy())
    assert maxabs(out, g["out"]) <= 1e-5
    assert maxabs(ops.project_to_2d(dev(g["out"]), dev(g["cam"])), g["proj"]) <= 2e-6
    # the swapped pose has exactly the template's lengths (PoseAug bone order)
    P, C = O.PA_PARENT, O.PA_CHILD
    L = (out[:, P] - out[:, C]).norm(dim=2).cpu()
    assert maxabs(L, torch.tensor(BL_TEMPLATES)[g["idx"].long()]) <= 2e-6


@pytest.mark.parametrize("N", [1, 63, 64, 1000, 65536])
def test_center_kcs_forward(ops, N):
    """one pass = center_flip(center) + kcs_forward on the result, bit for bit"""
    x = (GU.synth_pose16(N, seed=N) + torch.randn(N, 1, 3, generator=torch.Generator().manual_seed(N))).cuda()
    xc, kb = ops.center_kcs_forward(x, 32, True)
    ref_c = ops.center_flip(x, True, False).reshape(N, 48)
    _, ref_k = ops.kcs_forward(x, True, f32=False, bf16_ld=32)
    assert maxabs(xc, ref_c) == 0.0
    assert torch.equal(kb.view(torch.int16), ref_k.view(torch.int16))
    assert maxabs(xc.reshape(N, 16, 3), (x - x[:, :1]).cpu()) == 0.0
    # the oracle's KCS features of the same poses (bf16 operand: half an ulp of the value)
    o_k = O.kcs_features(x.cpu().double())
    assert maxabs(kb.float()[:, :30], o_k) <= 2.0 ** -8 * 1.01 * max(1.0, o_k.abs().max().item())


@pytest.mark.parametrize("M,act,slope,use_res,row0", [(4096, 1, 0.0, False, 0), (4096 + 32, 2, 0.01, True, 0), (2048, 1, 0.0, True, 1024)])
def test_gemm_nt_dbits_equals_dmask(ops, M, act, slope, use_res, row0):
    """the 256-wide backward / tangent step with the mask as a sign-bit array (dhaug_gemm_bf16_dbits) gives, bit for bit, what
    the bf16-mask form gives -- incl. a row slice that starts at a later 32-row tile (the tangent sweep's x_hat rows)"""
    from dhaug_amd import fused
    gen = torch.Generator().manual_seed(M + act)
    tot = M + row0
    A = _bf(torch.randn(tot, 256, generator=gen)).cuda()
    W = _bf(torch.randn(256, 256, generator=gen) / 16).cuda()
    Y = _bf(torch.randn(tot, 256, generator=gen)).cuda()
    Y[::7, ::5] = 0.0                                            # exact zeros: the mask is (y > 0), not (y >= 0)
    R = _bf(torch.randn(tot, 256, generator=gen)).cuda() if use_res else None
    ref = ops.gemm_nt_dmask(A[row0:], W, 256, 256, Y[row0:], act, slope, res_bf16=None if R is None else R[row0:])
    Yb = Y.clone()
    Yb._dhaug_bits = fused.encode_bits(Y.float() > 0)
    assert torch.equal(fused.decode_bits(Yb._dhaug_bits, tot), (Y.float() > 0).cpu())
    ys = ops.tail_rows(Yb, row0)
    assert getattr(ys, "_dhaug_bits", None) is not None
    calls = ops._lib.CALLS[0]
    got = ops.gemm_nt_dmask(A[row0:], W, 256, 256, ys, act, slope, res_bf16=None if R is None else R[row0:])
    assert ops._lib.CALLS[0] == calls + 1
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    # in place over the mask tensor (the tangent sweep writes u over y's x_hat rows): the bits, not y, are read
    out = ops.gemm_nt_dmask(A[row0:], W, 256, 256, ys, act, slope, res_bf16=None if R is None else R[row0:], out=ys)
    assert out.data_ptr() == ys.data_ptr() and torch.equal(out.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("M,act,slope,row0,ldx", [(32, 1, 0.0, 0, 256), (96, 2, 0.01, 0, 256), (32 * 37, 1, 0.0, 64, 512),
                                                  (32 * 256 * 3 + 32, 1, 0.0, 0, 256), (196608, 2, 0.01, 131072 - 64, 256)])
def test_gemm_block2_equals_two_layers(ops, M, act, slope, row0, ldx):
    """both layers of a residual block's backward / tangent step in one launch (dhaug_gemm_block2_bf16) give, bit for bit, what
    the two dhaug_gemm_bf16_dbits launches give -- one tile per workgroup, a ragged tile count, rows that start at a later tile,
    an operand that is a column block of a wider buffer, the critic step's full size; and in place over the mask tensors"""
    from dhaug_amd import fused
    gen = torch.Generator().manual_seed(M + act)
    tot = M + row0
    X = _bf(torch.randn(tot, ldx, generator=gen)).cuda()[:, ldx - 256:]
    W1 = _bf(torch.randn(256, 256, generator=gen) / 16).cuda()
    W2 = _bf(torch.randn(256, 256, generator=gen) / 16).cuda()
    m1 = torch.rand(tot, 256, generator=gen) > 0.45
    m2 = torch.rand(tot, 256, generator=gen) > 0.55
    Y1m = torch.zeros(tot, 256, dtype=torch.bfloat16, device="cuda"); Y1m._dhaug_bits = fused.encode_bits(m1.cuda())
    Y2m = torch.zeros(tot, 256, dtype=torch.bfloat16, device="cuda"); Y2m._dhaug_bits = fused.encode_bits(m2.cuda())
    t1, t2 = ops.tail_rows(Y1m, row0), ops.tail_rows(Y2m, row0)
    x = X[row0:]
    ref1 = ops.gemm_nt_dmask(x, W1, 256, 256, t1, act, slope)
    ref2 = ops.gemm_nt_dmask(ref1, W2, 256, 256, t2, act, slope, res_bf16=x)
    assert ops.block2_ok(x, t1, t2, M)
    calls = ops._lib.CALLS[0]
    y1, y2 = ops.gemm_block2(x, W1, W2, t1, t2, act, slope)
    assert ops._lib.CALLS[0] == calls + 1
    assert torch.equal(y1.view(torch.int16), ref1.view(torch.int16))
    assert torch.equal(y2.view(torch.int16), ref2.view(torch.int16))
    # against fp64 on a sample of rows (the two-launch form is itself tested against fp64 above)
    rows = torch.arange(0, M, max(1, M // 64))
    xs = x[rows].float().cpu().double()
    neg = 0.0 if act == 1 else slope
    z1 = xs @ W1.float().cpu().double().t()
    z1 = torch.where(m1[row0:][rows], z1, z1 * neg)
    assert maxabs(y1[rows], z1) <= 2e-2 * max(1.0, z1.abs().max().item())
    # in place over the mask tensors (the tangent sweep): their bits, not their values, are read
    o1, o2 = ops.gemm_block2(x, W1, W2, t1, t2, act, slope, out1=t1, out2=t2)
    assert o1.data_ptr() == t1.data_ptr() and torch.equal(o1.view(torch.int16), ref1.view(torch.int16))
    assert torch.equal(o2.view(torch.int16), ref2.view(torch.int16))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,K,nblk", [(64, 112, 2), (1000, 112, 2), (4096 + 32, 256, 2), (196608, 112, 2), (2048, 64, 1)])
def test_gemm_nt_dbits_wide_equals_dmask(ops, M, K, nblk):
    """the input-gradient step through a layer whose output is one or two 256-wide column blocks, each masked by ITS sign-bit
    array (dhaug_gemm_bf16_dbits_wide: the 3D critic's merge layer), equals the form that reads the bf16 mask image, bit for bit"""
    from dhaug_amd import fused
    gen = torch.Generator().manual_seed(M + K)
    N = 256 * nblk
    A = _bf(torch.randn(M, K, generator=gen)).cuda()
    A[:, 100:] = 0
    B = _bf(torch.randn(N, K, generator=gen) / 10).cuda()
    Y = _bf(torch.randn(M, N, generator=gen)).cuda()
    Y[::7, ::5] = 0.0
    ref = ops.gemm_nt_dmask(A, B, N, K, Y, 1, 0.0)
    Yb = Y.clone()
    Yb._dhaug_bits_cols = [fused.encode_bits(Y[:, 256 * b:256 * (b + 1)].float() > 0) for b in range(nblk)]
    calls = ops._lib.CALLS[0]
    got = ops.gemm_nt_dmask(A, B, N, K, Yb, 1, 0.0)
    assert ops._lib.CALLS[0] == calls + 1
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))
    got2 = ops.gemm_nt_dmask(A, B, N, K, Yb, 2, 0.01)
    ref2 = ops.gemm_nt_dmask(A, B, N, K, Y, 2, 0.01)
    assert torch.equal(got2.view(torch.int16), ref2.view(torch.int16))


@pytest.mark.parametrize("M,act,slope", [(64, 1, 0.0), (32 * 101, 2, 0.01)])
def test_rank1_on_sign_bits_equals_mask_image(ops, M, act, slope):
    """the first backward step through a 1-wide logit layer behind a 256-wide hidden layer: mask as a sign-bit array
    (dhaug_rank1_bits_bf16) against the mask image (dhaug_rank1_mask_bf16), bit for bit"""
    from dhaug_amd import fused
    gen = torch.Generator().manual_seed(M)
    seed = _bf(torch.randn(M, 16, generator=gen)).cuda()
    w = _bf(torch.randn(256, 16, generator=gen)).cuda()                       # the layer's weights along dim 0, strided
    y = _bf(torch.randn(M, 256, generator=gen)).cuda()
    y[::5, ::3] = 0.0
    ref = ops.rank1_mask(seed, w[:, 0], y, 256, act, slope)
    yb = torch.zeros_like(y)
    yb._dhaug_bits = fused.encode_bits(y.float() > 0)
    calls = ops._lib.CALLS[0]
    got = ops.rank1_mask(seed, w[:, 0], yb, 256, act, slope)
    assert ops._lib.CALLS[0] == calls + 1
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


def test_workgroup_cap_changes_nothing_but_the_grid(ops):
    """dhaug_set_workgroup_cap: the persistent launches (block kernel, 256-wide layer, grouped weight gradients) on a part of
    the card give the same bits as on the whole card; the previous cap comes back from the setter"""
    from dhaug_amd import fused
    gen = torch.Generator().manual_seed(5)
    M = 32 * 700
    x = _bf(torch.randn(M, 256, generator=gen)).cuda()
    W1 = _bf(torch.randn(256, 256, generator=gen) / 16).cuda(); W2 = _bf(torch.randn(256, 256, generator=gen) / 16).cuda()
    m1 = torch.zeros(M, 256, dtype=torch.bfloat16, device="cuda"); m1._dhaug_bits = fused.encode_bits((torch.rand(M, 256, generator=gen) > 0.5).cuda())
    m2 = torch.zeros(M, 256, dtype=torch.bfloat16, device="cuda"); m2._dhaug_bits = fused.encode_bits((torch.rand(M, 256, generator=gen) > 0.5).cuda())
    g = _bf(torch.randn(M, 256, generator=gen)).cuda()

    def run():
        y1, y2 = ops.gemm_block2(x, W1, W2, m1, m2, 1, 0.0)
        z = ops.gemm_nt_dmask(x, W1, 256, 256, m1, 1, 0.0)
        dw = ops.gemm_tn(g, x, 256, 256)
        return y1, y2, z, dw

    ref = run()
    L = ops._lib.lib()
    assert L.dhaug_set_workgroup_cap(0) == 0
    with ops.workgroup_cap(64):
        assert L.dhaug_set_workgroup_cap(64) == 64
        got = run()
    assert L.dhaug_set_workgroup_cap(0) == 0                                  # restored by the context manager
    for a, b in zip(got[:3], ref[:3]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    assert (got[3] - ref[3]).abs().max().item() <= 1e-4 * ref[3].abs().max().item()   # (other partial sums: fp32 rounding only)
    assert L.dhaug_set_workgroup_cap(300) != 0 and L.dhaug_set_workgroup_cap(0) == 0  # out of range: an error code, nothing set


@pytest.mark.parametrize("M,nb", [(32, 2), (32 * 37, 3), (32 * 256 * 2 + 64, 3), (65536, 3)])
def test_gemm_block2_stack_equals_single_blocks(ops, M, nb):
    """a chain of blocks in one launch (dhaug_gemm_block2_stack_bf16: a workgroup walks its row tiles through block 0, reloads
    its weights, reads back the rows it wrote ...) equals the blocks launched one after the other, bit for bit -- also in place"""
    from dhaug_amd import fused
    gen = torch.Generator().manual_seed(M + nb)
    x = _bf(torch.randn(M, 256, generator=gen)).cuda()
    Ws = [(_bf(torch.randn(256, 256, generator=gen) / 16).cuda(), _bf(torch.randn(256, 256, generator=gen) / 16).cuda()) for _ in range(nb)]
    masks = []
    for _ in range(nb):
        pair = []
        for thr in (0.45, 0.55):
            t = torch.zeros(M, 256, dtype=torch.bfloat16, device="cuda")
            t._dhaug_bits = fused.encode_bits((torch.rand(M, 256, generator=gen) > thr).cuda())
            pair.append(t)
        masks.append(pair)
    ref, cur = [], x
    for (W1, W2), (m1, m2) in zip(Ws, masks):
        y1, y2 = ops.gemm_block2(cur, W1, W2, m1, m2, 1, 0.0)
        ref.append((y1, y2)); cur = y2
    calls = ops._lib.CALLS[0]
    got = ops.gemm_block2_stack(x, [(W1, W2, m1, m2, None, None) for (W1, W2), (m1, m2) in zip(Ws, masks)], 1, 0.0)
    assert ops._lib.CALLS[0] == calls + 1
    for (g1, g2), (r1, r2) in zip(got, ref):
        assert torch.equal(g1.view(torch.int16), r1.view(torch.int16)) and torch.equal(g2.view(torch.int16), r2.view(torch.int16))
    # in place over the mask tensors (the tangent sweep)
    got = ops.gemm_block2_stack(x, [(W1, W2, m1, m2, m1, m2) for (W1, W2), (m1, m2) in zip(Ws, masks)], 1, 0.0)
    for (g1, g2), (r1, r2), (m1, m2) in zip(got, ref, masks):
        assert g1.data_ptr() == m1.data_ptr() and torch.equal(g1.view(torch.int16), r1.view(torch.int16))
        assert torch.equal(g2.view(torch.int16), r2.view(torch.int16))
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,act,slope,use_res", [(4096, 256, 256, 1, 0.0, False), (4096, 256, 256, 2, 0.01, False),
                                                   (1024, 256, 128, 1, 0.0, True), (1000, 104, 256, 1, 0.0, False)])
def test_gemm_nt_dmask(ops, M, N, K, act, slope, use_res):
    """(A B^T + res) * act'(y): mask in the 256-wide kernel's epilogue, GEMM + dhaug_act_backward_bf16 elsewhere"""
    gen = torch.Generator().manual_seed(M + N + act)
    A = _bf(torch.randn(M, K, generator=gen)).cuda()
    B = _bf(torch.randn(N, K, generator=gen) / K ** 0.5).cuda()
    y = _bf(torch.randn(M, N, generator=gen)).cuda()
    y[::7, ::5] = 0.0                                            # exact zeros take the "negative" branch, as in act_backward
    res = _bf(torch.randn(M, N, generator=gen)).cuda() if use_res else None
    z = A.float().cpu().double() @ B.float().cpu().double().t()
    if use_res:
        z = z + res.float().cpu().double()
    neg = 0.0 if act == 1 else slope
    ref = torch.where(y.float().cpu() > 0, z, z * neg)
    out = ops.gemm_nt_dmask(A, B, N, K, y, act, slope, res_bf16=res)
    Np = (N + 15) // 16 * 16                                     # zero-padded to the next GEMM's operand width
    assert out.shape == (M, Np) and (Np == N or out[:, N:].float().abs().max().item() == 0.0)
    out = out[:, :N]
    assert maxabs(out.float(), ref) <= 2.0 ** -7 * ref.abs().max().item()
    assert (out.float().cpu()[y.float().cpu() <= 0].abs().max().item() == 0.0) if act == 1 else True
    # into a column block of a wider buffer: nothing beyond the block is touched
    wide = torch.full((M, Np + 24), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt_dmask(A, B, N, K, y, act, slope, res_bf16=res, out=wide[:, 8:8 + N])
    assert torch.equal(wide[:, 8:8 + N], out) and (wide[:, :8] == 7).all() and (wide[:, 8 + N:] == 7).all()


@pytest.mark.parametrize("N,pre", [(1, True), (63, True), (1000, False), (65536, True)])
def test_gen_tail_forward_critics(ops, N, pre):
    """the generator tail with the critics' inputs emitted from the same launch == the separate passes"""
    gen = torch.Generator().manual_seed(N)
    head = torch.randn(N, 35, generator=gen).cuda()
    bl = (torch.rand(N, 15, generator=gen) * 0.4 + 0.1).cuda()
    sc = (torch.randint(-200, 200, (N, 8), generator=gen).float() / 1000.0).cuda()
    quat = [0.1407056450843811, -0.1500701755285263, -0.755240797996521, 0.6223280429840088]
    trans = [1.841107, 4.955284, 1.563445]
    cam9 = [2.29, 2.287, 0.0251, 0.0289, -0.207, 0.2478, -0.00307, -0.00097, -0.00142]
    fake, xc, kcs, p2 = ops.gen_tail_forward_critics(head, bl, sc, pre, (quat, trans, cam9))
    ref_fake, _ = ops.gen_tail_forward(head, bl, sc, pre)
    assert maxabs(fake, ref_fake) <= 1e-6                              # another instantiation: contraction order may differ by an ulp
    assert maxabs(xc, ops.center_flip(fake, True, False).reshape(N, 48)) == 0.0
    _, ref_k = ops.kcs_forward(ops.center_flip(fake, True, False), True, f32=False, bf16_ld=32)
    assert maxabs(kcs.float(), ref_k.float()) <= 2.0 ** -7            # same formula on world vs centred joints: <= 1 bf16 ulp
    c3, ref_p = ops.world_to_camera_project(fake, quat, trans, cam9)
    # x/z amplifies an ulp of the joint (1e-6 at |x| ~ 10) by f/|z|: compare in units of that conditioning
    z = c3[..., 2].abs().clamp_min(1e-3).unsqueeze(-1)
    assert ((p2 - ref_p).abs() * z).max().item() <= 1e-4
    assert maxabs(p2, ref_p) <= 1e-2                                  # and stays bounded everywhere (both clamp x/z to +-1)
    f2, x2, k2, none = ops.gen_tail_forward_critics(head, bl, None, pre, None)
    assert none is None and f2.shape == (N, 16, 3)
    if N <= 1000:
        # ... and the oracle's restatement of the reference (fp64 inputs: the arithmetic the fp32 reference approximates)
        o_fake, _ = O.gen_tail(head.cpu().double(), bl.cpu().double(), sc.cpu().double(), use_preangle=pre)
        o_fake = o_fake.reshape(N, 16, 3)
        assert maxabs(fake, o_fake) <= 1e-5
        o_c = (o_fake - o_fake[:, :1]).reshape(N, 48)
        assert maxabs(xc, o_c) <= 1e-5
        o_k = O.kcs_features(o_fake)
        assert maxabs(kcs.float()[:, :30], o_k) <= 2.0 ** -8 * 1.01 + 1e-5             # bf16 operand: half an ulp of <= 1 (cosines), of <= 2 (lengths: 2^-7)
        o_cam = O.world_to_camera(o_fake, torch.tensor([quat], dtype=torch.float64), torch.tensor([trans], dtype=torch.float64))
        o_p = O.project_to_2d(o_cam, torch.tensor([cam9], dtype=torch.float64))
        zc = o_cam[..., 2].abs().clamp_min(1e-3).unsqueeze(-1)
        assert ((p2.cpu().double() - o_p).abs() * zc).max().item() <= 1e-4


def test_gen_tail_in_kernel_jitter(ops):
    """bone-length jitter drawn inside the tail kernel (Philox4x32-10): integers in [-200, 200) / 1000, uniform,
    reproducible for a (seed, offset), different for another offset, and the pose equals the one computed from the
    same jitter passed explicitly"""
    N = 65536
    gen = torch.Generator().manual_seed(3)
    head = torch.randn(N, 35, generator=gen).cuda()
    bl = (torch.rand(N, 15, generator=gen) * 0.4 + 0.1).cuda()
    fake, xc, kcs, p2, sc = ops.gen_tail_forward_critics(head, bl, None, True, None, rng=(1234, 16), want_scaler=True)
    k = (sc * 1000.0).round()
    assert maxabs(k.cpu() / 1000.0, sc) == 0.0 and k.min().item() >= -200 and k.max().item() <= 199   # true division, as on the host
    counts = torch.bincount((k + 200).long().reshape(-1).cpu(), minlength=400).double()
    expect = N * 8 / 400.0
    chi2 = ((counts - expect) ** 2 / expect).sum().item()
    assert chi2 < 399 + 5 * (2 * 399) ** 0.5                          # chi-square, 399 dof, 5 sigma
    cols = k.cpu().double()
    assert (torch.corrcoef(cols.t()) - torch.eye(8, dtype=torch.float64)).abs().max().item() < 0.02
    again = ops.gen_tail_forward_critics(head, bl, None, True, None, rng=(1234, 16), want_scaler=True)
    assert maxabs(again[4], sc) == 0.0 and maxabs(again[0], fake) == 0.0
    other = ops.gen_tail_forward_critics(head, bl, None, True, None, rng=(1234, 24), want_scaler=True)
    assert (other[4] != sc).float().mean().item() > 0.99
    explicit = ops.gen_tail_forward_critics(head, bl, sc, True, None)
    assert maxabs(explicit[0], fake) == 0.0


@pytest.mark.parametrize("M,N", [(196608, 256), (65536 + 2, 100), (9999, 256), (8192, 512)])
def test_colsum_f32_vector_path(ops, M, N):
    """dhaug_colsum_f32 on rows of whole 16-byte pieces (four columns per thread, four row pairs requested together): against fp64, a column
    block of a wider buffer, accumulate, and the pairing of rows r and r + M/2 that keeps a WGAN critic's logit-bias gradient exactly zero."""
    gen = torch.Generator().manual_seed(17)
    X = torch.randn(M, N, generator=gen).cuda()
    ref = X.double().sum(0).cpu()
    tol = 3e-7 * M ** 0.5 * 4
    assert maxabs(ops.colsum(X), ref) <= tol * 8
    wide = torch.randn(M, N + 12, generator=gen).cuda()
    blk = wide[:, 4:4 + N]                                        # (16-byte aligned column block, row pitch % 4 == 0: vector path)
    assert maxabs(ops.colsum(blk), blk.double().sum(0).cpu()) <= tol * 8
    out = torch.full((N,), 2.0, device="cuda")
    ops.colsum(X, out=out, accumulate=True)
    assert maxabs(out, ref + 2.0) <= tol * 8
    if M % 2 == 0:
        Y = X.clone(); Y[M // 2:] = -Y[:M // 2]                     # second half the exact negative of the first: every column sums to exactly 0
        assert float(ops.colsum(Y).abs().max()) == 0.0
