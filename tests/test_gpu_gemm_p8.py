"""The 256 x 256 x 64 ping-pong NT kernel (csrc/dhaug_gemm_p8.hip): the DenseDim-1000 layers of long batches and of the grouped branch
layers, through the C-ABI (dhaug_gemm_bf16, dhaug_gemm_bf16_dmask_pad, dhaug_gemm_bf16_dmask_f32, dhaug_gemm_bf16_group), against fp64
products of the bf16 operands and against the kernels it replaces (DHAUG_GEMM_NOP8 / DHAUG_NT_GROUP_P8_ROWS=0)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


def maxabs(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import dhaug_amd
    from dhaug_amd import ops as o
    dhaug_amd._lib.lib()            # fail loudly if the HIP extension is missing
    return o


def _operands(M, N, K, Kp, seed):
    gen = torch.Generator().manual_seed(seed)
    A = torch.zeros(M, Kp); A[:, :K] = torch.randn(M, K, generator=gen) * 0.5
    W = torch.zeros(N, Kp); W[:, :K] = torch.randn(N, K, generator=gen) / K ** 0.5
    return _bf(A).cuda(), _bf(W).cuda(), gen


# (M, N, K, Kp): Kp is the operands' padded width = the K the kernel is given.  K tails of 48 / 16 / 32 / none; ragged row blocks; ragged
# column tiles; 256-wide layers of the split-operand arithmetic (K' = 6 K); the shortest K the kernel takes (two K-tiles).
SHAPES = [(13824, 1000, 1000, 1008), (40960 + 8, 1000, 1000, 1008), (41000, 512, 976, 976), (40970, 256, 1536, 1536), (20500, 1000, 160, 160),
          (41216, 264, 128, 128), (10241, 1000, 1024, 1024)]


@pytest.mark.parametrize("M,N,K,Kp", SHAPES)
def test_p8_plain_fp32_out(ops, M, N, K, Kp):
    A, W, _ = _operands(M, N, K, Kp, M + N + K)
    ref = A.float().cpu().double() @ W.float().cpu().double().t()
    _, cf = ops.gemm_nt(A, W, N, Kp, out_f32=True)
    assert maxabs(cf, ref) <= 2e-5 * max(1.0, ref.abs().max().item())
    os.environ["DHAUG_GEMM_NOP8"] = "1"
    try:
        _, cf_old = ops.gemm_nt(A, W, N, Kp, out_f32=True)
    finally:
        del os.environ["DHAUG_GEMM_NOP8"]
    # both sum in fp32; only the order of the k-steps differs
    assert maxabs(cf, cf_old) <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("act,slope", [(0, 0.0), (1, 0.0), (2, 0.01)])
def test_p8_epilogues(ops, act, slope):
    """bias + bf16 residual + activation with a zero-padded bf16 output AND an fp32 output in one launch; the fp32-residual form; the
    masked input-gradient forms (bf16 mask; fp32 mask with fp32 result: the split-operand arithmetic's dhaug_gemm_bf16_dmask_f32)"""
    M, N, K, Kp = 40960 + 77, 1000, 1000, 1008
    A, W, gen = _operands(M, N, K, Kp, 7 + act)
    R = torch.zeros(M, Kp); R[:, :N] = torch.randn(M, N, generator=gen)
    Y = torch.zeros(M, Kp); Y[:, :N] = torch.randn(M, N, generator=gen)
    R, Y = _bf(R).cuda(), _bf(Y).cuda()
    bias = torch.randn(N, generator=gen).cuda()
    pre = A.float().cpu().double() @ W.float().cpu().double().t()
    z = pre + bias.cpu().double() + R[:, :N].float().cpu().double()
    ref = z if act == 0 else (torch.relu(z) if act == 1 else torch.nn.functional.leaky_relu(z, slope))
    scale = max(1.0, ref.abs().max().item())
    cb, cf = ops.gemm_nt(A, W, N, Kp, bias=bias, res_bf16=R, act=act, slope=slope, out_bf16=True, n_pad=Kp, out_f32=True)
    assert maxabs(cf, ref) <= 3e-5 * scale
    assert cb.shape == (M, Kp) and cb[:, N:].abs().max().item() == 0.0
    assert maxabs(cb[:, :N].float(), cf.to(torch.bfloat16).float()) == 0.0
    resf = torch.randn(M, N, generator=gen).cuda()
    _, cf2 = ops.gemm_nt(A, W, N, Kp, bias=bias, res_f32=resf, act=0, out_f32=True)
    assert maxabs(cf2, pre + bias.cpu().double() + resf.cpu().double()) <= 3e-5 * scale
    if act != 0:
        dneg = 0.0 if act == 1 else slope
        g = ops.gemm_nt_dmask(A, W, N, Kp, Y, act, slope, res_bf16=R)
        m = (Y[:, :N].float().cpu() > 0).double()
        refg = (pre + R[:, :N].float().cpu().double()) * (m + (1 - m) * dneg)
        assert g.shape[1] == Kp and g[:, N:].abs().max().item() == 0.0
        assert maxabs(g[:, :N].float(), refg) <= 2.0 ** -8 * max(1.0, refg.abs().max().item())
        yf = torch.randn(M, N, generator=gen).cuda()
        gf = ops.gemm_nt_dmask_f32(A, W, N, Kp, yf, act, slope, res_f32=resf)
        mf = (yf.cpu() > 0).double()
        reff = (pre + resf.cpu().double()) * (mf + (1 - mf) * dneg)
        assert maxabs(gf, reff) <= 3e-5 * max(1.0, reff.abs().max().item())


def test_p8_column_block_views(ops):
    """operands and outputs as column blocks of wider buffers (the cotangent of a concatenation, the branches' column blocks of the
    merge layer's input): row strides larger than the widths, nothing outside the view's own columns is written"""
    M, N, K, Kp = 40960, 1000, 1000, 1008
    A, W, gen = _operands(M, N, K, Kp, 3)
    wide_in = torch.zeros(M, 2 * Kp, dtype=torch.bfloat16, device="cuda"); wide_in[:, Kp:] = A
    wide_out = torch.full((M, 2 * Kp), 7.0, dtype=torch.bfloat16, device="cuda")
    ref = torch.relu(A.float().cpu().double() @ W.float().cpu().double().t())
    ops.gemm_nt(wide_in[:, Kp:], W, N, Kp, act=1, c_bf16=wide_out[:, :Kp], n_pad=Kp)
    assert maxabs(wide_out[:, :N].float(), ref) <= 2.0 ** -8 * max(1.0, ref.abs().max().item())
    assert wide_out[:, N:Kp].abs().max().item() == 0.0 and (wide_out[:, Kp:] == 7.0).all()


@pytest.mark.parametrize("M,n", [(1536, 4), (1536, 2), (1100, 3), (2048 + 5, 1)])
def test_p8_group_equals_single_launches(ops, M, n):
    """dhaug_gemm_bf16_group on the ping-pong tiles (members of >= 1024 rows): bit-identical to one dhaug_gemm_bf16 launch per member
    on the same kernel where that launch takes it, and equal within bf16 rounding to the 128 x 128-tile group kernel it replaces"""
    N, K, Kp = 1000, 1000, 1008
    gen = torch.Generator().manual_seed(M + n)
    mem, refs = [], []
    for i in range(n):
        A, W, _ = _operands(M, N, K, Kp, 100 * i + M)
        R = torch.zeros(M, Kp); R[:, :N] = torch.randn(M, N, generator=gen)
        Y = torch.zeros(M, Kp); Y[:, :N] = torch.randn(M, N, generator=gen)
        R, Y = _bf(R).cuda(), _bf(Y).cuda()
        bias = torch.randn(N, generator=gen).cuda()
        if i % 2 == 0:
            mem.append(dict(A=A, B=W, N=N, K=Kp, bias=bias, res_bf16=R, act=1, n_pad=Kp))
            refs.append(torch.relu(A.float().cpu().double() @ W.float().cpu().double().t() + bias.cpu().double() + R[:, :N].float().cpu().double()))
        else:
            mem.append(dict(A=A, B=W, N=N, K=Kp, res_bf16=R, dmask=Y, dmask_act=1, n_pad=Kp))
            refs.append((A.float().cpu().double() @ W.float().cpu().double().t() + R[:, :N].float().cpu().double()) * (Y[:, :N].float().cpu() > 0).double())
    outs = ops.gemm_nt_group(mem)
    for o, r in zip(outs, refs):
        assert o.shape == (M, Kp) and o[:, N:].abs().max().item() == 0.0
        assert maxabs(o[:, :N].float(), r) <= 2.0 ** -8 * max(1.0, r.abs().max().item())
    os.environ["DHAUG_NT_GROUP_P8_ROWS"] = "0"
    try:
        old = ops.gemm_nt_group(mem)
    finally:
        del os.environ["DHAUG_NT_GROUP_P8_ROWS"]
    for o, q, r in zip(outs, old, refs):
        assert (o.float() - q.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, r.abs().max().item())
    again = ops.gemm_nt_group(mem)                               # deterministic: no atomics, one summation order
    for o, q in zip(outs, again):
        assert torch.equal(o, q)


def test_p8_reads_nothing_beyond_k(ops):
    """the operand rows are K = 1008 wide and the last K-tile covers k = 960 .. 1023: what lies behind a row's K columns (the next row,
    or for the LAST row the end of the allocation) must not reach the result -- NaN-filled neighbours, operands at the end of their
    allocations"""
    M, N, K, Kp = 40960, 1000, 1000, 1008
    A, W, _ = _operands(M, N, K, Kp, 11)
    ref = A.float().cpu().double() @ W.float().cpu().double().t()
    bufA = torch.full((M * Kp + 4096,), float("nan"), dtype=torch.bfloat16, device="cuda")
    bufW = torch.full((N * Kp + 4096,), float("nan"), dtype=torch.bfloat16, device="cuda")
    # the operands end exactly where their buffers' valid part ends; everything behind is NaN
    a = bufA[:M * Kp].view(M, Kp); a.copy_(A)
    w = bufW[:N * Kp].view(N, Kp); w.copy_(W)
    _, cf = ops.gemm_nt(a, w, N, Kp, out_f32=True)
    assert torch.isfinite(cf).all()
    assert maxabs(cf, ref) <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(4, 1000, 1000), (300, 1000, 432), (40960 + 3, 1000, 1000), (8200, 256, 48)])
def test_f16x3_layer_gemm(ops, M, N, K):
    """dhaug_split_f16 + dhaug_gemm_f16x3: x W^T + bias + residual, activation, in the fused parity programs' arithmetic (IEEE-half
    pairs x = hi + lo, Whi Xhi + Whi Xlo + Wlo Xhi on v_mfma_f32_16x16x32_f16) as a layer GEMM at any width: against fp64 on the fp32
    operands at 3e-6 of the result's scale (one bf16 pass: 4e-3; "bf16x3": 2e-4), the split itself exactly"""
    gen = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=gen) * 0.7).cuda()
    W = (torch.randn(N, K, generator=gen) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=gen).cuda()
    res = torch.randn(M, N, generator=gen).cuda()
    Kp = (K + 15) // 16 * 16
    x3, w3 = ops.split_f16(x, 0, Kp), ops.split_f16(W, 1, Kp)
    assert x3.shape == (M, 3 * Kp) and w3.shape == (N, 3 * Kp) and x3.dtype == torch.float16
    hi = x.half()
    lo = (x - hi.float()).half()
    assert torch.equal(x3[:, :K], hi) and torch.equal(x3[:, Kp:Kp + K], hi) and torch.equal(x3[:, 2 * Kp:2 * Kp + K], lo)
    assert torch.equal(w3[:, Kp:Kp + K], (W - W.half().float()).half()) and torch.equal(w3[:, 2 * Kp:2 * Kp + K], W.half())
    if Kp > K:
        assert x3[:, K:Kp].abs().max().item() == 0 and x3[:, 2 * Kp + K:].abs().max().item() == 0
    assert ops.gemm_f16x3_ok(N, 3 * Kp, bias, res)
    z = x.cpu().double() @ W.cpu().double().t() + bias.cpu().double() + res.cpu().double()
    for act, slope, ref in ((0, 0.0, z), (1, 0.0, torch.relu(z)), (2, 0.01, torch.nn.functional.leaky_relu(z, 0.01))):
        y = ops.gemm_nt_f16x3(x3, w3, N, 3 * Kp, bias=bias, res_f32=res, act=act, slope=slope)
        assert maxabs(y, ref) <= 3e-6 * max(1.0, ref.abs().max().item()), (act, maxabs(y, ref))
    y0 = ops.gemm_nt_f16x3(x3, w3, N, 3 * Kp)
    assert maxabs(y0, x.cpu().double() @ W.cpu().double().t()) <= 3e-6 * max(1.0, z.abs().max().item())
    # a shape the kernel does not take is refused, not computed wrongly
    assert not ops.gemm_f16x3_ok(100, 3 * Kp) and not ops.gemm_f16x3_ok(N, 96)
    with pytest.raises(RuntimeError):
        ops.gemm_nt_f16x3(x3, w3[:100], 100, 3 * Kp)


def test_p8_column_tile_wholly_in_the_pad(ops):
    """N = 256 with n_pad = 272: the second 256-column tile holds nothing but zero-pad columns -- it must write its zeros without
    reading a weight row that does not exist"""
    M, N, K = 40960, 256, 512                                   # (K > 256: the wide-layer path)
    A, W, gen = _operands(M, N, K, K, 5)
    bufW = torch.full((N * K + 8,), float("nan"), dtype=torch.bfloat16, device="cuda")
    w = bufW[:N * K].view(N, K); w.copy_(W)
    ref = torch.relu(A.float().cpu().double() @ W.float().cpu().double().t())
    out = torch.full((M, 272), 7.0, dtype=torch.bfloat16, device="cuda")
    ops.gemm_nt(A, w, N, K, act=1, c_bf16=out, n_pad=272)
    assert maxabs(out[:, :N].float(), ref) <= 2.0 ** -8 * max(1.0, ref.abs().max().item())
    assert out[:, N:].abs().max().item() == 0.0


@pytest.mark.parametrize("M,N,kp,order", [(3000, 256, 256, 0), (3000, 256, 256, 1), (777, 1000, 1024, 0), (513, 64, 64, 1), (4096, 256, 128, 0)])
def test_planes_operand_is_bit_identical_to_the_six_segment_operand(ops, M, N, kp, order):
    """dhaug_gemm_bf16x6_planes: the activation side of a "bf16x6" product as the three distinct pieces [hi|mid|lo] (dhaug_split_bf16 mode 2) --
    the same six product terms in the same order as the six-segment operand (mode 0 against mode 1 weights, or, x_order 1, mode 1 against mode 0
    weights: the backward chain), so the same bits; plain, with bias + fp32 residual + activation, and with the fp32 mask."""
    g = torch.Generator().manual_seed(5)
    k = kp - (0 if kp < 1024 else 24)                             # (DenseDim 1000 padded to 1 024)
    x = torch.randn(M, k, generator=g).cuda()
    W = (torch.randn(N, k, generator=g) / k ** 0.5).cuda()
    res, msk, bias = torch.randn(M, N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda(), torch.randn(N, generator=g).cuda()
    A6, B6 = ops.split_bf16(x, order, 6, kp), ops.split_bf16(W, 1 - order, 6, kp)
    A3 = ops.split_bf16(x, 2, 6, kp)
    assert A3.shape == (M, 3 * kp) and torch.equal(A3[:, :kp], A6[:, :kp])
    assert ops.gemm_planes_ok(N, kp, bias, res, msk)
    _, want = ops.gemm_nt(A6, B6, N, 6 * kp, out_f32=True)
    assert torch.equal(ops.gemm_nt_planes(A3, B6, N, kp, x_order=order), want)
    ref = x.double().cpu() @ W.double().cpu().t()
    assert maxabs(want, ref) <= 2e-6 * ref.abs().max().item()
    _, want = ops.gemm_nt(A6, B6, N, 6 * kp, bias=bias, res_f32=res, act=2, slope=0.01, out_f32=True)
    assert torch.equal(ops.gemm_nt_planes(A3, B6, N, kp, bias=bias, res_f32=res, act=2, slope=0.01, x_order=order), want)
    want = ops.gemm_nt_dmask_f32(A6, B6, N, 6 * kp, msk, 1, 0.0, res_f32=res)
    assert torch.equal(ops.gemm_nt_planes(A3, B6, N, kp, res_f32=res, dmask_f32=msk, dmask_act=1, x_order=order), want)
    got, planes = ops.gemm_nt_planes(A3, B6, N, kp, res_f32=res, dmask_f32=msk, dmask_act=1, x_order=order, planes_out=True)
    assert torch.equal(got, want) and torch.equal(planes, ops.split_bf16(want, 2, 6, N))     # the result's own planes, from the epilogue
    out = torch.full((M, N + 8), 7.0, device="cuda")              # into a column block of a wider buffer
    ops.gemm_nt_planes(A3, B6, N, kp, res_f32=res, dmask_f32=msk, dmask_act=1, x_order=order, out=out[:, :N])
    assert torch.equal(out[:, :N], want) and bool((out[:, N:] == 7.0).all())


@pytest.mark.parametrize("pa,pb", [(2, 1), (2, 0), (0, 1)])
def test_weight_gradient_contraction_over_planes_is_bit_identical(ops, pa, pb):
    """dhaug_tn_layer.planes_a / _b: the grouped weight-gradient launch contracts the three planes of a split operand over six virtual rows per
    tensor row, in the order of the six-segment operand's rows -- the same sums bit for bit, from half the bytes.  Several layers of a launch,
    split over the batch and not, narrow and full width; against fp64 too."""
    g = torch.Generator().manual_seed(9)
    Mr = 4096 + 64
    shapes = [(256, 256), (256, 64), (128, 256), (64, 128)]
    items_p, items_6, refs = [], [], []
    for N, K in shapes:
        gt, xt = (torch.randn(Mr, N, generator=g) * 0.01).cuda(), torch.randn(Mr, K, generator=g).cuda()
        g6, x6 = ops.split_bf16(gt, 1, 6, N), ops.split_bf16(xt, 0, 6, K)
        g3, x3 = ops.split_bf16(gt, 2, 6, N), ops.split_bf16(xt, 2, 6, K)
        o6, op = torch.zeros(N, K, device="cuda"), torch.zeros(N, K, device="cuda")
        items_6.append((g6.view(6 * Mr, N), x6.view(6 * Mr, K), N, K, o6, None, 0, True, 6 * Mr, None, None))
        items_p.append(((g3.view(3 * Mr, N) if pa else g6.view(6 * Mr, N)), (x3.view(3 * Mr, K) if pb else x6.view(6 * Mr, K)), N, K, op, None, 0,
                        True, 6 * Mr, None, None, pa, pb))
        refs.append((gt.double().cpu().t() @ xt.double().cpu(), o6, op))
    ops.gemm_tn_group(items_6)
    ops.gemm_tn_group(items_p)
    for ref, o6, op in refs:
        assert torch.equal(o6, op)
        assert maxabs(op, ref) <= 3e-6 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,k,kp", [(3000, 256, 30, 32), (2049, 256, 48, 48)])
def test_six_segment_input_layer_writes_its_results_planes(ops, M, N, k, kp):
    """dhaug_gemm_bf16x6_planes with x_order 2: a narrow input layer (30 / 48 -> DenseDim) on its ordinary six-segment operand, the result
    also as planes -- same bits as dhaug_gemm_bf16 on the same operands, planes = the split of the result."""
    g = torch.Generator().manual_seed(6)
    x, W, bias = torch.randn(M, k, generator=g).cuda(), (torch.randn(N, k, generator=g) / k ** 0.5).cuda(), torch.randn(N, generator=g).cuda()
    A6, B6 = ops.split_bf16(x, 0, 6, kp), ops.split_bf16(W, 1, 6, kp)
    assert ops.gemm_planes_ok(N, kp, bias, six=True)
    _, want = ops.gemm_nt(A6, B6, N, 6 * kp, bias=bias, act=1, out_f32=True)
    got, planes = ops.gemm_nt_planes(A6, B6, N, kp, bias=bias, act=1, x_order=2, planes_out=True)
    assert torch.equal(got, want) and torch.equal(planes, ops.split_bf16(want, 2, 6, N))
    ref = torch.relu(x.double().cpu() @ W.double().cpu().t() + bias.double().cpu())
    assert maxabs(want, ref) <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(3000, 1000, 1000), (513, 256, 256), (2048 + 7, 1000, 432)])
def test_f16x3_layer_chain_without_split_launches(ops, M, N, K):
    """dhaug_gemm_f16x3_planes: the activation side as its two distinct pieces [hi|lo] (piece width the next power of two: 1 024 at DenseDim
    1000), the result's pieces written by the epilogue with zero pad columns -- same bits as dhaug_gemm_f16x3 on the mode 0 operand of the
    same padded width; a second layer fed with the first one's planes equals the layer fed with a split launch."""
    g = torch.Generator().manual_seed(4)
    kp = 64
    while kp < K:
        kp *= 2
    npw = 64
    while npw < N:
        npw *= 2
    x = torch.randn(M, K, generator=g).cuda()
    W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias, res = torch.randn(N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    B3 = ops.split_f16(W, 1, kp)
    A3, A2 = ops.split_f16(x, 0, kp), ops.split_f16(x, 2, kp)
    assert A2.shape == (M, 2 * kp) and torch.equal(A2[:, :kp], A3[:, :kp]) and torch.equal(A2[:, kp:], A3[:, 2 * kp:])
    want = ops.gemm_nt_f16x3(A3, B3, N, 3 * kp, bias=bias, res_f32=res, act=1)
    got, planes = ops.gemm_nt_f16x3_planes(A2, B3, N, kp, True, bias=bias, res_f32=res, act=1, planes_kp=npw)
    assert torch.equal(got, want)
    assert torch.equal(planes, ops.split_f16(want, 2, npw)) and float(planes[:, N:npw].abs().max() if npw > N else 0.0) == 0.0
    got0, planes0 = ops.gemm_nt_f16x3_planes(A3, B3, N, kp, False, bias=bias, res_f32=res, act=1, planes_kp=npw)   # six... the mode 0 operand in, planes out
    assert torch.equal(got0, want) and torch.equal(planes0, planes)
    ref = torch.relu(x.double().cpu() @ W.double().cpu().t() + bias.double().cpu() + res.double().cpu())
    assert maxabs(want, ref) <= 3e-6 * max(1.0, ref.abs().max().item())
    # the next layer from those planes
    W2 = (torch.randn(256, N, generator=g) / N ** 0.5).cuda()
    B2 = ops.split_f16(W2, 1, npw)
    nxt = ops.gemm_nt_f16x3_planes(planes, B2, 256, npw, True)
    assert torch.equal(nxt, ops.gemm_nt_f16x3(ops.split_f16(want, 0, npw), B2, 256, 3 * npw))
